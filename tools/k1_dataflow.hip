// EXPERIMENT (never shipped): barrier-free "dataflow" execution of one full-k SGD epoch.
//
// The product runs one kernel launch per dependency level; at cfg 2 a level is latency bound
// (~9.7 us for 16 K ratings, of which ~5.6 us are launch + end-of-kernel L2 write-back).  Here ONE
// persistent launch per epoch processes level-aligned chunks of 8 ratings in level-major order,
// chunk c on wave c mod W; a rating waits on two per-row sequence counters (its user's and its
// item's chain position) instead of a global barrier.  Rows and counters move with agent-scope
// (sc1) loads/stores: producer = sc1 row stores -> s_waitcnt vmcnt(0) -> sc1 counter stores;
// consumer = sc1 poll -> sc1 row loads (MI355X_MICROARCH.md "Valid forms", measured, not an
// architectural guarantee).  Every spin is bounded (error flag, no hang).  The tool checks the
// result bit for bit against the level-per-launch kernel and times both.
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w tools/k1_dataflow.hip -o build/k1_dataflow
#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>
#include "../mfrec_amd/csrc/mfx_sgd_kernels.h"
namespace mfx {
void set_error(const char *, ...) {}
int fail(int c, const char *fmt, ...) { fprintf(stderr, "fail %d: %s\n", c, fmt); return c; }
int last_code() { return 0; }
}  // namespace mfx
#include "../mfrec_amd/csrc/mfx_schedule.cpp"
using namespace mfx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Chunk { int64_t start; int32_t cnt; int32_t pad; };

__device__ __forceinline__ int ld_sc1_i32(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_i32(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int B = 8;        // ratings per chunk
constexpr int K = 64;
constexpr int SPIN_LIMIT = 1 << 14;
#ifndef FLOW_SLEEP
#define FLOW_SLEEP 2
#endif
#ifndef FLOW_PREFETCH
#define FLOW_PREFETCH 0
#endif

__global__ __launch_bounds__(64) void k1_flow(const Rec *recs, const int2 *seq, const Chunk *chunks, int n_chunks,
                                              double *U, double *V, double *ib, double *ub, int *cnt_u, int *cnt_i,
                                              double *se_partial, int *err, double lr, double Ku, double Ki, double Kb)
{
    __shared__ __attribute__((aligned(16))) double prod[B * 66];
    const int lane = threadIdx.x;
    const int W = gridDim.x;
#if FLOW_PREFETCH
    Chunk nch = chunks[blockIdx.x < n_chunks ? blockIdx.x : 0];
    Rec nrec; nrec.user = 0; nrec.item = 0; nrec.r = 0.0;
    int2 nsq = make_int2(0, 0);
    if (blockIdx.x < n_chunks && lane < nch.cnt) { nrec = recs[nch.start + lane]; nsq = seq[nch.start + lane]; }
#endif
    for (int c = blockIdx.x; c < n_chunks; c += W) {
#if FLOW_PREFETCH
        const Chunk ch = nch;
        const int cnt = ch.cnt;
        Rec rec = nrec;
        int2 sq = nsq;
        if (c + W < n_chunks) {                       // next chunk's record: off the critical path
            nch = chunks[c + W];
            nrec.user = 0; nrec.item = 0; nrec.r = 0.0; nsq = make_int2(0, 0);
            if (lane < nch.cnt) { nrec = recs[nch.start + lane]; nsq = seq[nch.start + lane]; }
        }
#else
        const Chunk ch = chunks[c];
        const int cnt = ch.cnt;
        Rec rec; rec.user = 0; rec.item = 0; rec.r = 0.0;
        int2 sq = make_int2(0, 0);
        if (lane < cnt) { rec = recs[ch.start + lane]; sq = seq[ch.start + lane]; }
#endif
        // ---- wait for both chains (bounded) ----------------------------------------------
        bool ready = lane >= cnt;
        int spins = 0;
        for (;;) {
            if (!ready) {
                const int a = ld_sc1_i32(cnt_u + rec.user), b = ld_sc1_i32(cnt_i + rec.item);
                ready = (a == sq.x) && (b == sq.y);
            }
            if (__all(ready)) break;
            if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(err, 1); break; }
            if (FLOW_SLEEP) __builtin_amdgcn_s_sleep(FLOW_SLEEP);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- gather rows (sc1), products, in-order sums -------------------------------------
        double uval[B], vval[B];
        const double *ur[B], *vr[B];
#pragma unroll
        for (int s = 0; s < B; ++s) {
            const int usr = __builtin_amdgcn_readlane(rec.user, s), itm = __builtin_amdgcn_readlane(rec.item, s);
            ur[s] = U + (size_t)itm * K;
            vr[s] = V + (size_t)usr * K;
        }
#pragma unroll
        for (int s = 0; s < B; ++s) {
            uval[s] = 0.0; vval[s] = 0.0;
            if (s < cnt) { uval[s] = ld_sc1_f64(ur[s] + lane); vval[s] = ld_sc1_f64(vr[s] + lane); }
        }
        double ibv = 0.0, ubv = 0.0;
        if (lane < cnt) { ibv = ld_sc1_f64(ib + rec.item); ubv = ld_sc1_f64(ub + rec.user); }
#pragma unroll
        for (int s = 0; s < B; ++s) prod[s * 66 + lane] = uval[s] * vval[s];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double grad = 0.0, e2 = 0.0;
        if (lane < cnt) {
            double sum = 0.0 + ibv + ubv;
            const double *row = prod + lane * 66;
#pragma unroll 8
            for (int f = 0; f < K; ++f) sum += row[f];
            const double e = rec.r - sum;
            grad = e;
            e2 = e * e;
            st_sc1_f64(ub + rec.user, ubv + lr * (grad - Kb * ubv));
            st_sc1_f64(ib + rec.item, ibv + lr * (grad - Kb * ibv));
        }
        __builtin_amdgcn_wave_barrier();              // prod is reused by the next chunk
#pragma unroll
        for (int s = 0; s < B; ++s) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(grad), s), hi = __builtin_amdgcn_readlane(__double2hiint(grad), s);
            const double g = __hiloint2double(hi, lo);
            if (s < cnt) {
                const double mf = uval[s], cf = vval[s];
                st_sc1_f64(const_cast<double *>(ur[s]) + lane, mf + lr * (g * cf - Ki * mf));
                st_sc1_f64(const_cast<double *>(vr[s]) + lane, cf + lr * (g * mf - Ku * cf));
            }
        }
        const double wsum = wave_sum_fixed(e2);
        if (lane == 0) se_partial[c] = wsum;
        // ---- publish: every store of this wave has completed, then the chain counters ----------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < cnt) { st_sc1_i32(cnt_u + rec.user, sq.x + 1); st_sc1_i32(cnt_i + rec.item, sq.y + 1); }
    }
}

int main(int argc, char **argv)
{
    const int nu = 100000, ni = 50000;
    const int64_t nnz = argc > 1 ? atoll(argv[1]) : 10000000;
    const int waves_per_cu = argc > 2 ? atoi(argv[2]) : 8;
    const int epochs = 3;
    std::vector<int32_t> idx((size_t)nnz * 2);
    std::vector<double> r((size_t)nnz);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int64_t t = 0; t < nnz; ++t) { idx[2 * t] = (int32_t)(rnd() % nu); idx[2 * t + 1] = (int32_t)(rnd() % ni); r[t] = 1.0 + (double)(rnd() % 5); }
    std::vector<int32_t> level((size_t)nnz);
    int64_t n_levels = 0;
    if (mfx_schedule_levels(idx.data(), nnz, nu, ni, level.data(), &n_levels)) return 1;
    std::vector<int64_t> perm((size_t)nnz), off((size_t)n_levels + 1);
    if (mfx_schedule_permutation(level.data(), nnz, n_levels, perm.data(), off.data())) return 1;
    // chain positions in ORIGINAL order (= order of levels along each chain)
    std::vector<int32_t> su((size_t)nnz), si((size_t)nnz), cu((size_t)nu, 0), ci((size_t)ni, 0);
    for (int64_t t = 0; t < nnz; ++t) { su[t] = cu[idx[2 * t]]++; si[t] = ci[idx[2 * t + 1]]++; }
    std::vector<Rec> recs((size_t)nnz);
    std::vector<int2> seq((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) { const int64_t t = perm[j]; recs[j] = {idx[2 * t], idx[2 * t + 1], r[t]}; seq[j] = make_int2(su[t], si[t]); }
    std::vector<Chunk> chunks;
    for (int64_t l = 0; l < n_levels; ++l)
        for (int64_t s = off[l]; s < off[l + 1]; s += B) chunks.push_back({s, (int32_t)std::min<int64_t>(B, off[l + 1] - s), 0});
    printf("nnz %lld levels %lld chunks %zu\n", (long long)nnz, (long long)n_levels, chunks.size());

    std::vector<double> U0((size_t)ni * K), V0((size_t)nu * K);
    for (auto &x : U0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    for (auto &x : V0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    Rec *d_recs; int2 *d_seq; Chunk *d_chunks; double *dU[2], *dV[2], *dib[2], *dub[2], *d_se; int *d_cu, *d_ci, *d_err;
    CK(hipMalloc(&d_recs, nnz * sizeof(Rec))); CK(hipMalloc(&d_seq, nnz * sizeof(int2))); CK(hipMalloc(&d_chunks, chunks.size() * sizeof(Chunk)));
    CK(hipMemcpy(d_recs, recs.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_seq, seq.data(), nnz * sizeof(int2), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&dU[v], U0.size() * 8)); CK(hipMalloc(&dV[v], V0.size() * 8)); CK(hipMalloc(&dib[v], ni * 8)); CK(hipMalloc(&dub[v], nu * 8));
        CK(hipMemcpy(dU[v], U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV[v], V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dib[v], 0, ni * 8)); CK(hipMemset(dub[v], 0, nu * 8));
    }
    CK(hipMalloc(&d_se, (chunks.size() + 4096) * 8)); CK(hipMalloc(&d_cu, nu * 4)); CK(hipMalloc(&d_ci, ni * 4)); CK(hipMalloc(&d_err, 4));
    CK(hipMemset(d_err, 0, 4));
    const double lr = 0.01, Ku = 0.1, Ki = 0.1, Kb = 0.007;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_ref = 0, ms_flow = 0;

    // ---- reference: one launch per level (the product kernel) --------------------------------
    {
        K1Params P{}; P.recs = d_recs; P.U = dU[0]; P.V = dV[0]; P.ib = dib[0]; P.ub = dub[0]; P.se_partial = d_se; P.k = K;
        P.upd_u = P.upd_i = 1; P.lr = lr; P.Ku = Ku; P.Ki = Ki; P.Kb = Kb;
        const size_t lds = (size_t)kWaves * 8 * k1_stride(K) * 8;
        for (int ep = 0; ep < epochs; ++ep) {
            if (ep == epochs - 1) CK(hipEventRecord(e0));
            for (int64_t l = 0; l < n_levels; ++l) {
                const int cnt = (int)(off[l + 1] - off[l]);
                const int blocks = (cnt + 8 * kWaves - 1) / (8 * kWaves);
                hipLaunchKernelGGL((k1_level<64, 1, 8, K1_LINEAR, 0>), dim3(blocks), dim3(kWaves * 64), lds, 0, P, off[l], cnt, (int64_t)0);
            }
            if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
    }
    // ---- dataflow ------------------------------------------------------------------------
    const int W = 256 * waves_per_cu;
    for (int ep = 0; ep < epochs; ++ep) {
        CK(hipMemsetAsync(d_cu, 0, nu * 4)); CK(hipMemsetAsync(d_ci, 0, ni * 4));
        if (ep == epochs - 1) CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k1_flow, dim3(W), dim3(64), 0, 0, d_recs, d_seq, d_chunks, (int)chunks.size(), dU[1], dV[1], dib[1], dub[1],
                           d_cu, d_ci, d_se, d_err, lr, Ku, Ki, Kb);
        if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_flow, e0, e1)); }
    }
    CK(hipDeviceSynchronize());
    int err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    std::vector<double> a(U0.size()), b(U0.size()), c(V0.size()), d(V0.size());
    CK(hipMemcpy(a.data(), dU[0], a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dU[1], b.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), dV[0], c.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), dV[1], d.size() * 8, hipMemcpyDeviceToHost));
    size_t diffU = 0, diffV = 0;
    for (size_t i = 0; i < a.size(); ++i) diffU += (memcmp(&a[i], &b[i], 8) != 0);
    for (size_t i = 0; i < c.size(); ++i) diffV += (memcmp(&c[i], &d[i], 8) != 0);
    printf("level-per-launch (eager): %.3f ms/epoch = %.3f G/s | dataflow W=%d: %.3f ms/epoch = %.3f G/s | spin-timeout flag %d | differing elements U %zu V %zu\n",
           ms_ref, nnz / ms_ref / 1e6, W, ms_flow, nnz / ms_flow / 1e6, err, diffU, diffV);
    return 0;
}
