// EXPERIMENT (round 3): the user-resident full-k SGD epoch -- tools/k1_csr.hip with 16-byte coherent row
// accesses (two ratings side by side per wave instruction), the item bias and the item's sequence counter in
// ONE 16-byte granule (the poll returns the bias, the publish stores both), records two ahead.
// Measured against the product's level-per-launch kernel on the same user-major input, bit for bit.
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w tools/k1_ur.hip -o /tmp/k1_ur
// /tmp/k1_ur <nu> <ni> <nnz> [epochs]
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

struct __attribute__((aligned(16))) URec { int item; int seq; double r; };
struct __attribute__((aligned(16))) IHdr { double ib; int cnt; int pad; };
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

#ifndef UR_AUX
#define UR_AUX 16          // sc1
#endif
__device__ __forceinline__ v4u ld16c(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, UR_AUX); }
__device__ __forceinline__ void st16c(v4u v, __amdgpu_buffer_rsrc_t r, unsigned off) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, UR_AUX); }

constexpr int K = 64, KP = 32, SPW = 2, STRIDE = 66;
constexpr int SPIN_LIMIT = 1 << 18;

// one step of the B slots of a wave.  FULL: every slot advances (no branches around memory operations).
template <int B, bool FULL>
__device__ __forceinline__ void ur_step(unsigned long long mask, bool ready, const URec &rec, double ibv, double &ubv, double &se,
                                        d2 (&vv)[B / SPW], double *prod, __amdgpu_buffer_rsrc_t rU, int lane, int g, int q,
                                        double lr, double Ku, double Ki, double Kb, double &ibn)
{
    constexpr int NR = B / SPW;
    unsigned uoff[NR];
    bool rdy[NR];
    d2 uu[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int src = j * SPW + g;
        const int itm = __shfl(rec.item, src, 64);
        uoff[j] = (unsigned)itm * (unsigned)(K * 8) + (unsigned)q * 16u;
        rdy[j] = FULL || ((mask >> src) & 1ull);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        uu[j] = d2{0.0, 0.0};
        if (FULL || rdy[j]) uu[j] = __builtin_bit_cast(d2, ld16c(rU, uoff[j]));
    }
#pragma unroll
    for (int j = 0; j < NR; ++j)
        *reinterpret_cast<d2 *>(prod + (j * SPW + g) * STRIDE + 2 * q) = d2{uu[j].x * vv[j].x, uu[j].y * vv[j].y};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double grad = 0.0;
    if (FULL ? (lane < B) : ready) {
        double sum = 0.0 + ibv + ubv;                        // kmf_train.pyx:92
        const double *row = prod + lane * STRIDE;
#pragma unroll
        for (int f = 0; f < K; f += 8) {
            const d2 a = *reinterpret_cast<const d2 *>(row + f), b = *reinterpret_cast<const d2 *>(row + f + 2);
            const d2 c = *reinterpret_cast<const d2 *>(row + f + 4), d = *reinterpret_cast<const d2 *>(row + f + 6);
            sum += a.x; sum += a.y; sum += b.x; sum += b.y; sum += c.x; sum += c.y; sum += d.x; sum += d.y;
        }
        const double e = rec.r - sum;                        // kmf_train.pyx:251-256
        grad = e;
        se += e * e;
        ubv = ubv + lr * (grad - Kb * ubv);                  // :259-260
        ibn = ibv + lr * (grad - Kb * ibv);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double gr = shfl_f64(grad, j * SPW + g);
        if (FULL || rdy[j]) {
            const d2 mf = uu[j], cf = vv[j];
            d2 un, vn;
            un.x = mf.x + lr * (gr * cf.x - Ki * mf.x);      // :267-270
            un.y = mf.y + lr * (gr * cf.y - Ki * mf.y);
            vn.x = cf.x + lr * (gr * mf.x - Ku * cf.x);
            vn.y = cf.y + lr * (gr * mf.y - Ku * cf.y);
            st16c(__builtin_bit_cast(v4u, un), rU, uoff[j]);
            vv[j] = vn;
        }
    }
}

template <int B, bool NOPOLL>
__global__ __launch_bounds__(64) void k1_ur(const URec *__restrict__ recs, const int64_t *__restrict__ run_off,
                                            const int *__restrict__ run_user, int n_runs, double *U, double *V, IHdr *hdr,
                                            double *ub, int *ticket, double *se_run, int *err, double lr, double Ku,
                                            double Ki, double Kb)
{
    constexpr int NR = B / SPW;
    __shared__ __attribute__((aligned(16))) double prod[B * STRIDE];
    const int lane = threadIdx.x, g = lane / KP, q = lane % KP;
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(U, 0, 0x7fffffff, 0x27000);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(hdr, 0, 0x7fffffff, 0x27000);
    constexpr unsigned long long ALL = (B == 64) ? ~0ull : ((1ull << B) - 1ull);
    for (;;) {
        int b = 0;
        if (lane == 0) b = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0x7fffffff : atomicAdd(ticket, 1);
        b = __builtin_amdgcn_readfirstlane(b);
        if ((int64_t)b * B >= n_runs) break;
        int64_t pos = 0, end = 0;
        int my_user = 0;
        bool have = false;
        if (lane < B && b * B + lane < n_runs) { pos = run_off[b * B + lane]; end = run_off[b * B + lane + 1]; my_user = run_user[b * B + lane]; have = true; }
        URec rec{0, 0, 0.0}, nxt{0, 0, 0.0};
        if (pos < end) rec = __builtin_bit_cast(URec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(recs + pos)));
        if (pos + 1 < end) nxt = __builtin_bit_cast(URec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(recs + pos + 1)));
        double ubv = have ? ub[my_user] : 0.0;
        d2 vv[NR];
        int usr[NR], hv[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            usr[j] = __shfl(my_user, j * SPW + g, 64);
            hv[j] = __shfl((int)have, j * SPW + g, 64);
            vv[j] = d2{0.0, 0.0};
            if (hv[j]) vv[j] = *reinterpret_cast<const d2 *>(V + (size_t)usr[j] * K + 2 * q);
        }
        double se = 0.0;
        int spins = 0;
        bool bail = false;
        while (__any(pos < end)) {
            bool ready = false;
            double ibv = 0.0;
            if (pos < end) {
                const IHdr h = __builtin_bit_cast(IHdr, ld16c(rH, (unsigned)rec.item * 16u));
                ibv = h.ib;
                ready = NOPOLL || (h.cnt == rec.seq);
            }
            const unsigned long long mask = __ballot(ready);
            if (!mask) {
                if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(err, 1); bail = true; break; }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            double ibn = 0.0;
            if (mask == ALL) ur_step<B, true>(mask, ready, rec, ibv, ubv, se, vv, prod, rU, lane, g, q, lr, Ku, Ki, Kb, ibn);
            else ur_step<B, false>(mask, ready, rec, ibv, ubv, se, vv, prod, rU, lane, g, q, lr, Ku, Ki, Kb, ibn);
            // publish: every row store of this wave has completed, then bias + chain counter in one granule
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ready) {
                IHdr nh; nh.ib = ibn; nh.cnt = rec.seq + 1; nh.pad = 0;
                st16c(__builtin_bit_cast(v4u, nh), rH, (unsigned)rec.item * 16u);
                ++pos;
                rec = nxt;
                if (pos + 1 < end) nxt = __builtin_bit_cast(URec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(recs + pos + 1)));
            }
            spins = 0;
        }
        if (bail) break;
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (hv[j]) *reinterpret_cast<d2 *>(V + (size_t)usr[j] * K + 2 * q) = vv[j];
        if (have) { ub[my_user] = ubv; se_run[b * B + lane] = se; }
    }
}

__global__ void k_hdr_reset(IHdr *hdr, const double *ib, int ni)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ni) { IHdr h; h.ib = ib[i]; h.cnt = 0; h.pad = 0; hdr[i] = h; }
}
__global__ void k_hdr_finish(const IHdr *hdr, double *ib, int ni)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ni) ib[i] = hdr[i].ib;
}

template <int B, bool NOPOLL>
static int run_variant(const char *name, int wpc, int epochs, int64_t nnz, int nu, int ni, int n_runs, const URec *d_in, const int64_t *d_run,
                       const int *d_ruser, double *dU, double *dV, double *dib, double *dub, IHdr *d_hdr, int *d_ticket, double *d_se,
                       int *d_err, const std::vector<double> &U0, const std::vector<double> &V0, const std::vector<double> &a,
                       const std::vector<double> &c, float ms_ref)
{
    const double lr = 0.01, Ku = 0.1, Ki = 0.1, Kb = 0.007;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemcpy(dU, U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV, V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dib, 0, (size_t)ni * 8)); CK(hipMemset(dub, 0, (size_t)nu * 8)); CK(hipMemset(d_err, 0, 4));
    float ms = 0;
    for (int ep = 0; ep < epochs; ++ep) {
        CK(hipMemsetAsync(d_ticket, 0, 4));
        if (ep == epochs - 1) CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_hdr_reset, dim3((ni + 255) / 256), dim3(256), 0, 0, d_hdr, dib, ni);
        hipLaunchKernelGGL((k1_ur<B, NOPOLL>), dim3(256 * wpc), dim3(64), 0, 0, d_in, d_run, d_ruser, n_runs, dU, dV, d_hdr, dub, d_ticket, d_se, d_err, lr, Ku, Ki, Kb);
        hipLaunchKernelGGL(k_hdr_finish, dim3((ni + 255) / 256), dim3(256), 0, 0, d_hdr, dib, ni);
        if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    }
    CK(hipDeviceSynchronize());
    int err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    size_t diffU = 0, diffV = 0;
    if (!NOPOLL) {
        std::vector<double> U1(U0.size()), V1(V0.size());
        CK(hipMemcpy(U1.data(), dU, U1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(V1.data(), dV, V1.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); ++i) diffU += (memcmp(&a[i], &U1[i], 8) != 0);
        for (size_t i = 0; i < c.size(); ++i) diffV += (memcmp(&c[i], &V1[i], 8) != 0);
    }
    printf("%-22s %2d waves/CU: %8.3f ms/epoch = %.3f G/s (level-per-launch %.3f ms = %.3f G/s) | timeout %d | differing U %zu V %zu%s\n",
           name, wpc, ms, nnz / ms / 1e6, ms_ref, nnz / ms_ref / 1e6, err, diffU, diffV, NOPOLL ? "  [no waits: WRONG results, throughput bound only]" : "");
    fflush(stdout);
    return err ? 2 : 0;
}

int main(int argc, char **argv)
{
    const int nu = argc > 1 ? atoi(argv[1]) : 100000, ni = argc > 2 ? atoi(argv[2]) : 50000;
    const int64_t nnz = argc > 3 ? atoll(argv[3]) : 10000000;
    const int epochs = argc > 4 ? atoi(argv[4]) : 2;
    std::vector<int32_t> idx((size_t)nnz * 2);
    std::vector<double> r((size_t)nnz);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    {   // get_ratings(randomize_order=False): user-major, ascending items inside a user
        std::vector<int32_t> uu((size_t)nnz), ii((size_t)nnz);
        std::vector<int64_t> cnt((size_t)nu + 1, 0);
        for (int64_t t = 0; t < nnz; ++t) { uu[t] = (int32_t)(rnd() % nu); ii[t] = (int32_t)(rnd() % ni); ++cnt[uu[t] + 1]; }
        for (int u = 0; u < nu; ++u) cnt[u + 1] += cnt[u];
        std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
        for (int64_t t = 0; t < nnz; ++t) { const int64_t p = fill[uu[t]]++; idx[2 * p] = uu[t]; idx[2 * p + 1] = ii[t]; }
        for (int u = 0; u < nu; ++u) {
            std::vector<int32_t> tmp;
            for (int64_t p = cnt[u]; p < cnt[u + 1]; ++p) tmp.push_back(idx[2 * p + 1]);
            std::sort(tmp.begin(), tmp.end());
            for (int64_t p = cnt[u]; p < cnt[u + 1]; ++p) idx[2 * p + 1] = tmp[p - cnt[u]];
        }
        for (int64_t t = 0; t < nnz; ++t) r[t] = 1.0 + (double)(rnd() % 5);
    }
    std::vector<int32_t> level((size_t)nnz);
    int64_t n_levels = 0;
    if (mfx_schedule_levels(idx.data(), nnz, nu, ni, level.data(), &n_levels)) return 1;
    std::vector<int64_t> perm((size_t)nnz), off((size_t)n_levels + 1);
    if (mfx_schedule_permutation(level.data(), nnz, n_levels, perm.data(), off.data())) return 1;
    std::vector<Rec> recs_lvl((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) { const int64_t t = perm[j]; recs_lvl[j] = {idx[2 * t], idx[2 * t + 1], r[t]}; }
    std::vector<URec> recs_in((size_t)nnz);
    std::vector<int32_t> ci((size_t)ni, 0), ruser;
    std::vector<int64_t> run_off;
    for (int64_t t = 0; t < nnz; ++t) {
        recs_in[t] = {idx[2 * t + 1], ci[idx[2 * t + 1]]++, r[t]};
        if (t == 0 || idx[2 * t] != idx[2 * (t - 1)]) { run_off.push_back(t); ruser.push_back(idx[2 * t]); }
    }
    const int n_runs = (int)run_off.size();
    run_off.push_back(nnz);
    printf("nu %d ni %d nnz %lld: levels (= DAG depth) %lld, runs %d\n", nu, ni, (long long)nnz, (long long)n_levels, n_runs);

    std::vector<double> U0((size_t)ni * K), V0((size_t)nu * K);
    for (auto &x : U0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    for (auto &x : V0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    Rec *d_lvl; URec *d_in; int64_t *d_run; int *d_ruser; double *dU[2], *dV[2], *dib[2], *dub[2], *d_se; IHdr *d_hdr; int *d_err, *d_ticket;
    CK(hipMalloc(&d_lvl, nnz * sizeof(Rec))); CK(hipMalloc(&d_in, nnz * sizeof(URec))); CK(hipMalloc(&d_run, run_off.size() * 8)); CK(hipMalloc(&d_ruser, ruser.size() * 4));
    CK(hipMemcpy(d_lvl, recs_lvl.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_in, recs_in.data(), nnz * sizeof(URec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_run, run_off.data(), run_off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ruser, ruser.data(), ruser.size() * 4, hipMemcpyHostToDevice));
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&dU[v], U0.size() * 8)); CK(hipMalloc(&dV[v], V0.size() * 8)); CK(hipMalloc(&dib[v], (size_t)ni * 8)); CK(hipMalloc(&dub[v], (size_t)nu * 8));
    }
    CK(hipMemcpy(dU[0], U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV[0], V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dib[0], 0, (size_t)ni * 8)); CK(hipMemset(dub[0], 0, (size_t)nu * 8));
    CK(hipMalloc(&d_se, ((size_t)nnz / 8 + (size_t)n_runs + 8192) * 8)); CK(hipMalloc(&d_hdr, (size_t)ni * 16)); CK(hipMalloc(&d_err, 4)); CK(hipMalloc(&d_ticket, 4));
    const double lr = 0.01, Ku = 0.1, Ki = 0.1, Kb = 0.007;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_ref = 0;
    {   // reference: the product kernel, one launch per level, on the same input
        K1Params P{}; P.recs = d_lvl; P.U = dU[0]; P.V = dV[0]; P.ib = dib[0]; P.ub = dub[0]; P.se_partial = d_se; P.k = K;
        P.upd_u = P.upd_i = 1; P.lr = lr; P.Ku = Ku; P.Ki = Ki; P.Kb = Kb;
        const size_t lds = (size_t)kWaves * 8 * k1_stride(K) * 8;
        for (int ep = 0; ep < epochs; ++ep) {
            if (ep == epochs - 1) CK(hipEventRecord(e0));
            for (int64_t l = 0; l < n_levels; ++l) {
                const int cnt = (int)(off[l + 1] - off[l]);
                const int blocks = (cnt + 8 * kWaves - 1) / (8 * kWaves);
                hipLaunchKernelGGL((k1_level<32, 2, 8, K1_LINEAR, 2, double, true>), dim3(blocks), dim3(kWaves * 64), lds, 0, P, off[l], cnt, (int64_t)0);
            }
            if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
    }
    std::vector<double> a(U0.size()), c(V0.size());
    CK(hipMemcpy(a.data(), dU[0], a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), dV[0], c.size() * 8, hipMemcpyDeviceToHost));
#define RUN(B, NP, W) if (run_variant<B, NP>(#B " slots" #NP, W, epochs, nnz, nu, ni, n_runs, d_in, d_run, d_ruser, dU[1], dV[1], dib[1], dub[1], d_hdr, d_ticket, d_se, d_err, U0, V0, a, c, ms_ref)) return 2
    RUN(8, false, 8); RUN(8, false, 16); RUN(8, false, 24); RUN(8, false, 32);
    RUN(16, false, 8); RUN(16, false, 16);
    RUN(8, true, 16); RUN(16, true, 16);
    return 0;
}
