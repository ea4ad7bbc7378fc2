// EXPERIMENT: exact full-k SGD epoch for USER-GROUPED input order (CSR order = what
// get_ratings(randomize_order=False) produces) as ONE persistent launch per epoch.
//
// A wave owns 8 consecutive users at a time (dynamic ticket), keeps their factor rows and
// biases in REGISTERS for the whole run of each user, and walks the 8 runs side by side.  The
// only cross-wave dependency is the item row: item i's ratings must be applied in input order,
// which a per-item sequence counter enforces (rating waits until cnt[i] == its rank in item
// i's chain, then row loads (sc1) -> update -> row stores (sc1) -> s_waitcnt vmcnt(0) ->
// cnt[i] = rank + 1).  Tickets are handed out in increasing user order, so the earliest
// unfinished rating always belongs to a running wave: no deadlock, and every spin is bounded.
// Per rating only the item row moves (1 KB) instead of both rows twice (2 KB), and there is no
// per-level launch + L2 write-back.  Result must be bit-identical to the level schedule.
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w tools/k1_csr.hip -o /tmp/k1_csr
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

__device__ __forceinline__ int ld_sc1_i32(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifdef CSR_NOSC1    // diagnostic (WRONG results): plain cached accesses, to price the coherent path
__device__ __forceinline__ double ld_sc1_f64(const double *p) { return *p; }
__device__ __forceinline__ void st_sc1_f64(double *p, double v) { *p = v; }
#else
__device__ __forceinline__ double ld_sc1_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
__device__ __forceinline__ void st_sc1_i32(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int B = 8;        // user slots per wave
constexpr int K = 64;
constexpr int SPIN_LIMIT = 1 << 16;
#ifndef CSR_SLEEP
#define CSR_SLEEP 1
#endif

__global__ __launch_bounds__(64) void k1_csr(const Rec *__restrict__ recs, const int *__restrict__ seq,
                                             const int64_t *__restrict__ run_off, int n_runs, double *U, double *V,
                                             double *ib, double *ub, int *cnt_i, int *ticket, double *se_partial,
                                             int *err, double lr, double Ku, double Ki, double Kb)
{
    __shared__ __attribute__((aligned(16))) double prod[B * 66];
    const int lane = threadIdx.x;
    for (;;) {
        int b = 0;
        if (lane == 0) b = ld_sc1_i32(err) ? 0x7fffffff : atomicAdd(ticket, 1);    // a timed-out wave stops everyone
        b = __builtin_amdgcn_readfirstlane(b);
        if ((int64_t)b * B >= n_runs) break;
        // ---- slot state: lane j (< B) owns slot j ------------------------------------------
        int64_t pos = 0, end = 0;
        if (lane < B && b * B + lane < n_runs) { pos = run_off[b * B + lane]; end = run_off[b * B + lane + 1]; }
        Rec rec; rec.user = 0; rec.item = 0; rec.r = 0.0;
        int sq = 0;
        if (pos < end) { rec = recs[pos]; sq = seq[pos]; }
        const int my_user = rec.user;
        double ubv = (pos < end) ? ub[my_user] : 0.0;
        double vval[B];
        int usr[B];
#pragma unroll
        for (int s = 0; s < B; ++s) {
            usr[s] = __builtin_amdgcn_readlane(my_user, s);
            vval[s] = V[(size_t)usr[s] * K + lane];              // slot may be empty: user 0's row, unused
        }
        double se_acc = 0.0;
        int spins = 0;
        bool bail = false;
        while (__any(pos < end)) {
            // ---- which slots may go: item chain position reached? ------------------------------
            bool ready = false;
#ifdef CSR_NOPOLL   // diagnostic (WRONG results): no waiting at all
            ready = pos < end;
#else
            if (pos < end) ready = ld_sc1_i32(cnt_i + rec.item) == sq;
#endif
            const unsigned long long mask = __ballot(ready);
            if (!mask) {
                if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(err, 1); bail = true; break; }
                if (CSR_SLEEP) __builtin_amdgcn_s_sleep(CSR_SLEEP);
                continue;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double uval[B];
            double *ur[B];
            if (mask == 0xffull) {
                // ---------- common case: all eight slots advance; no branches around memory ops ------
#pragma unroll
                for (int s = 0; s < B; ++s) ur[s] = U + (size_t)__builtin_amdgcn_readlane(rec.item, s) * K;
#pragma unroll
                for (int s = 0; s < B; ++s) uval[s] = ld_sc1_f64(ur[s] + lane);
                double ibv = 0.0;
                if (lane < B) ibv = ld_sc1_f64(ib + rec.item);
#pragma unroll
                for (int s = 0; s < B; ++s) prod[s * 66 + lane] = uval[s] * vval[s];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                double grad = 0.0;
                if (lane < B) {
                    double sum = 0.0 + ibv + ubv;
                    const double *row = prod + lane * 66;
#pragma unroll 8
                    for (int f = 0; f < K; ++f) sum += row[f];
                    const double e = rec.r - sum;
                    grad = e;
                    se_acc += e * e;
                    ubv = ubv + lr * (grad - Kb * ubv);
                    st_sc1_f64(ib + rec.item, ibv + lr * (grad - Kb * ibv));
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < B; ++s) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(grad), s), hi = __builtin_amdgcn_readlane(__double2hiint(grad), s);
                    const double g = __hiloint2double(hi, lo);
                    const double mf = uval[s], cf = vval[s];
                    st_sc1_f64(ur[s] + lane, mf + lr * (g * cf - Ki * mf));
                    vval[s] = cf + lr * (g * mf - Ku * cf);
                }
            } else {
                // ---------- general case: only the ready slots ------------------------------------------
#pragma unroll
                for (int s = 0; s < B; ++s) {
                    ur[s] = U + (size_t)__builtin_amdgcn_readlane(rec.item, s) * K;
                    uval[s] = 0.0;
                    if ((mask >> s) & 1) uval[s] = ld_sc1_f64(ur[s] + lane);
                }
                double ibv = 0.0;
                if (ready) ibv = ld_sc1_f64(ib + rec.item);
#pragma unroll
                for (int s = 0; s < B; ++s) prod[s * 66 + lane] = uval[s] * vval[s];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                double grad = 0.0;
                if (ready) {
                    double sum = 0.0 + ibv + ubv;
                    const double *row = prod + lane * 66;
#pragma unroll 8
                    for (int f = 0; f < K; ++f) sum += row[f];
                    const double e = rec.r - sum;
                    grad = e;
                    se_acc += e * e;
                    ubv = ubv + lr * (grad - Kb * ubv);
                    st_sc1_f64(ib + rec.item, ibv + lr * (grad - Kb * ibv));
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < B; ++s) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(grad), s), hi = __builtin_amdgcn_readlane(__double2hiint(grad), s);
                    const double g = __hiloint2double(hi, lo);
                    if ((mask >> s) & 1) {
                        const double mf = uval[s], cf = vval[s];
                        st_sc1_f64(ur[s] + lane, mf + lr * (g * cf - Ki * mf));
                        vval[s] = cf + lr * (g * mf - Ku * cf);
                    }
                }
            }
            // ---- publish: all row stores of this wave have completed, then the chain counters -------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ready) {
                st_sc1_i32(cnt_i + rec.item, sq + 1);
                ++pos;
                if (pos < end) { rec = recs[pos]; sq = seq[pos]; }
            }
            spins = 0;
        }
        if (bail) break;
        // ---- the runs are over: user rows and biases go back (nobody else reads them this epoch) ---
#pragma unroll
        for (int s = 0; s < B; ++s)
            if (b * B + s < n_runs) V[(size_t)usr[s] * K + lane] = vval[s];
        if (lane < B && b * B + lane < n_runs) ub[my_user] = ubv;
        const double wsum = wave_sum_fixed(se_acc);
        if (lane == 0) se_partial[b] = wsum;
    }
}

int main(int argc, char **argv)
{
    const int nu = 100000, ni = 50000;
    const int64_t nnz = argc > 1 ? atoll(argv[1]) : 10000000;
    const int epochs = 3;
    std::vector<int32_t> idx((size_t)nnz * 2);
    std::vector<double> r((size_t)nnz);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    {   // the order of get_ratings(randomize_order=False): user-major, ascending items inside a user
        // (argv[2] = 1: items in generation order inside a user -- a 670 K-deep dependency chain)
        const bool gen_order = argc > 2 && atoi(argv[2]) == 1;
        struct T3 { int32_t u, i; int64_t t; };
        std::vector<T3> key((size_t)nnz);
        std::vector<double> rr((size_t)nnz);
        for (int64_t t = 0; t < nnz; ++t) { key[t].u = (int32_t)(rnd() % nu); key[t].i = (int32_t)(rnd() % ni); key[t].t = t; rr[t] = 1.0 + (double)(rnd() % 5); }
        std::sort(key.begin(), key.end(), [&](const T3 &a, const T3 &b) {
            if (a.u != b.u) return a.u < b.u;
            if (!gen_order && a.i != b.i) return a.i < b.i;
            return a.t < b.t;
        });
        for (int64_t j = 0; j < nnz; ++j) { idx[2 * j] = key[j].u; idx[2 * j + 1] = key[j].i; r[j] = rr[key[j].t]; }
    }
    std::vector<int32_t> level((size_t)nnz);
    int64_t n_levels = 0;
    if (mfx_schedule_levels(idx.data(), nnz, nu, ni, level.data(), &n_levels)) return 1;
    std::vector<int64_t> perm((size_t)nnz), off((size_t)n_levels + 1);
    if (mfx_schedule_permutation(level.data(), nnz, n_levels, perm.data(), off.data())) return 1;
    std::vector<Rec> recs_lvl((size_t)nnz), recs_in((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) { const int64_t t = perm[j]; recs_lvl[j] = {idx[2 * t], idx[2 * t + 1], r[t]}; }
    std::vector<int32_t> seq((size_t)nnz), ci((size_t)ni, 0);
    std::vector<int64_t> run_off;
    for (int64_t t = 0; t < nnz; ++t) {
        recs_in[t] = {idx[2 * t], idx[2 * t + 1], r[t]};
        seq[t] = ci[idx[2 * t + 1]]++;
        if (t == 0 || idx[2 * t] != idx[2 * (t - 1)]) run_off.push_back(t);
    }
    const int n_runs = (int)run_off.size();
    run_off.push_back(nnz);
    printf("nnz %lld levels %lld runs %d\n", (long long)nnz, (long long)n_levels, n_runs);

    std::vector<double> U0((size_t)ni * K), V0((size_t)nu * K);
    for (auto &x : U0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    for (auto &x : V0) x = ((double)(rnd() % 2001) - 1000.0) * 1e-4;
    Rec *d_lvl, *d_in; int *d_seq; int64_t *d_run; double *dU[2], *dV[2], *dib[2], *dub[2], *d_se; int *d_ci, *d_err, *d_ticket;
    CK(hipMalloc(&d_lvl, nnz * sizeof(Rec))); CK(hipMalloc(&d_in, nnz * sizeof(Rec))); CK(hipMalloc(&d_seq, nnz * 4)); CK(hipMalloc(&d_run, run_off.size() * 8));
    CK(hipMemcpy(d_lvl, recs_lvl.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_in, recs_in.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_seq, seq.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_run, run_off.data(), run_off.size() * 8, hipMemcpyHostToDevice));
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&dU[v], U0.size() * 8)); CK(hipMalloc(&dV[v], V0.size() * 8)); CK(hipMalloc(&dib[v], ni * 8)); CK(hipMalloc(&dub[v], nu * 8));
        CK(hipMemcpy(dU[v], U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV[v], V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dib[v], 0, ni * 8)); CK(hipMemset(dub[v], 0, nu * 8));
    }
    CK(hipMalloc(&d_se, ((size_t)nnz / 8 + 8192) * 8)); CK(hipMalloc(&d_ci, ni * 4)); CK(hipMalloc(&d_err, 4)); CK(hipMalloc(&d_ticket, 4));
    CK(hipMemset(d_err, 0, 4));
    const double lr = 0.01, Ku = 0.1, Ki = 0.1, Kb = 0.007;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_ref = 0;
    {   // reference: one launch per level (the product kernel) on the same user-major input
        K1Params P{}; P.recs = d_lvl; P.U = dU[0]; P.V = dV[0]; P.ib = dib[0]; P.ub = dub[0]; P.se_partial = d_se; P.k = K;
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
    std::vector<double> U1(U0.size()), V1(V0.size()), a(U0.size()), c(V0.size());
    CK(hipMemcpy(a.data(), dU[0], a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), dV[0], c.size() * 8, hipMemcpyDeviceToHost));
    const int wpcs[] = {2, 4, 8, 16};
    for (int wpc : wpcs) {
        CK(hipMemcpy(dU[1], U0.data(), U0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dV[1], V0.data(), V0.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dib[1], 0, ni * 8)); CK(hipMemset(dub[1], 0, nu * 8));
        CK(hipMemset(d_err, 0, 4));
        float ms = 0;
        for (int ep = 0; ep < epochs; ++ep) {
            CK(hipMemsetAsync(d_ci, 0, ni * 4)); CK(hipMemsetAsync(d_ticket, 0, 4));
            if (ep == epochs - 1) CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k1_csr, dim3(256 * wpc), dim3(64), 0, 0, d_in, d_seq, d_run, n_runs, dU[1], dV[1], dib[1], dub[1], d_ci, d_ticket, d_se, d_err, lr, Ku, Ki, Kb);
            if (ep == epochs - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
        int err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(U1.data(), dU[1], U1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(V1.data(), dV[1], V1.size() * 8, hipMemcpyDeviceToHost));
        size_t diffU = 0, diffV = 0;
        for (size_t i = 0; i < a.size(); ++i) diffU += (memcmp(&a[i], &U1[i], 8) != 0);
        for (size_t i = 0; i < c.size(); ++i) diffV += (memcmp(&c[i], &V1[i], 8) != 0);
        printf("user-major: level-per-launch %.3f ms/epoch = %.3f G/s | CSR dataflow %d waves/CU: %.3f ms/epoch = %.3f G/s | timeout flag %d | differing U %zu V %zu\n",
               ms_ref, nnz / ms_ref / 1e6, wpc, ms, nnz / ms / 1e6, err, diffU, diffV);
        if (err) break;
    }
    return 0;
}
