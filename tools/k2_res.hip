// EXPERIMENT (round 3): one Funk per-feature pass (estimator_loop_without_bias, gd_estimator.pyx:742-777) as ONE persistent
// launch over ALL CUs with the K1 resident-side idea in its scalar form.  Items are resident: ONE LANE owns one item, keeps
// U[f,item] in a register and walks the item's ratings in input order.  The user side travels, and since its payload is ONE
// scalar it fits the hand-off granule itself: {V[f,user], chain counter} in 16 bytes -- the poll RETURNS the value (no row
// load), the publish is ONE 16-byte sc1 store that carries value and counter together (no store-acknowledge wait): a hop is
// a poll round trip, not a level barrier.  Checked bit for bit against the product's k2_level (one launch per level).
//
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w tools/k2_res.hip -o /tmp/k2_res
#include <algorithm>
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

struct __attribute__((aligned(16))) G2 { double val; int cnt; int pad; };
struct __attribute__((aligned(16))) R2 { int user; int seq; double r; };
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u ld16c(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ __forceinline__ void st16c(v4u v, __amdgpu_buffer_rsrc_t r, unsigned off) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }

// RES_ITEM: lanes own items (user granules travel) or users (item granules travel); PRE: the next record + cache word are
// requested one rating ahead, so that the next poll can go out at once
template <bool RES_ITEM, bool PRE, int NAP = 0>
__global__ __launch_bounds__(64) void k2_res(const R2 *__restrict__ recs, const double *__restrict__ cache, const int *__restrict__ lo,
                                             const int *__restrict__ hi, int n_res, double *rf, G2 *hdr, double trail, double lr,
                                             double K, double mu, double *se_ent, int *err)
{
    const int ent = blockIdx.x * 64 + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(hdr, 0, 0x7fffffff, 0x27000);
    int pos = 0, end = 0;
    double x = 0.0, se = 0.0;
    if (ent < n_res) { pos = lo[ent]; end = hi[ent]; x = rf[ent]; }
    R2 rec{0, 0, 0.0}, nrec{0, 0, 0.0};
    double c = 0.0, nc = 0.0;
    if (pos < end) { rec = recs[pos]; c = cache[pos]; }
    if (PRE && pos + 1 < end) { nrec = recs[pos + 1]; nc = cache[pos + 1]; }
    int spins = 0;
    while (__any(pos < end)) {
        if (NAP) __builtin_amdgcn_s_sleep(NAP);
        bool ready = false;
        G2 g{0.0, -1, 0};
        if (pos < end) {
            g = __builtin_bit_cast(G2, ld16c(rH, (unsigned)rec.user * 16u));
            ready = g.cnt == rec.seq;
        }
        if (ready) {
            const double mf = RES_ITEM ? x : g.val, cf = RES_ITEM ? g.val : x;      // item value, user value
            const double base = mu + 0.0 + 0.0;
            const double p = funk_estimate(mf, cf, c, base, trail, true);
            const double e = rec.r - p;
            se += e * e;
            const double un = mf + lr * (e * cf - K * mf), vn = cf + lr * (e * mf - K * cf);
            G2 n; n.val = RES_ITEM ? vn : un; n.cnt = rec.seq + 1; n.pad = 0;
            st16c(__builtin_bit_cast(v4u, n), rH, (unsigned)rec.user * 16u);
            x = RES_ITEM ? un : vn;
            ++pos;
            if (PRE) {
                rec = nrec; c = nc;
                if (pos + 1 < end) { nrec = recs[pos + 1]; nc = cache[pos + 1]; }
            } else if (pos < end) { rec = recs[pos]; c = cache[pos]; }
        }
        if (!__any(ready)) {
            if (++spins > (1 << 20)) { if (threadIdx.x == 0) atomicOr(err, 1); break; }
        } else spins = 0;
    }
    if (ent < n_res) { rf[ent] = x; se_ent[ent] = se; }
}

__global__ void k_hdr_reset(G2 *hdr, const double *vf, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { G2 g; g.val = vf[i]; g.cnt = 0; g.pad = 0; hdr[i] = g; }
}
__global__ void k_hdr_finish(const G2 *hdr, double *vf, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vf[i] = hdr[i].val;
}

int main(int argc, char **argv)
{
    const int nu = 100000, ni = 50000;
    const int64_t nnz = 10000000;
    const int balanced = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<int32_t> idx((size_t)nnz * 2);
    std::vector<double> r((size_t)nnz);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int64_t t = 0; t < nnz; ++t) { idx[2 * t] = (int32_t)(rnd() % nu); idx[2 * t + 1] = (int32_t)(rnd() % ni); r[t] = 1.0 + (double)(rnd() % 5); }
    (void)balanced;
    std::vector<int32_t> level((size_t)nnz);
    int64_t n_levels = 0;
    if (mfx_schedule_levels(idx.data(), nnz, nu, ni, level.data(), &n_levels)) return 1;
    std::vector<int64_t> perm((size_t)nnz), off((size_t)n_levels + 1);
    if (mfx_schedule_permutation(level.data(), nnz, n_levels, perm.data(), off.data())) return 1;
    std::vector<Rec> recs_lvl((size_t)nnz);
    for (int64_t j = 0; j < nnz; ++j) { const int64_t t = perm[j]; recs_lvl[j] = {idx[2 * t], idx[2 * t + 1], r[t]}; }
    // grouped by the resident side (stable), ranked in the travelling side's chains; R2.user = the TRAVELLING entity
    auto group = [&](bool res_item, std::vector<R2> &recs, std::vector<int> &lo, std::vector<int> &hi) {
        const int n_res = res_item ? ni : nu, n_trav = res_item ? nu : ni, rc = res_item ? 1 : 0, tc = res_item ? 0 : 1;
        std::vector<int> cnt((size_t)n_res + 1, 0), ctr((size_t)n_trav, 0);
        for (int64_t t = 0; t < nnz; ++t) ++cnt[idx[2 * t + rc] + 1];
        for (int i = 0; i < n_res; ++i) cnt[i + 1] += cnt[i];
        std::vector<int> fill(cnt.begin(), cnt.end() - 1);
        recs.resize((size_t)nnz); lo.assign((size_t)n_res, 0); hi.assign((size_t)n_res, 0);
        for (int64_t t = 0; t < nnz; ++t) { const int tr = idx[2 * t + tc]; recs[(size_t)fill[idx[2 * t + rc]]++] = R2{tr, ctr[tr]++, r[t]}; }
        for (int i = 0; i < n_res; ++i) { lo[i] = cnt[i]; hi[i] = cnt[i + 1]; }
    };
    std::vector<R2> recs; std::vector<int> lo, hi;
    group(true, recs, lo, hi);
    printf("cfg 2 shuffled: %lld levels\n", (long long)n_levels);

    const double lr = 0.001, K = 0.05, init = 0.1, trail = 63 * init * init, mu = 1.0;
    Rec *d_lvl; R2 *d_recs; double *d_cache, *d_uf[2], *d_vf[2], *d_se, *d_cache_l; int *d_lo, *d_hi, *d_err; G2 *d_hdr; K2Dyn *d_dyn;
    CK(hipMalloc(&d_lvl, nnz * sizeof(Rec))); CK(hipMalloc(&d_recs, nnz * sizeof(R2))); CK(hipMalloc(&d_cache, nnz * 8)); CK(hipMalloc(&d_cache_l, nnz * 8));
    CK(hipMemcpy(d_lvl, recs_lvl.data(), nnz * sizeof(Rec), hipMemcpyHostToDevice)); CK(hipMemcpy(d_recs, recs.data(), nnz * sizeof(R2), hipMemcpyHostToDevice));
    CK(hipMemset(d_cache, 0, nnz * 8)); CK(hipMemset(d_cache_l, 0, nnz * 8));
    CK(hipMalloc(&d_lo, nu * 4)); CK(hipMalloc(&d_hi, nu * 4));
    std::vector<double> u0((size_t)ni, init), v0((size_t)nu, init);
    for (int v = 0; v < 2; ++v) { CK(hipMalloc(&d_uf[v], ni * 8)); CK(hipMalloc(&d_vf[v], nu * 8)); CK(hipMemcpy(d_uf[v], u0.data(), ni * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vf[v], v0.data(), nu * 8, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&d_se, (nnz / 64 + ni + 65536) * 8)); CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4)); CK(hipMalloc(&d_hdr, (size_t)nu * 16)); CK(hipMalloc(&d_dyn, sizeof(K2Dyn)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int E = 3;
    float ms_ref = 0;
    {   // reference: k2_level, one launch per level, E epochs of one feature
        K2Dyn dyn; dyn.uf = d_uf[0]; dyn.vf = d_vf[0]; dyn.trail = trail;
        CK(hipMemcpy(d_dyn, &dyn, sizeof(dyn), hipMemcpyHostToDevice));
        K2Params P{}; P.recs = d_lvl; P.dyn = d_dyn; P.cache = d_cache_l; P.se_partial = d_se; P.lr = lr; P.K = K; P.mu = mu; P.upd_u = P.upd_i = 1;
        for (int ep = 0; ep < E; ++ep) {
            if (ep == E - 1) CK(hipEventRecord(e0));
            for (int64_t l = 0; l < n_levels; ++l) {
                const int c = (int)(off[l + 1] - off[l]);
                hipLaunchKernelGGL(k2_level<true>, dim3((c + kK2Block - 1) / kK2Block), dim3(kK2Block), 0, 0, P, off[l], c, (int64_t)0);
            }
            if (ep == E - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
    }
    std::vector<double> ua((size_t)ni), va((size_t)nu), ub((size_t)ni), vb((size_t)nu);
    CK(hipMemcpy(ua.data(), d_uf[0], ni * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(va.data(), d_vf[0], nu * 8, hipMemcpyDeviceToHost));
    for (int var = 0; var < 5; ++var) {
        const bool res_item = true, pre = var & 1;
        const int n_res = res_item ? ni : nu, n_trav = res_item ? nu : ni;
        if (var == 0) {
            group(res_item, recs, lo, hi);
            CK(hipMemcpy(d_recs, recs.data(), nnz * sizeof(R2), hipMemcpyHostToDevice));
            CK(hipMemcpy(d_lo, lo.data(), n_res * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_hi, hi.data(), n_res * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemcpy(d_uf[1], u0.data(), ni * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vf[1], v0.data(), nu * 8, hipMemcpyHostToDevice));
        double *rf = res_item ? d_uf[1] : d_vf[1], *tf = res_item ? d_vf[1] : d_uf[1];
        float ms = 0;
        for (int ep = 0; ep < E; ++ep) {
            if (ep == E - 1) CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_hdr_reset, dim3((n_trav + 255) / 256), dim3(256), 0, 0, d_hdr, tf, n_trav);
            const dim3 grid((n_res + 63) / 64);
            if (var == 0) hipLaunchKernelGGL((k2_res<true, false>), grid, dim3(64), 0, 0, d_recs, d_cache, d_lo, d_hi, n_res, rf, d_hdr, trail, lr, K, mu, d_se, d_err);
            else if (var == 1) hipLaunchKernelGGL((k2_res<true, true>), grid, dim3(64), 0, 0, d_recs, d_cache, d_lo, d_hi, n_res, rf, d_hdr, trail, lr, K, mu, d_se, d_err);
            else if (var == 2) hipLaunchKernelGGL((k2_res<true, false, 2>), grid, dim3(64), 0, 0, d_recs, d_cache, d_lo, d_hi, n_res, rf, d_hdr, trail, lr, K, mu, d_se, d_err);
            else if (var == 3) hipLaunchKernelGGL((k2_res<true, false, 6>), grid, dim3(64), 0, 0, d_recs, d_cache, d_lo, d_hi, n_res, rf, d_hdr, trail, lr, K, mu, d_se, d_err);
            else hipLaunchKernelGGL((k2_res<true, false, 16>), grid, dim3(64), 0, 0, d_recs, d_cache, d_lo, d_hi, n_res, rf, d_hdr, trail, lr, K, mu, d_se, d_err);
            hipLaunchKernelGGL(k_hdr_finish, dim3((n_trav + 255) / 256), dim3(256), 0, 0, d_hdr, tf, n_trav);
            if (ep == E - 1) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
        }
        CK(hipDeviceSynchronize());
        int err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ub.data(), d_uf[1], ni * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(vb.data(), d_vf[1], nu * 8, hipMemcpyDeviceToHost));
        size_t du = 0, dv = 0;
        for (int i = 0; i < ni; ++i) du += memcmp(&ua[i], &ub[i], 8) != 0;
        for (int i = 0; i < nu; ++i) dv += memcmp(&va[i], &vb[i], 8) != 0;
        printf("%s resident, variant %d: pass %.3f ms = %.2f G scalar-updates/s (one launch per level: %.3f ms) | timeout %d | differing U %zu V %zu\n",
               res_item ? "items" : "users", var, ms, nnz / ms / 1e6, ms_ref, err, du, dv);
    }
    return 0;
}
