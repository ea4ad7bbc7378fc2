// estimator_loop_with_implicit_feedback (gd_estimator.pyx:785-898): the reference's SVD++-like prototype,
// reproduced AS WRITTEN (its `feedback_sum` is overwritten by y[f, item] at :873; its inner loops reuse the
// loop variable, harmlessly; the cache commit ignores the implicit term).
//
// This loop does not parallelise over ratings: every rating rewrites y[f, j] for ALL items j its user rated
// (:879-881) and reads y[f, item], so two ratings conflict whenever their users share any item -- with ~100
// items per user the conflict-free sets hold a few dozen ratings and an exact schedule degenerates to the
// sequential loop.  What is left is the parallelism INSIDE one rating (the ~deg(user)
// independent y updates).  So: ONE workgroup walks the ratings in order, every thread evaluates the (uniform)
// scalar part, thread 0 applies the u / v updates, all threads share the y updates; two workgroup barriers per
// rating.  The three feature vectors live in LDS when they fit (2 ni + nu doubles <= 150 KB: MovieLens-100K
// does), else in global memory behind sc1 loads.  A legal call is served with the reference's exact bits;
// it is NOT fast and cannot be: 0.52-0.57 us per rating with the vectors in LDS (0.61 before 1 / sqrt(count) became a
// per-user table; a single wave without barriers: 0.66-0.87) against 0.07-0.09 us for the
// reference's own C loop on one host core (tools/svdpp_timing.py: MovieLens-100K / -1M shapes) -- a chain of
// two workgroup barriers per rating is what a GPU has to offer a sequential loop.  Documented as such.
#include <math.h>
#include <vector>
#include "mfx_common.h"

namespace {

struct SParams {
    const int2 *idx;           // (user, item) in the given order
    const double *r;
    const int32_t *rated_item; // [n_rated]: column 1 of `rated`
    const int2 *hash;          // [nu]: (first row of the user in `rated`, count)
    const double *ib, *ub;
    const double *norm;        // [nu]: 1 / sqrt(count) (gd_estimator.pyx:853-854), formed once per call
    double *uf, *vf, *yf;      // U[f,:], V[f,:], y[f,:] (feature-major rows of the host layout)
    double *cache;             // [nnz]
    double *se_out;
    int64_t nnz;
    int nu, ni;
    double lr, K, mu, trail;
};

__device__ __forceinline__ double clamp15s(double x)
{
    if (x > 5.0) x = 5.0;                        // gd_estimator.pyx:26-35
    if (x < 1.0) x = 1.0;
    return x;
}
__device__ __forceinline__ double estimate(double uf, double vf, double cache, double base, double trail, bool trailing)
{
    double sum = (cache > 0) ? cache : base;     // gd_estimator.pyx:38-73
    sum += uf * vf;
    sum = clamp15s(sum);
    if (trailing) {
        sum += trail;
        sum = clamp15s(sum);
    }
    return sum;
}
__device__ __forceinline__ double ld_l2(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1: L2, never a stale L1 line
}

constexpr int kSThreads = 256;          // one wave per SIMD: every wave evaluates the uniform scalar part (sqrt, division,
                                       // estimate) without fighting for the VALU; 1 024 threads measured 1.1 us per rating

template <bool LDS>
__global__ __launch_bounds__(kSThreads) void k_svdpp_epoch(SParams P)
{
    extern __shared__ double sh[];               // LDS: [uf (ni) | vf (nu) | yf (ni)]
    const int tid = threadIdx.x;
    double *uf = P.uf, *vf = P.vf, *yf = P.yf;
    if constexpr (LDS) {
        for (int i = tid; i < P.ni; i += kSThreads) { sh[i] = P.uf[i]; sh[P.ni + P.nu + i] = P.yf[i]; }
        for (int i = tid; i < P.nu; i += kSThreads) sh[P.ni + i] = P.vf[i];
        uf = sh; vf = sh + P.ni; yf = sh + P.ni + P.nu;
        __syncthreads();
    }
    double se = 0.0;
    // Everything a rating needs that no update touches (its record, cache word, biases, the user's slice of
    // `rated`) is requested one rating ahead -- the record itself two ahead, because the rest hangs off its
    // ids -- so that a step costs the two barriers and the LDS traffic, not a chain of three global loads;
    // the user's 1 / sqrt(count) is formed there too (same bits, off the chain).
    struct Step {
        int2 ui, h;
        double r, c, base, norm;
        int jj;                                  // this thread's first item of the user's rated list (-1: none)
    };
    auto load_step = [&](int64_t t, int2 ui, Step &x) {
        x.ui = ui; x.h = make_int2(0, 0); x.r = 0.0; x.c = 0.0; x.base = 0.0; x.norm = 0.0; x.jj = -1;
        if (t < P.nnz) {
            x.h = P.hash[ui.x];
            x.r = P.r[t];
            x.c = P.cache[t];
            x.base = P.mu + P.ib[ui.y] + P.ub[ui.x];
            x.norm = P.norm[ui.x];                                           // :853-854: same operations, once per user
            if (tid < x.h.y) x.jj = P.rated_item[x.h.x + tid];
        }
    };
    Step cur, nxt;
    int2 ui2 = make_int2(0, 0);                  // record of rating t + 2
    load_step(0, P.nnz > 0 ? P.idx[0] : make_int2(0, 0), cur);
    load_step(1, P.nnz > 1 ? P.idx[1] : make_int2(0, 0), nxt);
    if (P.nnz > 2) ui2 = P.idx[2];
    for (int64_t t = 0; t < P.nnz; ++t) {
        const int2 ui = cur.ui, h = cur.h;
        const double norm = cur.norm;
        double cf, fb, mf;
        if constexpr (LDS) { cf = vf[ui.x]; fb = yf[ui.y]; mf = uf[ui.y]; }
        else { cf = ld_l2(vf + ui.x); fb = ld_l2(yf + ui.y); mf = ld_l2(uf + ui.y); }
        const double v_implicit = cf + (norm * fb);                          // :855
        const double p = estimate(mf, v_implicit, cur.c, cur.base, P.trail, true);
        const double err = 1.0 * cur.r - p;                                  // :860
        __syncthreads();                         // every thread has read y[f,item], u, v: now they may change
        if (tid == 0) {
            se += err * err;
            uf[ui.y] = mf + P.lr * (err * (cf + norm * fb) - P.K * mf);     // :873 + :876 (feedback_sum = y[f,item])
            vf[ui.x] = cf + P.lr * (err * mf - P.K * cf);                    // :877
        }
        for (int j = tid; j < h.y; j += kSThreads) {                         // :879-881, the items are distinct
            const int jj = (j == tid) ? cur.jj : P.rated_item[h.x + j];
            double yv;
            if constexpr (LDS) yv = yf[jj]; else yv = ld_l2(yf + jj);
            yf[jj] = yv + P.lr * (err * norm * mf - P.K * yv);
        }
        if constexpr (!LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores are in L2
        cur = nxt;
        load_step(t + 2, ui2, nxt);
        if (t + 3 < P.nnz) ui2 = P.idx[t + 3];
        __syncthreads();                         // the updates are visible to the next rating
    }
    if constexpr (LDS) {
        for (int i = tid; i < P.ni; i += kSThreads) { P.uf[i] = sh[i]; P.yf[i] = sh[P.ni + P.nu + i]; }
        for (int i = tid; i < P.nu; i += kSThreads) P.vf[i] = sh[P.ni + i];
    }
    if (tid == 0) *P.se_out = se;
}

// 1 / sqrt(count) per user: the reference forms it for every rating (:853-854); it only depends on the user, and the
// division + square root cost ~60 instructions of every wave's in-order stream per rating
__global__ __launch_bounds__(256) void k_svdpp_norm(const int2 *hash, int nu, double *norm)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u < nu) norm[u] = hash[u].y > 0 ? 1.0 / sqrt((double)hash[u].y) : 0.0;
}

// cache commit (:891-896): no implicit term, no trailing term
__global__ __launch_bounds__(256) void k_svdpp_commit(SParams P)
{
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < P.nnz; t += (int64_t)gridDim.x * 256) {
        const int2 ui = P.idx[t];
        P.cache[t] = estimate(P.uf[ui.y], P.vf[ui.x], P.cache[t], P.mu + P.ib[ui.y] + P.ub[ui.x], 0.0, false);
    }
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    int get(size_t bytes)
    {
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) {
            (void)hipGetLastError();
            return mfx::fail(MFX_E_NOMEM, "mfx_funk_implicit: out of device memory (%zu bytes)", bytes);
        }
        return MFX_OK;
    }
    template <typename T> T *as() { return (T *)p; }
};

}  // namespace

extern "C" int mfx_funk_implicit(int min_epochs, double min_improvement, int dim, double f_init, double lr, double K,
                                 double mu, double *u, double *v, double *y, const int32_t *idx, const double *r,
                                 const int32_t *rated, int64_t n_rated, const int32_t *rated_hash, const double *ib,
                                 const double *ub, int64_t nnz, int nu, int ni, int epoch_cap, double *rmse_hist,
                                 int64_t hist_cap, int64_t *n_hist)
{
    mfx::TraceRange trace_("mfx:funk_implicit (sequential SVD++-like loop)");
    if (!u || !v || !y || !ib || !ub || !rated_hash || dim < 1 || nu <= 0 || ni <= 0 || nnz < 0 || n_rated < 0 ||
        (nnz > 0 && (!idx || !r)) || (n_rated > 0 && !rated))
        return mfx::fail(MFX_E_ARG, "mfx_funk_implicit: bad argument");
    if (n_hist) *n_hist = 0;
    int rc = mfx::ensure_init();
    if (rc) return rc;
    // validate what the reference would index blindly (boundscheck(False))
    for (int64_t t = 0; t < nnz; ++t) {
        const int a = idx[2 * t], b = idx[2 * t + 1];
        if (a < 0 || a >= nu || b < 0 || b >= ni)
            return mfx::fail(MFX_E_RANGE, "mfx_funk_implicit: rating %lld: (user %d, item %d) outside [0,%d) x [0,%d)",
                             (long long)t, a, b, nu, ni);
        const int64_t s = rated_hash[2 * a], c = rated_hash[2 * a + 1];
        if (c < 1 || s < 0 || s + c > n_rated)
            return mfx::fail(MFX_E_RANGE, "mfx_funk_implicit: rated_hash[%d] = (%lld, %lld) does not index `rated` (%lld rows)",
                             a, (long long)s, (long long)c, (long long)n_rated);
    }
    std::vector<int32_t> ritem((size_t)(n_rated ? n_rated : 1));
    for (int64_t j = 0; j < n_rated; ++j) {
        ritem[(size_t)j] = rated[2 * j + 1];
        if (ritem[(size_t)j] < 0 || ritem[(size_t)j] >= ni)
            return mfx::fail(MFX_E_RANGE, "mfx_funk_implicit: rated[%lld] item %d outside [0,%d)", (long long)j, ritem[(size_t)j], ni);
    }
    if (nnz == 0) {                                  // empty epochs, nothing changes (as mfx_funk)
        int per = min_epochs > 0 ? min_epochs : 0;
        if (epoch_cap > 0 && per > epoch_cap) per = epoch_cap;
        const int64_t n = (int64_t)per * dim;
        if (rmse_hist) for (int64_t i = 0; i < n && i < hist_cap; ++i) rmse_hist[i] = NAN;
        if (n_hist) *n_hist = n;
        return MFX_OK;
    }
    Buf dU, dV, dY, dIdx, dR, dRi, dHash, dIb, dUb, dCache, dSe, dNorm;
    const size_t su = (size_t)dim * ni * 8, sv = (size_t)dim * nu * 8;
    if ((rc = dU.get(su)) || (rc = dV.get(sv)) || (rc = dY.get(su)) || (rc = dIdx.get((size_t)nnz * 8)) ||
        (rc = dR.get((size_t)nnz * 8)) || (rc = dRi.get((size_t)n_rated * 4)) || (rc = dHash.get((size_t)nu * 8)) ||
        (rc = dIb.get((size_t)ni * 8)) || (rc = dUb.get((size_t)nu * 8)) || (rc = dCache.get((size_t)nnz * 8)) ||
        (rc = dSe.get(8)) || (rc = dNorm.get((size_t)nu * 8)))
        return rc;
    hipStream_t s = mfx::stream();
    MFX_HIP(hipMemcpy(dU.p, u, su, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dV.p, v, sv, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dY.p, y, su, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dIdx.p, idx, (size_t)nnz * 8, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dR.p, r, (size_t)nnz * 8, hipMemcpyHostToDevice));
    if (n_rated) MFX_HIP(hipMemcpy(dRi.p, ritem.data(), (size_t)n_rated * 4, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dHash.p, rated_hash, (size_t)nu * 8, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dIb.p, ib, (size_t)ni * 8, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dUb.p, ub, (size_t)nu * 8, hipMemcpyHostToDevice));
    MFX_HIP(hipMemsetAsync(dCache.p, 0, (size_t)nnz * 8, s));
    SParams P;
    P.idx = dIdx.as<int2>(); P.r = dR.as<double>(); P.rated_item = dRi.as<int32_t>(); P.hash = dHash.as<int2>();
    P.ib = dIb.as<double>(); P.ub = dUb.as<double>(); P.cache = dCache.as<double>(); P.se_out = dSe.as<double>();
    P.nnz = nnz; P.nu = nu; P.ni = ni; P.lr = lr; P.K = K; P.mu = mu;
    P.norm = dNorm.as<double>();
    hipLaunchKernelGGL(k_svdpp_norm, dim3((nu + 255) / 256), dim3(256), 0, s, P.hash, nu, dNorm.as<double>());
    MFX_HIP(hipGetLastError());
    const size_t lds = ((size_t)2 * ni + nu) * sizeof(double);
    const bool use_lds = lds <= 150 * 1024;
    if (use_lds && lds > 64 * 1024)
        MFX_HIP(hipFuncSetAttribute((const void *)k_svdpp_epoch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double rmse = 2.0, rmse_last = 0.0;
    int64_t nh = 0;
    const int commit_blocks = (int)((nnz + 255) / 256 < 4096 ? (nnz + 255) / 256 : 4096);
    for (int f = 0; f < dim; ++f) {
        P.uf = dU.as<double>() + (size_t)f * ni;
        P.vf = dV.as<double>() + (size_t)f * nu;
        P.yf = dY.as<double>() + (size_t)f * ni;
        P.trail = (dim - f - 1) * f_init * f_init;                       // gd_estimator.pyx:70
        int epoch = 0;
        while (epoch < min_epochs || rmse <= rmse_last - min_improvement) {   // :844 (max_epochs ignored)
            if (epoch_cap > 0 && epoch >= epoch_cap) break;
            rmse_last = rmse;
            if (use_lds) hipLaunchKernelGGL(k_svdpp_epoch<true>, dim3(1), dim3(kSThreads), lds, s, P);
            else hipLaunchKernelGGL(k_svdpp_epoch<false>, dim3(1), dim3(kSThreads), 0, s, P);
            MFX_HIP(hipGetLastError());
            double se = 0.0;
            MFX_HIP(hipMemcpyAsync(&se, dSe.p, 8, hipMemcpyDeviceToHost, s));
            MFX_HIP(hipStreamSynchronize(s));
            rmse = sqrt(se / (double)nnz);
            if (rmse_hist && nh < hist_cap) rmse_hist[nh] = rmse;
            ++nh;
            ++epoch;
        }
        hipLaunchKernelGGL(k_svdpp_commit, dim3(commit_blocks), dim3(256), 0, s, P);
        MFX_HIP(hipGetLastError());
    }
    MFX_HIP(hipStreamSynchronize(s));
    MFX_HIP(hipMemcpy(u, dU.p, su, hipMemcpyDeviceToHost));
    MFX_HIP(hipMemcpy(v, dV.p, sv, hipMemcpyDeviceToHost));
    MFX_HIP(hipMemcpy(y, dY.p, su, hipMemcpyDeviceToHost));
    if (n_hist) *n_hist = nh;
    return MFX_OK;
}
