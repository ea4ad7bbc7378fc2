// Batched predict + RMSE over rating triples: the step right after train()
// (mfrec/recommendation/metrics.py:51-82 calling the family's predict()).
#include <math.h>
#include <vector>
#include "mfx_common.h"

namespace {

// one wave per 64 triples; lane t computes triple t's dot sequentially over f from the
// feature-major arrays (reads of one feature across 64 triples are independent gathers).
__global__ __launch_bounds__(256) void k_predict(int kind, int k, const double *u, const double *v,
                                                 const int32_t *idx, const double *r,
                                                 const double *ib, const double *ub, double mu,
                                                 int64_t nnz, int nu, int ni, double *pred,
                                                 double *se_partial)
{
    __shared__ double sh[256];
    double e2 = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * 256) {
        const int user = idx[2 * t], item = idx[2 * t + 1];
        double dot = 0.0;
        for (int f = 0; f < k; ++f) dot += u[(size_t)f * ni + item] * v[(size_t)f * nu + user];
        double p;
        switch (kind) {
        case 0: p = dot + 1.0; break;                                              // gradient_descent.py:629
        case 1: p = dot + (ib[item] + ub[user]); break;                            // kmf.py:92-94
        case 2: p = 1.0 + (1.0 / (1.0 + exp(-(dot + (ib[item] + ub[user]))))) * 4.0; break;  // kmf.py:83-85
        case 3: p = dot + mu + (ib[item] + ub[user]); break;                       // gradient_descent.py:645-646
        default: p = dot; break;                                                   // wrmf.py:68
        }
        if (pred) pred[t] = p;
        const double e = r[t] - p;
        e2 += e * e;
    }
    sh[threadIdx.x] = e2;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) se_partial[blockIdx.x] = sh[0];
}

}  // namespace

extern "C" int mfx_predict_rmse(int kind, int dim, const double *u, const double *v,
                                const int32_t *idx, const double *r, const double *ib,
                                const double *ub, double mu, int64_t nnz, int nu, int ni,
                                double *pred_out, double *rmse_out)
{
    if (!u || !v || dim < 1 || nu <= 0 || ni <= 0 || nnz < 0 || (nnz > 0 && (!idx || !r)) ||
        kind < 0 || kind > 4 || ((kind >= 1 && kind <= 3) && (!ib || !ub)))
        return mfx::fail(MFX_E_ARG, "mfx_predict_rmse: bad argument");
    int rc = mfx::ensure_init();
    if (rc) return rc;
    for (int64_t t = 0; t < nnz; ++t)
        if (idx[2 * t] < 0 || idx[2 * t] >= nu || idx[2 * t + 1] < 0 || idx[2 * t + 1] >= ni)
            return mfx::fail(MFX_E_RANGE, "triple %lld: (user %d, item %d) out of range",
                             (long long)t, idx[2 * t], idx[2 * t + 1]);
    if (nnz == 0) {
        if (rmse_out) *rmse_out = NAN;
        return MFX_OK;
    }
    const int blocks = (int)((nnz + 255) / 256 < 2048 ? (nnz + 255) / 256 : 2048);
    double *du = nullptr, *dv = nullptr, *dr = nullptr, *dib = nullptr, *dub = nullptr, *dp = nullptr,
           *dse = nullptr;
    int32_t *di = nullptr;
    auto freeall = [&]() {
        void *q[] = {du, dv, dr, dib, dub, dp, dse, di};
        for (void *x : q) if (x) (void)hipFree(x);
    };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { freeall(); \
        return mfx::fail(MFX_E_HIP, "mfx_predict_rmse: %s -> %s", #call, hipGetErrorString(e_)); } } while (0)
    const size_t su = (size_t)dim * ni * 8, sv = (size_t)dim * nu * 8;
    TRY(hipMalloc((void **)&du, su)); TRY(hipMalloc((void **)&dv, sv));
    TRY(hipMalloc((void **)&di, (size_t)nnz * 8)); TRY(hipMalloc((void **)&dr, (size_t)nnz * 8));
    TRY(hipMalloc((void **)&dse, (size_t)blocks * 8));
    TRY(hipMemcpy(du, u, su, hipMemcpyHostToDevice)); TRY(hipMemcpy(dv, v, sv, hipMemcpyHostToDevice));
    TRY(hipMemcpy(di, idx, (size_t)nnz * 8, hipMemcpyHostToDevice));
    TRY(hipMemcpy(dr, r, (size_t)nnz * 8, hipMemcpyHostToDevice));
    if (ib) { TRY(hipMalloc((void **)&dib, (size_t)ni * 8)); TRY(hipMemcpy(dib, ib, (size_t)ni * 8, hipMemcpyHostToDevice)); }
    if (ub) { TRY(hipMalloc((void **)&dub, (size_t)nu * 8)); TRY(hipMemcpy(dub, ub, (size_t)nu * 8, hipMemcpyHostToDevice)); }
    if (pred_out) TRY(hipMalloc((void **)&dp, (size_t)nnz * 8));
    hipLaunchKernelGGL(k_predict, dim3(blocks), dim3(256), 0, mfx::stream(), kind, dim, du, dv, di,
                       dr, dib, dub, mu, nnz, nu, ni, dp, dse);
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(mfx::stream()));
    std::vector<double> se((size_t)blocks);
    TRY(hipMemcpy(se.data(), dse, (size_t)blocks * 8, hipMemcpyDeviceToHost));
    if (pred_out) TRY(hipMemcpy(pred_out, dp, (size_t)nnz * 8, hipMemcpyDeviceToHost));
#undef TRY
    freeall();
    double tot = 0.0;
    for (double x : se) tot += x;
    if (rmse_out) *rmse_out = sqrt(tot / (double)nnz);
    return MFX_OK;
}
