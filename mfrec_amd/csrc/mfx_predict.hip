// Batched predict + RMSE over rating triples: the step right after train()
// (mfrec/recommendation/metrics.py:51-82 calling the family's predict()).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <atomic>
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

// [k][n] feature-major -> [n][kp] row-major, rows padded to an even number of features (16-byte row reads)
__global__ __launch_bounds__(256) void k_to_rows(const double *in, double *out, int k, int n, int kp)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int tiles_n = (n + 31) / 32;
    const int f0 = (blockIdx.x / tiles_n) * 32, i0 = (blockIdx.x % tiles_n) * 32;
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, i = i0 + tx;
        tile[r][tx] = (f < k && i < n) ? in[(size_t)f * n + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, f = f0 + tx;
        if (i < n && f < kp) out[(size_t)i * kp + f] = tile[tx][r];
    }
}

// The same prediction from ROW-major factors: a lane walks its two rows 16 bytes at a time (eight loads per 128-byte
// line instead of one 8-byte gather per feature from a different line each): 10 M triples at k = 64 in ~5 ms instead of
// 19.  The dot product is the same left-to-right sum, bit for bit.
__global__ __launch_bounds__(256) void k_predict_rows(int kind, int k, int kp, const double *ut, const double *vt,
                                                      const int32_t *idx, const double *r, const double *ib,
                                                      const double *ub, double mu, int64_t nnz, double *pred,
                                                      double *se_partial)
{
    __shared__ double sh[256];
    double e2 = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * 256) {
        const int user = idx[2 * t], item = idx[2 * t + 1];
        const double2 *pu = reinterpret_cast<const double2 *>(ut + (size_t)item * kp);
        const double2 *pv = reinterpret_cast<const double2 *>(vt + (size_t)user * kp);
        double dot = 0.0;
        int f = 0;
        for (; f + 2 <= k; f += 2) {
            const double2 a = pu[f >> 1], b = pv[f >> 1];
            dot += a.x * b.x;
            dot += a.y * b.y;
        }
        if (f < k) dot += pu[f >> 1].x * pv[f >> 1].x;
        double p;
        switch (kind) {
        case 0: p = dot + 1.0; break;
        case 1: p = dot + (ib[item] + ub[user]); break;
        case 2: p = 1.0 + (1.0 / (1.0 + exp(-(dot + (ib[item] + ub[user]))))) * 4.0; break;
        case 3: p = dot + mu + (ib[item] + ub[user]); break;
        default: p = dot; break;
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
    double *dut = nullptr, *dvt = nullptr;          // row-major copies of the factors (many triples)
    auto freeall = [&]() {
        void *q[] = {du, dv, dr, dib, dub, dp, dse, di, dut, dvt};
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
    // many triples: one pass to row-major factors pays for itself (each row is then read as a few whole lines)
    const char *pk = getenv("MFX_PREDICT_ROWS");
    const bool rows = pk ? pk[0] == '1' : nnz >= 4 * ((int64_t)nu + ni);
    if (rows) {
        const int kp = (dim + 1) & ~1;
        if (hipMalloc((void **)&dut, (size_t)ni * kp * 8) != hipSuccess || hipMalloc((void **)&dvt, (size_t)nu * kp * 8) != hipSuccess) {
            (void)hipGetLastError();
            if (dut) (void)hipFree(dut);
            if (dvt) (void)hipFree(dvt);
            dut = dvt = nullptr;
        }
        if (dut) {
            const int tf = (kp + 31) / 32;
            hipLaunchKernelGGL(k_to_rows, dim3(tf * ((ni + 31) / 32)), dim3(256), 0, mfx::stream(), du, dut, dim, ni, kp);
            hipLaunchKernelGGL(k_to_rows, dim3(tf * ((nu + 31) / 32)), dim3(256), 0, mfx::stream(), dv, dvt, dim, nu, kp);
            hipLaunchKernelGGL(k_predict_rows, dim3(blocks), dim3(256), 0, mfx::stream(), kind, dim, kp, dut, dvt, di, dr, dib,
                               dub, mu, nnz, dp, dse);
        }
    }
    if (!dut)
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


// ------------------------------------------------------------------------------------------------
// Batched top-N (the ranking of mf.py:144-193 / gradient_descent.py:768-800 for MANY users at once):
// scores of items 0 .. n_cand-1 for every user of the batch, the reference's exclusions (the user's rated
// items, the item whose id equals the user id, NaN -> 0, zero scores dropped), descending score with ties
// by ascending item.  The dot product is summed feature by feature exactly as k_predict does, so a batch
// row carries the same bits as the single-user query.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kUB = 8;          // users per thread in k_scores: one load of U[f, item] feeds eight sums

// vs: [k][n_sel] feature-major columns of the selected users; sub: [n_sel] their biases (nullable)
__global__ __launch_bounds__(256) void k_scores(int kind, int k, const double *__restrict__ u, const double *__restrict__ vs,
                                                const double *__restrict__ ib, const double *__restrict__ sub, double mu,
                                                const int32_t *__restrict__ users, int n_sel, int n_pad, int j_base,
                                                int n_chunk, int ni, int n_cand, double *scores, int32_t *items)
{
    const int item = blockIdx.x * 256 + threadIdx.x;
    const int j0 = j_base + blockIdx.y * kUB;                 // first user (position in the batch) of this thread
    if (item >= n_cand) return;
    double acc[kUB];
#pragma unroll
    for (int q = 0; q < kUB; ++q) acc[q] = 0.0;
#pragma unroll 8
    for (int f = 0; f < k; ++f) {
        const double uf = u[(size_t)f * ni + item];
        const double *vrow = vs + (size_t)f * n_pad + j0;                   // eight users: contiguous, uniform, padded
#pragma unroll
        for (int q = 0; q < kUB; ++q) acc[q] += uf * vrow[q];
    }
    const double bi = ib ? ib[item] : 0.0;
#pragma unroll
    for (int q = 0; q < kUB; ++q) {
        const int j = j0 + q;
        if (j >= j_base + n_chunk || j >= n_sel) break;
        const double bu = sub ? sub[j] : 0.0;
        double p;
        switch (kind) {
        case 0: p = acc[q] + 1.0; break;
        case 1: p = acc[q] + (bi + bu); break;
        case 2: p = 1.0 + (1.0 / (1.0 + exp(-(acc[q] + (bi + bu))))) * 4.0; break;
        case 3: p = acc[q] + mu + (bi + bu); break;
        default: p = acc[q]; break;
        }
        if (p != p || item == users[j]) p = 0.0;              // NaN -> 0 (mf.py:176); the item named like the user (mf.py:162)
        const size_t o = (size_t)(j - j_base) * n_cand + item;
        scores[o] = p;
        items[o] = item;
    }
}

// rated items of the chunk's users score 0 (mf.py:161,171-173)
__global__ __launch_bounds__(256) void k_clear_rated(const int64_t *__restrict__ lo, const int64_t *__restrict__ hi,
                                                     const int32_t *__restrict__ rated, int j_base, int n_chunk, int n_cand,
                                                     double *scores)
{
    const int j = j_base + blockIdx.y;
    for (int64_t p = lo[j] + blockIdx.x * 256 + threadIdx.x; p < hi[j]; p += (int64_t)gridDim.x * 256) {
        const int it = rated[p];
        if (it >= 0 && it < n_cand) scores[(size_t)(j - j_base) * n_cand + it] = 0.0;
    }
}

__global__ void k_iota_offsets(unsigned int *off, int n, unsigned int stride)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i <= n) off[i] = (unsigned int)i * stride;
}

// a sorted (descending) segment is  positives | zeros | negatives ; the zeros are dropped (mf.py:178-182)
__global__ __launch_bounds__(64) void k_take_top(const double *__restrict__ keys, const int32_t *__restrict__ vals, int n_cand,
                                                 int N, int j_base, int32_t *out_items, double *out_scores)
{
    const int jl = blockIdx.x;
    const double *kk = keys + (size_t)jl * n_cand;
    const int32_t *vv = vals + (size_t)jl * n_cand;
    __shared__ int sP, sE;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_cand;                                 // first index with key <= 0
        while (lo < hi) { const int m = (lo + hi) / 2; if (kk[m] > 0.0) lo = m + 1; else hi = m; }
        sP = lo;
        hi = n_cand;                                             // first index with key < 0
        while (lo < hi) { const int m = (lo + hi) / 2; if (kk[m] >= 0.0) lo = m + 1; else hi = m; }
        sE = lo;
    }
    __syncthreads();
    const int P = sP, Z = sE - sP;
    for (int t = threadIdx.x; t < N; t += 64) {
        const int src = t < P ? t : t + Z;
        const size_t o = (size_t)(j_base + jl) * N + t;
        if (src < n_cand) { out_items[o] = vv[src]; out_scores[o] = kk[src]; }
        else { out_items[o] = -1; out_scores[o] = 0.0; }
    }
}


// ---- fused top-N for N <= kTopMaxN: the scores are never written out ------------------------------------------
// A workgroup owns kUB users and streams all candidates; every thread keeps, per user, the kTM best of the items
// it visits (score descending, item ascending: a thread visits its items in ascending order, so a later equal
// score never displaces an earlier one) and the best score it ever DROPPED.  The 256 x kTM kept entries are
// merged by N rounds of a workgroup-wide arg-max.  The result is exact whenever no thread dropped anything that
// could belong to the list -- every thread's best dropped score is strictly below the N-th score taken (or
// nothing valid was dropped when fewer than N qualify); otherwise the user is flagged and redone by the sort
// path (a thread would have to visit more than kTM of the top N: ~1e-4 of the users at N = 16, 50 K items).
// Tried: one WAVE per 8 users with the four waves of a workgroup walking U together (L1 reuse) -- 351 vs 151 ms for
// 100 K users x 50 K items: four times the visits per lane and 196 VGPRs cost more than the shared reads save..
// The reference's exclusions are applied before an entry is kept or counted as dropped; the rated-item test is a
// binary search in the user's ascending list and only runs for the few entries that reach that point.
constexpr int kTM = 3;
constexpr int kTopMaxN = 16;

__device__ __forceinline__ bool is_rated(const int64_t *lo_, const int64_t *hi_, const int32_t *rated, int j, int item)
{
    if (!lo_) return false;
    int64_t lo = lo_[j], hi = hi_[j];
    while (lo < hi) {
        const int64_t m = (lo + hi) >> 1;
        const int x = rated[m];
        if (x == item) return true;
        if (x < item) lo = m + 1; else hi = m;
    }
    return false;
}

// (s, it) ranks before (s2, it2)?
__device__ __forceinline__ bool ranks_before(double s, int it, double s2, int it2)
{
    return s > s2 || (s == s2 && it < it2);
}

__global__ __launch_bounds__(256) void k_topn_fused(int kind, int k, const double *__restrict__ u,
                                                    const double *__restrict__ vs, const double *__restrict__ ib,
                                                    const double *__restrict__ sub, double mu,
                                                    const int32_t *__restrict__ users, int n_sel, int n_pad, int ni,
                                                    int n_cand, int N, const int64_t *__restrict__ off, const int64_t *__restrict__ off_hi,
                                                    const int32_t *__restrict__ rated, int force_unsafe,
                                                    int32_t *out_items, double *out_scores, int32_t *unsafe)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j0 = blockIdx.x * kUB;
    double ks[kUB][kTM];
    int ki[kUB][kTM];
    double dropped[kUB];
#pragma unroll
    for (int q = 0; q < kUB; ++q) {
        dropped[q] = -INFINITY;
#pragma unroll
        for (int m = 0; m < kTM; ++m) { ks[q][m] = -INFINITY; ki[q][m] = 0x7fffffff; }
    }
    for (int item = tid; item < n_cand; item += 256) {
        double acc[kUB];
#pragma unroll
        for (int q = 0; q < kUB; ++q) acc[q] = 0.0;
        // four features per round: their column loads and their four 64-byte scalar loads (eight users: contiguous,
        // uniform, padded) are issued together and waited for once -- one scalar load per feature, waited for before
        // its multiplies, left ~200 clocks of latency between every eight multiplies
        int f = 0;
        for (; f + 4 <= k; f += 4) {
            double uf[4], vv[4][kUB];
#pragma unroll
            for (int r = 0; r < 4; ++r) uf[r] = u[(size_t)(f + r) * ni + item];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double *vrow = vs + (size_t)(f + r) * n_pad + j0;
#pragma unroll
                for (int q = 0; q < kUB; ++q) vv[r][q] = vrow[q];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < kUB; ++q) acc[q] += uf[r] * vv[r][q];
        }
        for (; f < k; ++f) {
            const double uf = u[(size_t)f * ni + item];
            const double *vrow = vs + (size_t)f * n_pad + j0;
#pragma unroll
            for (int q = 0; q < kUB; ++q) acc[q] += uf * vrow[q];
        }
        const double bi = ib ? ib[item] : 0.0;
#pragma unroll
        for (int q = 0; q < kUB; ++q) {
            const int j = j0 + q;
            if (j >= n_sel) continue;
            const double bu = sub ? sub[j] : 0.0;
            double p;
            switch (kind) {
            case 0: p = acc[q] + 1.0; break;
            case 1: p = acc[q] + (bi + bu); break;
            case 2: p = 1.0 + (1.0 / (1.0 + exp(-(acc[q] + (bi + bu))))) * 4.0; break;
            case 3: p = acc[q] + mu + (bi + bu); break;
            default: p = acc[q]; break;
            }
            if (p != p || p == 0.0 || item == users[j]) continue;          // never listed (mf.py:162,176-182)
            // a score of -inf is listable (non-zero, not NaN) but cannot be told from the empty-slot sentinel: flag the
            // user (best dropped score = +inf), the sort path ranks it
            if (p == -INFINITY) { if (!is_rated(off, off_hi, rated, j, item)) dropped[q] = INFINITY; continue; }
            if (p > ks[q][kTM - 1]) {                                       // would be kept
                if (is_rated(off, off_hi, rated, j, item)) continue;
                double cs = p;                                              // sorted insertion by compare-and-swap
                int ci = item;                                              // (static indices: the lists stay in registers)
#pragma unroll
                for (int t = 0; t < kTM; ++t)
                    if (ranks_before(cs, ci, ks[q][t], ki[q][t])) {
                        const double ts = ks[q][t]; const int ti = ki[q][t];
                        ks[q][t] = cs; ki[q][t] = ci;
                        cs = ts; ci = ti;
                    }
                if (cs > dropped[q]) dropped[q] = cs;                       // the entry pushed out (valid, or -inf)
            } else if (p > dropped[q]) {                                    // rejected: remember the best one
                if (!is_rated(off, off_hi, rated, j, item)) dropped[q] = p;
            }
        }
    }
    // ---- merge: N rounds of a workgroup-wide arg-max over the kept entries, user by user ------------------
    __shared__ double ws[4];
    __shared__ int wi[4], wo[4];
    __shared__ int s_unsafe;
#pragma unroll                                                              // q static: the kept lists stay in registers
    for (int q = 0; q < kUB; ++q) {
        const int j = j0 + q;
        if (j >= n_sel) continue;                                           // uniform
        if (tid == 0) s_unsafe = 0;
        int taken = 0;                                                      // this thread's entries taken so far (a prefix)
        double T = -INFINITY;
        int n_out = 0;
        for (int t = 0; t < N; ++t) {
            double s = -INFINITY;
            int it = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < kTM; ++m)
                if (m == taken) { s = ks[q][m]; it = ki[q][m]; }
            int owner = tid;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double s2 = __shfl_xor(s, d, 64);
                const int it2 = __shfl_xor(it, d, 64), o2 = __shfl_xor(owner, d, 64);
                if (ranks_before(s2, it2, s, it)) { s = s2; it = it2; owner = o2; }
            }
            __syncthreads();                                                // previous round's ws[] consumed
            if (lane == 0) { ws[wid] = s; wi[wid] = it; wo[wid] = owner; }
            __syncthreads();
            s = ws[0]; it = wi[0]; owner = wo[0];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (ranks_before(ws[w], wi[w], s, it)) { s = ws[w]; it = wi[w]; owner = wo[w]; }
            if (s == -INFINITY) break;                                      // nothing left (uniform)
            if (owner == tid) ++taken;
            if (tid == 0) { out_items[(size_t)j * N + t] = it; out_scores[(size_t)j * N + t] = s; }
            T = s;
            ++n_out;
        }
        if (tid == 0)
            for (int t = n_out; t < N; ++t) { out_items[(size_t)j * N + t] = -1; out_scores[(size_t)j * N + t] = 0.0; }
        // a thread that dropped something which could belong to the list makes the user unsafe
        const bool bad = n_out == N ? dropped[q] >= T : dropped[q] > -INFINITY;
        if (bad || (force_unsafe && (j % 3) == 0)) s_unsafe = 1;
        __syncthreads();
        if (tid == 0) unsafe[j] = s_unsafe;
        __syncthreads();
    }
}

// ---- the same ranking with U staged through LDS ---------------------------------------------------------------
// k_topn_fused re-reads all of U once per 8 users with 400-KB-strided column loads; it measured 8x off its VALU bound.
// Here a workgroup owns 32 users (wave w: users 8w .. 8w+7) and walks U in tiles of 64 features x 64 items that the
// 256 threads load together, fully coalesced, one tile ahead (registers -> the other LDS buffer, one barrier per
// tile).  lane = item inside the tile; a wave's eight user values per feature sit in its own LDS block (vs is padded to a
// multiple of 8 users, so the block is a contiguous copy).  Same sums in the same order, same keep / drop bookkeeping per lane, the merge is a
// wave-wide arg-max.  A lane visits n_cand / 64 items, so a few more users are flagged (< 1 % at N = 16).
constexpr int kTF = 64, kTI = 64;                   // tile: features x items

__global__ __launch_bounds__(256) void k_topn_tiled(int kind, int k, const double *__restrict__ u,
                                                    const double *__restrict__ vs, const double *__restrict__ ib,
                                                    const double *__restrict__ sub, double mu,
                                                    const int32_t *__restrict__ users, int n_sel, int n_pad, int ni,
                                                    int n_cand, int N, const int64_t *__restrict__ off, const int64_t *__restrict__ off_hi,
                                                    const int32_t *__restrict__ rated, int force_unsafe,
                                                    int32_t *out_items, double *out_scores, int32_t *unsafe, double *pdrop)
{
    __shared__ double tile[2][kTF][kTI];                                    // 64 KB
    __shared__ __attribute__((aligned(16))) double vt[4][kTF][kUB];         // 16 KB: per wave, its users' values of a feature chunk
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wid)) * kUB;   // uniform; < n_pad by construction of the grid
    const bool wave_on = j0 < n_sel;
    double ks[kUB][kTM];
    int ki[kUB][kTM];
    double dropped[kUB];
#pragma unroll
    for (int q = 0; q < kUB; ++q) {
        dropped[q] = -INFINITY;
#pragma unroll
        for (int m = 0; m < kTM; ++m) { ks[q][m] = -INFINITY; ki[q][m] = 0x7fffffff; }
    }
    // gridDim.y > 1 (small batches): the candidate tiles are split over blockIdx.y, every split writes its own top-N
    // (row j * splits + y of the outputs) and k_topn_merge picks the N best of the splits' lists
    const int n_it_all = (n_cand + kTI - 1) / kTI, n_fc = (k + kTF - 1) / kTF;
    const int it_lo = (int)((int64_t)n_it_all * blockIdx.y / gridDim.y), it_hi = (int)((int64_t)n_it_all * (blockIdx.y + 1) / gridDim.y);
    const int n_tiles = (it_hi - it_lo) * n_fc;
    constexpr int PER = kTF * kTI / 256;                                    // 16 doubles per thread per tile
    double pre[PER];
    auto fetch = [&](int t) {                                               // tile t -> registers
        const int i0 = (it_lo + t / n_fc) * kTI, f0 = (t % n_fc) * kTF;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int f = f0 + r * 4 + wid, it = i0 + lane;
            pre[r] = (f < k && it < n_cand) ? u[(size_t)f * ni + it] : 0.0;
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int r = 0; r < PER; ++r) tile[b][r * 4 + wid][lane] = pre[r];
    };
    if (n_tiles > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    double acc[kUB];
    for (int t = 0; t < n_tiles; ++t) {
        const int fc = t % n_fc, i0 = (it_lo + t / n_fc) * kTI, f0 = fc * kTF;
        if (t + 1 < n_tiles) fetch(t + 1);
        if (fc == 0) {
#pragma unroll
            for (int q = 0; q < kUB; ++q) acc[q] = 0.0;
        }
        if (wave_on) {
            const int nf = k - f0 < kTF ? k - f0 : kTF;
            const double(*tb)[kTI] = tile[t & 1];
            // the wave's eight user values of every feature of this chunk go from (padded, contiguous) global memory into
            // its own LDS block once per chunk and are read back as broadcast b128 reads the scheduler can hoist.
            // Measured alternatives for 100 K users x 50 K items: one scalar load per (feature, user) 150 ms; one 64-byte
            // scalar load per feature 123 ms; four of them batched per round 104 ms (SMEM shares its counter with the
            // LDS tile reads and returns out of order: every wait is a full one); this form 90 ms -- now LDS-bound, a
            // broadcast read costs its full 64-lane bandwidth.
            if (n_fc > 1 || t == 0) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < kTF * kUB / 64; ++r) {
                    const int e = r * 64 + lane, f = e / kUB, q = e % kUB;
                    vt[wid][f][q] = f0 + f < k ? vs[(size_t)(f0 + f) * n_pad + j0 + q] : 0.0;
                }
                __builtin_amdgcn_wave_barrier();
            }
            const double(*vb)[kUB] = vt[wid];
#pragma unroll 4
            for (int f = 0; f < nf; ++f) {
                const double uf = tb[f][lane];
                double vv[kUB];
#pragma unroll
                for (int q = 0; q < kUB; q += 2) {
                    const double2 w = *reinterpret_cast<const double2 *>(&vb[f][q]);
                    vv[q] = w.x; vv[q + 1] = w.y;
                }
#pragma unroll
                for (int q = 0; q < kUB; ++q) acc[q] += uf * vv[q];
            }
            const int item = i0 + lane;
            if (fc == n_fc - 1 && item < n_cand) {
                const double bi = ib ? ib[item] : 0.0;
#pragma unroll
                for (int q = 0; q < kUB; ++q) {
                    const int j = j0 + q;
                    if (j >= n_sel) continue;
                    const double bu = sub ? sub[j] : 0.0;
                    double p;
                    switch (kind) {
                    case 0: p = acc[q] + 1.0; break;
                    case 1: p = acc[q] + (bi + bu); break;
                    case 2: p = 1.0 + (1.0 / (1.0 + exp(-(acc[q] + (bi + bu))))) * 4.0; break;
                    case 3: p = acc[q] + mu + (bi + bu); break;
                    default: p = acc[q]; break;
                    }
                    if (p != p || p == 0.0 || item == users[j]) continue;  // never listed (mf.py:162,176-182)
                    // (-inf: listable, but indistinguishable from the empty-slot sentinel -> flag the user for the sort path)
                    if (p == -INFINITY) { if (!is_rated(off, off_hi, rated, j, item)) dropped[q] = INFINITY; continue; }
                    if (p > ks[q][kTM - 1]) {                               // would be kept
                        if (is_rated(off, off_hi, rated, j, item)) continue;
                        double cs = p;
                        int ci = item;
#pragma unroll
                        for (int m = 0; m < kTM; ++m)
                            if (ranks_before(cs, ci, ks[q][m], ki[q][m])) {
                                const double ts = ks[q][m]; const int ti = ki[q][m];
                                ks[q][m] = cs; ki[q][m] = ci;
                                cs = ts; ci = ti;
                            }
                        if (cs > dropped[q]) dropped[q] = cs;
                    } else if (p > dropped[q]) {
                        if (!is_rated(off, off_hi, rated, j, item)) dropped[q] = p;
                    }
                }
            }
        }
        if (t + 1 < n_tiles) stash((t + 1) & 1);                            // that buffer was last read in round t - 1
        __syncthreads();
    }
    if (!wave_on) return;
    // ---- merge: N rounds of a wave-wide arg-max over the kept entries, user by user ---------------------------
#pragma unroll
    for (int q = 0; q < kUB; ++q) {
        const int j = j0 + q;
        if (j >= n_sel) continue;                                           // uniform
        const size_t orow = (size_t)j * gridDim.y + blockIdx.y;
        int taken = 0;
        double T = -INFINITY;
        int n_out = 0;
        for (int t = 0; t < N; ++t) {
            double s = -INFINITY;
            int it = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < kTM; ++m)
                if (m == taken) { s = ks[q][m]; it = ki[q][m]; }
            int owner = lane;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double s2 = __shfl_xor(s, d, 64);
                const int it2 = __shfl_xor(it, d, 64), o2 = __shfl_xor(owner, d, 64);
                if (ranks_before(s2, it2, s, it)) { s = s2; it = it2; owner = o2; }
            }
            if (s == -INFINITY) break;                                      // nothing left (uniform)
            if (owner == lane) ++taken;
            if (lane == 0) { out_items[orow * N + t] = it; out_scores[orow * N + t] = s; }
            T = s;
            ++n_out;
        }
        if (lane == 0)
            for (int t = n_out; t < N; ++t) { out_items[orow * N + t] = -1; out_scores[orow * N + t] = 0.0; }
        if (gridDim.y > 1) {
            // split mode: whether something dropped here could belong to the list is decided against the MERGED list's
            // N-th score (k_topn_merge) -- a split's own N-th score is lower and would flag users needlessly
            double dm = dropped[q];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(dm, d, 64); dm = o > dm ? o : dm; }
            if (lane == 0) pdrop[orow] = dm;
        } else {
            const bool bad = n_out == N ? dropped[q] >= T : dropped[q] > -INFINITY;
            const bool any_bad = __any(bad || (force_unsafe && (j % 3) == 0));
            if (lane == 0 && any_bad) unsafe[j] = 1;                        // zeroed by the host
        }
    }
}


// ---- the same ranking with the scores on the MATRIX cores (round 5) -------------------------------------------------
// mf.py:161-173 is a dense contraction, scores = V^T U: 100 K users x 50 K items x 64 features = 0.64 TFLOP, which
// k_topn_tiled did at 7 TFLOP/s on unfused vector f64.  Here a workgroup owns 32 users (their feature values in LDS: the A
// operands of v_mfma_f64_16x16x4_f64) and its four waves walk the candidates in blocks of 32 items each, U straight from
// global memory / L2 (B operands: one 16-byte load per lane and 4-feature step = items 2c, 2c + 1 -> two column sets), four
// accumulator tiles per wave (2 x 16 users x 2 x 16 items).  Lane (c, g) of a wave then holds the scores of EIGHT users
// (g + 4 q of both user tiles) for the items i0 + 2c, i0 + 2c + 1 of every block: the per-lane keep / drop bookkeeping is
// k_topn_tiled's, a user's 64 holders (16 lanes x 4 waves) are merged by the same wave-wide arg-max through LDS.
// EXACTNESS.  The matrix instruction does not add in the reference's order: these scores are APPROXIMATE (a few ulps).  They
// only SELECT: k_topn_exactify recomputes the N selected scores of a user in the reference's order (k_scores' loop: the
// bits the single-user query returns), orders them by (score descending, item ascending), and flags the user for the sort
// path unless everything that was not selected lies below the N-th exact score by more than the rigorous error bound of the
// approximate scores, 2 (k + 4) 2^-53 (||u||_max ||v_user|| + |biases|) -- so a flagged-or-exact result, as before.
constexpr int kMK = 128;                            // features held in LDS (k <= kMK; more: k_topn_tiled)
constexpr int kVS = 48;                             // row stride of the user tile: 32 users + pad (A operand reads conflict-free)
constexpr int kRL = 176;                            // rated items per user kept in LDS
typedef double double4_mf __attribute__((ext_vector_type(4)));
typedef double v2f64_u8 __attribute__((ext_vector_type(2), aligned(8)));

#ifdef MFX_TOPN_KO
// diagnostic build only (tools/r05_topn_ko.sh): knock single phases out of k_topn_mfma to price them; results are garbage
__device__ int g_topn_ko = 0;
#define TOPN_KO(bit) (g_topn_ko & (bit))
#else
#define TOPN_KO(bit) 0
#endif
__global__ __launch_bounds__(256, 2) void k_topn_mfma(int kind, int k, const double *__restrict__ u,
                                                      const double *__restrict__ vs, const double *__restrict__ ib,
                                                      const double *__restrict__ sub, double mu,
                                                      const int32_t *__restrict__ users, int n_sel, int n_pad, int ni,
                                                      int n_cand, int N, const int64_t *__restrict__ off, const int64_t *__restrict__ off_hi,
                                                      const int32_t *__restrict__ rated, const unsigned long long *__restrict__ unorm2_max,
                                                      int32_t *out_items, double *out_scores, double *pdrop, int g_off)
{
    // g_off: the first user group of this launch (the lists go out relative to it: row (j - 32 g_off) gridDim.y + blockIdx.y)
    // LDS: the user tile while the candidates are walked; the holders' kept lists of 16 users at a time afterwards
    constexpr int kMergeU = 16;
    constexpr size_t kVtBytes = (size_t)kMK * kVS * 8, kMgBytes = (size_t)kMergeU * 64 * (kTM * 12 + 8);
    __shared__ __attribute__((aligned(16))) unsigned char smem[kVtBytes > kMgBytes ? kVtBytes : kMgBytes];
    double(*vt)[kVS] = reinterpret_cast<double(*)[kVS]>(smem);
    // the users' rated lists (ascending) in LDS: the kept / dropped bookkeeping asks "is this item rated?" a few hundred times per
    // user and lane, and a binary search of seven DEPENDENT global loads each was what the whole kernel waited for (both this kernel
    // and k_topn_tiled ran 85-95 ms whatever the arithmetic cost); a list longer than kRL stays in global memory
    __shared__ int32_t rl[32][kRL];
    __shared__ int rn[32];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int j0 = (blockIdx.x + g_off) * 32;                                // < n_pad (a multiple of 32)
    const int nks = (k + 3) >> 2;                                            // 4-feature steps
    for (int e = tid; e < kMK * 32; e += 256) {                              // (rows k .. kMK - 1: zeros, the padded steps of the last group)
        const int f = e >> 5, q = e & 31;
        vt[f][q] = f < k ? vs[(size_t)f * n_pad + j0 + q] : 0.0;
    }
    for (int q = wid; q < 32; q += 4) {                                      // a wave copies a user's list
        const int j = j0 + q;
        int n = -1;                                                          // -1: no lists at all / list too long -> global search
        if (off && j < n_sel && !TOPN_KO(32)) {
            const int64_t lo = off[j], hi = off_hi[j];
            if (hi - lo <= kRL) {
                n = (int)(hi - lo);
                for (int e = lane; e < n; e += 64) rl[q][e] = rated[lo + e];
            }
        } else if (!off) n = 0;
        if (lane == 0) rn[q] = n;
    }
    __shared__ unsigned rmask[4][32];                                        // per wave: bit i of [q] = item i0 + i of the block is rated by user q
    // THE USER'S BAR.  A lane's own bookkeeping rejects what lies below ITS weakest dropped score, but a user's scores sit in 64 holders
    // (16 lanes x 4 waves): with 64 lanes per wave SOME lane took the slow path in 59 % of the (item, user) steps and the selection cost
    // as much as the matrix instructions (knock-outs, profiles/r05_topn_knockouts.txt).  After 4, 8, 16, ... blocks per wave the
    // workgroup therefore takes, per user, the N-th largest of the 64 holders' BEST kept scores -- N distinct candidates score at least
    // that, so nothing below it can enter the user's top N -- lowers it by a margin above the approximate scores' error bound
    // (k_topn_exactify's delta; too small a margin would only flag the user for the sort path, never return a wrong list) and raises
    // every holder's `dropped` to it: the fast reject below then passes ~N / (items seen) of the scores instead of ~3 x 64 / (items seen).
    // `dropped` stays what the merge needs: an upper bound of every score the lane did not keep.
    __shared__ double xb[16][64];                                            // heads of 16 users' holders
    __shared__ double xbar[32], mg[32];                                      // the users' bars; their margins
    int cur = 0;                                                             // lane q < 32: cursor into user q's (ascending) list
    // this lane's eight users: slot s = 4 t + q <-> user j0 + 16 t + g + 4 q
    // the five predictors as ONE expression, p = (a + xk) + (bi + bu) (+ the logistic map for kind 2): a branch nest on `kind` per
    // score was ~10 scalar branches x 16 scores per block.  -0.0 is the additive identity for every double (x + -0.0 == x bit for
    // bit), so kind 1 is a + (bi + bu), kind 3 (a + mu) + (bi + bu), kind 0 (a + 1.0) + -0.0, default (a + -0.0) + -0.0: the very
    // operations of k_scores' switch (these scores only select, but the error bound of k_topn_exactify is stated for them).
    const bool biased = kind >= 1 && kind <= 3;
    const double xk = kind == 0 ? 1.0 : (kind == 3 ? mu : -0.0);
    double bu[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int j = j0 + 16 * (s >> 2) + g + 4 * (s & 3);
        bu[s] = !biased ? -0.0 : ((sub && j < n_sel) ? sub[j] : 0.0);
    }
    double ks[8][kTM];
    int ki[8][kTM];
    double dropped[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        dropped[s] = -INFINITY;
#pragma unroll
        for (int m = 0; m < kTM; ++m) { ks[s][m] = -INFINITY; ki[s][m] = 0x7fffffff; }
    }
    __syncthreads();
    if (tid < 32 && !TOPN_KO(128)) {
        double vn2 = 0.0;
        for (int f = 0; f < k; ++f) vn2 += vt[f][tid] * vt[f][tid];
        const int j = j0 + tid;
        const double unmax = sqrt(__longlong_as_double((long long)*unorm2_max));
        // 8 x (k + 4) 2^-53 x the magnitude bound (k_topn_exactify: 2 x), the |score| term is added where the bar is taken
        mg[tid] = 8.0 * (k + 4) * 1.1102230246251565e-16 * (unmax * sqrt(vn2) + fabs((sub && j < n_sel) ? sub[j] : 0.0) + fabs(mu) + 1.0);
        xbar[tid] = -INFINITY;
    }
    auto raise_bars = [&]() __attribute__((always_inline)) {                 // (workgroup-uniform call sites only)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if ((s >> 2) == half) xb[g + 4 * (s & 3)][wid * 16 + c] = ks[s][0];
            __syncthreads();
#pragma unroll 1
            for (int uq = 0; uq < 4; ++uq) {
                const int lu16 = wid * 4 + uq;
                const double v = xb[lu16][lane];
                int cnt = 0;                                                 // holders whose head ranks before this one's
#pragma unroll 2
                for (int t = 0; t < 64; t += 2) {
                    const double w0 = xb[lu16][t], w1 = xb[lu16][t + 1];
                    cnt += ((w0 > v) || (w0 == v && t < lane)) ? 1 : 0;
                    cnt += ((w1 > v) || (w1 == v && t + 1 < lane)) ? 1 : 0;
                }
                if (cnt == N - 1) {                                          // one lane at most (none for N > 64); -inf stays -inf
                    const int lu = 16 * half + lu16;
                    const double bar = v - (mg[lu] + fabs(v) * ((k + 4) * 8.881784197001252e-16));
                    if (bar > xbar[lu]) xbar[lu] = bar;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const double bar = xbar[16 * (s >> 2) + g + 4 * (s & 3)];
            if (bar > dropped[s]) dropped[s] = bar;
        }
    };
    // gridDim.y > 1 (small batches): the candidate blocks are split over blockIdx.y (k_topn_merge joins the lists)
    const int n_blk_all = (n_cand + 31) / 32;
    const int b_lo = (int)((int64_t)n_blk_all * blockIdx.y / gridDim.y), b_hi = (int)((int64_t)n_blk_all * (blockIdx.y + 1) / gridDim.y);
    // B operands: groups of eight 4-feature steps, the NEXT group's loads (the next block's first group after a block's last) in
    // flight under this group's 32 matrix instructions
    constexpr int GS = 4;
    const int ngr = (nks + GS - 1) / GS;
    // (two groups ahead measured slower: 50 against 43 ms for 100 K users; groups of eight steps spill)
    // BRANCH-FREE loads through a buffer resource per feature group (base = the group's first feature row, extent = what is left of
    // u + 8 pad bytes): rows beyond k and items beyond the array read 0 (or, inside the array, a neighbour's finite-or-not value in a
    // COLUMN that is never looked at: items >= n_cand), so no `if` guards a load (one branch nest per guarded load was what the
    // compiler generated before).  The raw 16-byte quads stay where they land, in TWO register sets that take turns (no copies), and
    // the matrix instructions read their halves in place.  (host: 16 ni 8 < 2^31, u allocated with 16 pad bytes, ZEROED: the first 8
    // are addressable as item 0 of row k)
    // The loads are the compiler's (`raw_buffer_load` builtins) and so are the wait counts.  They are conservative -- it asks for
    // `vmcnt(3..0)` in front of set A's matrix instructions right behind set B's four loads -- and `asm volatile` loads with hand-written
    // `vmcnt(4)` ran 19.5 instead of 25 ms, but returned a garbage list about once in 170 small queries (then k_topn_exactify faulted on
    // it): measured, not understood, NOT in the product (profiles/experiments/r05_topn_asm_loads.hip.txt, profiles/r05_topn_asm_loads.txt).
    typedef unsigned v4u_mf __attribute__((ext_vector_type(4)));
    typedef unsigned v2u_mf __attribute__((ext_vector_type(2)));
    v4u_mf wA[GS], wB[GS];
    auto loadB = [&](v4u_mf (&w)[GS], int blk, int gr) __attribute__((always_inline)) {
        const int f0 = gr * GS * 4;                                          // uniform; < k
        const size_t left = (size_t)(k - f0) * ni * 8 + 8;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(u) + (size_t)f0 * ni, 0,
                                                                           left > 0x7fffffffu ? 0x7fffffff : (int)left, 0x27000);
        const unsigned off = ((unsigned)g * (unsigned)ni + (unsigned)(blk * 32 + 2 * c)) * 8u;   // feature f0 + 4 s + g of step s
#pragma unroll
        for (int s = 0; s < GS; ++s) w[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)(4 * s) * (unsigned)ni * 8u, 0, 0);
    };
    loadB(wA, b_lo + wid, 0);                                                // (a block beyond the range: loaded, never used)
    const int n_it = (b_hi - b_lo + 3) >> 2;                                 // the same trip count for the four waves (the bars need barriers)
    for (int nb = 0; nb < n_it; ++nb) {
        const int b = b_lo + wid + 4 * nb;
        if (nb >= 4 && (nb & (nb - 1)) == 0 && !TOPN_KO(16)) raise_bars();
        if (b >= b_hi) continue;
        const int i0 = b * 32, it0 = i0 + 2 * c;                            // this lane's items: it0, it0 + 1
        // which of the block's 32 items each user has rated: the lists ascend and so do a wave's blocks, so lane q walks user q's
        // list once per launch (a couple of LDS reads per block; the tests below are then bit tests, no search, no divergence)
        if (lane < 32 && !TOPN_KO(8)) {
            const int n = rn[lane];
            unsigned m = 0u;
            while (cur < n && rl[lane][cur] < i0) ++cur;
            for (int t = cur; t < n && rl[lane][t] < i0 + 32; ++t) m |= 1u << (rl[lane][t] - i0);
            rmask[wid][lane] = m;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double4_mf acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (double4_mf){0.0, 0.0, 0.0, 0.0};
        auto mma = [&](const v4u_mf (&w)[GS], int gr) __attribute__((always_inline)) {
            if (TOPN_KO(2)) return;
#pragma unroll
            for (int s = 0; s < GS; ++s) {
                const int f = 4 * (gr * GS + s) + g;                        // < kMK: the tile's rows beyond k hold zeros
                const double a0 = vt[f][c], a1 = vt[f][16 + c];
                const double bx = __builtin_bit_cast(double, (v2u_mf){w[s].x, w[s].y}), by = __builtin_bit_cast(double, (v2u_mf){w[s].z, w[s].w});
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bx, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, by, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bx, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, by, acc[3], 0, 0, 0);
            }
        };
        // set A holds group `gr` on entry (of the loop and of every pair); the next group's (after a block's last: the next block's
        // first) loads go out before this group's matrix instructions
        for (int gr = 0; gr < ngr; gr += 2) {
            if (gr + 1 < ngr) loadB(wB, b, gr + 1); else loadB(wB, b + 4, 0);
            mma(wA, gr);
            if (gr + 1 < ngr) {
                if (gr + 2 < ngr) loadB(wA, b, gr + 2); else loadB(wA, b + 4, 0);
                mma(wB, gr + 1);
            } else {                                                         // an odd number of groups: the next block's first group sits in B
#pragma unroll
                for (int s = 0; s < GS; ++s) wA[s] = wB[s];
            }
        }
        if (TOPN_KO(2)) {                                                   // no matrix instructions: scores with the same kind of order statistics
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned h = (unsigned)(i0 + 2 * c + (t & 1)) * 2654435761u ^ (unsigned)(j0 + 16 * (t >> 1) + g + 4 * q) * 40503u;
                    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                    acc[t][q] = (double)(h >> 8) * (1.0 / 16777216.0) + (double)wA[0].x * 1e-300;
                }
        }
        if (TOPN_KO(1)) {                                                   // no selection: keep the scores alive
#pragma unroll
            for (int t = 0; t < 4; ++t) dropped[0] = fmax(dropped[0], (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]));
            continue;
        }
#pragma unroll
        for (int par = 0; par < 2; ++par) {                                 // the lane's two items, ascending
            const int item = it0 + par;
            if (item >= n_cand) continue;
            const double bi = !biased ? -0.0 : (ib ? ib[item] : 0.0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int lu = 16 * (s >> 2) + g + 4 * (s & 3), j = j0 + lu;
                if (j >= n_sel) continue;
                const double a = acc[2 * (s >> 2) + par][s & 3];
                auto rated_here = [&]() -> bool {
                    if (rn[lu] < 0) return is_rated(off, off_hi, rated, j, item);      // a list too long for LDS
                    return (rmask[wid][lu] >> (2 * c + par)) & 1u;
                };
                double p = (a + xk) + (bi + bu[s]);
                if (kind == 2) p = 1.0 + (1.0 / (1.0 + exp(-p))) * 4.0;
                // fast reject: nothing at or below `dropped` (the lane's best dropped score <= its weakest kept entry, or the user's bar)
                // changes the bookkeeping: it is not kept and `dropped` already bounds it -- a NaN compares false and is never listed; an
                // approximate 0.0 below the bar is one more unselected score under `dropped`, which k_topn_exactify's margin test covers
                if (!(p > dropped[s]) && p != -INFINITY) continue;
                if (p != p || p == 0.0) {                                    // never listed (mf.py:162,176-182) -- if the EXACT score says so too:
                    if (p == 0.0 && a != 0.0) dropped[s] = INFINITY;         // a sum that cancels to zero here need not in the reference's order
                    continue;
                }
                if (item == users[j]) continue;
                // (-inf: listable, but indistinguishable from the empty-slot sentinel -> flag the user for the sort path)
                if (p == -INFINITY) { if (!rated_here()) dropped[s] = INFINITY; continue; }
                if (p > ks[s][kTM - 1]) {                                    // would be kept
                    if (rated_here()) continue;
                    double cs = p;
                    int ci = item;
#pragma unroll
                    for (int m = 0; m < kTM; ++m)
                        if (ranks_before(cs, ci, ks[s][m], ki[s][m])) {
                            const double ts = ks[s][m]; const int ti = ki[s][m];
                            ks[s][m] = cs; ki[s][m] = ci;
                            cs = ts; ci = ti;
                        }
                    if (cs > dropped[s]) dropped[s] = cs;
                } else if (p > dropped[s]) {
                    if (!rated_here()) dropped[s] = p;
                }
            }
        }
    }
    // ---- merge: a user's 64 holders (lanes (c, g) of the four waves) through LDS, 16 users at a time; wave w then takes four users:
    //      lane = holder, N rounds of a wave-wide arg-max over the sorted kept lists (k_topn_tiled's merge) -------------------------
    double *m_s = reinterpret_cast<double *>(smem);                          // [16][64][kTM]
    double *m_d = m_s + kMergeU * 64 * kTM;                                  // [16][64]
    int *m_i = reinterpret_cast<int *>(m_d + kMergeU * 64);                  // [16][64][kTM]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (TOPN_KO(64)) break;
        __syncthreads();                                                     // the user tile / the previous half has been read
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if ((s >> 2) != half) continue;                                  // user tile `half`: local users 16 half + g + 4 q
            const int lu = g + 4 * (s & 3), holder = wid * 16 + c;
#pragma unroll
            for (int m = 0; m < kTM; ++m) { m_s[(lu * 64 + holder) * kTM + m] = ks[s][m]; m_i[(lu * 64 + holder) * kTM + m] = ki[s][m]; }
            m_d[lu * 64 + holder] = dropped[s];
        }
        __syncthreads();
        for (int uq = 0; uq < 4; ++uq) {
            const int lu = wid * 4 + uq, j = j0 + 16 * half + lu;
            if (j >= n_sel) continue;                                        // uniform
            double ls[kTM];
            int li[kTM];
#pragma unroll
            for (int m = 0; m < kTM; ++m) { ls[m] = m_s[(lu * 64 + lane) * kTM + m]; li[m] = m_i[(lu * 64 + lane) * kTM + m]; }
            double dm = m_d[lu * 64 + lane];
            const size_t orow = (size_t)(j - 32 * g_off) * gridDim.y + blockIdx.y;
            int taken = 0, n_out = 0;
            for (int t = 0; t <= N; ++t) {                                   // round N: the best entry NOT taken counts as dropped
                double s = -INFINITY;
                int it = 0x7fffffff;
#pragma unroll
                for (int m = 0; m < kTM; ++m)
                    if (m == taken) { s = ls[m]; it = li[m]; }
                int owner = lane;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const double s2 = __shfl_xor(s, d, 64);
                    const int it2 = __shfl_xor(it, d, 64), o2 = __shfl_xor(owner, d, 64);
                    if (ranks_before(s2, it2, s, it)) { s = s2; it = it2; owner = o2; }
                }
                if (s == -INFINITY) break;                                   // nothing left (uniform)
                if (t == N) { dm = s > dm ? s : dm; break; }
                if (owner == lane) ++taken;
                if (lane == 0) { out_items[orow * N + t] = it; out_scores[orow * N + t] = s; }
                ++n_out;
            }
            if (lane == 0)
                for (int t = n_out; t < N; ++t) { out_items[orow * N + t] = -1; out_scores[orow * N + t] = 0.0; }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(dm, d, 64); dm = o > dm ? o : dm; }
            if (lane == 0) pdrop[orow] = dm;
        }
    }
}

// ||u_item||^2 over the candidates -> its maximum (bit pattern of a non-negative double orders like the integer)
__global__ __launch_bounds__(256) void k_item_norm_max(int k, const double *__restrict__ u, int ni, int n_cand, unsigned long long *out)
{
    const int item = blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (item < n_cand)
        for (int f = 0; f < k; ++f) { const double x = u[(size_t)f * ni + item]; s += x * x; }
    if (!(s == s)) s = 0.0;              // a NaN factor: every score of that item is NaN in ANY order of the sum (never listed): no bound needed
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(s, d, 64); s = o > s ? o : s; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(s));
}

// One wave per user: the N selected items (approximate scores, any order) -> their EXACT scores (k_scores' loop: the reference's
// order of operations), ordered by (score descending, item ascending); the user is flagged unless every score that was not selected
// is below the N-th exact score by more than the error bound of the approximate scores.
__global__ __launch_bounds__(64) void k_topn_exactify(int kind, int k, const double *__restrict__ u, const double *__restrict__ vs,
                                                      const double *__restrict__ ib, const double *__restrict__ sub, double mu,
                                                      int n_sel, int n_pad, int ni, int N, const int32_t *__restrict__ in_items,
                                                      const double *__restrict__ pdrop, const unsigned long long *__restrict__ unorm2_max,
                                                      int force_unsafe, int32_t *out_items, double *out_scores, int32_t *unsafe)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    const int item = lane < N ? in_items[(size_t)j * N + lane] : -1;
    const double bu = sub ? sub[j] : 0.0;
    double p = -INFINITY, acc = 0.0, vn2 = 0.0;
    for (int f = 0; f < k; ++f) {
        const double vf = vs[(size_t)f * n_pad + j];
        vn2 += vf * vf;
        if (item >= 0) acc += u[(size_t)f * ni + item] * vf;
    }
    bool lost = false;
    double bi = 0.0;
    if (item >= 0) {
        bi = ib ? ib[item] : 0.0;
        switch (kind) {
        case 0: p = acc + 1.0; break;
        case 1: p = acc + (bi + bu); break;
        case 2: p = 1.0 + (1.0 / (1.0 + exp(-(acc + (bi + bu))))) * 4.0; break;
        case 3: p = acc + mu + (bi + bu); break;
        default: p = acc; break;
        }
        if (p != p || p == 0.0 || p == -INFINITY) lost = true;               // the exact score is one the list never holds: redo
    }
    // rank among the selected (distinct items): entries that rank before this one
    int rank = 0, n_out = 0;
    double T = INFINITY, bmax = 0.0;
    for (int t = 0; t < kTopMaxN; ++t) {
        const double s2 = __shfl(p, t, 64), b2 = __shfl(bi, t, 64);
        const int it2 = __shfl(item, t, 64);
        if (it2 < 0) continue;                                               // uniform
        ++n_out;
        T = s2 < T ? s2 : T;
        bmax = fabs(b2) > bmax ? fabs(b2) : bmax;
        if (item >= 0 && ranks_before(s2, it2, p, item)) ++rank;
    }
    if (item >= 0) { out_items[(size_t)j * N + rank] = item; out_scores[(size_t)j * N + rank] = p; }
    else if (lane < N) { out_items[(size_t)j * N + lane] = -1; out_scores[(size_t)j * N + lane] = 0.0; }   // (ranks fill 0 .. n_out - 1)
    // |approximate - exact| <= gamma (sum |u_f v_f| + |biases|) <= gamma (||u|| ||v|| + ...), gamma = (k + 4) 2^-53, doubled for slack
    const double unmax = sqrt(__longlong_as_double((long long)*unorm2_max));
    const double delta = 2.0 * (k + 4) * 1.1102230246251565e-16 * (unmax * sqrt(vn2) + bmax + fabs(bu) + fabs(mu) + 1.0);
    const double dm = pdrop[j];
    const bool margin_ok = n_out == N ? dm < T - delta : dm == -INFINITY;
    const bool bad = !margin_ok || !(delta == delta) || delta == INFINITY || (force_unsafe && (j % 3) == 0);
    const bool any_bad = __any(bad || lost);
    if (lane == 0) unsafe[j] = any_bad ? 1 : 0;
}

// small batches: the N best of `splits` lists of N (each sorted: score descending, item ascending; -1 = no entry).
// One wave per user, lane y holds split y's list; the same arg-max rounds as above (splits <= 64, N <= kTopMaxN).
// drop_out != null (lists of APPROXIMATE scores, k_topn_mfma): instead of deciding, hand k_topn_exactify the best score that is NOT
// in the merged list -- the splits' dropped scores and the merged list's own runner-up.
__global__ __launch_bounds__(64) void k_topn_merge(const int32_t *__restrict__ pit, const double *__restrict__ psc,
                                                   const double *__restrict__ pdrop, int splits, int N, int force_unsafe,
                                                   int32_t *out_items, double *out_scores, int32_t *unsafe, double *drop_out)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    double ls[kTopMaxN];
    int li[kTopMaxN];
#pragma unroll
    for (int m = 0; m < kTopMaxN; ++m) {
        const bool have = lane < splits && m < N && pit[((size_t)j * splits + lane) * N + m] >= 0;
        li[m] = have ? pit[((size_t)j * splits + lane) * N + m] : 0x7fffffff;
        ls[m] = have ? psc[((size_t)j * splits + lane) * N + m] : -INFINITY;
    }
    int taken = 0, n_out = 0;
    double T = -INFINITY, runner_up = -INFINITY;
    for (int t = 0; t < N + (drop_out ? 1 : 0); ++t) {
        double s = -INFINITY;
        int it = 0x7fffffff;
#pragma unroll
        for (int m = 0; m < kTopMaxN; ++m)
            if (m == taken) { s = ls[m]; it = li[m]; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double s2 = __shfl_xor(s, d, 64);
            const int it2 = __shfl_xor(it, d, 64);
            if (ranks_before(s2, it2, s, it)) { s = s2; it = it2; }
        }
        if (s == -INFINITY) break;
        if (t == N) { runner_up = s; break; }
        if (taken < kTopMaxN) {
            double hs = -INFINITY; int hi = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < kTopMaxN; ++m)
                if (m == taken) { hs = ls[m]; hi = li[m]; }
            if (hs == s && hi == it) ++taken;                               // an item belongs to exactly one split
        }
        if (lane == 0) { out_items[(size_t)j * N + t] = it; out_scores[(size_t)j * N + t] = s; }
        T = s;
        ++n_out;
    }
    if (lane == 0)
        for (int t = n_out; t < N; ++t) { out_items[(size_t)j * N + t] = -1; out_scores[(size_t)j * N + t] = 0.0; }
    // a split that dropped something which could belong to the merged list makes the user unsafe
    double dm = lane < splits ? pdrop[(size_t)j * splits + lane] : -INFINITY;
    if (drop_out) {
        dm = runner_up > dm ? runner_up : dm;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(dm, d, 64); dm = o > dm ? o : dm; }
        if (lane == 0) drop_out[j] = dm;
        return;
    }
    const bool bad = n_out == N ? dm >= T : dm > -INFINITY;
    const bool any_bad = __any(bad || (force_unsafe && (j % 3) == 0));
    if (lane == 0) unsafe[j] = any_bad ? 1 : 0;
}

struct DevBufs {
    std::vector<void *> p;
    ~DevBufs() { for (void *x : p) if (x) (void)hipFree(x); }
    template <typename T> hipError_t get(T **out, size_t n)
    {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
        if (e == hipSuccess) { p.push_back(q); *out = (T *)q; }
        return e;
    }
};

}  // namespace

static std::atomic<int64_t> g_topn_stats[2];   // users served by the fused kernel / users redone by the sort path

// Device-resident model for top-N queries: the one-shot mfx_topn_batch builds one for the call (V = the selected
// users' columns, a user's position = its place in the batch), mfx_topn_plan_* keep one across calls (V = every
// user, position = user id) -- the serving form: a query then costs a 64-byte-per-feature gather and the kernel,
// not an upload of the item factors.
struct mfx_topn_plan {
    int kind = 0, k = 0, nu = 0, ni = 0;
    double mu = 0.0;
    double *dU = nullptr, *dV = nullptr, *dib = nullptr, *dub = nullptr;   // [k][ni], [k][nu], [ni], [nu]
    int64_t *d_off = nullptr;                                              // [nu + 1] rated lists by position (CSR), or null
    int32_t *d_rated = nullptr;
    bool ascending = true;                                                 // every rated list ascending (fused kernel's search)
    ~mfx_topn_plan()
    {
        void *q[] = {dU, dV, dib, dub, d_off, d_rated};
        for (void *x : q) if (x) (void)hipFree(x);
    }
};

namespace {

__global__ __launch_bounds__(256) void k_gather_users(const double *__restrict__ V, const double *__restrict__ ub, int nu, int k,
                                                      const int32_t *__restrict__ pos, int n_sel, int n_pad, double *vs,
                                                      double *ubs)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pad) return;
    const int p = j < n_sel ? pos[j] : -1;
    for (int f = 0; f < k; ++f) vs[(size_t)f * n_pad + j] = p >= 0 ? V[(size_t)f * nu + p] : 0.0;
    if (ubs && j < n_sel) ubs[j] = ub[p];
}

// per-batch CSR of the rated lists: off_sel[j] = start of the list of the user at position pos[j]
__global__ __launch_bounds__(256) void k_batch_lists(const int64_t *__restrict__ off, const int32_t *__restrict__ pos, int n_sel,
                                                     int64_t *lo, int64_t *hi)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n_sel) { lo[j] = off[pos[j]]; hi[j] = off[pos[j] + 1]; }
}

}  // namespace

#define TOPN_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); \
        return mfx::fail(e_ == hipErrorOutOfMemory ? MFX_E_NOMEM : MFX_E_HIP, "mfx_topn: %s -> %s", #call, hipGetErrorString(e_)); } } while (0)

static int topn_state_init(mfx_topn_plan &S, int kind, int dim, const double *u, const double *v, const double *ib,
                           const double *ub, double mu, int nu, int ni, const int64_t *rated_off, const int32_t *rated_items)
{
    hipStream_t st = mfx::stream();
    S.kind = kind; S.k = dim; S.nu = nu; S.ni = ni; S.mu = mu;
    TOPN_TRY(hipMalloc((void **)&S.dU, (size_t)dim * ni * 8 + 16));   // (+ pad: k_topn_mfma's 16-byte loads of an odd catalogue's last item)
    TOPN_TRY(hipMalloc((void **)&S.dV, (size_t)dim * nu * 8));
    TOPN_TRY(hipMemcpyAsync(S.dU, u, (size_t)dim * ni * 8, hipMemcpyHostToDevice, st));
    // the pad is READ: as item 0 of feature row k by the matrix-core kernel's padded feature steps (its A operand is 0 there, and
    // 0 x a NaN left over in the allocation is NaN: a whole item silently unlisted -- tests/test_gpu_topn_golden.py caught it)
    TOPN_TRY(hipMemsetAsync(reinterpret_cast<char *>(S.dU) + (size_t)dim * ni * 8, 0, 16, st));
    TOPN_TRY(hipMemcpyAsync(S.dV, v, (size_t)dim * nu * 8, hipMemcpyHostToDevice, st));
    if (ib) { TOPN_TRY(hipMalloc((void **)&S.dib, (size_t)ni * 8)); TOPN_TRY(hipMemcpyAsync(S.dib, ib, (size_t)ni * 8, hipMemcpyHostToDevice, st)); }
    if (ub) { TOPN_TRY(hipMalloc((void **)&S.dub, (size_t)nu * 8)); TOPN_TRY(hipMemcpyAsync(S.dub, ub, (size_t)nu * 8, hipMemcpyHostToDevice, st)); }
    const int64_t n_rated = rated_off ? rated_off[nu] : 0;
    if (n_rated > 0) {
        for (int j = 0; j < nu && S.ascending; ++j)
            for (int64_t q = rated_off[j] + 1; q < rated_off[j + 1]; ++q)
                if (rated_items[q] <= rated_items[q - 1]) { S.ascending = false; break; }
        TOPN_TRY(hipMalloc((void **)&S.d_off, ((size_t)nu + 1) * 8));
        TOPN_TRY(hipMalloc((void **)&S.d_rated, (size_t)n_rated * 4));
        TOPN_TRY(hipMemcpyAsync(S.d_off, rated_off, ((size_t)nu + 1) * 8, hipMemcpyHostToDevice, st));
        TOPN_TRY(hipMemcpyAsync(S.d_rated, rated_items, (size_t)n_rated * 4, hipMemcpyHostToDevice, st));
    }
    TOPN_TRY(hipStreamSynchronize(st));
    return MFX_OK;
}

// pos[j]: the user's position in the state (row of V, list of the CSR); label[j]: its id for the "item named like the
// user" rule (mf.py:162)
static int topn_query(mfx_topn_plan &S, const int32_t *pos, const int32_t *label, int n_sel, int n_cand, int N,
                      int32_t *out_items, double *out_scores, int level)
{
    // level 0: the fused path; 1: the users level 0 flagged, once more through the fused path (few users: their candidates are split
    // over up to 64 workgroups, so the items that met in one holder at level 0 -- the usual reason, ~8e-4 of the users at N = 10 --
    // meet different neighbours); 2: the sort path
    if (n_sel == 0) return MFX_OK;
    hipStream_t st = mfx::stream();
    DevBufs B;
    const int dim = S.k, ni = S.ni, kind = S.kind;
    const int n_pad = (n_sel + 4 * kUB - 1) / (4 * kUB) * (4 * kUB);
    double *dv, *dubs = nullptr, *d_scores;
    int32_t *d_pos, *d_label, *d_items;
    int64_t *d_lo = nullptr;
    TOPN_TRY(B.get(&dv, (size_t)dim * n_pad)); TOPN_TRY(B.get(&d_pos, (size_t)n_sel)); TOPN_TRY(B.get(&d_label, (size_t)n_sel));
    TOPN_TRY(B.get(&d_items, (size_t)n_sel * N)); TOPN_TRY(B.get(&d_scores, (size_t)n_sel * N));
    if (S.dub) TOPN_TRY(B.get(&dubs, (size_t)n_sel));
    TOPN_TRY(hipMemcpyAsync(d_pos, pos, (size_t)n_sel * 4, hipMemcpyHostToDevice, st));
    TOPN_TRY(hipMemcpyAsync(d_label, label, (size_t)n_sel * 4, hipMemcpyHostToDevice, st));
    k_gather_users<<<(n_pad + 255) / 256, 256, 0, st>>>(S.dV, S.dub, S.nu, dim, d_pos, n_sel, n_pad, dv, dubs);
    if (S.d_off) {                                  // the batch's lists: [lo[j], hi[j]) -> one array lo[0 .. 2 n_sel)
        TOPN_TRY(B.get(&d_lo, (size_t)2 * n_sel));
        k_batch_lists<<<(n_sel + 255) / 256, 256, 0, st>>>(S.d_off, d_pos, n_sel, d_lo, d_lo + n_sel);
    }

    // ---- fused path (N <= 16, every rated list ascending): no score matrix at all ---------------------------
    const char *ef = getenv("MFX_TOPN_FUSED");
    if (level < 2 && N <= kTopMaxN && S.ascending && !(ef && ef[0] == '0')) {
        int32_t *d_unsafe;
        TOPN_TRY(B.get(&d_unsafe, (size_t)n_sel));
        const char *fu = getenv("MFX_TOPN_FORCE_UNSAFE");          // test hook: every third user goes the long way
        const char *tv = getenv("MFX_TOPN_KERNEL");                // "strided": the first fused kernel; "tiled": the vector-f64 tiled kernel (A/B, tests)
        const int force = (fu && fu[0] == '1') ? 1 : 0;
        TOPN_TRY(hipMemsetAsync(d_unsafe, 0, (size_t)n_sel * 4, st));
        // few users: one workgroup per 32 users would walk all candidates alone (4.8 ms for ONE user at 50 K items);
        // the candidate tiles are split over up to 64 workgroups per user group and the lists merged
        const int groups = n_pad / (4 * kUB), n_it = (n_cand + kTI - 1) / kTI;
        const char *es = getenv("MFX_TOPN_SPLITS");
        int splits = es ? atoi(es) : 1024 / groups;
        if (splits > 64) splits = 64;
        if (splits > n_it) splits = n_it;
        if (splits < 1 || (tv && tv[0] == 's')) splits = 1;
        int32_t *p_items = d_items;
        double *p_scores = d_scores;
        double *p_drop = nullptr;
        if (splits > 1) {
            TOPN_TRY(B.get(&p_items, (size_t)n_sel * splits * N)); TOPN_TRY(B.get(&p_scores, (size_t)n_sel * splits * N));
            TOPN_TRY(B.get(&p_drop, (size_t)n_sel * splits));
        }
        if (tv && tv[0] == 's')
            k_topn_fused<<<(n_sel + kUB - 1) / kUB, 256, 0, st>>>(kind, dim, S.dU, dv, S.dib, dubs, S.mu, d_label, n_sel, n_pad, ni,
                                                                   n_cand, N, d_lo, d_lo ? d_lo + n_sel : nullptr, S.d_rated, force,
                                                                   d_items, d_scores, d_unsafe);
        else if (dim <= kMK && (size_t)ni * 128 < 0x7fffffffu && !(tv && tv[0] == 't'))
        {
            // scores on the matrix cores (approximate: they only select), then the selected scores again in the reference's order
            int32_t *a_items;                                       // the merged approximate lists
            double *a_scores, *a_drop;
            unsigned long long *d_un2;
            TOPN_TRY(B.get(&a_items, (size_t)n_sel * N)); TOPN_TRY(B.get(&a_scores, (size_t)n_sel * N)); TOPN_TRY(B.get(&a_drop, (size_t)n_sel));
            TOPN_TRY(B.get(&d_un2, (size_t)1));
            TOPN_TRY(hipMemsetAsync(d_un2, 0, 8, st));
#ifdef MFX_TOPN_KO
            { const char *ko = getenv("MFX_TOPN_KO"); const int kv = ko ? atoi(ko) : 0; TOPN_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_topn_ko), &kv, 4, 0, hipMemcpyHostToDevice, st)); }
#endif
            k_item_norm_max<<<(n_cand + 255) / 256, 256, 0, st>>>(dim, S.dU, ni, n_cand, d_un2);
            const int n_blk = (n_cand + 31) / 32;
            if (splits > n_blk) splits = n_blk;
            if (splits > 1 && !p_drop) TOPN_TRY(B.get(&p_drop, (size_t)n_sel * splits));
            if (splits > 1) {
                k_topn_mfma<<<dim3(groups, splits), 256, 0, st>>>(kind, dim, S.dU, dv, S.dib, dubs, S.mu, d_label, n_sel, n_pad, ni, n_cand,
                                                                  N, d_lo, d_lo ? d_lo + n_sel : nullptr, S.d_rated, d_un2, p_items, p_scores, p_drop, 0);
                k_topn_merge<<<n_sel, 64, 0, st>>>(p_items, p_scores, p_drop, splits, N, force, a_items, a_scores, d_unsafe, a_drop);
            } else {
                // Whole rounds of workgroups first, one workgroup per 32 users; the LAST, partial round's user groups with their candidates
                // split over the idle slots (3 125 groups on 512 slots = 6.1 rounds used to cost 7: 98 304 users 19.8 ms, 100 000 22.3 ms)
                int per_cu = 0, cus = 0, dev = 0;
                TOPN_TRY(hipGetDevice(&dev));
                TOPN_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                TOPN_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_topn_mfma, 256, 0));
                const int slots = per_cu * cus;
                const char *et = getenv("MFX_TOPN_TAIL");                    // "0": one launch (A/B, tests)
                int full = groups, rem = 0, tsplits = 1;
                if (slots > 0 && groups > slots && !(et && et[0] == '0')) {
                    rem = groups % slots;
                    tsplits = rem ? slots / rem : 1;
                    if (tsplits > 64) tsplits = 64;
                    if (tsplits > n_blk) tsplits = n_blk;
                    if (rem && tsplits >= 2) full = groups - rem; else rem = 0;
                }
                k_topn_mfma<<<dim3(full, 1), 256, 0, st>>>(kind, dim, S.dU, dv, S.dib, dubs, S.mu, d_label, n_sel, n_pad, ni, n_cand,
                                                          N, d_lo, d_lo ? d_lo + n_sel : nullptr, S.d_rated, d_un2, a_items, a_scores, a_drop, 0);
                if (rem) {
                    const int jb = full * 32, n_tail = n_sel - jb;           // (> 0: only the last group may be padded)
                    int32_t *t_items;
                    double *t_scores, *t_drop;
                    TOPN_TRY(B.get(&t_items, (size_t)rem * 32 * tsplits * N)); TOPN_TRY(B.get(&t_scores, (size_t)rem * 32 * tsplits * N));
                    TOPN_TRY(B.get(&t_drop, (size_t)rem * 32 * tsplits));
                    k_topn_mfma<<<dim3(rem, tsplits), 256, 0, st>>>(kind, dim, S.dU, dv, S.dib, dubs, S.mu, d_label, n_sel, n_pad, ni, n_cand,
                                                                    N, d_lo, d_lo ? d_lo + n_sel : nullptr, S.d_rated, d_un2, t_items, t_scores, t_drop, full);
                    k_topn_merge<<<n_tail, 64, 0, st>>>(t_items, t_scores, t_drop, tsplits, N, force, a_items + (size_t)jb * N,
                                                        a_scores + (size_t)jb * N, d_unsafe + jb, a_drop + jb);
                }
            }
            k_topn_exactify<<<n_sel, 64, 0, st>>>(kind, dim, S.dU, dv, S.dib, dubs, S.mu, n_sel, n_pad, ni, N, a_items, a_drop, d_un2, force,
                                                  d_items, d_scores, d_unsafe);
        }
        else
        {
            k_topn_tiled<<<dim3(groups, splits), 256, 0, st>>>(kind, dim, S.dU, dv, S.dib, dubs, S.mu, d_label, n_sel, n_pad, ni,
                                                               n_cand, N, d_lo, d_lo ? d_lo + n_sel : nullptr, S.d_rated, force,
                                                               p_items, p_scores, d_unsafe, p_drop);
            if (splits > 1) k_topn_merge<<<n_sel, 64, 0, st>>>(p_items, p_scores, p_drop, splits, N, force, d_items, d_scores, d_unsafe, nullptr);
        }
        TOPN_TRY(hipGetLastError());
        std::vector<int32_t> flag((size_t)n_sel);
        TOPN_TRY(hipMemcpyAsync(flag.data(), d_unsafe, (size_t)n_sel * 4, hipMemcpyDeviceToHost, st));
        TOPN_TRY(hipMemcpyAsync(out_items, d_items, (size_t)n_sel * N * 4, hipMemcpyDeviceToHost, st));
        TOPN_TRY(hipMemcpyAsync(out_scores, d_scores, (size_t)n_sel * N * 8, hipMemcpyDeviceToHost, st));
        TOPN_TRY(hipStreamSynchronize(st));
        std::vector<int32_t> rpos, rlab;
        std::vector<int> redo;
        for (int j = 0; j < n_sel; ++j)
            if (flag[(size_t)j]) { redo.push_back(j); rpos.push_back(pos[j]); rlab.push_back(label[j]); }
        g_topn_stats[0] += n_sel - (int64_t)redo.size();
        if (redo.empty()) return MFX_OK;
        // the flagged users once more as a batch of their own: through the fused path if that splits their candidates differently
        const int m = (int)redo.size();
        const int next = (level == 0 && !force && (size_t)m * 8 <= (size_t)n_sel) ? 1 : 2;
        if (next == 2) g_topn_stats[1] += (int64_t)m;
        std::vector<int32_t> it2((size_t)m * N);
        std::vector<double> sc2((size_t)m * N);
        int rc2 = topn_query(S, rpos.data(), rlab.data(), m, n_cand, N, it2.data(), sc2.data(), next);
        if (rc2) return rc2;
        for (int a = 0; a < m; ++a) {
            memcpy(out_items + (size_t)redo[(size_t)a] * N, it2.data() + (size_t)a * N, (size_t)N * 4);
            memcpy(out_scores + (size_t)redo[(size_t)a] * N, sc2.data() + (size_t)a * N, (size_t)N * 8);
        }
        return MFX_OK;
    }

    // ---- sort path: score matrix by chunks of users (it and its sorted copy stay below ~3 GB) -----------------
    int chunk = (int)(((size_t)1 << 27) / (size_t)n_cand);
    if (chunk < 1) chunk = 1;
    if (chunk > n_sel) chunk = n_sel;
    chunk = (chunk + kUB - 1) / kUB * kUB;
    const size_t cells = (size_t)chunk * n_cand;
    if (cells >= ((size_t)1 << 32)) return mfx::fail(MFX_E_ARG, "mfx_topn: n_cand too large");
    double *k_in, *k_out;
    int32_t *v_in, *v_out;
    unsigned int *d_seg;
    TOPN_TRY(B.get(&k_in, cells)); TOPN_TRY(B.get(&k_out, cells)); TOPN_TRY(B.get(&v_in, cells)); TOPN_TRY(B.get(&v_out, cells));
    TOPN_TRY(B.get(&d_seg, (size_t)chunk + 1));
    k_iota_offsets<<<(chunk + 256) / 256, 256, 0, st>>>(d_seg, chunk, (unsigned int)n_cand);
    size_t tmp_bytes = 0;
    TOPN_TRY(rocprim::segmented_radix_sort_pairs_desc(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (unsigned int)cells,
                                                      (unsigned int)chunk, d_seg, d_seg + 1, 0u, 64u, st));
    char *tmp;
    TOPN_TRY(B.get(&tmp, tmp_bytes));
    for (int jb = 0; jb < n_sel; jb += chunk) {
        const int nc = n_sel - jb < chunk ? n_sel - jb : chunk;
        k_scores<<<dim3((n_cand + 255) / 256, (nc + kUB - 1) / kUB), 256, 0, st>>>(kind, dim, S.dU, dv, S.dib, dubs, S.mu, d_label,
                                                                                   n_sel, n_pad, jb, nc, ni, n_cand, k_in, v_in);
        if (d_lo) k_clear_rated<<<dim3(4, nc), 256, 0, st>>>(d_lo, d_lo + n_sel, S.d_rated, jb, nc, n_cand, k_in);
        TOPN_TRY(rocprim::segmented_radix_sort_pairs_desc(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (unsigned int)((size_t)nc * n_cand),
                                                          (unsigned int)nc, d_seg, d_seg + 1, 0u, 64u, st));
        k_take_top<<<nc, 64, 0, st>>>(k_out, v_out, n_cand, N, jb, d_items, d_scores);
    }
    TOPN_TRY(hipGetLastError());
    TOPN_TRY(hipMemcpyAsync(out_items, d_items, (size_t)n_sel * N * 4, hipMemcpyDeviceToHost, st));
    TOPN_TRY(hipMemcpyAsync(out_scores, d_scores, (size_t)n_sel * N * 8, hipMemcpyDeviceToHost, st));
    TOPN_TRY(hipStreamSynchronize(st));
    return MFX_OK;
}

extern "C" int mfx_topn_batch(int kind, int dim, const double *u, const double *v_sel, const double *ib,
                              const double *ub_sel, double mu, int ni, const int32_t *users, int n_sel, int n_cand,
                              int N, const int64_t *rated_off, const int32_t *rated_items, int32_t *out_items,
                              double *out_scores)
{
    if (!u || !v_sel || !users || !out_items || !out_scores || dim < 1 || ni <= 0 || n_sel < 0 || n_cand < 1 ||
        n_cand > ni || N < 1 || kind < 0 || kind > 4 || ((kind >= 1 && kind <= 3) && (!ib || !ub_sel)) ||
        (rated_off && !rated_items && n_sel > 0 && rated_off[n_sel] > 0))
        return mfx::fail(MFX_E_ARG, "mfx_topn_batch: bad argument");
    int rc = mfx::ensure_init();
    if (rc) return rc;
    if (n_sel == 0) return MFX_OK;
    if (rated_off) {
        if (rated_off[0] != 0) return mfx::fail(MFX_E_ARG, "mfx_topn_batch: rated_off[0] must be 0");
        for (int j = 0; j < n_sel; ++j)
            if (rated_off[j + 1] < rated_off[j]) return mfx::fail(MFX_E_ARG, "mfx_topn_batch: rated_off must ascend");
    }
    mfx_topn_plan S;                                           // for this call: the batch's users are "all users"
    if ((rc = topn_state_init(S, kind, dim, u, v_sel, ib, ub_sel, mu, n_sel, ni, rated_off, rated_items))) return rc;
    std::vector<int32_t> pos((size_t)n_sel);
    for (int j = 0; j < n_sel; ++j) pos[(size_t)j] = j;
    return topn_query(S, pos.data(), users, n_sel, n_cand, N, out_items, out_scores, 0);
}

extern "C" mfx_topn_plan *mfx_topn_plan_create(int kind, int dim, const double *u, const double *v, const double *ib,
                                               const double *ub, double mu, int nu, int ni, const int64_t *rated_off,
                                               const int32_t *rated_items)
{
    if (!u || !v || dim < 1 || nu <= 0 || ni <= 0 || kind < 0 || kind > 4 || ((kind >= 1 && kind <= 3) && (!ib || !ub)) ||
        (rated_off && (rated_off[0] != 0 || (!rated_items && rated_off[nu] > 0)))) {
        mfx::fail(MFX_E_ARG, "mfx_topn_plan_create: bad argument");
        return nullptr;
    }
    if (rated_off)
        for (int j = 0; j < nu; ++j)
            if (rated_off[j + 1] < rated_off[j]) { mfx::fail(MFX_E_ARG, "mfx_topn_plan_create: rated_off must ascend"); return nullptr; }
    if (mfx::ensure_init()) return nullptr;
    mfx_topn_plan *p = new (std::nothrow) mfx_topn_plan();
    if (!p) { mfx::fail(MFX_E_NOMEM, "mfx_topn_plan_create: out of host memory"); return nullptr; }
    if (topn_state_init(*p, kind, dim, u, v, ib, ub, mu, nu, ni, rated_off, rated_items)) { delete p; return nullptr; }
    return p;
}

extern "C" int mfx_topn_plan_query(mfx_topn_plan *p, const int32_t *users, int n_sel, int n_cand, int N, int32_t *out_items,
                                   double *out_scores)
{
    if (!p || n_sel < 0 || (n_sel > 0 && (!users || !out_items || !out_scores)) || n_cand < 1 || n_cand > p->ni || N < 1)
        return mfx::fail(MFX_E_ARG, "mfx_topn_plan_query: bad argument");
    for (int j = 0; j < n_sel; ++j)
        if (users[j] < 0 || users[j] >= p->nu) return mfx::fail(MFX_E_RANGE, "mfx_topn_plan_query: user %d outside [0,%d)", users[j], p->nu);
    return topn_query(*p, users, users, n_sel, n_cand, N, out_items, out_scores, 0);
}

extern "C" void mfx_topn_plan_destroy(mfx_topn_plan *p)
{
    if (p && mfx::ready()) (void)hipStreamSynchronize(mfx::stream());
    delete p;
}

// stats[0] = users answered by the fused kernel, stats[1] = users it flagged and the sort path redid (process totals)
extern "C" int mfx_topn_stats(int64_t stats[2])
{
    if (!stats) return mfx::fail(MFX_E_ARG, "mfx_topn_stats: NULL");
    stats[0] = g_topn_stats[0].load();
    stats[1] = g_topn_stats[1].load();
    return MFX_OK;
}
