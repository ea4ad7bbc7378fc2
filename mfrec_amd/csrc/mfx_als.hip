// K3: ALS / WRMF for implicit feedback -- als_wrmf, mfrec/lib/als_implicit.pyx:211-352.
//
// Per half-epoch (solve X from Y):      G = Y^T Y                      (k x k, all rows of Y)
//   per active row j with neighbours N(j):   A = G + c * Y_N^T Y_N + lambda I
//                                            b = (1 + c) * sum_{n in N(j)} y_n
//                                            x_j = A^{-1} b
// The reference forms A with scalar loops and calls np.linalg.inv per row; here one
// workgroup owns one row: neighbour rows are gathered into LDS in chunks, Y_N^T Y_N is
// accumulated on the f64 matrix cores (v_mfma_f64_16x16x4_f64, 16x16 tiles of A in
// registers), A is Cholesky-factored in LDS and x_j comes from two triangular solves.
// (SPD solve instead of an explicit inverse: ~1e-13 relative difference, pinned at 1e-9.)
//
// Device layout: factors row-major [n][k] so that a neighbour row is one contiguous
// k*8-byte gather; k is padded to KP = 16*ceil(k/16) inside LDS only.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mfx_common.h"

namespace mfx {
bool comm_active();
int comm_rank();
int comm_world();
int comm_allgather_f64(double *buf, size_t count);
}  // namespace mfx

namespace {

using mfx::ceil_div;
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int CH = 16;                 // neighbour rows per LDS chunk (multiple of 4)

__host__ __device__ inline int ych_stride(int KP) { return (KP % 32 == 16) ? KP : KP + 16; }

// ------------------------------------------------------------------------------------
// Gramian: partial[b] = sum over a slab of rows of y y^T (full k x k, row-major, KP padded)
// ------------------------------------------------------------------------------------
template <int NT>   // NT = KP/16 tiles per dimension; block = 256 threads (4 waves)
__global__ __launch_bounds__(256) void k_gram_partial(const double *__restrict__ Y, int ny, int k,
                                                      int rows_per_block, double *partial)
{
    constexpr int KP = NT * 16;
    constexpr int TPW = (NT * NT + 3) / 4;           // tiles per wave
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int S = ych_stride(KP);
    double *ych = lds;                               // [CH][S]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(ny, r0 + rows_per_block);

    double4_t acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};

    for (int c0 = r0; c0 < r1; c0 += CH) {
        for (int e = threadIdx.x; e < CH * KP; e += 256) {
            const int n = e / KP, f = e % KP;
            const int row = c0 + n;
            ych[n * S + f] = (row < r1 && f < k) ? Y[(size_t)row * k + f] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int n0 = 0; n0 < CH; n0 += 4) {
            const double *base = ych + (n0 + (lane >> 4)) * S + (lane & 15);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tile = wid * TPW + t;
                if (tile < NT * NT) {
                    const int I = tile / NT, J = tile % NT;
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[I * 16], base[J * 16], acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    double *out = partial + (size_t)blockIdx.x * KP * KP;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wid * TPW + t;
        if (tile < NT * NT) {
            const int I = tile / NT, J = tile % NT;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = I * 16 + (lane >> 4) + 4 * q, c = J * 16 + (lane & 15);
                out[r * KP + c] = acc[t][q];
            }
        }
    }
}

// G[e] = sum_b partial[b][e] in block order (deterministic)
__global__ __launch_bounds__(256) void k_gram_reduce(const double *partial, int nblocks, int n,
                                                     double *G)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
#pragma unroll 16                         // the adds stay one ordered chain; the loads no longer wait for them
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * n + e];
    G[e] = s;
}

// ------------------------------------------------------------------------------------
// per-row assemble + solve, one 256-thread workgroup per active row: symmetric-half MFMA
// tiles, software-prefetched neighbour gather, MFMA-blocked Cholesky (16-wide panels) with
// look-ahead, register back substitution.
// ------------------------------------------------------------------------------------
constexpr int CH2 = 32;                // neighbour rows per chunk

// Diagnostic builds only (-DMFX_ALS_STAMPS): s_memtime at phase boundaries of the first rows.
// Diagnostic builds only (-DMFX_ALS_KO): knock-out bits in `tune` remove ONE phase of the row kernel (wrong results) -- what
// that phase costs in THROUGHPUT under the real load, as opposed to its share of one row's latency (tools/als_ab.py)
#ifdef MFX_ALS_KO
#define ALS_KO(bit) ((tune & (bit)) != 0)
#else
#define ALS_KO(bit) false
#endif
enum { KO_BACK = 4, KO_FACTOR = 8, KO_SOLVE = 16, KO_SYRK = 32, KO_GATHER = 64, KO_TRAIL = 128, KO_BSUM = 256, KO_REST = 512 };

#ifdef MFX_ALS_STAMPS
#ifndef MFX_ALS_STAMP_BASE
#define MFX_ALS_STAMP_BASE 0         // first stamped workgroup: a window in the MIDDLE of a launch shows the steady state
#endif
__device__ unsigned long long g_als_stamps[4096 * 16];
#define ALS_STAMPED() ((unsigned)(blockIdx.x - (MFX_ALS_STAMP_BASE)) < 4096u)
#define ALS_STAMP(i)                                                                        \
    do {                                                                                    \
        __syncthreads();                                                                    \
        if (threadIdx.x == 0 && ALS_STAMPED()) {                                            \
            unsigned long long t_;                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
            g_als_stamps[(blockIdx.x - (MFX_ALS_STAMP_BASE)) * 16 + (i)] = t_;               \
        }                                                                                   \
    } while (0)
// wave-local stamp (no barrier): thread 0 only
#define ALS_STAMP_W0(i)                                                                     \
    do {                                                                                    \
        if (threadIdx.x == 0 && ALS_STAMPED()) {                                            \
            unsigned long long t_;                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
            g_als_stamps[(blockIdx.x - (MFX_ALS_STAMP_BASE)) * 16 + (i)] = t_;              \
        }                                                                                   \
    } while (0)
// where the workgroup ran: HW_ID and XCC_ID (slots 12, 13)
#define ALS_STAMP_WHERE()                                                                   \
    do {                                                                                    \
        if (threadIdx.x == 0 && ALS_STAMPED()) {                                            \
            unsigned h_, x_;                                                                \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h_));                 \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_));                \
            g_als_stamps[(blockIdx.x - (MFX_ALS_STAMP_BASE)) * 16 + 12] = h_;               \
            g_als_stamps[(blockIdx.x - (MFX_ALS_STAMP_BASE)) * 16 + 13] = x_ & 15u;         \
        }                                                                                   \
    } while (0)
#ifdef MFX_ALS_STAMP_SYRK           // slots 5..8, 14: inside the gather + SYRK phase (instead of the panel-step stamps)
#define ALS_STAMP_SYRK_W0(i) ALS_STAMP_W0(i)
#define ALS_STAMP_CHOL_W0(i) do { } while (0)
#else
#define ALS_STAMP_SYRK_W0(i) do { } while (0)
#define ALS_STAMP_CHOL_W0(i) ALS_STAMP_W0(i)
#endif
#else
#define ALS_STAMP_SYRK_W0(i) do { } while (0)
#define ALS_STAMP_CHOL_W0(i) do { } while (0)
#define ALS_STAMP(i) do { } while (0)
#define ALS_STAMP_W0(i) do { } while (0)
#define ALS_STAMP_WHERE() do { } while (0)
#endif
#ifdef MFX_ALS_STAMPS
#define ALS_STAMP_ROW(i)                                                                    \
    do {                                                                                    \
        if (lane == 0 && (unsigned)(row - (MFX_ALS_STAMP_BASE)) < 4096u) {                  \
            unsigned long long t_;                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
            g_als_stamps[(row - (MFX_ALS_STAMP_BASE)) * 16 + (i)] = t_;                      \
        }                                                                                   \
    } while (0)
#else
#define ALS_STAMP_ROW(i) do { } while (0)
#endif

__device__ __forceinline__ void tile_ij(int tt, int &I, int &J)
{
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= tt) ++i;
    I = i;
    J = tt - i * (i + 1) / 2;
}

__device__ __forceinline__ double readlane_f64(double x, int l)      // l must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to ~1 ulp: hardware estimate + two Newton steps (the result only has to meet
// the 1e-9 ALS tolerance; a correctly rounded sqrt/divide pair costs ~4x the instructions)
__device__ __forceinline__ double rsqrt_nr(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}

// ---- f64 DPP row_newbcast (gfx90a+): operand = lane N of the caller's own 16-lane row ----------
// One VALU instruction replaces two v_readlane + the arithmetic.  All 64 lanes must be active.
// "s_nop 1" covers the VALU-write -> DPP-read hazard, which the compiler cannot see inside asm.
template <int N>
__device__ __forceinline__ double dpp_bcast(double v)
{
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "=v"(r) : "v"(v), "n"(N));
    return r;
}
// acc -= l[lane N] * a  (fused).  NOP = false only where l was not written by the two
// preceding instructions (a run of updates with the same l).
template <int N, bool NOP = true>
__device__ __forceinline__ void dpp_fnma(double &acc, double l, double a)
{
    if constexpr (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(l), "v"(a), "n"(N));
    else
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(l), "v"(a), "n"(N));
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
// 1/sqrt(d), fused Newton steps
__device__ __forceinline__ double rsqrt_fma(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    double e = __builtin_fma(-(d * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-(d * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    return y;
}

// offset (doubles) of row r of the packed lower-block-triangular A (r == 16*NT: border row)
template <int NT>
__host__ __device__ constexpr int als_row_off(int r)
{
    const int I = r >> 4;
    return 128 * I * (I + 1) + 32 * I + (r & 15) * (16 * (I + 1) + 2);
}

// waves per SIMD (= workgroups per CU) the register allocation must leave room for: LDS admits
// 2 workgroups at k = 128 and more below
// NW = waves per workgroup: 4 for k > 32; ONE wave per row for k <= 32, where a 256-thread
// workgroup would idle through the 16x16 / 32x32 factorisation (rows are then independent
// single-wave workgroups: barriers degenerate to wave barriers, many rows share a CU)
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 1 ? 2 : (NT >= 5 ? 2 : 3))) void k_als_rows2(const double *__restrict__ Y, int k,
                                                   const double *__restrict__ G,
                                                   const int64_t *__restrict__ off,
                                                   const int32_t *__restrict__ col, double c_pos,
                                                   double one_plus_c, double lambda, double *X,
                                                   int *status, int tune)
{
    constexpr int KP = NT * 16;
    // A is stored PACKED: only the lower block triangle.  A row of block-row I holds its
    // 16*(I+1) columns + 2 pad doubles (row stride = 2 or 18 mod 32 doubles: conflict-free
    // MFMA operand reads of the panels); 77 KB instead of 134 KB at k = 128, so two
    // workgroups share a CU and one's Cholesky overlaps the other's gather + MFMA phase.
    constexpr int NTT = NT * (NT + 1) / 2;           // lower-triangle tiles
    constexpr int TPW = (NTT + NW - 1) / NW;
    constexpr int NTHR = 64 * NW;
    constexpr int PF = CH2 * KP / NTHR;              // gathered doubles per thread per chunk
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int S = ych_stride(KP);
    constexpr int BORDER = als_row_off<NT>(KP);      // right-hand side row
    constexpr int A_ELEMS = BORDER + KP + 2;
    const int stage_elems = CH2 * S;
    const int region = (A_ELEMS > stage_elems ? A_ELEMS : stage_elems) + 1 & ~1;
    double *A = lds;                                 // packed lower triangle + border row
    double *ych = lds;                               // aliases A while Y_N^T Y_N is accumulated
    double *dinv = lds + region;                     // [KP]  1 / L[c][c]
    int *cidx = (int *)(dinv + KP);                  // [2][CH2] neighbour ids of two chunks
    int *bad = cidx + 2 * CH2;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = blockIdx.x;
    const int64_t start = off[j];
    const int span = (int)(off[j + 1] - start);
    double *xrow = X + (size_t)j * k;
    if (span == 0) {                                 // b = 0  =>  x = 0 (als_implicit.pyx:298-302)
        for (int f = tid; f < k; f += NTHR) xrow[f] = 0.0;
        return;
    }
    if (tid == 0) *bad = 0;

    int tI[TPW], tJ[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        tI[t] = 0; tJ[t] = 0;
        if (t * NW + wid < NTT) tile_ij(t * NW + wid, tI[t], tJ[t]);
    }
    // accumulators start at G / c, so that A = c * acc needs no second pass over G (c != 0)
    const bool from_g = (c_pos != 0.0);
    const double inv_c = from_g ? 1.0 / c_pos : 0.0;
    double4_t acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = tI[t] * 16 + (lane >> 4) + 4 * q, c = tJ[t] * 16 + (lane & 15);
            acc[t][q] = (from_g && t * NW + wid < NTT) ? G[r * KP + c] * inv_c : 0.0;
        }
    }
    double bsum = 0.0;

    ALS_STAMP(0);
    // ---- Y_N^T Y_N and b: chunked gather, next chunk in flight under the MFMAs -----------
    const int nchunks = (span + CH2 - 1) / CH2;
    if (tid < CH2) cidx[tid] = (tid < span) ? col[start + tid] : -1;
    __syncthreads();
    // feature PAIRS: one 16-byte load per pair when k is even (see k_als_rows3), 16-byte staging stores
    typedef double v2f64_a8 __attribute__((ext_vector_type(2), aligned(8)));
    constexpr int PF2 = PF / 2;
    static_assert(PF % 2 == 0, "pairs");
    const bool keven = (k & 1) == 0;
    double2 pre[PF2];
    auto fetch = [&](int cbase) {
#pragma unroll
        for (int i = 0; i < PF2; ++i) {
            const int e = 2 * (tid + NTHR * i), n = e / KP, f = e % KP;
            const int id = cidx[cbase + n];
            double2 v = make_double2(0.0, 0.0);
            if (id >= 0 && f < k) {
                const double *src = Y + (size_t)id * k + f;
                if (keven) {
                    const v2f64_a8 w = *reinterpret_cast<const v2f64_a8 *>(src);
                    v = make_double2(w.x, w.y);
                } else {
                    v.x = src[0];
                    if (f + 1 < k) v.y = src[1];
                }
            }
            pre[i] = v;
        }
    };
    fetch(0);
    int cnext = (tid < CH2 && CH2 + tid < span) ? col[start + CH2 + tid] : -1;
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int i = 0; i < PF2; ++i) {
            const int e = 2 * (tid + NTHR * i), n = e / KP, f = e % KP;
            *reinterpret_cast<double2 *>(ych + n * S + f) = pre[i];
        }
        const int nb = ((c + 1) & 1) * CH2;
        if (tid < CH2) cidx[nb + tid] = cnext;
        __syncthreads();
        if (c + 1 < nchunks) {
            fetch(nb);
            const int t2 = (c + 2) * CH2 + tid;
            cnext = (tid < CH2 && t2 < span) ? col[start + t2] : -1;
        }
        if (tid < KP) {
#pragma unroll
            for (int n = 0; n < CH2; ++n) bsum += ych[n * S + tid] * one_plus_c;
        }
        // a ragged last chunk (span = 100: 4 valid rows of 32) skips its all-zero 4-row steps
        const int nv = span - c * CH2, nv4 = nv >= CH2 ? CH2 : ((nv + 3) & ~3);
#pragma unroll
        for (int n0 = 0; n0 < CH2; n0 += 4) {
            if (n0 < nv4) {                                       // wave-uniform
                const double *base = ych + (n0 + (lane >> 4)) * S + (lane & 15);
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    if (t * NW + wid < NTT)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[tI[t] * 16], base[tJ[t] * 16],
                                                                      acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    ALS_STAMP(1);
    // ---- A = G + c * acc + lambda I (identity on the padding), b as border row -----------
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (t * NW + wid < NTT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = tI[t] * 16 + (lane >> 4) + 4 * q, c = tJ[t] * 16 + (lane & 15);
                double a;
                if (r < k && c < k) {
                    a = from_g ? acc[t][q] * c_pos : G[r * KP + c];
                    if (r == c) a += lambda;
                } else {
                    a = (r == c) ? 1.0 : 0.0;
                }
                A[als_row_off<NT>(r) + c] = a;
            }
        }
    }
    if (tid < KP) A[BORDER + tid] = (tid < k) ? bsum : 0.0;
    __syncthreads();

    ALS_STAMP(2);
    // ---- blocked Cholesky of the bordered matrix, 16-wide panels, with look-ahead: while
    //      waves 1-3 apply panel Jb to the trailing tiles, wave 0 updates the next diagonal
    //      tile first and factors it, so the serial 16-step block factorisation is hidden -----
    auto factor_diag = [&](int c0) {
        // lane r (< 16) holds row r of the 16x16 block in registers; right-looking by columns
        const int r = lane & 15;
        double d[16];
        const int drow = als_row_off<NT>(c0 + r) + c0;
#pragma unroll
        for (int c = 0; c < 16; ++c) d[c] = A[drow + c];
        double myinv = 1.0;
        bool ok = true;
        // DPP row broadcasts instead of v_readlane (all 64 lanes active: four copies of the block); the
        // next pivot's rsqrt chain runs under the remaining updates of the current step.  Only the
        // factorisation: the panel solve keeps reading L11 from LDS (its DPP form costs 64 VGPRs).
        double pj = dpp_bcast<0>(d[0]);
        double inv = rsqrt_fma(pj);
        static_for<0, 16>([&](auto JJ) {
            constexpr int jj = JJ;
            ok = ok && (pj > 0.0);
            if (r == jj) myinv = inv;
            d[jj] *= inv;
            if constexpr (jj < 15) {
                dpp_fnma<jj + 1>(d[jj + 1], d[jj], d[jj]);
                pj = dpp_bcast<jj + 1>(d[jj + 1]);
                inv = rsqrt_fma(pj);
                static_for<jj + 2, 16>([&](auto C) {
                    constexpr int c = C;
                    dpp_fnma<c, false>(d[c], d[jj], d[jj]);
                });
            }
        });
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c <= r) A[drow + c] = d[c];
            dinv[c0 + r] = myinv;
        }
        if (!ok && lane == 0) *bad = 1;
    };
    auto trailing_tile = [&](int Jb, int tt) {
        const int c0 = Jb * 16;
        int I, J2;
        tile_ij(tt, I, J2);
        I += Jb + 1;
        J2 += Jb + 1;
        double4_t cacc;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            cacc[q] = A[als_row_off<NT>(I * 16 + (lane >> 4) + 4 * q) + J2 * 16 + (lane & 15)];
        const double *pa = A + als_row_off<NT>(I * 16 + (lane & 15)) + c0 + (lane >> 4);
        const double *pb = A + als_row_off<NT>(J2 * 16 + (lane & 15)) + c0 + (lane >> 4);
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4)
            cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[s4], pb[s4], cacc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            A[als_row_off<NT>(I * 16 + (lane >> 4) + 4 * q) + J2 * 16 + (lane & 15)] = cacc[q];
    };

    if (wid == 0) factor_diag(0);
    __syncthreads();
    for (int Jb = 0; Jb < NT; ++Jb) {
        const int c0 = Jb * 16;
        // (b) panel: rows below the block and the border row:  X = A21 * L11^{-T}
        {
            const int pr = c0 + 16 + tid;            // physical row; pr == KP is the border row
            if (pr <= KP) {
                const int prow = als_row_off<NT>(pr) + c0;
                const int lrow = als_row_off<NT>(c0) + c0;       // L11; its rows share one stride
                const int lstride = 16 * (Jb + 1) + 2;
                double a[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = A[prow + c];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    double x = a[jj];
#pragma unroll
                    for (int m = 0; m < jj; ++m) x -= a[m] * A[lrow + jj * lstride + m];
                    a[jj] = x * dinv[c0 + jj];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) A[prow + c] = a[c];
            }
        }
        __syncthreads();
        // (c) trailing update: border row (VALU), then the lower tiles on the matrix cores
        {
            const int c = c0 + 16 + tid;
            if (c < KP) {
                double sum = 0.0;
                const int crow = als_row_off<NT>(c) + c0;
#pragma unroll
                for (int m = 0; m < 16; ++m) sum += A[BORDER + c0 + m] * A[crow + m];
                A[BORDER + c] -= sum;
            }
        }
        {
            const int nt = NT - Jb - 1;              // trailing tile rows
            const int ntt = nt * (nt + 1) / 2;
            if (NW == 1) {
                for (int tt = 0; tt < ntt; ++tt) trailing_tile(Jb, tt);
                if (ntt > 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    factor_diag(c0 + 16);
                }
            } else if (wid == 0) {
                if (ntt > 0) {
                    trailing_tile(Jb, 0);            // tile (Jb+1, Jb+1) ...
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    factor_diag(c0 + 16);            // ... factored right away (look-ahead)
                }
            } else {
                for (int tt = wid; tt < ntt; tt += NW - 1) trailing_tile(Jb, tt);
            }
        }
        __syncthreads();
    }
    if (*bad) {
        if (tid == 0) atomicExch(status, j + 1);
        return;
    }
    ALS_STAMP(3);

    // ---- back substitution L^T x = z (z = border row) by wave 0, x in registers; the rows
    //      of L and the reciprocal pivots of 16 steps are fetched ahead of the serial chain ---
    if (wid == 0) {
        double z0 = A[BORDER + (lane < KP ? lane : 0)];
        double z1 = (KP > 64) ? A[BORDER + 64 + (lane < KP - 64 ? lane : 0)] : 0.0;
        if (KP <= 64 && lane >= KP) z0 = 0.0;
        if (KP > 64 && lane >= KP - 64) z1 = 0.0;
        constexpr int PB = 8;                        // steps fetched ahead (register budget)
        for (int cb = KP - PB; cb >= 0; cb -= PB) {
            double lo[PB], hi[KP > 64 ? PB : 1], dv[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int c = cb + i;
                const int crow = als_row_off<NT>(c);
                lo[i] = (lane < c) ? A[crow + lane] : 0.0;
                if constexpr (KP > 64) hi[i] = (lane + 64 < c) ? A[crow + lane + 64] : 0.0;
                dv[i] = dinv[c];
            }
#pragma unroll
            for (int i = PB - 1; i >= 0; --i) {
                const int c = cb + i;
                double x;
                if (cb >= 64) {                      // wave-uniform: pivots owned by z1
                    x = readlane_f64(z1, c - 64) * dv[i];
                    if (lane == c - 64) z1 = x;
                } else {
                    x = readlane_f64(z0, c) * dv[i];
                    if (lane == c) z0 = x;
                }
                z0 -= lo[i] * x;
                if constexpr (KP > 64) z1 -= hi[i] * x;
            }
        }
        if (lane < k) xrow[lane] = z0;
        if (lane + 64 < k) xrow[lane + 64] = z1;
    }
    ALS_STAMP(4);
}

// ------------------------------------------------------------------------------------
// rows3: the same algorithm with the lower-triangle tiles of A kept IN REGISTERS (MFMA C layout)
// through the whole factorisation.  LDS then only holds the current 16-wide panel (Xs), the
// factored diagonal blocks (Dg) and the right-hand side: 37 KB instead of 77 KB at k = 128, so
// three workgroups share a CU and the serial phases of one row hide behind the MFMA phases of
// the others.  Wave 0 owns the diagonal tiles (it also does the serial block factorisation),
// waves 1-3 own the off-diagonal tiles column by column.  Used for k > 64.
// ------------------------------------------------------------------------------------
// Which wave holds which 16x16 tile of the lower block triangle.  Role 0: the diagonal tiles.
// Roles 1-3: whole block rows (tiles of a row share their A operand), rows dealt greedily,
// longest first, to the lightest role -- NT = 8: {7,2,1} {6,3} {5,4}.
template <int NT>
struct Rows3Map {
    static constexpr int NOFF = NT * (NT - 1) / 2;
    static constexpr int TS = NT > (NOFF + 2) / 3 + 1 ? NT : (NOFF + 2) / 3 + 1;
    int I[4][TS] = {}, J[4][TS] = {}, n[4] = {};
    constexpr Rows3Map()
    {
        for (int s = 0; s < NT; ++s) { I[0][s] = s; J[0][s] = s; }
        n[0] = NT;
        for (int r = NT - 1; r >= 1; --r) {
            int g = 1;
            for (int w = 2; w <= 3; ++w) if (n[w] < n[g]) g = w;
            for (int c = 0; c < r; ++c) { I[g][n[g]] = r; J[g][n[g]] = c; ++n[g]; }
        }
    }
    constexpr int slots(int role) const { return n[role]; }
};

template <int NT, int ROLE>
__device__ __forceinline__ void als_rows3_body(const double *__restrict__ Y, int k,
                                               const double *__restrict__ G,
                                               const int64_t *__restrict__ off,
                                               const int32_t *__restrict__ col, double c_pos,
                                               double one_plus_c, double lambda, double *X,
                                               int *status, int tune)
{
    constexpr int KP = NT * 16;
    constexpr Rows3Map<NT> TM{};
    constexpr int TS = TM.slots(ROLE);               // tile slots of this wave
    constexpr int wid = ROLE;
    constexpr int PF = CH2 * KP / 256;
    constexpr int XS = 18;                           // row stride of Xs / Dg: conflict-free MFMA operands
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int S = ych_stride(KP);
    const int stage_elems = CH2 * S;
    const int region = (stage_elems > 2 * KP * XS ? stage_elems : 2 * KP * XS);
    double *ych = lds;                               // SYRK staging; later:
    double *Xs = lds;                                //   [KP][XS] current panel (rows below the block)
    double *Dg = lds + KP * XS;                      //   [KP][XS] factored diagonal blocks, block J at rows 16J..
    double *dinv = lds + region;                     // [KP]
    double *bz = dinv + KP;                          // [KP] right-hand side -> z -> x
    double *scratch = bz + KP;                       // [4][16]
    int *cidx = (int *)(scratch + 64);               // [2][CH2]
    int *bad = cidx + 2 * CH2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int j = blockIdx.x;
    const int64_t start = off[j];
    const int span = (int)(off[j + 1] - start);
    double *xrow = X + (size_t)j * k;
    if (span == 0) {
        for (int f = tid; f < k; f += 256) xrow[f] = 0.0;
        return;
    }
    if (tid == 0) *bad = 0;

#define sI(s) (TM.I[ROLE][s])
#define sJ(s) (TM.J[ROLE][s])
    const bool from_g = (c_pos != 0.0);
    const double inv_c = from_g ? 1.0 / c_pos : 0.0;
    double4_t acc[TS > 0 ? TS : 1];
#pragma unroll
    for (int s = 0; s < TS; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = sI(s) * 16 + (lane >> 4) + 4 * q, c = sJ(s) * 16 + (lane & 15);
            acc[s][q] = from_g ? G[r * KP + c] * inv_c : 0.0;
        }
    }
    double bsum = 0.0;

    ALS_STAMP(0);
    ALS_STAMP_WHERE();
    // ---- Y_N^T Y_N and b (identical to rows2) ----------------------------------------------------
    const int nchunks = (span + CH2 - 1) / CH2;
    if (tid < CH2) cidx[tid] = (tid < span) ? col[start + tid] : -1;
    __syncthreads();
    ALS_STAMP_SYRK_W0(14);
    // every thread moves PF / 2 feature PAIRS per chunk: one 16-byte load per pair when k is even
    // (rows are then 16-byte aligned; 8-byte accesses run at 0.54-0.70x the 16-byte rate), two scalar
    // loads otherwise; the staging stores are 16 bytes either way (S and f are even)
    typedef double v2f64_a8 __attribute__((ext_vector_type(2), aligned(8)));
    constexpr int PF2 = PF / 2;
    static_assert(PF % 2 == 0, "pairs");
    const bool keven = (k & 1) == 0;
    double2 pre[PF2];
    auto fetch = [&](int cbase) {
#pragma unroll
        for (int i = 0; i < PF2; ++i) {
            const int e = 2 * (tid + 256 * i), n = e / KP, f = e % KP;
            const int id = cidx[cbase + n];
            double2 v = make_double2(0.0, 0.0);
            if (id >= 0 && f < k && !ALS_KO(KO_GATHER)) {
                const double *src = Y + (size_t)id * k + f;
                if (keven) {
                    const v2f64_a8 w = *reinterpret_cast<const v2f64_a8 *>(src);
                    v = make_double2(w.x, w.y);
                } else {
                    v.x = src[0];
                    if (f + 1 < k) v.y = src[1];
                }
            }
            pre[i] = v;
        }
    };
    fetch(0);
    int cnext = (tid < CH2 && CH2 + tid < span) ? col[start + CH2 + tid] : -1;
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int i = 0; i < PF2; ++i) {
            const int e = 2 * (tid + 256 * i), n = e / KP, f = e % KP;
            *reinterpret_cast<double2 *>(ych + n * S + f) = pre[i];
        }
        const int nb = ((c + 1) & 1) * CH2;
        if (tid < CH2) cidx[nb + tid] = cnext;
        __syncthreads();
        if (c < 4) ALS_STAMP_SYRK_W0(5 + c);
        if (c + 1 < nchunks) {
            fetch(nb);
            const int t2 = (c + 2) * CH2 + tid;
            cnext = (tid < CH2 && t2 < span) ? col[start + t2] : -1;
        }
        if (tid < KP && !ALS_KO(KO_BSUM)) {
#pragma unroll
            for (int n = 0; n < CH2; ++n) bsum += ych[n * S + tid] * one_plus_c;
        }
        // a ragged last chunk (span = 100: 4 valid rows of 32) skips its all-zero 4-row steps
        // (round 5: `b` summed from the gather registers and straight-line 4-row steps for full chunks were measured here -- 27 against
        //  18 ms: the kernel sits at 165 of its 168 registers and any addition spills; profiles/r05_als_diagnosis.md)
        const int nv = span - c * CH2, nv4 = nv >= CH2 ? CH2 : ((nv + 3) & ~3);
#pragma unroll
        for (int n0 = 0; n0 < CH2; n0 += 4) {
            if (n0 < nv4 && !ALS_KO(KO_SYRK)) {                   // wave-uniform
                const double *base = ych + (n0 + (lane >> 4)) * S + (lane & 15);
#pragma unroll
                for (int s = 0; s < TS; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[sI(s) * 16], base[sJ(s) * 16], acc[s], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    ALS_STAMP(1);
    // ---- A = c * acc + lambda I (identity on the padding), in registers; b -> LDS -----------------
#pragma unroll
    for (int s = 0; s < TS; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = sI(s) * 16 + (lane >> 4) + 4 * q, c = sJ(s) * 16 + (lane & 15);
            double a;
            if (r < k && c < k) {
                a = from_g ? acc[s][q] * c_pos : G[r * KP + c];
                if (r == c) a += lambda;
            } else {
                a = (r == c) ? 1.0 : 0.0;
            }
            acc[s][q] = a;
        }
    }
    if (tid < KP) bz[tid] = (tid < k) ? bsum : 0.0;

    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // C-layout register tile -> Dg block Jd, then factor it in row-per-lane registers (wave 0)
    auto factor_block = [&](const double4_t &t, int Jd) {
        double *blk = Dg + Jd * 16 * XS;
#pragma unroll
        for (int q = 0; q < 4; ++q) blk[((lane >> 4) + 4 * q) * XS + (lane & 15)] = t[q];
        wave_sync();
        const int r = lane & 15;
        double d[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) d[c] = blk[r * XS + c];
        double myinv = 1.0;
        bool ok = true;
        // software-pipelined: the next pivot's rsqrt chain starts as soon as its column is
        // updated and runs under the remaining updates of the current step
        double pj = dpp_bcast<0>(d[0]);
        double inv = rsqrt_fma(pj);
        if (!ALS_KO(KO_FACTOR))
        static_for<0, 16>([&](auto JJ) {
            constexpr int jj = JJ;
            ok = ok && (pj > 0.0);
            if (r == jj) myinv = inv;
            d[jj] *= inv;                                        // column jj of L (lane = row)
            if constexpr (jj < 15) {
                dpp_fnma<jj + 1>(d[jj + 1], d[jj], d[jj]);       // d[.][c] -= L[.][jj] * L[c][jj]
                pj = dpp_bcast<jj + 1>(d[jj + 1]);
                inv = rsqrt_fma(pj);
                static_for<jj + 2, 16>([&](auto C) {
                    constexpr int c = C;
                    dpp_fnma<c, false>(d[c], d[jj], d[jj]);
                });
            }
        });
        wave_sync();                                 // all lanes have read the raw block
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c <= r) blk[r * XS + c] = d[c];
            dinv[Jd * 16 + r] = myinv;
        }
        if (!ok && lane == 0 && !ALS_KO(~0)) *bad = 1;
    };
    auto trailing_reg = [&](double4_t &t, int I, int J2) {
        const double *pa = Xs + (I * 16 + (lane & 15)) * XS + (lane >> 4);
        const double *pb = Xs + (J2 * 16 + (lane & 15)) * XS + (lane >> 4);
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4)
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[s4], pb[s4], t, 0, 0, 0);
    };

    ALS_STAMP(2);
    // ---- blocked Cholesky on the register tiles ----------------------------------------------------
    if constexpr (wid == 0) factor_block(acc[0], 0);
#pragma unroll
    for (int Jb = 0; Jb < NT; ++Jb) {
        const int c0 = Jb * 16;
        __syncthreads();                             // Dg[Jb], dinv ready; Xs free
        if (Jb == 0) ALS_STAMP_CHOL_W0(5);
        if (Jb == 1) ALS_STAMP_CHOL_W0(11);
        if constexpr (wid != 0) {
#pragma unroll
            for (int s = 0; s < TS; ++s)
                if (sJ(s) == Jb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        Xs[(sI(s) * 16 + (lane >> 4) + 4 * q) * XS + (lane & 15)] = acc[s][q];
                }
        }
        __syncthreads();
        if (Jb == 0) ALS_STAMP_CHOL_W0(6);
        // panel rows and the border row: x = a * L11^{-T}, one lane per row.  L11 sits in registers,
        // lane = its row, and reaches the other lanes by DPP broadcast (every lane of a taking-part
        // wave stays active: surplus lanes redo the border row and skip the store).
        if (c0 + 16 + wid * 64 <= KP) {                  // wave-uniform
            const int pr = c0 + 16 + tid;
            const bool mine = pr <= KP;
            double *row = (pr < KP) ? Xs + pr * XS : bz + c0;
            const double *L11 = Dg + (c0 + (lane & 15)) * XS;
            double Lr[16], a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) Lr[c] = L11[c];
            const double dv = dinv[c0 + (lane & 15)];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = row[c];
            if (!ALS_KO(KO_SOLVE))
            static_for<0, 16>([&](auto JJ) {
                constexpr int jj = JJ;
                a[jj] *= dpp_bcast<jj>(dv);
                static_for<jj + 1, 16>([&](auto M) {             // a[m] -= L[m][jj] * x_jj
                    constexpr int m = M;
                    dpp_fnma<m, false>(a[m], Lr[jj], a[jj]);
                });
            });
            if (mine) {
#pragma unroll
                for (int c = 0; c < 16; ++c) row[c] = a[c];
            }
        }
        if (Jb == 0) ALS_STAMP_CHOL_W0(7);
        __syncthreads();
        if (Jb == 0) ALS_STAMP_CHOL_W0(8);
        if constexpr (wid >= 2) {   // border row of the trailing update (waves 2-3: off wave 0's chain)
            const int c = c0 + 16 + tid - 128;
            if (c < KP) {
                double sum = 0.0;
#pragma unroll
                for (int m = 0; m < 16; ++m) sum += bz[c0 + m] * Xs[c * XS + m];
                bz[c] -= sum;
            }
        }
        if constexpr (wid != 0) {
#pragma unroll
            for (int s = 0; s < TS; ++s) {
                if (sJ(s) == Jb) {         // the solved panel tile comes back to registers
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[s][q] = Xs[(sI(s) * 16 + (lane >> 4) + 4 * q) * XS + (lane & 15)];
                } else if (sJ(s) > Jb && !ALS_KO(KO_TRAIL)) {
                    trailing_reg(acc[s], sI(s), sJ(s));
                }
            }
        } else if (Jb + 1 < NT) {   // role 0
            trailing_reg(acc[Jb + 1], Jb + 1, Jb + 1);   // next diagonal tile first ...
            if (Jb == 0) ALS_STAMP_CHOL_W0(9);
            factor_block(acc[Jb + 1], Jb + 1);           // ... and factor it while waves 1-3 update
            if (Jb == 0) ALS_STAMP_CHOL_W0(10);
            if (!ALS_KO(KO_REST)) {
#pragma unroll
                for (int J2 = Jb + 2; J2 < NT; ++J2) trailing_reg(acc[J2], J2, J2);
            }
        }
    }
    __syncthreads();
    if (*bad) {
        if (tid == 0) atomicExch(status, j + 1);
        return;
    }

    ALS_STAMP(3);
    // ---- back substitution: x_J = L_JJ^{-T} (z_J - sum_{I>J} L_IJ^T x_I), L_IJ in registers -------
    if (!ALS_KO(KO_BACK))
#pragma unroll
    for (int Jb = NT - 1; Jb >= 0; --Jb) {
        const int c0 = Jb * 16;
        double cs = 0.0;
        if constexpr (wid != 0) {
#pragma unroll
            for (int s = 0; s < TS; ++s)
                if (sJ(s) == Jb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) cs += acc[s][q] * bz[sI(s) * 16 + (lane >> 4) + 4 * q];
                }
        }
        {
            int lo = __double2loint(cs), hi = __double2hiint(cs);
            cs += __hiloint2double(__shfl_xor(hi, 16, 64), __shfl_xor(lo, 16, 64));
            lo = __double2loint(cs); hi = __double2hiint(cs);
            cs += __hiloint2double(__shfl_xor(hi, 32, 64), __shfl_xor(lo, 32, 64));
        }
        if (lane < 16) scratch[wid * 16 + lane] = cs;
        __syncthreads();
        if constexpr (wid == 0) {
            const int m = lane & 15;
            double rhs = bz[c0 + m] - scratch[16 + m] - scratch[32 + m] - scratch[48 + m];
            double Lc[16];                                       // lane = column m of L11
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) Lc[jj] = Dg[(c0 + jj) * XS + m];
            const double dv = dinv[c0 + m];
            static_for<0, 16>([&](auto T) {
                constexpr int jj = 15 - T;
                const double x = dpp_bcast<jj>(rhs) * dpp_bcast<jj>(dv);
                const double upd = __builtin_fma(-Lc[jj], x, rhs);
                rhs = (m < jj) ? upd : (m == jj ? x : rhs);
            });
            if (lane < 16) bz[c0 + lane] = rhs;
        }
        __syncthreads();
    }
    ALS_STAMP(4);
    for (int f = tid; f < k; f += 256) xrow[f] = bz[f];
}
#undef sI
#undef sJ

template <int NT, int OCC>
__global__ __launch_bounds__(256, OCC) void k_als_rows3(const double *__restrict__ Y, int k,
                                                      const double *__restrict__ G,
                                                      const int64_t *__restrict__ off,
                                                      const int32_t *__restrict__ col, double c_pos,
                                                      double one_plus_c, double lambda, double *X,
                                                      int *status, int tune)
{
    // Issue priority by the wave's hardware slot on its SIMD: the rows that share a CU otherwise fall into step -- all in their
    // matrix phase together, sharing the pipe, then all in their serial phases together with the pipe idle.  Distinct
    // priorities serve the matrix phases one after the other and the rows drift apart.
    if (tune & 1) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        switch ((tune & 2) ? 3 - ((hw >> 16) & 3) : ((hw >> 16) & 3)) {      // TG_ID: the workgroup's slot on its CU
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
    // every wave runs the same sequence of barriers; only its tile set differs
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: als_rows3_body<NT, 0>(Y, k, G, off, col, c_pos, one_plus_c, lambda, X, status, tune); break;
    case 1: als_rows3_body<NT, 1>(Y, k, G, off, col, c_pos, one_plus_c, lambda, X, status, tune); break;
    case 2: als_rows3_body<NT, 2>(Y, k, G, off, col, c_pos, one_plus_c, lambda, X, status, tune); break;
    default: als_rows3_body<NT, 3>(Y, k, G, off, col, c_pos, one_plus_c, lambda, X, status, tune); break;
    }
}

// ------------------------------------------------------------------------------------
// k > 128 (als_implicit.pyx:211-225 takes any dim): the same algorithm with the bordered matrix
// [A; b^T] in a per-workgroup GLOBAL scratch slot (L2 resident: (KP+1) x (KP+2) doubles) instead of
// LDS / registers, KP a run-time multiple of 16.  Y_N^T Y_N is accumulated on the matrix cores in
// ROUNDS of 8 tiles per wave (the neighbour chunks are re-streamed every round), the blocked
// Cholesky and the blocked back substitution walk the scratch with workgroup barriers in between.
// A fallback, far from the tuned kernels' speed -- but a legal argument is served, not refused.
// ------------------------------------------------------------------------------------
constexpr int BIG_NW = 8;              // waves per workgroup
constexpr int BIG_TPR = 8;             // tiles per wave per accumulation round
constexpr int BIG_CH = 16;             // neighbour rows per LDS chunk
constexpr int kBigMaxK = 512;

// partial[slab][KP*KP] of G = Y^T Y: one block per (tile, slab), operands straight from global
__global__ __launch_bounds__(256) void k_gram_big(const double *__restrict__ Y, int ny, int k, int KP,
                                                  int rows_per_slab, double *partial)
{
    __shared__ double red[3][64][4];
    const int NT = KP / 16;
    const int I = blockIdx.x / NT, J = blockIdx.x % NT;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(ny, r0 + rows_per_slab);
    const int ca = I * 16 + (lane & 15), cb = J * 16 + (lane & 15);
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int n0 = r0 + wid * 4; n0 < r1; n0 += 16) {
        const int row = n0 + (lane >> 4);
        double a = 0.0, b = 0.0;
        if (row < r1) {
            if (ca < k) a = Y[(size_t)row * k + ca];
            if (cb < k) b = Y[(size_t)row * k + cb];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    if (wid > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wid - 1][lane][q] = acc[q];
    }
    __syncthreads();
    if (wid == 0) {
        double *out = partial + (size_t)blockIdx.y * KP * KP;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v = ((acc[q] + red[0][lane][q]) + red[1][lane][q]) + red[2][lane][q];
            out[(I * 16 + (lane >> 4) + 4 * q) * KP + J * 16 + (lane & 15)] = v;
        }
    }
}

__global__ __launch_bounds__(64 * BIG_NW) void k_als_rows_big(const double *__restrict__ Y, int k, int KP,
                                                             const double *__restrict__ G,
                                                             const int64_t *__restrict__ off,
                                                             const int32_t *__restrict__ col, double c_pos,
                                                             double one_plus_c, double lambda, double *X,
                                                             int *status, double *scratch, int n_rows)
{
    constexpr int NW = BIG_NW, NTHR = 64 * BIG_NW;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int NT = KP / 16, LD = KP + 2, S = ych_stride(KP);
    double *ych = lds;                               // [BIG_CH][S]
    double *dinv = ych + BIG_CH * S;                 // [KP]
    double *z = dinv + KP;                           // [KP] right-hand side / solution
    double *dblk = z + KP;                           // [16][17] current diagonal block
    int *cidx = (int *)(dblk + 16 * 17);             // [BIG_CH]
    int *bad = cidx + BIG_CH;
    double *A = scratch + (size_t)blockIdx.x * (size_t)(KP + 1) * LD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int NTT = NT * (NT + 1) / 2;

    for (int j = blockIdx.x; j < n_rows; j += gridDim.x) {
        const int64_t start = off[j];
        const int span = (int)(off[j + 1] - start);
        double *xrow = X + (size_t)j * k;
        if (span == 0) {                             // b = 0 => x = 0 (als_implicit.pyx:298-302)
            for (int f = tid; f < k; f += NTHR) xrow[f] = 0.0;
            continue;
        }
        if (tid == 0) *bad = 0;
        // ---- A = G + c * Y_N^T Y_N + lambda I, b: rounds of NW * BIG_TPR tiles ----------------
        const int nchunks = (span + BIG_CH - 1) / BIG_CH;
        double bsum = 0.0;
        for (int t0 = 0; t0 < NTT; t0 += NW * BIG_TPR) {
            int tI[BIG_TPR], tJ[BIG_TPR];
            double4_t acc[BIG_TPR];
#pragma unroll
            for (int t = 0; t < BIG_TPR; ++t) {
                tI[t] = 0; tJ[t] = 0;
                const int tt = t0 + t * NW + wid;
                if (tt < NTT) tile_ij(tt, tI[t], tJ[t]);
                acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
            }
            for (int c = 0; c < nchunks; ++c) {
                __syncthreads();                     // the previous chunk's readers are done
                if (tid < BIG_CH) cidx[tid] = (c * BIG_CH + tid < span) ? col[start + c * BIG_CH + tid] : -1;
                __syncthreads();
                for (int e = tid; e < BIG_CH * KP; e += NTHR) {
                    const int n = e / KP, f = e % KP;
                    const int id = cidx[n];
                    ych[n * S + f] = (id >= 0 && f < k) ? Y[(size_t)id * k + f] : 0.0;
                }
                __syncthreads();
                if (t0 == 0 && tid < KP) {
#pragma unroll
                    for (int n = 0; n < BIG_CH; ++n) bsum += ych[n * S + tid] * one_plus_c;
                }
#pragma unroll
                for (int n0 = 0; n0 < BIG_CH; n0 += 4) {
                    const double *base = ych + (n0 + (lane >> 4)) * S + (lane & 15);
#pragma unroll
                    for (int t = 0; t < BIG_TPR; ++t)
                        if (t0 + t * NW + wid < NTT)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[tI[t] * 16], base[tJ[t] * 16], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < BIG_TPR; ++t) {
                if (t0 + t * NW + wid < NTT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = tI[t] * 16 + (lane >> 4) + 4 * q, c = tJ[t] * 16 + (lane & 15);
                        double a;
                        if (r < k && c < k) {
                            a = G[(size_t)r * KP + c] + c_pos * acc[t][q];
                            if (r == c) a += lambda;
                        } else {
                            a = (r == c) ? 1.0 : 0.0;
                        }
                        A[(size_t)r * LD + c] = a;
                    }
                }
            }
        }
        if (tid < KP) A[(size_t)KP * LD + tid] = (tid < k) ? bsum : 0.0;     // border row = b
        __syncthreads();

        // ---- blocked Cholesky of the bordered matrix, 16-wide panels ---------------------------
        for (int Jb = 0; Jb < NT; ++Jb) {
            const int c0 = Jb * 16;
            // (a) diagonal block: lane r (< 16) factors row r, pivots broadcast with shuffles
            if (wid == 0) {
                const int r = lane & 15;
                double d[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) d[c] = A[(size_t)(c0 + r) * LD + c0 + c];
                double myinv = 1.0;
                bool ok = true;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const double pj = __shfl(d[jj], jj, 16);          // L-updated pivot
                    ok = ok && (pj > 0.0);
                    const double inv = rsqrt_fma(pj);
                    if (r == jj) myinv = inv;
                    d[jj] *= inv;                                     // column jj of L (rows >= jj matter)
#pragma unroll
                    for (int c = jj + 1; c < 16; ++c) {
                        const double lc = __shfl(d[jj], c, 16);       // L[c][jj]
                        d[c] -= d[jj] * lc;
                    }
                }
                if (lane < 16) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        if (c <= r) A[(size_t)(c0 + r) * LD + c0 + c] = d[c];
                        dblk[r * 17 + c] = (c <= r) ? d[c] : 0.0;
                    }
                    dinv[c0 + r] = myinv;
                }
                if (!ok && lane == 0) *bad = 1;
            }
            __syncthreads();
            // (b) panel: rows below the block and the border row:  X = A21 * L11^{-T}
            for (int pr = c0 + 16 + tid; pr <= KP; pr += NTHR) {
                double *prow = A + (size_t)pr * LD + c0;
                double a[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = prow[c];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    double x = a[jj];
#pragma unroll
                    for (int m = 0; m < jj; ++m) x -= a[m] * dblk[jj * 17 + m];
                    a[jj] = x * dinv[c0 + jj];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) prow[c] = a[c];
            }
            __syncthreads();
            // (c) trailing update: border row (VALU), then the lower tiles on the matrix cores
            for (int c = c0 + 16 + tid; c < KP; c += NTHR) {
                double sum = 0.0;
                const double *crow = A + (size_t)c * LD + c0, *brow = A + (size_t)KP * LD + c0;
#pragma unroll
                for (int m = 0; m < 16; ++m) sum += brow[m] * crow[m];
                A[(size_t)KP * LD + c] -= sum;
            }
            const int nt = NT - Jb - 1, ntt = nt * (nt + 1) / 2;
            for (int tt = wid; tt < ntt; tt += NW) {
                int I, J2;
                tile_ij(tt, I, J2);
                I += Jb + 1;
                J2 += Jb + 1;
                double4_t cacc;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cacc[q] = A[(size_t)(I * 16 + (lane >> 4) + 4 * q) * LD + J2 * 16 + (lane & 15)];
                const double *pa = A + (size_t)(I * 16 + (lane & 15)) * LD + c0 + (lane >> 4);
                const double *pb = A + (size_t)(J2 * 16 + (lane & 15)) * LD + c0 + (lane >> 4);
#pragma unroll
                for (int s4 = 0; s4 < 16; s4 += 4)
                    cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[s4], pb[s4], cacc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    A[(size_t)(I * 16 + (lane >> 4) + 4 * q) * LD + J2 * 16 + (lane & 15)] = cacc[q];
            }
            __syncthreads();
        }
        if (*bad) {
            if (tid == 0) atomicExch(status, j + 1);
            __syncthreads();
            continue;
        }
        // ---- blocked back substitution L^T x = z (z = the border row after the forward sweep) ----
        for (int c = tid; c < KP; c += NTHR) z[c] = A[(size_t)KP * LD + c];
        __syncthreads();
        for (int Jb = NT - 1; Jb >= 0; --Jb) {
            const int c0 = Jb * 16;
            if (wid == 0) {
                // lane m (< 16) owns x_m of the block; 16 dependent steps, broadcast with shuffles
                const int m = lane & 15;
                double zm = z[c0 + m];
                double lcol[16];                                      // L[c0 + jj][c0 + m], jj = 0..15
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) lcol[jj] = (jj > m) ? A[(size_t)(c0 + jj) * LD + c0 + m] : 0.0;
                const double di = dinv[c0 + m];
#pragma unroll
                for (int jj = 15; jj >= 0; --jj) {
                    const double xj = __shfl(zm * di, jj, 16);       // x_jj is final when its turn comes
                    if (m == jj) zm = xj;
                    else if (m < jj) zm -= lcol[jj] * xj;
                }
                if (lane < 16) z[c0 + m] = zm;
            }
            __syncthreads();
            for (int r = tid; r < c0; r += NTHR) {
                double sum = 0.0;
#pragma unroll
                for (int m = 0; m < 16; ++m) sum += A[(size_t)(c0 + m) * LD + r] * z[c0 + m];
                z[r] -= sum;
            }
            __syncthreads();
        }
        for (int f = tid; f < k; f += NTHR) xrow[f] = z[f];
        __syncthreads();
    }
}

size_t rows_big_lds_bytes(int KP)
{
    return ((size_t)BIG_CH * ych_stride(KP) + 2 * (size_t)KP + 16 * 17) * sizeof(double) + (BIG_CH + 2) * sizeof(int);
}

#ifdef MFX_ALS_STAMPS
extern "C" int mfx_debug_als_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_als_stamps), sizeof(unsigned long long) * 4096 * 16) == hipSuccess ? 0 : -4;
}
#endif

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void k_transpose_als(const double *in, double *out, int rows, int cols)
{
    __shared__ double tile[32][33];
    // 1-D grid of tiles (a 2-D grid would cap rows at 65535 * 32)
    const unsigned nbx = (unsigned)(cols + 31) / 32;
    const int bx = (int)(blockIdx.x % nbx) * 32, by = (int)(blockIdx.x / nbx) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = by + i, c = bx + tx;
        if (r < rows && c < cols) tile[i][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = bx + i, r = by + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][i];
    }
}

struct AlsKernels {
    void *gram;
    void *rows2;
    int rows_waves;     // waves per workgroup of the row kernel
    bool reg_tiles = false;   // k_als_rows3: tiles in registers, small LDS
};

size_t rows3_lds_bytes(int KP)
{
    const size_t st = (size_t)CH2 * ych_stride(KP), xd = (size_t)2 * KP * 18;
    return ((st > xd ? st : xd) + 2 * KP + 64) * sizeof(double) + (2 * CH2 + 2) * sizeof(int);
}

template <int NT>
AlsKernels als_kernels_nt()
{
    if constexpr (NT == 1) return {(void *)k_gram_partial<NT>, (void *)k_als_rows2<NT, 1>, 1};
    else if constexpr (NT == 2) {
        const char *e = getenv("MFX_ALS_NW");                 // tuning hook
        const int nw = e ? atoi(e) : 2;
        if (nw == 1) return {(void *)k_gram_partial<NT>, (void *)k_als_rows2<NT, 1>, 1};
        if (nw == 4) return {(void *)k_gram_partial<NT>, (void *)k_als_rows2<NT, 4>, 4};
        return {(void *)k_gram_partial<NT>, (void *)k_als_rows2<NT, 2>, 2};
    } else if constexpr (NT >= 5) {
        const char *e = getenv("MFX_ALS_ROWS2");              // tuning / cross-check hook
        if (e && e[0] == '1') return {(void *)k_gram_partial<NT>, (void *)k_als_rows2<NT, 4>, 4, false};
        // occupancy hint: k = 128 needs the cap to fit three workgroups per CU; below that the
        // kernel fits anyway and the uncapped schedule measured 4 % faster
        const char *o = getenv("MFX_ALS_OCC");
        const bool occ3 = o ? o[0] == '3' : NT == 8;
        if (!occ3) return {(void *)k_gram_partial<NT>, (void *)k_als_rows3<NT, 2>, 4, true};
        return {(void *)k_gram_partial<NT>, (void *)k_als_rows3<NT, 3>, 4, true};
    } else {                                                  // NT = 3, 4
        const char *e = getenv("MFX_ALS_ROWS2");
        const bool r2 = e && e[0] == '1';
        if (r2) return {(void *)k_gram_partial<NT>, (void *)k_als_rows2<NT, 4>, 4, false};
        const char *o = getenv("MFX_ALS_OCC");
        switch (o ? o[0] - '0' : (NT == 4 ? 4 : 3)) {   // measured: k=64 5.6 (rows2) / 5.9 / 5.2 / 5.3 / 5.7 ms at 3..6
        case 4: return {(void *)k_gram_partial<NT>, (void *)k_als_rows3<NT, 4>, 4, true};
        case 5: return {(void *)k_gram_partial<NT>, (void *)k_als_rows3<NT, 5>, 4, true};
        case 6: return {(void *)k_gram_partial<NT>, (void *)k_als_rows3<NT, 6>, 4, true};
        default: return {(void *)k_gram_partial<NT>, (void *)k_als_rows3<NT, 3>, 4, true};
        }
    }
}

size_t rows2_lds_bytes(int KP)
{
    const int NTv = KP / 16;
    const size_t a = (size_t)(128 * NTv * (NTv + 1) + 32 * NTv) + KP + 2, st = (size_t)CH2 * ych_stride(KP);
    const size_t region = ((a > st ? a : st) + 1) & ~(size_t)1;
    return (region + KP) * sizeof(double) + (2 * CH2 + 2) * sizeof(int);
}

AlsKernels als_kernels(int nt)
{
    switch (nt) {
    case 1: return als_kernels_nt<1>();
    case 2: return als_kernels_nt<2>();
    case 3: return als_kernels_nt<3>();
    case 4: return als_kernels_nt<4>();
    case 5: return als_kernels_nt<5>();
    case 6: return als_kernels_nt<6>();
    case 7: return als_kernels_nt<7>();
    default: return als_kernels_nt<8>();
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) return mfx::fail(MFX_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return MFX_OK;
    }
    template <typename T> T *as() { return (T *)p; }
};

constexpr int kGramBlocks = 512;

constexpr int kBigSlabs = 64;
double *g_big_scratch = nullptr;       // per-workgroup bordered matrices of the k > 128 path (one call at a time)
size_t g_big_scratch_bytes = 0;

// the k > 128 half-epoch: k_gram_big + k_als_rows_big (persistent over rows, global scratch)
int als_half_big(int k, int KP, const double *dY, int ny, double *dX, const int64_t *d_off, int64_t n_active,
                 const int32_t *d_col, int c_pos, double lambda, double *d_partial, double *dG, int *d_status,
                 int rank, int world, bool gather)
{
    hipStream_t s = mfx::stream();
    const int NT = KP / 16;
    {
        int slabs = ny < kBigSlabs * 64 ? ceil_div(ny, 64) : kBigSlabs;
        if (slabs < 1) slabs = 1;
        const int rps = ceil_div(ny, slabs);
        hipLaunchKernelGGL(k_gram_big, dim3(NT * NT, slabs), dim3(256), 0, s, dY, ny, k, KP, rps, d_partial);
        hipLaunchKernelGGL(k_gram_reduce, dim3(ceil_div(KP * KP, 256)), dim3(256), 0, s, d_partial, slabs, KP * KP, dG);
        MFX_HIP(hipGetLastError());
    }
    if (n_active <= 0) return MFX_OK;
    const size_t lds = rows_big_lds_bytes(KP);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_als_rows_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t B = ceil_div(n_active, (int64_t)world);
    const int r_first = rank < 0 ? 0 : rank, r_last = rank < 0 ? world - 1 : rank;
    for (int rr = r_first; rr <= r_last; ++rr) {
        const int64_t lo = (int64_t)rr * B < n_active ? (int64_t)rr * B : n_active;
        const int64_t hi = lo + B < n_active ? lo + B : n_active;
        if (hi <= lo) continue;
        const int n = (int)(hi - lo);
        const int grid = n < 512 ? n : 512;
        const size_t need = (size_t)grid * (size_t)(KP + 1) * (KP + 2) * sizeof(double);
        if (need > g_big_scratch_bytes) {
            if (g_big_scratch) (void)hipFree(g_big_scratch);
            g_big_scratch = nullptr;
            g_big_scratch_bytes = 0;
            if (hipMalloc((void **)&g_big_scratch, need) != hipSuccess)
                return mfx::fail(MFX_E_NOMEM, "mfx_als_wrmf: %zu bytes of scratch for k = %d", need, k);
            g_big_scratch_bytes = need;
        }
        hipLaunchKernelGGL(k_als_rows_big, dim3(grid), dim3(64 * BIG_NW), lds, s, dY, k, KP, (const double *)dG,
                           d_off + lo, d_col, (double)c_pos, (double)(1 + c_pos), lambda, dX + (size_t)lo * k, d_status,
                           g_big_scratch, n);
        MFX_HIP(hipGetLastError());
    }
    if (gather) {
        int rc = mfx::comm_allgather_f64(dX, (size_t)B * k);
        if (rc) return rc;
    }
    return MFX_OK;
}

// one half-epoch: solve X (n_active rows) from Y
// rank/world: this process solves rows [rank*B, (rank+1)*B) with B = ceil(n_active / world) and
// the slices are all-gathered in place (dX has room for world*B rows); world == 1: all rows.
int als_half(int k, int NT, const AlsKernels &K, const double *dY, int ny, double *dX,
             const int64_t *d_off, int64_t n_active, const int32_t *d_col, int c_pos, double lambda,
             double *d_partial, double *dG, int *d_status, int rank = 0, int world = 1,
             bool gather = false)
{
    mfx::TraceRange trace_("mfx:als_half (Gramian + row solves)");
    const int KP = NT * 16;
    if (NT > 8) return als_half_big(k, KP, dY, ny, dX, d_off, n_active, d_col, c_pos, lambda, d_partial, dG, d_status,
                                    rank, world, gather);
    hipStream_t s = mfx::stream();
    {   // Gramian over ALL rows of Y (als_implicit.pyx:257-262)
        int blocks = ny < kGramBlocks * CH ? ceil_div(ny, CH) : kGramBlocks;
        if (blocks < 1) blocks = 1;
        int rpb = ceil_div(ceil_div(ny, blocks), CH) * CH;
        if (rpb < CH) rpb = CH;
        blocks = ceil_div(ny, rpb);
        if (blocks < 1) blocks = 1;
        const size_t lds = (size_t)CH * ych_stride(KP) * sizeof(double);
        const double *Yc = dY;
        int kk = k, nyy = ny;
        void *args[] = {&Yc, &nyy, &kk, &rpb, &d_partial};
        MFX_HIP(hipLaunchKernel(K.gram, dim3(blocks), dim3(256), args, lds, s));
        hipLaunchKernelGGL(k_gram_reduce, dim3(ceil_div(KP * KP, 256)), dim3(256), 0, s, d_partial,
                           blocks, KP * KP, dG);
        MFX_HIP(hipGetLastError());
    }
    if (n_active > 0) {
        const size_t lds = K.reg_tiles ? rows3_lds_bytes(KP) : rows2_lds_bytes(KP);
        void *rows_fn = K.rows2;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(rows_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const double *Yc = dY;
        const double *Gc = dG;
        int kk = k;
        double c = (double)c_pos, opc = (double)(1 + c_pos), lam = lambda;
        const char *te = getenv("MFX_ALS_TUNE");                 // tuning hook (A/B runs inside one process)
        int tune = te ? atoi(te) : 0;
        void *args[] = {&Yc, &kk, &Gc, &d_off, &d_col, &c, &opc, &lam, &dX, &d_status, &tune};
        const int64_t B = ceil_div(n_active, (int64_t)world);
        // rank < 0 (test hook MFX_ALS_FAKE_WORLD): run every rank's slice on this GPU in turn
        const int r_first = rank < 0 ? 0 : rank, r_last = rank < 0 ? world - 1 : rank;
        for (int rr = r_first; rr <= r_last; ++rr) {
        const int64_t lo = (int64_t)rr * B < n_active ? (int64_t)rr * B : n_active;
        const int64_t hi = lo + B < n_active ? lo + B : n_active;
        // the grid is one block per active row; launch in slices of <= 2^20 rows
        for (int64_t r0 = lo; r0 < hi; r0 += (1 << 20)) {
            const int n = (int)((hi - r0) < (1 << 20) ? (hi - r0) : (1 << 20));
            const int64_t *off_s = d_off + r0;
            double *X_s = dX + (size_t)r0 * k;
            args[3] = &off_s;
            args[8] = &X_s;
            MFX_HIP(hipLaunchKernel(rows_fn, dim3(n), dim3(64 * K.rows_waves), args, lds, s));
        }
        }
        if (gather) {
            int rc = mfx::comm_allgather_f64(dX, (size_t)B * k);
            if (rc) return rc;
        }
    }
    return MFX_OK;
}

int transpose_to(const double *in, double *out, int rows, int cols)
{
    if (!rows || !cols) return MFX_OK;
    const int64_t tiles = (int64_t)ceil_div(cols, 32) * ceil_div(rows, 32);
    if (tiles > 0x7fffffff) return mfx::fail(MFX_E_ARG, "transpose: %d x %d is too large", rows, cols);
    hipLaunchKernelGGL(k_transpose_als, dim3((unsigned)tiles), dim3(256), 0, mfx::stream(), in, out, rows, cols);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

// run-length [0,c0,c1,...] -> exclusive offsets (the reference's running sum,
// als_implicit.pyx:269-270), validating the column ids.
int offsets_from_runlength(const int32_t *row, int64_t n_row, const int32_t *col, int n_rows_max,
                           int n_cols, std::vector<int64_t> &off, const char *what)
{
    const int64_t n_active = n_row > 0 ? n_row - 1 : 0;
    if (n_active > n_rows_max)
        return mfx::fail(MFX_E_RANGE, "%s: %lld active rows > %d", what, (long long)n_active, n_rows_max);
    off.assign((size_t)n_active + 1, 0);
    int64_t start = 0;
    for (int64_t j = 0; j < n_active; ++j) {
        start += row[j];
        if (row[j + 1] < 0) return mfx::fail(MFX_E_RANGE, "%s: negative count", what);
        off[j] = start;
    }
    off[n_active] = n_active ? off[n_active - 1] + row[n_active] : 0;
    for (int64_t t = 0; t < off[n_active]; ++t)
        if (col[t] < 0 || col[t] >= n_cols)
            return mfx::fail(MFX_E_RANGE, "%s: neighbour id %d outside [0,%d)", what, col[t], n_cols);
    return MFX_OK;
}

}  // namespace

static int als_wrmf_impl(int nbr_epochs, int dim, double *u, double *v, const int32_t *urow,
                         int64_t n_urow, const int32_t *ucol, const int32_t *irow, int64_t n_irow,
                         const int32_t *icol, int nu, int ni, int c_pos, double lambda, bool sharded)
{
    if (!u || !v || !urow || !irow || dim < 1 || dim > kBigMaxK || nu <= 0 || ni <= 0 || n_urow < 1 ||
        n_irow < 1)
        return mfx::fail(MFX_E_ARG, "mfx_als_wrmf: bad argument (1 <= k <= %d)", kBigMaxK);
    int rc = mfx::ensure_init();
    if (rc) return rc;
    if (nbr_epochs <= 0) return MFX_OK;
    std::vector<int64_t> uoff, ioff;
    try {
        rc = offsets_from_runlength(urow, n_urow, ucol, nu, ni, uoff, "ratings_users_row");
        if (!rc) rc = offsets_from_runlength(irow, n_irow, icol, ni, nu, ioff, "ratings_items_row");
    } catch (const std::bad_alloc &) {
        return mfx::fail(MFX_E_NOMEM, "mfx_als_wrmf: out of host memory");
    }
    if (rc) return rc;
    const int64_t nau = (int64_t)uoff.size() - 1, nai = (int64_t)ioff.size() - 1;
    const int64_t nnz_u = uoff[nau], nnz_i = ioff[nai];
    if ((nnz_u > 0 && !ucol) || (nnz_i > 0 && !icol)) return mfx::fail(MFX_E_ARG, "mfx_als_wrmf: NULL column array");

    const int NT = ceil_div(dim, 16), KP = NT * 16;
    const AlsKernels K = als_kernels(NT > 8 ? 8 : NT);      // NT > 8: als_half_big, K unused
    const size_t su = (size_t)dim * ni * sizeof(double), sv = (size_t)dim * nu * sizeof(double);
    // sharded: every rank holds full replicas, solves a contiguous slice of the rows of each
    // half-epoch and all-gathers the slices (row solves are independent: same result as 1 GPU)
    const bool gather = sharded && mfx::comm_active();
    int rank = gather ? mfx::comm_rank() : 0, world = gather ? mfx::comm_world() : 1;
    if (!gather) {                                  // slicing arithmetic check on one GPU
        const char *fw = getenv("MFX_ALS_FAKE_WORLD");
        if (fw && atoi(fw) > 1) { world = atoi(fw); rank = -1; }
    }
    if (sharded && !mfx::comm_active() && mfx::comm_world() > 1)
        return mfx::fail(MFX_E_STATE, "mfx_als_wrmf_sharded: mfx_comm_init has not been called");
    const size_t pad = (size_t)world * dim * sizeof(double);   // in-place all-gather slack rows
    DevBuf dUkm, dVkm, dU, dV, dUoff, dIoff, dUcol, dIcol, dPart, dG, dStatus;
    if ((rc = dUkm.alloc(su)) || (rc = dVkm.alloc(sv)) || (rc = dU.alloc(su + pad)) || (rc = dV.alloc(sv + pad)) ||
        (rc = dUoff.alloc((nau + 1) * 8)) || (rc = dIoff.alloc((nai + 1) * 8)) ||
        (rc = dUcol.alloc((size_t)nnz_u * 4)) || (rc = dIcol.alloc((size_t)nnz_i * 4)) ||
        (rc = dPart.alloc((size_t)(NT > 8 ? kBigSlabs : kGramBlocks) * KP * KP * 8)) || (rc = dG.alloc((size_t)KP * KP * 8)) ||
        (rc = dStatus.alloc(4)))
        return rc;
    hipStream_t s = mfx::stream();
    MFX_HIP(hipMemcpy(dUkm.p, u, su, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dVkm.p, v, sv, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dUoff.p, uoff.data(), (nau + 1) * 8, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(dIoff.p, ioff.data(), (nai + 1) * 8, hipMemcpyHostToDevice));
    if (nnz_u) MFX_HIP(hipMemcpy(dUcol.p, ucol, (size_t)nnz_u * 4, hipMemcpyHostToDevice));
    if (nnz_i) MFX_HIP(hipMemcpy(dIcol.p, icol, (size_t)nnz_i * 4, hipMemcpyHostToDevice));
    MFX_HIP(hipMemset(dStatus.p, 0, 4));
    rc = transpose_to(dUkm.as<double>(), dU.as<double>(), dim, ni);      // [k][ni] -> [ni][k]
    if (!rc) rc = transpose_to(dVkm.as<double>(), dV.as<double>(), dim, nu);
    struct EvPair {                                 // destroyed on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evp;
    MFX_HIP(hipEventCreate(&evp.a));
    MFX_HIP(hipEventCreate(&evp.b));
    hipEvent_t ev0 = evp.a, ev1 = evp.b;
    MFX_HIP(hipEventRecord(ev0, s));
    for (int e = 0; e < nbr_epochs && !rc; ++e) {
        // user pass (als_implicit.pyx:257-302), then item pass (:307-352)
        rc = als_half(dim, NT, K, dU.as<double>(), ni, dV.as<double>(), dUoff.as<int64_t>(), nau,
                      dUcol.as<int32_t>(), c_pos, lambda, dPart.as<double>(), dG.as<double>(),
                      dStatus.as<int>(), rank, world, gather);
        if (!rc)
            rc = als_half(dim, NT, K, dV.as<double>(), nu, dU.as<double>(), dIoff.as<int64_t>(), nai,
                          dIcol.as<int32_t>(), c_pos, lambda, dPart.as<double>(), dG.as<double>(),
                          dStatus.as<int>(), rank, world, gather);
    }
    if (!rc) {
        MFX_HIP(hipEventRecord(ev1, s));
        MFX_HIP(hipEventSynchronize(ev1));
        float ms = 0.f;
        MFX_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        mfx::set_last_compute_ms((double)ms);
    }
    if (rc) return rc;
    rc = transpose_to(dU.as<double>(), dUkm.as<double>(), ni, dim);
    if (!rc) rc = transpose_to(dV.as<double>(), dVkm.as<double>(), nu, dim);
    if (rc) return rc;
    MFX_HIP(hipStreamSynchronize(s));
    int status = 0;
    MFX_HIP(hipMemcpy(&status, dStatus.p, 4, hipMemcpyDeviceToHost));
    if (status)
        return mfx::fail(MFX_E_SINGULAR, "mfx_als_wrmf: row %d: k x k system is not positive definite", status - 1);
    MFX_HIP(hipMemcpy(u, dUkm.p, su, hipMemcpyDeviceToHost));
    MFX_HIP(hipMemcpy(v, dVkm.p, sv, hipMemcpyDeviceToHost));
    return MFX_OK;
}

extern "C" int mfx_als_wrmf(int nbr_epochs, int dim, double *u, double *v, const int32_t *urow,
                            int64_t n_urow, const int32_t *ucol, const int32_t *irow,
                            int64_t n_irow, const int32_t *icol, int nu, int ni, int c_pos,
                            double lambda)
{
    return als_wrmf_impl(nbr_epochs, dim, u, v, urow, n_urow, ucol, irow, n_irow, icol, nu, ni, c_pos,
                         lambda, false);
}

extern "C" int mfx_als_wrmf_sharded(int nbr_epochs, int dim, double *u, double *v,
                                    const int32_t *urow, int64_t n_urow, const int32_t *ucol,
                                    const int32_t *irow, int64_t n_irow, const int32_t *icol,
                                    int nu, int ni, int c_pos, double lambda)
{
    return als_wrmf_impl(nbr_epochs, dim, u, v, urow, n_urow, ucol, irow, n_irow, icol, nu, ni, c_pos,
                         lambda, true);
}
