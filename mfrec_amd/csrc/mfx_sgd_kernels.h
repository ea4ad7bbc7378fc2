// Device kernels of the SGD families (gfx950 / wave64 only).
//
// K1  full-k update  : train_linear_kernel / train_logistic_kernel (kmf_train.pyx:195-277,
//                      :103-189) and estimator_loop_with_learned_bias (gd_estimator.pyx:401-483)
// K2  Funk pass      : estimator_loop_without_bias / _with_bias / _with_bias_dev
//                      (gd_estimator.pyx:691-779, :489-582, :588-685) + cache-commit pass
//
// Exactness contract: one launch processes one dependency level (ratings with pairwise
// distinct users and items), every floating-point operation is the reference's operation
// in the reference's order (no FMA contraction: built with -ffp-contract=off), and the
// k-term dot product is summed strictly left to right by ONE lane per rating.  The factor
// arrays therefore come out bit-identical to the sequential C loop.
#pragma once
#include "mfx_common.h"

// Diagnostic builds only (tools/k1_stamps.hip defines MFX_K1_STAMPS): per-wave s_memtime
// stamps written to a side buffer that nothing else reads.  The product build has none.
#ifdef MFX_K1_STAMPS
#define MFX_STAMP(i)                                                                          \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (lane == 0) P.stamps[(wave_off + wave_g) * 8 + (i)] = t_;                          \
    } while (0)
#else
#define MFX_STAMP(i) do { } while (0)
#endif

namespace mfx {

enum { K1_LINEAR = 0, K1_LOGISTIC = 1, K1_LEARNED = 2 };

struct K1Params {
    const Rec *recs;
    void *U;             // [ni][k] row-major, element type TS (f64, or f32 in the fast mode)
    void *V;             // [nu][k]
    double *ib;          // [ni]
    double *ub;          // [nu]
    double *se_partial;  // one slot per wave per epoch
    int k;
    int upd_u, upd_i;
    double lr, Ku, Ki, Kb;
    // learned-bias family only
    double lr_u, lr_i, mu, trail;
    int f;               // the single feature that is updated
#ifdef MFX_K1_STAMPS
    unsigned long long *stamps;
#endif
};

#ifndef MFX_KWAVES
#define MFX_KWAVES 4
#endif
static constexpr int kWaves = MFX_KWAVES;   // waves per block
// ratings per wave for rank k (measured on cfg 2, k = 64: 8 per wave = 2 waves per SIMD beats
// 16 per wave by 19 %; 4 is no better); the product tile stays at <= 8 KB per wave
static inline int k1_batch(int k) { return k <= 32 ? 16 : (k <= 128 ? 8 : (k <= 256 ? 4 : 2)); }
// LDS row stride (doubles) of the product tile: rows 16-B aligned so the summing lanes can
// read two doubles per ds_read_b128, and stride/2 odd so 16 lanes walking one column pair
// hit 16 different 4-dword bank groups.
__host__ __device__ static inline int k1_stride(int k) { return ((k + 1) / 2 * 2) | 2; }

__device__ __forceinline__ double clamp15(double x)
{
    // gd_estimator.pyx:26-35 -- two separate compares, NaN passes through
    if (x > 5.0) x = 5.0;
    if (x < 1.0) x = 1.0;
    return x;
}

__device__ __forceinline__ double shfl_f64(double x, int src)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_fixed(double x)
{
    // fixed butterfly: the same association on every run (deterministic RMSE)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __shfl_xor(lo, m, 64);
        hi = __shfl_xor(hi, m, 64);
        x = x + __hiloint2double(hi, lo);
    }
    return x;
}

// ---- tolerance mode (DOT = 1): the k-term dot as north_star words it -- per-lane partial products,
// then a wave butterfly over the KP lanes of a rating.  Still f64, still deterministic (a FIXED
// association: quad swaps, half-row and row mirrors on DPP, then the cross-row swaps), FMA used; NOT the
// reference's left-to-right sum (kmf_train.pyx:75-97), so results differ from it at the 1e-16 level per
// operation -- measured final-RMSE delta in DESIGN.md.  Selected per plan (mfx_sgd_plan_set_dot).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the KP lanes of an aligned group; every lane of the group gets the same bits
template <int KP>
__device__ __forceinline__ double group_sum(double x)
{
    if constexpr (KP >= 2) x = x + dpp_f64<0xB1>(x);     // quad_perm [1,0,3,2]
    if constexpr (KP >= 4) x = x + dpp_f64<0x4E>(x);     // quad_perm [2,3,0,1]
    if constexpr (KP >= 8) x = x + dpp_f64<0x141>(x);    // row_half_mirror: lane i <-> 7 - i
    if constexpr (KP >= 16) x = x + dpp_f64<0x140>(x);   // row_mirror: lane i <-> 15 - i
    if constexpr (KP >= 32) x = x + shfl_f64(x, (int)(threadIdx.x & 63) ^ 16);
    if constexpr (KP >= 64) x = x + shfl_f64(x, (int)(threadIdx.x & 63) ^ 32);
    return x;
}

// KP  lanes per rating slot (8/16/32/64), SPW = 64/KP slots side by side
// FPL features per lane (f = j*KP + q), NR ratings per slot in flight; SPW*NR == BATCH
//
// FULL = the wave holds a complete batch and both update gates are on: no per-rating
// branches remain, so the 2*BATCH row loads and the 2*BATCH row stores are each ONE basic
// block.  (With a branch around every store hipcc loses count of the outstanding memory
// operations and emits s_waitcnt vmcnt(0) before each one: every store then waits for the
// previous store's round trip -- measured 300 cycles per store instruction.)
// TS = storage type of the factor rows: double (exact), or float ("f32 storage, f64
// arithmetic" fast mode: rows are widened on load and rounded once on store -- NOT bit-exact,
// measured RMSE delta in DESIGN.md)
// VEC = each lane owns CONTIGUOUS 16-byte chunks of its rows (feature pairs at f64, k even) and moves
// each with one access: half the memory instructions per row, and 16-byte
// write-through (sc1) stores cost what plain stores cost (8-byte ones are 2.7x dearer per byte,
// MI355X_MICROARCH.md price list), so NT == 2 then leaves no dirty L2 lines for the
// end-of-kernel write-back.  The arithmetic and the summation order do not change.
// (f32 storage: the 16-byte chunk is FOUR features -- VW = 16 / sizeof(TS) -- k a multiple of 4.)
template <typename TS> struct Chunk16;
template <> struct Chunk16<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };
template <> struct Chunk16<float> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };
// bf16 row storage (SURVEY 8(f)4's other storage option; a measured one: DESIGN.md): EIGHT features per chunk
template <> struct Chunk16<__bf16> { typedef __bf16 type __attribute__((ext_vector_type(8), aligned(2))); };
// f64 -> storage type, one rounding per step (f64 -> f32 -> bf16 for bf16: the mode is not exact anyway)
template <typename TS> __host__ __device__ __forceinline__ TS narrow(double x) { return (TS)x; }
template <> __host__ __device__ __forceinline__ __bf16 narrow<__bf16>(double x) { return (__bf16)(float)x; }
template <typename TS> __host__ __device__ __forceinline__ double widen(TS x) { return (double)x; }
template <> __host__ __device__ __forceinline__ double widen<__bf16>(__bf16 x) { return (double)(float)x; }
template <typename TS>
__device__ __forceinline__ void store16_sc1(TS *p, typename Chunk16<TS>::type c)
{
    typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
    const v4u32 v = __builtin_bit_cast(v4u32, c);
    // s_nop: a VMEM store of more than 8 bytes reads its data VGPRs after issue; a VALU write to them
    // in the next wait state corrupts the stored value (gfx9 "12-dword store" hazard).  The compiler
    // inserts that wait state for its own stores but cannot see inside inline asm -- without it the
    // generated code had v_mul_f64 overwrite the data registers one instruction later (intermittent
    // wrong rows at full size).  tools/check_dpp_hazard.py lints for the nop.
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// PRE: the lane's record comes in `pre` (requested one level ahead by k1_tail) instead of being loaded here
template <int KP, int FPL, int BATCH, int MODE, int NT, bool FULL, typename TS, bool VEC = false, int DOT = 0, bool PRE = false>
__device__ __forceinline__ void k1_body(const K1Params &P, int64_t start, int cnt, int64_t base,
                                        int64_t slot, double *prod, int lane, Rec pre = Rec())
{
    constexpr int SPW = 64 / KP;
    constexpr int NR = BATCH / SPW;
    const int k = P.k;
    const int stride = k1_stride(k);
    const int g = lane / KP;                         // slot
    const int q = lane % KP;                         // lane in slot
#ifdef MFX_K1_STAMPS
    const int64_t wave_off = 0, wave_g = slot;
#endif

    MFX_STAMP(0);
    // ---- the rating owned by this lane (lanes 0..cnt-1) -------------------------------
    Rec rec;
    rec.user = 0; rec.item = 0; rec.r = 0.0;
    if constexpr (PRE) {
        rec = pre;
    } else if (FULL ? (lane < BATCH) : (lane < cnt)) {
        // records are read once per epoch: non-temporal, so that the stream does not displace factor
        // rows from L2 / Infinity Cache (cfg 2 +0.1 %, cfg-3 shard shape +2.5 %, cfg-5 shard shape +0.6 %)
        typedef int v4i32 __attribute__((ext_vector_type(4)));
        const v4i32 raw = __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(P.recs + start + base + lane));
        rec = __builtin_bit_cast(Rec, raw);
    }
#ifdef MFX_K1_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MFX_STAMP(1);

    // ---- gather both factor rows of every rating of the batch -------------------------
    double uval[NR][FPL], vval[NR][FPL];
    TS *ur[NR];
    TS *vr[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
        int usr, itm;
        if constexpr (KP == 64) {
            usr = __builtin_amdgcn_readlane(rec.user, s);
            itm = __builtin_amdgcn_readlane(rec.item, s);
        } else {
            usr = __shfl(rec.user, s * SPW + g, 64);
            itm = __shfl(rec.item, s * SPW + g, 64);
        }
        ur[s] = (TS *)P.U + (size_t)itm * k;
        vr[s] = (TS *)P.V + (size_t)usr * k;
    }
    constexpr int VW = 16 / (int)sizeof(TS);             // features per 16-byte chunk
    static_assert(!VEC || FPL % VW == 0, "VEC: whole 16-byte chunks per lane");
    typedef typename Chunk16<TS>::type chunk_t;
    if constexpr (VEC) {
        // chunk j of lane q = features VW*q + VW*KP*j ...: every wave instruction moves 64 x 16 contiguous bytes
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
#pragma unroll
            for (int s = 0; s < NR; ++s) { uval[s][j] = 0.0; vval[s][j] = 0.0; }
        }
#pragma unroll
        for (int j = 0; j < FPL / VW; ++j) {
            const int f0 = VW * q + VW * KP * j;
            if (f0 < k) {                                 // k is a multiple of VW
#pragma unroll
                for (int s = 0; s < NR; ++s) {
                    if (FULL || (s * SPW + g < cnt)) {
                        const chunk_t a = *reinterpret_cast<const chunk_t *>(ur[s] + f0);
                        const chunk_t b = *reinterpret_cast<const chunk_t *>(vr[s] + f0);
#pragma unroll
                        for (int e = 0; e < VW; ++e) {
                            uval[s][VW * j + e] = widen<TS>(a[e]);
                            vval[s][VW * j + e] = widen<TS>(b[e]);
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const int f = j * KP + q;
    #pragma unroll
            for (int s = 0; s < NR; ++s) { uval[s][j] = 0.0; vval[s][j] = 0.0; }
            if (f < k) {
    #pragma unroll
                for (int s = 0; s < NR; ++s) {
                    if (FULL || (s * SPW + g < cnt)) {
                        uval[s][j] = (double)ur[s][f];
                        vval[s][j] = (double)vr[s][f];
                    }
                }
            }
        }
    }
    double ibv = 0.0, ubv = 0.0;
    if (FULL ? (lane < BATCH) : (lane < cnt)) {
        ibv = P.ib[rec.item];
        ubv = P.ub[rec.user];
    }
#ifdef MFX_K1_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MFX_STAMP(2);

    // ---- tolerance mode: per-lane partials (features ascending, FMA) + butterfly over the slot's lanes ----
    double dotv = 0.0;
    if constexpr (DOT == 1) {
        double part[NR];
#pragma unroll
        for (int s = 0; s < NR; ++s) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < FPL; ++j) a = __builtin_fma(uval[s][j], vval[s][j], a);
            part[s] = a;
        }
#pragma unroll
        for (int s = 0; s < NR; ++s) part[s] = group_sum<KP>(part[s]);
        // lane n (< BATCH) owns rating n = s * SPW + g: its dot sits in every lane of slot g, register s
#pragma unroll
        for (int s = 0; s < NR; ++s) {
            const double t = shfl_f64(part[s], (lane % SPW) * KP);
            if (lane / SPW == s) dotv = t;
        }
    }
    // ---- products to LDS: prod[n][f] -------------------------------------------------
    if constexpr (DOT == 1) {
    } else if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < FPL / VW; ++j) {
            const int f0 = VW * q + VW * KP * j;
            if (f0 < k) {
#pragma unroll
                for (int s = 0; s < NR; ++s) {
                    double *pp = prod + (s * SPW + g) * stride + f0;         // rows 16-B aligned, f0 even
#pragma unroll
                    for (int e = 0; e < VW; e += 2)
                        *reinterpret_cast<double2 *>(pp + e) = make_double2(uval[s][VW * j + e] * vval[s][VW * j + e],
                                                                            uval[s][VW * j + e + 1] * vval[s][VW * j + e + 1]);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            const int f = j * KP + q;
            if (f < k) {
    #pragma unroll
                for (int s = 0; s < NR; ++s) prod[(s * SPW + g) * stride + f] = uval[s][j] * vval[s][j];
            }
        }
    }
    if constexpr (DOT != 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    // ---- lane n: strictly sequential k-term sum, error, gradient, bias update ---------
    double grad = 0.0, e2 = 0.0, ubn = 0.0, ibn = 0.0;
    if (FULL ? (lane < BATCH) : (lane < cnt)) {
        double sum;
        if constexpr (MODE == K1_LEARNED) sum = P.mu + ibv + ubv;     // gd_estimator.pyx:138
        else sum = 0.0 + ibv + ubv;                                   // kmf_train.pyx:92
        if constexpr (DOT == 1) {
            sum += dotv;
        } else {
            const double *row = prod + lane * stride;
            // strictly left-to-right; LDS reads are issued 8 columns ahead of the dependent adds
            int f = 0;
            for (; f + 8 <= k; f += 8) {
                const double2 a = *reinterpret_cast<const double2 *>(row + f);
                const double2 b = *reinterpret_cast<const double2 *>(row + f + 2);
                const double2 c = *reinterpret_cast<const double2 *>(row + f + 4);
                const double2 d = *reinterpret_cast<const double2 *>(row + f + 6);
                sum += a.x; sum += a.y; sum += b.x; sum += b.y;
                sum += c.x; sum += c.y; sum += d.x; sum += d.y;
            }
            for (; f < k; ++f) sum += row[f];
        }
        double err;
        if constexpr (MODE == K1_LINEAR) {
            err = rec.r - sum;                                        // kmf_train.pyx:251-256
            grad = err;
        } else if constexpr (MODE == K1_LOGISTIC) {
            const double sig = 1.0 / (1.0 + exp(-sum));               // kmf_train.pyx:159-165
            const double p = 1.0 + sig * 4.0;
            err = rec.r - p;
            grad = err * sig * (1.0 - sig) * 4.0;
        } else {
            sum = clamp15(sum);                                       // gd_estimator.pyx:142-146
            sum += P.trail;
            sum = clamp15(sum);
            err = rec.r - sum;
            grad = err;
        }
        e2 = err * err;
        if constexpr (MODE == K1_LEARNED) {
            P.ub[rec.user] = ubv + P.lr_u * (err - P.Kb * ubv);       // gd_estimator.pyx:473-474
            P.ib[rec.item] = ibv + P.lr_i * (err - P.Kb * ibv);
        } else {
            // linear: unconditional (kmf_train.pyx:259-260); logistic: gated (:168-171).  The stores are
            // issued at the very end of the body: behind this branch hipcc drains the memory pipe
            // (s_waitcnt vmcnt(0)) before it goes on -- a store round trip in the middle of the wave
            // (+1.3 % from moving them; write-through or padded whole-granule bias stores: slower).
            ubn = ubv + P.lr * (grad - P.Kb * ubv);
            ibn = ibv + P.lr * (grad - P.Kb * ibv);
        }
    }
    MFX_STAMP(3);

    // ---- row updates ------------------------------------------------------------------
    double gr[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
        if constexpr (KP == 64) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(grad), s);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(grad), s);
            gr[s] = __hiloint2double(hi, lo);
        } else {
            gr[s] = shfl_f64(grad, s * SPW + g);
        }
    }
    if constexpr (VEC && MODE != K1_LEARNED) {
#pragma unroll
        for (int j = 0; j < FPL / VW; ++j) {
            const int f0 = VW * q + VW * KP * j;
            if (f0 < k) {
#pragma unroll
                for (int s = 0; s < NR; ++s) {
                    if (FULL || (s * SPW + g < cnt)) {
                        chunk_t un, vn;
#pragma unroll
                        for (int e = 0; e < VW; ++e) {
                            const double mf = uval[s][VW * j + e], cf = vval[s][VW * j + e];
                            if constexpr (DOT == 1) {
                                un[e] = narrow<TS>(__builtin_fma(P.lr, __builtin_fma(gr[s], cf, -(P.Ki * mf)), mf));
                                vn[e] = narrow<TS>(__builtin_fma(P.lr, __builtin_fma(gr[s], mf, -(P.Ku * cf)), cf));
                            } else {
                                un[e] = narrow<TS>(mf + P.lr * (gr[s] * cf - P.Ki * mf));       // kmf_train.pyx:267-270
                                vn[e] = narrow<TS>(cf + P.lr * (gr[s] * mf - P.Ku * cf));
                            }
                        }
                        TS *up = ur[s] + f0, *vp = vr[s] + f0;
                        if constexpr (NT == 2) {               // 16-byte write-through stores
                            if (FULL || P.upd_i) store16_sc1<TS>(up, un);
                            if (FULL || P.upd_u) store16_sc1<TS>(vp, vn);
                        } else {
                            if (FULL || P.upd_i) *reinterpret_cast<chunk_t *>(up) = un;
                            if (FULL || P.upd_u) *reinterpret_cast<chunk_t *>(vp) = vn;
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < FPL; ++j) {
        const int f = VEC ? (VW * q + VW * KP * (j / VW) + (j % VW)) : (j * KP + q);
        if constexpr (MODE == K1_LEARNED) {
            if (f == P.f) {                                           // gd_estimator.pyx:476-477
#pragma unroll
                for (int s = 0; s < NR; ++s) {
                    if (FULL || (s * SPW + g < cnt)) {
                        const double mf = uval[s][j], cf = vval[s][j];
                        ur[s][f] = (TS)(mf + P.lr * (gr[s] * cf - P.Ki * mf));
                        vr[s][f] = (TS)(cf + P.lr * (gr[s] * mf - P.Ku * cf));
                    }
                }
            }
        } else if (f < k) {
#pragma unroll
            for (int s = 0; s < NR; ++s) {
                if (FULL || (s * SPW + g < cnt)) {
                    const double mf = uval[s][j], cf = vval[s][j];
                    const TS un = (TS)(mf + P.lr * (gr[s] * cf - P.Ki * mf));     // kmf_train.pyx:267-270
                    const TS vn = (TS)(cf + P.lr * (gr[s] * mf - P.Ku * cf));
                    TS *up = ur[s] + f, *vp = vr[s] + f;
                    if constexpr (NT == 1) {
                        if (FULL || P.upd_i) __builtin_nontemporal_store(un, up);
                        if (FULL || P.upd_u) __builtin_nontemporal_store(vn, vp);
                    } else if constexpr (NT == 2) {        // write-through (sc1) stores
                        if (FULL || P.upd_i) __hip_atomic_store(up, un, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (FULL || P.upd_u) __hip_atomic_store(vp, vn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if constexpr (NT == 3) {        // diagnostic: no row stores
                        asm volatile("" ::"v"(un), "v"(vn));
                    } else {
                        if (FULL || P.upd_i) *up = un;
                        if (FULL || P.upd_u) *vp = vn;
                    }
                }
            }
        }
    }
    }
    MFX_STAMP(4);
#ifdef MFX_K1_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MFX_STAMP(5);

    // ---- bias stores (linear / logistic), after every other memory operation of the wave ------
    if constexpr (MODE != K1_LEARNED) {
        if (FULL ? (lane < BATCH) : (lane < cnt)) {
            if (FULL || MODE == K1_LINEAR || P.upd_u) P.ub[rec.user] = ubn;
            if (FULL || MODE == K1_LINEAR || P.upd_i) P.ib[rec.item] = ibn;
        }
    }

    // ---- squared-error partial of this wave (deterministic) ---------------------------
    const double wsum = wave_sum_fixed(e2);
    if (lane == 0) P.se_partial[slot] = wsum;
}

template <int KP, int FPL, int BATCH, int MODE, int NT, typename TS = double, bool VEC = false, int DOT = 0>
__global__ __launch_bounds__(kWaves * 64) void k1_level(K1Params P, int64_t start, int count,
                                                       int64_t wave_off)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int64_t wave_g = (int64_t)blockIdx.x * kWaves + wid;
    const int64_t base = wave_g * BATCH;
    if (base >= count) {                             // wave-uniform; padding wave of the last block
        if (lane == 0) P.se_partial[wave_off + wave_g] = 0.0;
        return;
    }
    const int cnt = (count - base) < BATCH ? (int)(count - base) : BATCH;
    double *prod = lds + (size_t)wid * BATCH * k1_stride(P.k);
    if (cnt == BATCH && P.upd_u && P.upd_i)
        k1_body<KP, FPL, BATCH, MODE, NT, true, TS, VEC, DOT>(P, start, cnt, base, wave_off + wave_g, prod, lane);
    else
        k1_body<KP, FPL, BATCH, MODE, NT, false, TS, VEC, DOT>(P, start, cnt, base, wave_off + wave_g, prod, lane);
}

// Tail kernel: a run of consecutive SMALL levels (each <= kTailWaves*BATCH ratings) in ONE
// launch by ONE workgroup, __syncthreads() between levels (same-CU visibility of the row
// stores is guaranteed at workgroup scope).  With power-law data most levels are the long
// chain of a few hot items/users with a handful of ratings each; a launch per level would
// cost ~9 us for ~no work.
static constexpr int kTailWaves = 16;
template <int KP, int FPL, int BATCH, int MODE, int NT, typename TS = double, bool VEC = false, int DOT = 0>
__global__ __launch_bounds__(kTailWaves * 64) void k1_tail(K1Params P, const int64_t *level_off,
                                                          const int64_t *wave_off, int l0, int l1)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    double *prod = lds + (size_t)wid * BATCH * k1_stride(P.k);
    // Small levels are latency chains: offsets (scalar loads) -> record -> row gathers -> stores -> barrier.  The
    // offsets of level l + 3 and the records of level l + 1 are requested while level l runs (records never change).
    typedef int v4i32 __attribute__((ext_vector_type(4)));
    const int base = wid * BATCH;
    int64_t s0 = level_off[l0], s1 = level_off[l0 + 1], w0 = wave_off[l0], w1 = wave_off[l0 + 1];
    int64_t s2 = s1, w2 = w1;                            // level l + 1's end: read one level before it is needed
    if (l0 + 1 < l1) { s2 = level_off[l0 + 2]; w2 = wave_off[l0 + 2]; }
    Rec cur;
    cur.user = 0; cur.item = 0; cur.r = 0.0;
    if (base + lane < (int)(s1 - s0) && lane < BATCH)
        cur = __builtin_bit_cast(Rec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(P.recs + s0 + base + lane)));
    for (int l = l0; l < l1; ++l) {
        const int64_t start = s0;
        const int count = (int)(s1 - s0);
        const int64_t woff = w0;
        const int nslots = (int)(w1 - w0);
        int64_t s3 = s2, w3 = w2;
        if (l + 2 < l1) { s3 = level_off[l + 3]; w3 = wave_off[l + 3]; }
        Rec nxt;
        nxt.user = 0; nxt.item = 0; nxt.r = 0.0;
        if (l + 1 < l1 && base + lane < (int)(s2 - s1) && lane < BATCH)
            nxt = __builtin_bit_cast(Rec, __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(P.recs + s1 + base + lane)));
        if (wid < nslots) {
            if (base >= count) {
                if (lane == 0) P.se_partial[woff + wid] = 0.0;
            } else {
                const int cnt = (count - base) < BATCH ? (count - base) : BATCH;
                if (cnt == BATCH && P.upd_u && P.upd_i)
                    k1_body<KP, FPL, BATCH, MODE, NT, true, TS, VEC, DOT, true>(P, start, cnt, base, woff + wid, prod, lane, cur);
                else
                    k1_body<KP, FPL, BATCH, MODE, NT, false, TS, VEC, DOT, true>(P, start, cnt, base, woff + wid, prod, lane, cur);
            }
        }
        cur = nxt;
        s0 = s1; s1 = s2; s2 = s3; w0 = w1; w1 = w2; w2 = w3;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// K2: Funk per-feature pass.  One lane per rating of one level; the feature vectors
// uf = U[f, :], vf = V[f, :] are feature-major slices (L2 resident).
// ------------------------------------------------------------------------------------
// per-feature state, read through a device pointer so that ONE captured epoch graph serves
// every feature of the run (only this 24-byte record changes between features)
struct K2Dyn {
    double *uf;           // U + f*ni
    double *vf;           // V + f*nu
    double trail;         // (k - f - 1) * f_init^2, gd_estimator.pyx:70
};

struct K2Params {
    const Rec *recs;
    const K2Dyn *dyn;
    double *cache;        // level-major, one per rating
    const int32_t *slot;  // nullable: dense-cache semantics (gd_estimator.pyx:250-255) -- ratings
                          // with equal (user, item) share the cell of their first occurrence
    const double *ib;     // nullable (without_bias)
    const double *ub;
    double *se_partial;
    double lr, K, mu;
    int upd_u, upd_i;
};

__device__ __forceinline__ double funk_estimate(double uf, double vf, double cache, double base,
                                                double trail, bool trailing)
{
    // gd_estimator.pyx:38-73
    double sum = (cache > 0) ? cache : base;
    sum += uf * vf;
    sum = clamp15(sum);
    if (trailing) {
        sum += trail;
        sum = clamp15(sum);
    }
    return sum;
}

// PRE: record and cache word come in (rec0, c0), requested one level ahead by k2_tail
template <bool WT = false, bool PRE = false>
__device__ __forceinline__ double k2_one(const K2Params &P, double *uf, double *vf, double trail,
                                         int64_t start, int t, int count, Rec rec0 = Rec(), double c0 = 0.0)
{
    double e2 = 0.0;
    if (t < count) {
        Rec rec;
        double c;
        if constexpr (PRE) { rec = rec0; c = c0; }
        else { rec = P.recs[start + t]; c = P.cache[P.slot ? (int64_t)P.slot[start + t] : start + t]; }
        double base;
        if (P.ib) base = P.mu + P.ib[rec.item] + P.ub[rec.user];
        else base = P.mu + 0.0 + 0.0;
        const double mf = uf[rec.item], cf = vf[rec.user];
        const double p = funk_estimate(mf, cf, c, base, trail, true);
        const double err = rec.r - p;
        e2 = err * err;
        const double un = mf + P.lr * (err * cf - P.K * mf), vn = cf + P.lr * (err * mf - P.K * cf);
        if constexpr (WT) {
            // one launch per level: write-through (sc1) stores leave no dirty L2 lines for the
            // end-of-kernel write-back (2.35 -> 2.08 ms per pass at cfg 2)
            if (P.upd_i) __hip_atomic_store(uf + rec.item, un, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (P.upd_u) __hip_atomic_store(vf + rec.user, vn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (P.upd_i) uf[rec.item] = un;
            if (P.upd_u) vf[rec.user] = vn;
        }
    }
    return e2;
}

#ifndef MFX_K2_BLOCK
#define MFX_K2_BLOCK 64
#endif
static constexpr int kK2Block = MFX_K2_BLOCK;    // threads per block of the Funk level kernel
// WT: write-through scalar stores -- faster for small levels (16 K ratings: 2.35 -> 2.08 ms per pass),
// slower for large ones (37 K ratings: 1.23 -> 1.34 ms per pass); the host picks per level
static constexpr int kK2WriteThroughMax = 24576;
template <bool WT>
__global__ __launch_bounds__(kK2Block) void k2_level(K2Params P, int64_t start, int count,
                                                     int64_t wave_off)
{
    const int t = blockIdx.x * kK2Block + threadIdx.x;
    double *const uf = P.dyn->uf, *const vf = P.dyn->vf;
    const double e2 = k2_one<WT>(P, uf, vf, P.dyn->trail, start, t, count);
    const double wsum = wave_sum_fixed(e2);
    if ((threadIdx.x & 63) == 0)
        P.se_partial[wave_off + (int64_t)blockIdx.x * (kK2Block / 64) + (threadIdx.x >> 6)] = wsum;
}

// run of small levels (each <= 1024 ratings) by one 1024-thread workgroup
static constexpr int kTail2 = 1024;
__global__ __launch_bounds__(kTail2) void k2_tail(K2Params P, const int64_t *level_off,
                                                  const int64_t *wave_off, int l0, int l1)
{
    double *const uf = P.dyn->uf, *const vf = P.dyn->vf;
    const double trail = P.dyn->trail;
    const int wid = threadIdx.x >> 6, t = (int)threadIdx.x;
    // as in k1_tail: offsets two levels ahead, the next level's records and cache words (no update touches them
    // inside a pass) requested while this level runs
    int64_t s0 = level_off[l0], s1 = level_off[l0 + 1], w0 = wave_off[l0], w1 = wave_off[l0 + 1];
    int64_t s2 = s1, w2 = w1;
    if (l0 + 1 < l1) { s2 = level_off[l0 + 2]; w2 = wave_off[l0 + 2]; }
    auto fetch = [&](int64_t st, int cnt, Rec &rc, double &c) {
        rc.user = 0; rc.item = 0; rc.r = 0.0; c = 0.0;
        if (t < cnt) {
            rc = P.recs[st + t];
            c = P.cache[P.slot ? (int64_t)P.slot[st + t] : st + t];
        }
    };
    Rec cur, nxt;
    double ccur, cnxt;
    fetch(s0, (int)(s1 - s0), cur, ccur);
    for (int l = l0; l < l1; ++l) {
        const int64_t start = s0;
        const int count = (int)(s1 - s0);
        const int64_t woff = w0;
        const int nslots = (int)(w1 - w0);                       // one slot per launched wave of the level
        int64_t s3 = s2, w3 = w2;
        if (l + 2 < l1) { s3 = level_off[l + 3]; w3 = wave_off[l + 3]; }
        fetch(s1, l + 1 < l1 ? (int)(s2 - s1) : 0, nxt, cnxt);
        const double e2 = k2_one<false, true>(P, uf, vf, trail, start, t, count, cur, ccur);
        const double wsum = wave_sum_fixed(e2);
        if ((threadIdx.x & 63) == 0 && wid < nslots) P.se_partial[woff + wid] = wsum;
        cur = nxt; ccur = cnxt;
        s0 = s1; s1 = s2; s2 = s3; w0 = w1; w1 = w2; w2 = w3;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K2, XCD-resident form: ONE persistent launch per (feature, epoch) pass that runs on a SINGLE XCD.
//
// A pass is a chain of dependency levels of a few thousand scalar updates each; with a launch per level it
// costs ~3.4 us per level whatever the level's size.  The two feature vectors U[f,:] + V[f,:] are ~1 MB:
// inside ONE XCD the L2 is the coherence point (plain stores stay there, sc1 loads bypass the CU's L1),
// so a level hand-off costs L2 round trips instead of a kernel boundary.  Users are cut into kXGroups
// groups; a workgroup keeps its groups' slice of V[f,:] in LDS for the whole pass (a user's ratings all
// belong to one workgroup), only the item side travels through L2; the ratings of a level are ordered
// (group, item) (mfx_schedule_dev.hip: levels_group_sort).  Workgroup = 1 sync wave + kXWorkers worker
// waves; flat barrier: every workgroup stores its flag (own 128-B line), every sync wave polls all flags.
// Measured (tools/k2_xcd.hip, cfg 2): 2.2 us per level against 3.4; per-rating tagged hand-offs instead
// of the barrier: 2.9-3.6 us (poll storms), both vectors through L2: 2.5-3.5 us.
// The 256 launched workgroups register by XCC id, the XCD with the most is chosen, the others leave.
// Nothing is modified before every participant has passed a first barrier: a failed registration (grid
// not co-resident, LDS too small) leaves the data untouched and the host falls back to the launch path.
static constexpr int kXWorkers = 15;
static constexpr int kXGroups = 32;
static constexpr unsigned kXSpinSetup = 1u << 22;     // ~1 s: registration / first barrier
static constexpr unsigned kXSpinRun = 0x7fffffffu;    // minutes: only a lost workgroup ends here

struct K2XParams {
    const Rec *recs;
    const K2Dyn *dyn;
    const double *cache;
    const double *ib;          // nullable (without_bias)
    const double *ub;
    const int32_t *seg_off;    // [n_levels * kXGroups + 1]: segment (level, group) = [seg_off[q], seg_off[q+1])
    const int32_t *seg_mid;    // [n_levels * kXGroups]: its LATE ratings [seg_off[q], seg_mid[q]), EARLY the rest
    double *se_partial;        // [kXGroups][kXWorkers][64]: one accumulator per (group, worker, lane)
    unsigned *ctl;             // [x*32] registrations of XCD x, [512] status, [1024 + slot*32] flags
    int n_levels, nu;
    int lds_doubles;
    int tune;                  // bit 1: no early phase -- every rating runs behind its own level's barrier (A/B hook,
                               // MFX_K2_TUNE=2; same-process measurement: tools/k2_tune.py)
    unsigned gen_base;         // flags count up across passes: gen = gen_base + level + 1
    double lr, K, mu;
    int upd_u, upd_i;
};

__device__ __forceinline__ unsigned ld_sc1_u32(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__((kXWorkers + 1) * 64) void k2_xcd_pass(K2XParams P)
{
    extern __shared__ double vlds[];           // this workgroup's slices of vf
    __shared__ unsigned sh[6];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // ---- registration ------------------------------------------------------------------------
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7u;
        const unsigned slot = atomicAdd(P.ctl + x * 32, 1u);
        unsigned spins = 0, cnt[8], tot = 0;
        bool ok = true;
        for (;;) {
            tot = 0;
            for (int y = 0; y < 8; ++y) { cnt[y] = ld_sc1_u32(P.ctl + y * 32); tot += cnt[y]; }
            if (tot >= gridDim.x) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kXSpinSetup || ld_sc1_u32(P.ctl + 512)) { ok = false; break; }
        }
        unsigned best = 0;
        for (int y = 1; y < 8; ++y) if (cnt[y] > cnt[best]) best = y;
        if (!ok) atomicExch(P.ctl + 512, 1u);
        sh[0] = x; sh[1] = slot; sh[2] = cnt[best]; sh[3] = ok ? 1u : 0u; sh[4] = best;
    }
    __syncthreads();
    if (!sh[3] || sh[0] != sh[4]) return;               // not on the chosen XCD (or failed): leave
    const unsigned slot = sh[1], nx = sh[2] > 64 ? 64 : sh[2];
    if (slot >= nx) return;
    unsigned *flags = P.ctl + 1024;
    double *const uf = P.dyn->uf, *const vf = P.dyn->vf;
    double trail = P.dyn->trail, lr = P.lr, K = P.K, mu = P.mu;
    // uniform, but kept in VECTOR registers on purpose: the kernel sits at the SGPR limit (the per-level prefetch state
    // is uniform), and a spill to scratch memory costs 10-60 % (measured); 54 of 128 VGPRs are in use
    asm volatile("" : "+v"(trail), "+v"(lr), "+v"(K), "+v"(mu));
    const int NG = kXGroups, nu = P.nu;
    auto glo = [nu](int g) { return (int)(((int64_t)nu * g + kXGroups - 1) / kXGroups); };   // first user of group g
    // (group(u) = u * NG / nu  <=>  u in [ceil(nu * g / NG), ceil(nu * (g+1) / NG)))

    // my groups' user slices -> LDS (groups slot, slot + nx, ...)
    int need = 0, ngroups = 0;
    for (int g = slot; g < NG; g += nx, ++ngroups) need += glo(g + 1) - glo(g);
    if (need > P.lds_doubles || ngroups > 2) {           // uniform over the workgroup
        if (threadIdx.x == 0) atomicExch(P.ctl + 512, 2u);
        return;
    }
    {
        int base = 0;
        for (int g = slot; g < NG; g += nx) {
            const int lo = glo(g), n = glo(g + 1) - lo;
            for (int i = threadIdx.x; i < n; i += (kXWorkers + 1) * 64) vlds[base + i] = vf[lo + i];
            base += n;
        }
    }
    // (measured with run-time hooks, same process: 8 or 12 worker waves instead of 15: +1 %; 4: +16 %; raising the
    // sync wave's priority: nothing -- the level is a latency chain, not a throughput problem)
    const int worker = wid - 1;                          // -1 = the sync wave
    // barrier(gen): A = my workgroup is done, flag, poll everybody's flag, B
    auto barrier = [&](unsigned gen, unsigned limit) -> bool {
        __syncthreads();
        if (worker == -1) {
            if (lane == 0) flags[slot * 32] = gen;       // plain store: stays in this XCD's L2
            unsigned spins = 0;
            for (;;) {
                const bool ok = lane >= (int)nx || (int)(ld_sc1_u32(flags + lane * 32) - gen) >= 0;
                if (__all(ok)) break;
                if (++spins > limit || ld_sc1_u32(P.ctl + 512)) {
                    if (lane == 0) atomicExch(P.ctl + 512, limit == kXSpinSetup ? 1u : 3u);
                    break;
                }
            }
        }
        __syncthreads();
        return ld_sc1_u32(P.ctl + 512) == 0;
    };
    if (!barrier(P.gen_base, kXSpinSetup)) return;      // nothing has been modified yet

    // squared errors: one accumulator per (user group, worker, lane), summed in processing order and written
    // once at the end of the pass -- a layout fixed by the schedule alone (a segment's late / early part is cut
    // into 64-rating chunks dealt to the workers in turn), so the epoch RMSE has the same bits whatever the
    // placement; and no wave reduction + store sits on a level's critical path
    double se_acc[2] = {0.0, 0.0};                       // a workgroup holds at most two groups (LDS, above)
    // One part (late or early) of level l for my groups.  LATE ratings (item updated in level l - 1) can only
    // run once level l - 1 is complete everywhere; EARLY ones (item final since level l - 2 or before) run one
    // barrier earlier, while that barrier is still in flight.  Records and cache words never change: the first
    // late chunk of the next level is requested before the barrier (pre).
    struct Pre {
        Rec rec;
        double cache;
    };
    // the segment bounds of my first group for the level about to open are read together with the prefetch, one level
    // ahead of their use: a scalar load in front of every part sat on the level's critical path (same-process A/B,
    // MFX_K2_TUNE bit 4 = the old way: 1.580 -> 1.466 ms per shuffled pass), and so is the first chunk of the EARLY part
    // of that level (bit 8 = off: 1.466 -> 1.354 ms), which runs right behind barrier A with a ~1 us record load in front
    // otherwise.  Two traps on the way, both measured: a prefetch struct handed to run_part through a pointer that may
    // be null is pinned in scratch memory (+10-60 %), and so is per-level state kept in an indexed array.
    int nb_level = -1, nb_off = 0, nb_mid = 0, nb_next = 0;
    const bool pre_early = !(P.tune & (2 | 4 | 8));      // bit 8: no prefetch of the early part's first chunk
    auto fetch_first_late = [&](int l, Pre &x, Pre &e) {
        x.cache = 0.0;
        x.rec.user = 0; x.rec.item = 0; x.rec.r = 0.0;
        e = x;
        if (l < P.n_levels && (int)slot < NG) {
            const int q = l * NG + (int)slot;
            const int o = P.seg_off[q], m = P.seg_mid[q], nx_ = P.seg_off[q + 1];
            if (!(P.tune & 4)) { nb_level = l; nb_off = o; nb_mid = m; nb_next = nx_; }
            const int t = o + worker * 64 + lane;
            if (worker >= 0 && t < m) {
                x.rec = P.recs[t];
                x.cache = P.cache[t];
            }
            const int te = m + worker * 64 + lane;       // early(l) runs right behind barrier A: its records too
            if (pre_early && worker >= 0 && te < nx_) {
                e.rec = P.recs[te];
                e.cache = P.cache[te];
            }
        }
    };
    auto run_part = [&](int l, bool late, const Pre &pre, bool have) {   // (a pointer that may be null would pin the
        int base = 0, gi = 0;                                            //  struct in scratch memory: measured)
        bool first = have;
        for (int g = slot; g < NG; g += nx, ++gi) {
            const int lo = glo(g);
            const int q = l * NG + g;
            int s0, s1;
            if (gi == 0 && nb_level == l) {
                s0 = late ? nb_off : nb_mid;
                s1 = late ? nb_mid : nb_next;
            } else {
                s0 = late ? P.seg_off[q] : P.seg_mid[q];
                s1 = late ? P.seg_mid[q] : P.seg_off[q + 1];
            }
            for (int tb = s0 + worker * 64; tb < s1; tb += kXWorkers * 64) {      // wave-uniform
                const int t = tb + lane;
                if (t < s1) {
                    const Rec rec = first ? pre.rec : P.recs[t];
                    const double c = first ? pre.cache : P.cache[t];
                    double bs;
                    if (P.ib) bs = mu + P.ib[rec.item] + P.ub[rec.user];
                    else bs = mu + 0.0 + 0.0;
                    const double mf = __hip_atomic_load(uf + rec.item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double cf = vlds[base + rec.user - lo];
                    const double p = funk_estimate(mf, cf, c, bs, trail, true);
                    const double err = rec.r - p;
                    if (P.upd_i) uf[rec.item] = mf + lr * (err * cf - K * mf);
                    if (P.upd_u) vlds[base + rec.user - lo] = cf + lr * (err * mf - K * cf);
                    if (gi == 0) se_acc[0] += err * err;
                    else se_acc[1] += err * err;
                }
                first = false;
            }
            first = false;
            base += glo(g + 1) - lo;
        }
    };
    // (requesting it two levels ahead instead of one: measured neutral)
    Pre cur, pe;
    if (worker >= 0) {
        run_part(0, false, cur, false);                  // level 0 has no late ratings: all of it is early
        fetch_first_late(0, cur, pe);                    // (empty; keeps the loop uniform)
    }
    for (int l = 0; l < P.n_levels; ++l) {
        if (worker >= 0) {
            run_part(l, true, cur, true);                         // late(l): needs barrier l - 1, which has passed
            if ((P.tune & 2) && l > 0) run_part(l, false, cur, false);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my stores of level l (late and early) are in L2
            fetch_first_late(l + 1, cur, pe);
        }
        // level l is complete in this workgroup: signal; while the flags travel, the workers run early(l + 1)
        __syncthreads();                                          // A
        if (worker == -1) {
            const unsigned gen = P.gen_base + (unsigned)l + 1u;
            if (lane == 0) flags[slot * 32] = gen;               // plain store: stays in this XCD's L2
            unsigned spins = 0;
            for (;;) {
                const bool ok = lane >= (int)nx || (int)(ld_sc1_u32(flags + lane * 32) - gen) >= 0;
                if (__all(ok)) break;
                if (++spins > kXSpinRun || ld_sc1_u32(P.ctl + 512)) {
                    if (lane == 0) atomicExch(P.ctl + 512, 3u);
                    break;
                }
            }
        } else if (worker >= 0 && l + 1 < P.n_levels && !(P.tune & 2)) {
            run_part(l + 1, false, pe, pre_early);                // early(l + 1): items final since level l - 1
        }
        __syncthreads();                                          // B
        if (ld_sc1_u32(P.ctl + 512)) return;
    }
    if (wid >= 1) {                                      // every worker slot is written every pass (idle ones: 0)
        int gi = 0;
        for (int g = slot; g < NG; g += nx, ++gi)
            P.se_partial[((size_t)g * kXWorkers + (wid - 1)) * 64 + lane] = gi == 0 ? se_acc[0] : se_acc[1];
    }
    // LDS slices back to V[f,:]
    if (P.upd_u) {
        int base = 0;
        for (int g = slot; g < NG; g += nx) {
            const int lo = glo(g), n = glo(g + 1) - lo;
            for (int i = threadIdx.x; i < n; i += (kXWorkers + 1) * 64) vf[lo + i] = vlds[base + i];
            base += n;
        }
    }
}

// cache-commit pass (gd_estimator.pyx:771-777): embarrassingly parallel over all ratings
__global__ __launch_bounds__(256) void k2_commit(K2Params P, int64_t nnz)
{
    const double *const uf = P.dyn->uf, *const vf = P.dyn->vf;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nnz;
         t += (int64_t)gridDim.x * 256) {
        const Rec rec = P.recs[t];
        double base;
        if (P.ib) base = P.mu + P.ib[rec.item] + P.ub[rec.user];
        else base = P.mu + 0.0 + 0.0;
        P.cache[t] = funk_estimate(uf[rec.item], vf[rec.user], P.cache[t], base, 0.0, false);
    }
}

// cache-commit of ONE level (dense-cache variants: duplicates of a cell are in different
// levels and must be applied in order, gd_estimator.pyx:295-301)
__global__ __launch_bounds__(256) void k2_commit_level(K2Params P, int64_t start, int count)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const double *const uf = P.dyn->uf, *const vf = P.dyn->vf;
    const Rec rec = P.recs[start + t];
    const int64_t cs = P.slot ? (int64_t)P.slot[start + t] : start + t;
    double base;
    if (P.ib) base = P.mu + P.ib[rec.item] + P.ub[rec.user];
    else base = P.mu + 0.0 + 0.0;
    P.cache[cs] = funk_estimate(uf[rec.item], vf[rec.user], P.cache[cs], base, 0.0, false);
}

// ------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------
// Deterministic two-stage sum of the per-wave squared-error partials.
// stage 1: block b sums its contiguous chunk (fixed per-thread stride + fixed LDS tree)
static constexpr int kReduceBlocks = 128;
__global__ __launch_bounds__(256) void k_reduce_se1(const double *partial, int64_t n, double *tmp)
{
    __shared__ double sh[256];
    const int64_t chunk = (n + kReduceBlocks - 1) / kReduceBlocks;
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) tmp[blockIdx.x] = sh[0];
}
// stage 2: out[*counter] = tmp[0] + tmp[1] + ... (in order); ++*counter
__global__ void k_reduce_se2(const double *tmp, double *out, int *counter)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < kReduceBlocks; ++i) s += tmp[i];
        out[*counter] = s;
        *counter = *counter + 1;
    }
}

// out[c][r] = in[r][c]; in is rows x cols row-major; element types converted on the way.
// 1-D grid of ceil(cols/32) * ceil(rows/32) tiles (a 2-D grid would cap rows at 65535 * 32).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_transpose(const TI *in, TO *out, int rows, int cols)
{
    __shared__ double tile[32][33];
    const unsigned nbx = (unsigned)(cols + 31) / 32;
    const int bx = (int)(blockIdx.x % nbx) * 32, by = (int)(blockIdx.x / nbx) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = by + i, c = bx + tx;
        if (r < rows && c < cols) tile[i][tx] = widen<TI>(in[(size_t)r * cols + c]);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = bx + i, r = by + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = narrow<TO>(tile[tx][i]);
    }
}

}  // namespace mfx
