// Resident-side full-k SGD epoch (round 3): train_linear_kernel / train_logistic_kernel
// (kmf_train.pyx:195-277, :103-189) as ONE persistent launch per epoch instead of one launch per
// dependency level.
//
// The sequential loop's result is fixed by two orders only: the ratings of a user in input order, the
// ratings of an item in input order.  Here ONE side of every rating is RESIDENT: a
// wave keeps the factor rows and biases of its entities (users or items) in registers for a whole run of
// their ratings and walks each run in input order, so that side's order needs no synchronisation at all
// and its rows move once per run instead of twice per rating.  The other side TRAVELS: its rows go
// through the coherent path (16-byte sc1 loads and write-through stores, the level kernel's accesses), and
// a rating waits until a per-entity chain counter says that its predecessor in that entity's input order
// has been applied.  Counter and bias share one 16-byte granule: the poll returns the bias, the publish
// stores both behind the wave's `s_waitcnt vmcnt(0)` (every row store acknowledged: MI355X_MICROARCH.md,
// hand-offs with sc1 loads in place of the acquire).  Arithmetic per rating is k1_body's, operation for
// operation: the factors come out bit-identical to the level schedule and to the reference loop.
//
// Two bindings of entities to waves:
//   TICKET  (input grouped by user, e.g. get_ratings(randomize_order=False), base.py:1115-1126: north_star's
//           "ratings streamed in CSR blocks, hot rows staged"): a wave takes the next B user runs by ticket.
//           Tickets go out in input order, so the earliest unfinished rating always belongs to a running
//           wave: no deadlock for any grid size, any residency.  Any number of users.
//   STATIC  (ANY input order): every entity of the smaller side has a fixed owner (wave, slot) for the whole
//           epoch; the ratings are grouped by owner entity (stable).  Needs the whole side in registers
//           (<= waves x B rows) and the whole grid co-resident: the launch opens with a registration barrier
//           and NOTHING is modified unless every wave has arrived (status word -> the host falls back to
//           the level schedule and stays there).
// Every spin is bounded; a time-out mid-epoch sets the status word and the host returns MFX_E_HIP.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <type_traits>
#include <vector>
#include "mfx_common.h"
#include "mfx_sgd_internal.h"

namespace mfx {

#ifdef MFX_BURST_STATS
__device__ unsigned long long g_burst_stats[16];
#endif

typedef ResRec URec;          // {travelling entity, its chain rank, rating}: 16 bytes, built by res_build_device
struct __attribute__((aligned(16))) THdr { double bias; int32_t cnt; int32_t pad; };  // travelling entity: bias + chain counter

struct ResParams {
    const URec *recs;           // grouped by resident entity (stable)
    const int32_t *lo, *hi;     // [n_res] record range of a resident entity
    const int32_t *run_ent;     // ticket: [n_runs] resident entity of a run, in input order; static: null (run = entity)
    double *Tm, *Rm;            // travelling / resident rows, [n][k]
    THdr *hdr;                  // [n_trav]
    double *rb;                 // resident biases
    unsigned *ctl;              // [0] ticket, [32] status, [64] registration
    double *se_run;             // squared error per run (fixed layout: deterministic epoch RMSE)
    int n_runs, k, res_item;
    int reg_extra;              // test hook: waves that will never register (the co-residency check must fail)
    int burst_run;              // burst walk: ratings a slot must have ahead for a burst to be taken (below)
    double lr, Kres, Ktrav, Kb;
};

namespace {

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kCtlStatus = 32, kCtlDone = 33, kCtlReg = 64, kCtlWords = 96;   // [33]: static launches of the batch that passed their check
constexpr unsigned kRegFail = 0x80000000u;
constexpr int kSpinRun = 1 << 22;        // polls of ~1 us: seconds
constexpr int kSpinReg = 1 << 15;        // registration: tens of milliseconds

// 16-byte agent-scope (sc1) accesses the compiler can see and count
__device__ __forceinline__ v4u ld16c(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ __forceinline__ void st16c(v4u v, __amdgpu_buffer_rsrc_t r, unsigned off) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
// one lane's feature pair of a row in the plan's storage type TS: f64 rows move 16 bytes per lane (exact mode), f32 rows 8 bytes
// ("f32 storage, f64 arithmetic": widened on load, rounded once on store -- the level kernels' narrow-storage option,
// mfx_sgd_plan_set_storage(32); NOT the reference's precision)
template <typename TS> struct RowIO;
template <> struct RowIO<double> {
    static constexpr unsigned lane_bytes = 16u;
    static __device__ __forceinline__ d2 ld(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(d2, ld16c(r, off)); }
    static __device__ __forceinline__ void st(d2 v, __amdgpu_buffer_rsrc_t r, unsigned off) { st16c(__builtin_bit_cast(v4u, v), r, off); }
    static __device__ __forceinline__ d2 round(d2 v) { return v; }
    static __device__ __forceinline__ d2 ldp(const double *rows, size_t e) { return *reinterpret_cast<const d2 *>(rows + e); }
    static __device__ __forceinline__ void stp(double *rows, size_t e, d2 v) { *reinterpret_cast<d2 *>(rows + e) = v; }
};
template <> struct RowIO<float> {
    typedef float f2 __attribute__((ext_vector_type(2)));
    static constexpr unsigned lane_bytes = 8u;
    static __device__ __forceinline__ d2 ld(__amdgpu_buffer_rsrc_t r, unsigned off)
    {
        const f2 w = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16));
        return d2{(double)w.x, (double)w.y};
    }
    static __device__ __forceinline__ void st(d2 v, __amdgpu_buffer_rsrc_t r, unsigned off)
    {
        const f2 w = f2{(float)v.x, (float)v.y};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, w), r, off, 0, 16);
    }
    static __device__ __forceinline__ d2 round(d2 v) { return d2{(double)(float)v.x, (double)(float)v.y}; }   // what the next load would return
    static __device__ __forceinline__ d2 ldp(const double *rows, size_t e)
    {
        const f2 w = *reinterpret_cast<const f2 *>(reinterpret_cast<const float *>(rows) + e);
        return d2{(double)w.x, (double)w.y};
    }
    static __device__ __forceinline__ void stp(double *rows, size_t e, d2 v)
    {
        *reinterpret_cast<f2 *>(reinterpret_cast<float *>(rows) + e) = f2{(float)v.x, (float)v.y};
    }
};
// a slot's next record.  Plain (cached) load: a lane walks its run 16 bytes at a time and eight records share a 128-byte line.
// (Measured against non-temporal loads: FETCH_SIZE and the epoch time are the same within the box-to-box spread.)
#ifndef MFX_RES_REC_NT
#define MFX_RES_REC_NT 0
#endif
__device__ __forceinline__ ResRec ldrec(const ResRec *p)
{
#if MFX_RES_REC_NT
    return __builtin_bit_cast(ResRec, __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p)));
#else
    return *p;
#endif
}
// every outstanding vector-memory operation of the wave has returned -- as a REAL s_waitcnt instruction (vmcnt 0, the other counters
// untouched), which the compiler's wait-count pass sees and accounts for; an `asm volatile` wait is opaque to it.  Placed in uniform
// control flow behind a block of exec-masked loads: without it the compiler only waits INSIDE the masked blocks that use a loaded
// register, must assume the load still pending where such a block was skipped, and protects the register's next write with a
// vmcnt(0) of its own -- which in a step of k1_res sat right behind the FIRST row store and made every step wait for that store's
// acknowledge before the other stores were even issued (a fourth memory round trip per step; round 4, from the ISA).
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// Diagnostic builds only (MFX_EXTRA_FLAGS=-DMFX_RES_STAMPS -> libmfx_diag.so, tools/res_stamps.py): where a step of the resident-side
// epoch spends its time.  Every 64th wave accumulates the shader-clock ticks between the phase boundaries of its steps.
#ifdef MFX_RES_STAMPS
__device__ unsigned long long g_res_stamps[64 * 16];
#define RS_DECL unsigned long long rs_t = clock64(), rs_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long rs_steps = 0, rs_spins = 0, rs_moved = 0
#define RS(i) do { const unsigned long long t_ = clock64(); rs_acc[i] += t_ - rs_t; rs_t = t_; } while (0)
#define RS_STEP(n) do { ++rs_steps; rs_moved += (n); } while (0)
#define RS_SPIN() do { ++rs_spins; } while (0)
#define RS_PARAM , unsigned long long &rs_t, unsigned long long (&rs_acc)[10]
#define RS_ARG , rs_t, rs_acc
#define RS_FLUSH() do { if ((blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 64 && threadIdx.x == 0) {                  \
        unsigned long long *o_ = g_res_stamps + (blockIdx.x >> 6) * 16;                                               \
        for (int i_ = 0; i_ < 10; ++i_) o_[i_] = rs_acc[i_];                                                          \
        o_[10] = rs_steps; o_[11] = rs_spins; o_[12] = rs_moved; } } while (0)
#else
#define RS_DECL
#define RS(i)
#define RS_STEP(n)
#define RS_SPIN()
#define RS_FLUSH()
#define RS_PARAM
#define RS_ARG
#endif
__device__ __forceinline__ unsigned ldc_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ double shfl64(double x, int src)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

// Registration barrier of the static launches (k1_res STATIC, k2_res, k2_res_box): lane 0 of every wave calls this; NOTHING is
// modified unless it returns 1 in EVERY wave of the grid.  One word: arrivals count up; a wave that has waited kSpinReg polls gives
// up by raising kRegFail -- with a compare-and-swap that only succeeds on an INCOMPLETE count, so the flag and a complete count
// exclude each other: either every wave sees the complete count (whenever it looks) and passes, or the flag goes up first and every
// wave -- the pollers, and the late arrivals through the value their own increment returns -- leaves with 0.  (Round 3 raised the
// flag with an atomicOr and let the wave go on when the count had just completed: pollers that then read count | flag left, and
// the grid ran the epoch without them until its chain counters timed out.)
// reg_extra > 0 (test hook): that many waves more are expected than the grid has -- the check must fail; < 0 (test hook): wave 0
// takes the give-up path at the very moment it sees the count complete -- the check must still pass everywhere.
__device__ __forceinline__ int register_wave(unsigned *ctl, int reg_extra)
{
    if (ldc_u32(ctl + kCtlStatus)) return 0;
    const unsigned want = gridDim.x + (reg_extra > 0 ? (unsigned)reg_extra : 0u);
    unsigned v = atomicAdd(ctl + kCtlReg, 1u) + 1u;
    int spins = 0;
    for (;;) {
        if (v & kRegFail) return 0;
        const bool late = reg_extra < 0 && blockIdx.x == 0 && v == want;
        if (v == want && !late) return 1;
        if (late || ++spins > kSpinReg) {
            unsigned cur = late ? v : ldc_u32(ctl + kCtlReg);
            for (;;) {
                if (cur == want) return 1;
                if (cur & kRegFail) return 0;
                const unsigned old = atomicCAS(ctl + kCtlReg, cur, cur | kRegFail);
                if (old == cur) { atomicOr(ctl + kCtlStatus, 1u); return 0; }
                cur = old;
            }
        }
        __builtin_amdgcn_s_sleep(8);
        v = ldc_u32(ctl + kCtlReg);
    }
}

// value of lane (base + j * SPW + g) for every lane of half-group g, j and base compile-time / uniform: v_readlane (scalar result,
// no LDS traffic, no address register per set, reads a lane whatever EXEC says) + one select per extra slot of the set
template <int SPW>
__device__ __forceinline__ int set_bcast_i(int v, int first, int g)
{
    int out = __builtin_amdgcn_readlane(v, first);
#pragma unroll
    for (int x = 1; x < SPW; ++x) {
        const int t = __builtin_amdgcn_readlane(v, first + x);
        out = g == x ? t : out;
    }
    return out;
}
template <int SPW>
__device__ __forceinline__ double set_bcast_d(double v, int first, int g)
{
    const int lo = set_bcast_i<SPW>(__double2loint(v), first, g), hi = set_bcast_i<SPW>(__double2hiint(v), first, g);
    return __hiloint2double(hi, lo);
}

// kmf_train.pyx:92-95: the k products of one rating, added strictly left to right to `sum` by ONE lane, from its row of the LDS
// product tile.  The chain of k dependent f64 adds is the longest serial piece of a step; the reads of the next eight products
// are issued before the current eight are added, so the chain waits for the adder only (round 3's loop waited for LDS in every
// iteration: ~17 clocks per add; tools/res_stamps.py).
__device__ __forceinline__ double ordered_sum(double sum, const double *row, int k)
{
    int f = 0;
    if (k >= 8) {
        d2 a = *reinterpret_cast<const d2 *>(row), b = *reinterpret_cast<const d2 *>(row + 2);
        d2 c = *reinterpret_cast<const d2 *>(row + 4), d = *reinterpret_cast<const d2 *>(row + 6);
        for (; f + 16 <= k; f += 8) {
            const d2 na = *reinterpret_cast<const d2 *>(row + f + 8), nb = *reinterpret_cast<const d2 *>(row + f + 10);
            const d2 nc = *reinterpret_cast<const d2 *>(row + f + 12), nd = *reinterpret_cast<const d2 *>(row + f + 14);
            sum += a.x; sum += a.y; sum += b.x; sum += b.y; sum += c.x; sum += c.y; sum += d.x; sum += d.y;
            a = na; b = nb; c = nc; d = nd;
        }
        sum += a.x; sum += a.y; sum += b.x; sum += b.y; sum += c.x; sum += c.y; sum += d.x; sum += d.y;
        f += 8;
    }
    for (; f < k; ++f) sum += row[f];
    return sum;
}

__device__ __forceinline__ double readlane64(double x, int l)          // l wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// ---- tolerance mode (DOT): the k-term dot as north_star words it -- per-lane partial products (FMA), reduced over the KP lanes of a
// slot with the association of the level kernels' wave butterfly (mfx_sgd_kernels.h group_sum / k1_body<DOT = 1>), and FMA in the row
// updates: a plan gives the same bits whichever kernel serves it; NOT the reference's left-to-right sum (kmf_train.pyx:75-97).
// Tolerance mode: the KP per-lane partials of one rating, from its row of the LDS tile, added as the BALANCED pairwise tree in index
// order -- ((p0 + p1) + (p2 + p3)) + ... -- by ONE lane.  Floating-point addition is commutative, so this tree gives the very bits of
// the wave butterfly of the level kernels' tolerance mode (group_sum: quad swaps, half-row / row mirrors, cross-row swaps -- every
// stage adds a lane's value and its partner's), at a fraction of its cost here: the butterflies of a step's register sets run one
// after the other behind the uniform skips (0.76 us per step, tools/res_stamps.py), this tree sums all slots of the wave at once.
template <int KP>
__device__ __forceinline__ double tree_sum(const double *row)
{
    double c[KP / 8];
#pragma unroll
    for (int i = 0; i < KP / 8; ++i) {
        const d2 a = *reinterpret_cast<const d2 *>(row + 8 * i), b = *reinterpret_cast<const d2 *>(row + 8 * i + 2);
        const d2 e = *reinterpret_cast<const d2 *>(row + 8 * i + 4), f = *reinterpret_cast<const d2 *>(row + 8 * i + 6);
        c[i] = ((a.x + a.y) + (b.x + b.y)) + ((e.x + e.y) + (f.x + f.y));
    }
    if constexpr (KP == 16) return c[0] + c[1];
    else if constexpr (KP == 32) return (c[0] + c[1]) + (c[2] + c[3]);
    else return ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
}

// One step of a wave: the slots in `mask` apply their head rating.  FULL: every slot (no branches around
// memory operations: one basic block of loads, one of stores).  Lane s < B owns slot s's scalar state.
// (Round 4 also split this step into a load half and an apply half for a pipelined epoch that took turns over two groups of
//  register sets -- measured slower on every order and dropped: profiles/experiments/r04_k1_res_pipe.hip.txt; the split itself cost
//  1.5 % through register allocation, so the step is one function again.)
template <int KP, int NR, int MODE, bool FULL, bool DOT, typename TS>
__device__ __forceinline__ void res_step(const ResParams &P, unsigned long long mask, bool ready, const URec &rec, double tbv,
                                         double &rbv, double &se, d2 (&vv)[NR], double *prod, __amdgpu_buffer_rsrc_t rT,
                                         int lane, int g, int q, double &tbn RS_PARAM)
{
    constexpr int SPW = 64 / KP, B = NR * SPW;
    constexpr int stride = 2 * KP + 2;                       // rows 16-B aligned, stride / 2 odd (k1_stride at k = 2 KP)
    const int k = P.k;
    const bool fl = 2 * q < k;                               // this lane holds a feature pair
    bool rdy[NR];
    d2 uu[NR];
    const unsigned qoff = (unsigned)q * RowIO<TS>::lane_bytes, rowb = (unsigned)(k * (int)sizeof(TS));
    constexpr unsigned SETM = (1u << SPW) - 1u;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int src = j * SPW + g;
        rdy[j] = fl && (FULL || ((mask >> src) & 1ull));
        uu[j] = d2{0.0, 0.0};
        // a set none of whose slots moves is skipped as a whole -- a UNIFORM branch (the ballot is a scalar): no broadcast, no
        // address, no memory instruction for it.  (The cross-lane reads are v_readlane: they do not depend on EXEC.)
        if (FULL || ((unsigned)(mask >> (j * SPW)) & SETM)) {
            const unsigned o = (unsigned)set_bcast_i<SPW>(rec.trav, j * SPW, g) * rowb + qoff;
            if (rdy[j]) uu[j] = RowIO<TS>::ld(rT, o);
        }
    }
    wait_vm0();
    RS(2);
    if constexpr (DOT) {
        // per-lane partials (features ascending, FMA) of every moving slot into the first KP entries of its tile row (a lane that holds
        // no feature pair of the row writes its 0), then ONE lane per slot adds them as the balanced pairwise tree (tree_sum)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (!(FULL || ((unsigned)(mask >> (j * SPW)) & SETM))) continue;                 // uniform
            const double a = __builtin_fma(uu[j].y, vv[j].y, __builtin_fma(uu[j].x, vv[j].x, 0.0));
            if (FULL || ((mask >> (j * SPW + g)) & 1ull)) prod[(j * SPW + g) * stride + q] = a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        RS(7);
    } else {
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (rdy[j]) *reinterpret_cast<d2 *>(prod + (j * SPW + g) * stride + 2 * q) = d2{uu[j].x * vv[j].x, uu[j].y * vv[j].y};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        RS(7);                                               // 7: products to the LDS tile
    }
    double grad = 0.0;
    if (FULL ? (lane < B) : ready) {
        // kmf_train.pyx:92: 0 + item bias + user bias, then the products left to right
        double sum = P.res_item ? (0.0 + rbv + tbv) : (0.0 + tbv + rbv);
        if constexpr (DOT) sum += tree_sum<KP>(prod + lane * stride);
        else sum = ordered_sum(sum, prod + lane * stride, k);
        double err;
        if constexpr (MODE == 0) {
            err = rec.r - sum;                               // kmf_train.pyx:251-256
            grad = err;
        } else {
            const double sig = 1.0 / (1.0 + exp(-sum));      // kmf_train.pyx:159-165
            const double p = 1.0 + sig * 4.0;
            err = rec.r - p;
            grad = err * sig * (1.0 - sig) * 4.0;
        }
        se += err * err;
        rbv = rbv + P.lr * (grad - P.Kb * rbv);              // :259-260 (both gates on in this path)
        tbn = tbv + P.lr * (grad - P.Kb * tbv);
    }
    __builtin_amdgcn_wave_barrier();
    RS(8);                                                   // 8: ordered sums, errors, bias updates
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        if (!(FULL || ((unsigned)(mask >> (j * SPW)) & SETM))) continue;                     // uniform
        const double gr = set_bcast_d<SPW>(grad, j * SPW, g);
        const unsigned o = (unsigned)set_bcast_i<SPW>(rec.trav, j * SPW, g) * rowb + qoff;   // formed again rather than kept: a register per slot pair less
        if (rdy[j]) {
            const d2 mf = uu[j], cf = vv[j];
            d2 un, vn;
            if constexpr (DOT) {
                un.x = __builtin_fma(P.lr, __builtin_fma(gr, cf.x, -(P.Ktrav * mf.x)), mf.x);
                un.y = __builtin_fma(P.lr, __builtin_fma(gr, cf.y, -(P.Ktrav * mf.y)), mf.y);
                vn.x = __builtin_fma(P.lr, __builtin_fma(gr, mf.x, -(P.Kres * cf.x)), cf.x);
                vn.y = __builtin_fma(P.lr, __builtin_fma(gr, mf.y, -(P.Kres * cf.y)), cf.y);
            } else {
                un.x = mf.x + P.lr * (gr * cf.x - P.Ktrav * mf.x);   // :267-270, both rows from the OLD values
                un.y = mf.y + P.lr * (gr * cf.y - P.Ktrav * mf.y);
                vn.x = cf.x + P.lr * (gr * mf.x - P.Kres * cf.x);
                vn.y = cf.y + P.lr * (gr * mf.y - P.Kres * cf.y);
            }
            RowIO<TS>::st(un, rT, o);
            vv[j] = RowIO<TS>::round(vn);
        }
    }
}

// Diagnostic builds only (-DMFX_BURST_STATS): what the burst walk covers and how it ends (tools/burst_stats.py)
#ifdef MFX_BURST_STATS
#ifdef MFX_BURST_STATS_WAVE0          // the counters of wave 0 alone: the wave that holds entity 0 (the hottest item of the synthetic shapes)
#define BST(i, n) do { if (lane == 0 && blockIdx.x == 0) atomicAdd(&g_burst_stats[i], (unsigned long long)(n)); } while (0)
#else
#define BST(i, n) do { if (lane == 0) atomicAdd(&g_burst_stats[i], (unsigned long long)(n)); } while (0)
#endif
#else
#define BST(i, n) do { } while (0)
#endif

// BURST (static binding, skewed data): a resident entity with a long run ahead -- the hot item of a power-law catalogue:
// its ratings come from different, mostly idle travelling entities -- walks up to eight of its ratings in ONE step.  The
// three memory round trips of a step (granule polls, row loads, store acknowledge) are paid once for the eight; the
// arithmetic stays one rating after the other on the resident row in registers (same operations, same order: same bits).
// Lane i < 8 holds rating i's record, granule and new travelling bias; slot s's own lanes hold the rows.
// Returns the number of ratings applied (0: nothing was ready after all -- the caller takes the ordinary step).
template <int KP, int NR, int MODE>
__device__ __forceinline__ int res_burst(const ResParams &P, int s, int &pos, int end, const URec *rp, URec &rec, URec &nxt, THdr &h,
                                         double &rbv, double &se, d2 (&vv)[NR], double *prod, __amdgpu_buffer_rsrc_t rT,
                                         __amdgpu_buffer_rsrc_t rH, int lane, int g, int q)
{
    constexpr int SPW = 64 / KP, D = 8, stride = 2 * KP + 2;
    const int k = P.k;
    const unsigned qoff = (unsigned)q * 16u, rowb = (unsigned)(k * 8);
    int pos_s = __builtin_amdgcn_readlane(pos, s);
    const int end_s = __builtin_amdgcn_readlane(end, s);
    const unsigned long long rpb = (unsigned long long)(uintptr_t)rp;
    const URec *rp_s = (const URec *)(uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(rpb >> 32), s) << 32) |
                                                 (unsigned)__builtin_amdgcn_readlane((int)rpb, s));
    const int js = s / SPW, gs = s % SPW;
    const bool ing = (2 * q < k) && g == gs;                 // the lanes that hold slot s's rows
    double *ubuf = prod + NR * SPW * stride;                 // [D][2 KP]: the travelling rows of a window, behind the product tile
    d2 cur = d2{0.0, 0.0};                                    // slot s's resident row: out of its register set, back at the end
#pragma unroll
    for (int j = 0; j < NR; ++j)
        if (j == js) cur = vv[j];
    int nd = end_s - pos_s < D ? end_s - pos_s : D;
    URec br{0, -2, 0.0};
    THdr bh; bh.bias = 0.0; bh.cnt = -1; bh.pad = 0;
    if (lane < nd) {
        br = ldrec(rp_s + pos_s + lane);
        bh = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)br.trav * 16u));
    }
    int total = 0;
    for (int w = 0; w < 8 && nd > 0; ++w) {                   // (bounded: the wave's other slots get their turn again)
        const unsigned long long okm = __ballot(lane < nd && bh.cnt == br.seq);
        const int depth = __builtin_ctzll(~okm);               // leading ready ratings of the window
        if (w == 0) BST(0, 1);
        if (depth == 0) { BST(w == 0 ? 1 : 5, 1); break; }
        BST(2, 1); BST(3, depth); if (depth < nd) BST(4, 1); if (depth == 8) BST(6, 1);
        // the travelling rows of the window: one block of loads, parked in LDS so that the walk below is ONE loop body with a
        // run-time rating index (registers cannot be indexed at run time)
        {
            d2 uu[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                uu[i] = d2{0.0, 0.0};
                if (i < depth) {                              // uniform
                    const unsigned o = (unsigned)__builtin_amdgcn_readlane(br.trav, i) * rowb + qoff;
                    if (ing) uu[i] = __builtin_bit_cast(d2, ld16c(rT, o));
                }
            }
            // the NEXT window's records travel under this window's walk
            const int pos_n = pos_s + depth, nd_n = end_s - pos_n < D ? end_s - pos_n : D;
            URec brn{0, -2, 0.0};
            if (lane < nd_n) brn = ldrec(rp_s + pos_n + lane);
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (i < depth && ing) *reinterpret_cast<d2 *>(ubuf + i * 2 * KP + 2 * q) = uu[i];
            double tbn = 0.0;
            for (int i = 0; i < depth; ++i) {
                d2 ui = d2{0.0, 0.0};
                if (ing) {
                    ui = *reinterpret_cast<const d2 *>(ubuf + i * 2 * KP + 2 * q);
                    *reinterpret_cast<d2 *>(prod + s * stride + 2 * q) = d2{ui.x * cur.x, ui.y * cur.y};
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const double tbv = readlane64(bh.bias, i), rr = readlane64(br.r, i);
                double grad = 0.0;
                if (lane == s) {
                    double sum = P.res_item ? (0.0 + rbv + tbv) : (0.0 + tbv + rbv);      // kmf_train.pyx:92
                    // (round 5: the step's ordered_sum -- the next eight products' reads issued before the current eight are added -- was
                    //  never brought into the burst walk: its loop waited for LDS in every iteration, 0.65 us of a rating's 0.83; the counters
                    //  of the hottest item's wave showed FULL windows, i.e. a chain bound by exactly this arithmetic)
                    sum = ordered_sum(sum, prod + s * stride, k);
                    double err;
                    if constexpr (MODE == 0) {
                        err = rr - sum;
                        grad = err;
                    } else {
                        const double sig = 1.0 / (1.0 + exp(-sum));
                        const double pp = 1.0 + sig * 4.0;
                        err = rr - pp;
                        grad = err * sig * (1.0 - sig) * 4.0;
                    }
                    se += err * err;
                    rbv = rbv + P.lr * (grad - P.Kb * rbv);
                }
                __builtin_amdgcn_wave_barrier();
                const double gr = readlane64(grad, s);
                if (lane == i) tbn = tbv + P.lr * (gr - P.Kb * tbv);
                const unsigned o = (unsigned)__builtin_amdgcn_readlane(br.trav, i) * rowb + qoff;
                if (ing) {
                    d2 un, vn;
                    un.x = ui.x + P.lr * (gr * cur.x - P.Ktrav * ui.x);
                    un.y = ui.y + P.lr * (gr * cur.y - P.Ktrav * ui.y);
                    vn.x = cur.x + P.lr * (gr * ui.x - P.Kres * cur.x);
                    vn.y = cur.y + P.lr * (gr * ui.y - P.Kres * cur.y);
                    st16c(__builtin_bit_cast(v4u, un), rT, o);
                    cur = vn;
                }
            }
            // the next window's granule polls go out BEFORE the acknowledge wait: loads return in issue order behind the
            // stores, so one round trip serves both (a rating whose entity is also in THIS window polls stale and ends the burst)
            THdr bhn; bhn.bias = 0.0; bhn.cnt = -1; bhn.pad = 0;
            if (lane < nd_n) bhn = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)brn.trav * 16u));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every row store acknowledged, then the granules
            if (lane < depth) {
                THdr nh; nh.bias = tbn; nh.cnt = br.seq + 1; nh.pad = 0;
                st16c(__builtin_bit_cast(v4u, nh), rH, (unsigned)br.trav * 16u);
            }
            total += depth;
            pos_s = pos_n;
            br = brn; bh = bhn; nd = nd_n;
        }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j)
        if (j == js && ing) vv[j] = cur;
    if (total == 0) return 0;
    if (lane == s) {                                          // slot s moves on
        pos += total;
        h.cnt = -1;
        if (pos < end) {
            rec = ldrec(rp + pos);
            h = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)rec.trav * 16u));
        }
        if (pos + 1 < end) nxt = ldrec(rp + pos + 1);
    }
    return total;
}

// KP lanes per row (16 bytes = one feature pair per lane: k <= 2 KP), NR register sets per lane, B = NR * 64/KP slots
template <int KP, int NR, int MODE, bool STATIC, bool BURST = false, bool DOT = false, typename TS = double>
__global__ __launch_bounds__(64, STATIC ? ((NR <= 9 && !BURST) ? 3 : 2) : 4) void k1_res(ResParams P)
{
    static_assert(!(BURST && DOT), "the burst walk keeps the ordered sum");
    static_assert(!(BURST || DOT) || std::is_same<TS, double>::value, "narrow rows: ordered sum, no bursts");
    constexpr int SPW = 64 / KP, B = NR * SPW;
    extern __shared__ __attribute__((aligned(16))) double prod[];       // [B][stride]
    const int lane = threadIdx.x, g = lane / KP, q = lane % KP;
    const int k = P.k;
    const bool fl = 2 * q < k;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(P.Tm, 0, 0x7fffffff, 0x27000);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(P.hdr, 0, 0x7fffffff, 0x27000);
    constexpr unsigned long long ALL = (B == 64) ? ~0ull : ((1ull << B) - 1ull);
    if (STATIC) {
        // ---- registration: nothing is modified unless every wave of the grid is running ----------------
        int ok = 1;
        if (lane == 0) ok = register_wave(P.ctl, P.reg_extra);
        if (!__builtin_amdgcn_readfirstlane(ok)) return;
        if (blockIdx.x == 0 && lane == 0) atomicAdd(P.ctl + kCtlDone, 1u);       // this launch of the batch is being applied
    }
    for (int pass = 0;; ++pass) {
        int64_t run = -1;
        if (STATIC) {
            if (pass) break;
            run = (int64_t)lane * gridDim.x + blockIdx.x;
        } else {
            unsigned b = 0;
            if (lane == 0) b = ldc_u32(P.ctl + kCtlStatus) ? 0x7fffffffu : atomicAdd(P.ctl, 1u);   // a timed-out wave stops everyone
            b = __builtin_amdgcn_readfirstlane(b);
            if ((int64_t)b * B >= P.n_runs) break;
            run = (int64_t)b * B + lane;
        }
        int pos = 0, end = 0;                               // relative to the run's first record
        int64_t base = 0;
        int my_ent = 0;
        bool have = false;
        if (lane < B && run < P.n_runs) {
            my_ent = STATIC ? (int)run : P.run_ent[run];
            base = P.lo[my_ent];
            end = P.hi[my_ent] - (int)base;
            have = true;
        }
        const URec *rp = P.recs + base;
        URec rec{0, 0, 0.0}, nxt{0, 0, 0.0};
        // records are read once per epoch: non-temporal
        if (pos < end) rec = ldrec(rp + pos);
        if (pos + 1 < end) nxt = ldrec(rp + pos + 1);
        double rbv = have ? P.rb[my_ent] : 0.0;
        d2 vv[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int ent = __shfl(my_ent, j * SPW + g, 64);
            const int hv = __shfl((int)have, j * SPW + g, 64);
            vv[j] = d2{0.0, 0.0};
            if (hv && fl) vv[j] = RowIO<TS>::ldp(P.Rm, (size_t)ent * k + 2 * q);
        }
        double se = 0.0;
        int spins = 0;
        bool bail = false;
        THdr h; h.bias = 0.0; h.cnt = -1; h.pad = 0;
        if (pos < end) h = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)rec.trav * 16u));
        RS_DECL;
        while (__any(pos < end)) {
            const bool act = pos < end;
            const bool ready = act && h.cnt == rec.seq;
            const unsigned long long mask = __ballot(ready);
            RS(0);
            if (!mask) {
                if (++spins > kSpinRun || ((spins & 1023) == 0 && ldc_u32(P.ctl + kCtlStatus))) {
                    if (lane == 0) atomicOr(P.ctl + kCtlStatus, 3u);
                    bail = true;
                    break;
                }
                BST(9, 1);
                __builtin_amdgcn_s_sleep(1);
                if (act) h = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)rec.trav * 16u));
                RS_SPIN();
                RS(1);
                continue;
            }
            if constexpr (BURST) {
                // at most two slots can move and one of them has a run ahead: walk up to eight of ITS ratings in this step
                if (__popcll(mask) <= 2) {
                    const unsigned long long cm = __ballot(ready && end - pos >= P.burst_run);
                    if (cm) {
                        const int s = __builtin_amdgcn_readfirstlane(__builtin_ctzll(cm));
                        if (res_burst<KP, NR, MODE>(P, s, pos, end, rp, rec, nxt, h, rbv, se, vv, prod, rT, rH, lane, g, q) > 0) {
                            spins = 0;
                            continue;
                        }
                    }
                }
            }
            BST(7, 1); BST(8, __popcll(mask));
            double tbn = 0.0;
            if (!STATIC && mask == ALL) res_step<KP, NR, MODE, true, DOT, TS>(P, mask, ready, rec, h.bias, rbv, se, vv, prod, rT, lane, g, q, tbn RS_ARG);
            else res_step<KP, NR, MODE, false, DOT, TS>(P, mask, ready, rec, h.bias, rbv, se, vv, prod, rT, lane, g, q, tbn RS_ARG);
            RS(3);                                                        // 2 + 3: loads, products, sums, updates, stores issued (res_step stamps 2)
            // publish: every row store of this wave has been acknowledged, then bias + chain counter in one granule
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RS(5);
            const bool more = ready ? (pos + 1 < end) : act;
            const int ntrav = ready ? nxt.trav : rec.trav;
            if (ready) {
                THdr nh; nh.bias = tbn; nh.cnt = rec.seq + 1; nh.pad = 0;
                st16c(__builtin_bit_cast(v4u, nh), rH, (unsigned)rec.trav * 16u);
                ++pos;
                rec = nxt;
                if (pos + 1 < end) nxt = ldrec(rp + pos + 1);
            }
            h.cnt = -1;
            if (more) h = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)ntrav * 16u));
            spins = 0;
            RS_STEP(__popcll(mask));
            RS(6);
        }
        RS_FLUSH();
        if (bail) break;
        // the runs are over: resident rows and biases go back (nobody else touches them this epoch)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int ent = __shfl(my_ent, j * SPW + g, 64);
            const int hv = __shfl((int)have, j * SPW + g, 64);
            if (hv && fl) RowIO<TS>::stp(P.Rm, (size_t)ent * k + 2 * q, vv[j]);
        }
        if (have) { P.rb[my_ent] = rbv; P.se_run[run] = se; }
    }
}

__global__ void k_hdr_reset(THdr *hdr, const double *tb, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { THdr h; h.bias = tb[i]; h.cnt = 0; h.pad = 0; hdr[i] = h; }
}
__global__ void k_hdr_finish(const THdr *hdr, double *tb, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tb[i] = hdr[i].bias;
}

int env_i(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
// MFX_TEST_REG_FAIL=n together with MFX_TESTING=1 (tests only): the static launches wait for one wave more than their grid has,
// i.e. their co-residency check fails the way it does on a GPU shared with another process -- exercises the fall-backs
// (MFX_TEST_REG_FAIL=n: the n-th static launch of the process and every later one fail -- n > 1 reaches the fall-back in the middle
//  of a run, with state to hand over)
int test_reg_extra()
{
    static int launches = 0;
    if (env_i("MFX_TESTING", 0) != 1) return 0;
    const int n = env_i("MFX_TEST_REG_FAIL", 0);
    if (n == -1) return -1;                                  // the give-up path fires on a COMPLETE count: every launch must still pass
    if (n <= 0) { launches = 0; return 0; }
    return ++launches >= n ? 1 : 0;
}

struct Sel { void *fn; int kp, nr, id; };       // nr: register sets per lane; id: what select() was asked for
// product tile [B][2 KP + 2]; the burst variant (id 13) parks eight travelling rows behind it
size_t res_lds_bytes(int kp, int id) { return ((size_t)id * (64 / kp) * (2 * kp + 2) + (id == 13 ? 8 * 2 * kp : 0)) * 8; }

// ticket: 4 register sets (4 waves per SIMD); static: 9 sets at 3 waves per SIMD (18 slots x 3 072 waves at k = 64: measured 3 %
// faster than 26 x 2 048 and 5 % faster than 32 x 2 048 on cfg 2) or 16 sets at 2 waves per SIMD (the largest side that fits)
template <int MODE, int NR, bool STATIC, bool BURST = false, bool DOT = false, typename TS = double>
Sel pick(int k)
{
    const int id = NR;
    if (k <= 32) return {(void *)k1_res<16, NR, MODE, STATIC, BURST, DOT, TS>, 16, NR, id};
    if (k <= 64) return {(void *)k1_res<32, NR, MODE, STATIC, BURST, DOT, TS>, 32, NR, id};
    return {(void *)k1_res<64, NR, MODE, STATIC, BURST, DOT, TS>, 64, NR, id};
}
// id: 4 = ticket; 9 / 16 = static; 13 = static with bursts (skewed data: 2 waves per SIMD, 26 slots x 2 048 waves at k = 64).
// variant 0: f64 rows, ordered dot (exact); 1: the butterfly-dot / FMA tolerance mode (mfx_sgd_plan_set_dot); 2: f32 row storage
// (mfx_sgd_plan_set_storage(32)) -- both on every binding but the burst variant, whose walk keeps f64 rows and the ordered sum
// (res_variants_ok: such a plan serves its tolerance modes on the level kernels)
template <int MODE, bool DOT, typename TS>
Sel select_m(int k, int id)
{
    return id == 4 ? pick<MODE, 4, false, false, DOT, TS>(k) : (id == 9 ? pick<MODE, 9, true, false, DOT, TS>(k) : pick<MODE, 16, true, false, DOT, TS>(k));
}
Sel select(int k, int mode, int id, int variant = 0)
{
    if (id == 13) return mode == 0 ? pick<0, 13, true, true>(k) : pick<1, 13, true, true>(k);
    if (variant == 1) return mode == 0 ? select_m<0, true, double>(k, id) : select_m<1, true, double>(k, id);
    if (variant == 2) return mode == 0 ? select_m<0, false, float>(k, id) : select_m<1, false, float>(k, id);
    return mode == 0 ? select_m<0, false, double>(k, id) : select_m<1, false, double>(k, id);
}

}  // namespace

struct ResPlan {
    int kind = 0;               // 1 = ticket, 2 = static
    bool res_item = false;
    int k = 0, nu = 0, ni = 0, n_res = 0, n_trav = 0, n_runs = 0;
    int kp = 0, nr = 0, B = 0, waves = 0;
    int64_t nnz = 0, se_slots = 0;
    URec *d_recs = nullptr;
    int32_t *d_lo = nullptr, *d_hi = nullptr;
    int32_t *d_run_ent = nullptr;
    THdr *d_hdr = nullptr;
    unsigned *d_ctl = nullptr;
    double *d_se_run = nullptr;
    size_t dev_bytes = 0;
    double est_ms = 0.0, build_ms = 0.0;
};

void res_destroy(ResPlan *rp)
{
    if (!rp) return;
    void *ptrs[] = {rp->d_recs, rp->d_lo, rp->d_hi, rp->d_run_ent, rp->d_hdr, rp->d_ctl, rp->d_se_run};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete rp;
}

namespace {
template <typename T>
int ralloc(ResPlan *rp, T **ptr, size_t n)
{
    *ptr = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)ptr, bytes);
    if (e != hipSuccess) return fail(MFX_E_NOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    rp->dev_bytes += bytes;
    return MFX_OK;
}
}  // namespace

// Cost model (measured, DESIGN.md section 4 "K1 resident"): a chain of `levels` dependent steps of ~4.1 us
// against a stream of nnz ratings at the coherent path's rate; the level schedule costs ~3.7 us per level
// plus 20 line operations per rating at ~67 G lines/s.
// (levels too small for a launch of their own run in the tail kernel: 2.47 us each)
static double est_level_ms(int64_t nnz, int64_t levels) { return levels * (nnz / levels >= 128 ? 3.7e-3 : 2.47e-3) + nnz * 0.30e-6; }
// (grouped = user-major input on the static binding: the travelling user's hops follow each other closely -- measured 3.8 us per
//  level at cfg 2 against 4.15 on the ticket binding, tools/k1_pipe.hip's reference runs, profiles/r03_k1_pipe_cfg2_user_major.log)
static double est_res_ms(int64_t nnz, int64_t levels, bool stat, bool grouped = false)
{
    const double chain = levels * (stat ? (grouped ? 3.8e-3 : 4.6e-3) : 4.15e-3), stream = nnz * (stat ? 0.30e-6 : 0.265e-6);
    return chain > stream ? chain : stream;
}

// Decide whether a resident-side epoch applies to this plan and prepare it.  *out stays nullptr when it does
// not.  want: MFX_EPOCH_AUTO (cost model; the environment variable MFX_K1_RES = levels / ticket / static overrides
// it: tuning hook), MFX_EPOCH_LEVELS (never), MFX_EPOCH_TICKET / MFX_EPOCH_STATIC (that binding, or MFX_E_ARG
// when the plan is not eligible for it).
int res_prepare(ResPlan **out, const int32_t *idx, const Rec *d_recs, int64_t nnz, int nu, int ni, int k, int64_t n_levels, int want,
                bool headroom)
{
    *out = nullptr;
    bool forced = want == MFX_EPOCH_TICKET || want == MFX_EPOCH_STATIC;
    if (want == MFX_EPOCH_AUTO) {
        if (const char *e = getenv("MFX_K1_RES")) {
            if (e[0] == 'l' || e[0] == '0') want = MFX_EPOCH_LEVELS;
            else if (e[0] == 't') want = MFX_EPOCH_TICKET;
            else if (e[0] == 's') want = MFX_EPOCH_STATIC;
        }
    }
    if (want == MFX_EPOCH_LEVELS) return MFX_OK;
    const bool shape_ok = nnz > 0 && nnz < ((int64_t)1 << 31) - 1 && n_levels > 0 && k % 2 == 0 && k > 16 && k <= 128;
    if (!shape_ok) {
        if (forced) return fail(MFX_E_ARG, "mfx_sgd_plan_create3: the resident-side epoch needs an even k in 18..128 and 0 < nnz < 2^31 (k=%d, nnz=%lld)", k, (long long)nnz);
        return MFX_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipDeviceProp_t prop;
    int dev = 0;
    MFX_HIP(hipGetDevice(&dev));
    MFX_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    // ---- ticket: is the input grouped by user (every user one contiguous run)? ---------------------------
    bool grouped = true;
    int64_t n_runs_u = 0;
    std::vector<int32_t> run_ent;                            // the users of the runs, in input order (= the ticket order)
    {
        std::vector<unsigned char> seen((size_t)nu, 0);
        int32_t prev = -1;
        for (int64_t t = 0; t < nnz; ++t) {
            const int32_t u = idx[2 * t];
            if (u != prev) {
                if (seen[(size_t)u]) { grouped = false; break; }
                seen[(size_t)u] = 1;
                run_ent.push_back(u);
                ++n_runs_u;
                prev = u;
            }
        }
    }
    const bool ticket_ok = grouped && (int64_t)ni * k * 8 < ((int64_t)1 << 31) && nnz / (n_runs_u ? n_runs_u : 1) >= 4;
    // ---- static: the smaller side in registers, the whole grid resident -----------------------------------
    const bool res_item = ni <= nu;
    const int n_res = res_item ? ni : nu, n_trav = res_item ? nu : ni;
    const Sel st = select(k, 0, 4);
    const int Bt = st.nr * (64 / st.kp);
    const size_t lds_t = (size_t)Bt * (2 * st.kp + 2) * 8;
    Sel ss = select(k, 0, 9);
    int Bs = 0, waves_s = 0;
    for (int nr : {9, 16}) {                                 // the smallest register image that holds the side
        ss = select(k, 0, nr);
        Bs = ss.nr * (64 / ss.kp);
        int occ = 0;
        MFX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ss.fn, 64, res_lds_bytes(ss.kp, ss.id)));
        waves_s = occ * cus;
        if (headroom) {                                      // leave an eighth of the slots to whatever runs beside the launch
            const int64_t need = ceil_div((int64_t)n_res, (int64_t)Bs), spare = (int64_t)waves_s * 7 / 8;
            if (need <= spare) waves_s = (int)spare;
        }
        if ((int64_t)waves_s * Bs >= n_res) break;
    }
    const bool static_ok = waves_s > 0 && (int64_t)waves_s * Bs >= n_res && (int64_t)n_trav * k * 8 < ((int64_t)1 << 31);
    const double e_level = est_level_ms(nnz, n_levels);
    const double e_ticket = ticket_ok ? est_res_ms(nnz, n_levels, false) : 1e30;
    double e_static = static_ok ? est_res_ms(nnz, n_levels, true, grouped && res_item) : 1e30;
    // ---- skewed data: the chain is a hot entity's own run.  When that entity is on the resident side, the burst variant walks it
    //      eight ratings per step (measured ~0.7 us per rating against 2.1 without, 2.47 per level in the tail kernel); a hot
    //      TRAVELLING entity hops from wave to wave at the full step.  MFX_K1_BURST = 0 / 1: off / forced (tuning, tests).
    bool burst = false;
    {
        const int benv = env_i("MFX_K1_BURST", -1);
        const double avg = (double)nnz / (double)(nu < ni ? nu : ni);
        if (benv != 0 && (benv == 1 || (double)n_levels > 16.0 * avg)) {
            const Sel sb = select(k, 0, 13);
            const int Bb = sb.nr * (64 / sb.kp);
            int occ = 0;
            MFX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sb.fn, 64, res_lds_bytes(sb.kp, sb.id)));
            int wb = occ * cus;
            if (headroom && ceil_div((int64_t)n_res, (int64_t)Bb) <= (int64_t)wb * 7 / 8) wb = wb * 7 / 8;
            if (wb > 0 && (int64_t)wb * Bb >= n_res && (int64_t)n_trav * k * 8 < ((int64_t)1 << 31)) {
                int mu = 0, mi = 0;
                int rc = res_max_degrees(d_recs, nnz, nu, ni, &mu, &mi);
                if (rc) return rc;
                const int max_res = res_item ? mi : mu, max_trav = res_item ? mu : mi;
                const double stream = nnz * 0.30e-6;
                // (round 5, MovieLens-10M shape: 1.16 us per rating of the hottest resident entity, 6.2 us per hand-off of the heaviest travelling one)
                double e = max_res * 1.15e-3 > max_trav * 6.2e-3 ? max_res * 1.15e-3 : max_trav * 6.2e-3;
                e = e > stream ? e : stream;
                if (benv == 1 || (2 * (int64_t)max_res >= n_levels && e < e_static)) {
                    burst = true;
                    e_static = e;
                    ss = sb; Bs = Bb; waves_s = wb;
                }
            }
        }
    }
    const bool static_ok2 = static_ok || burst;
    int kind = 0;
    if (want == MFX_EPOCH_TICKET) kind = ticket_ok ? 1 : 0;
    else if (want == MFX_EPOCH_STATIC) kind = static_ok2 ? 2 : 0;
    else if (e_ticket <= e_static && e_ticket < 0.9 * e_level) kind = 1;
    else if (e_static < e_ticket && e_static < 0.9 * e_level) kind = 2;
    if (!kind) {
        if (forced && want == MFX_EPOCH_TICKET)
            return fail(MFX_E_ARG, "mfx_sgd_plan_create3: MFX_EPOCH_TICKET needs ratings grouped by user (every user one contiguous run of >= 4 ratings on average) and item rows below 2 GB");
        if (forced)
            return fail(MFX_E_ARG, "mfx_sgd_plan_create3: MFX_EPOCH_STATIC needs the smaller side in registers (min(users, items) = %d > %lld rows at k = %d) and the other side's rows below 2 GB",
                        n_res, (long long)waves_s * Bs, k);
        return MFX_OK;
    }

    ResPlan *rp = new (std::nothrow) ResPlan();
    if (!rp) return fail(MFX_E_NOMEM, "mfx_sgd_plan_create: out of host memory");
    rp->kind = kind;
    rp->k = k; rp->nu = nu; rp->ni = ni; rp->nnz = nnz;
    rp->est_ms = kind == 1 ? e_ticket : e_static;
    auto bail = [&](int rc) { res_destroy(rp); return rc; };
    {
        const Sel sel = kind == 1 ? st : ss;
        rp->kp = sel.kp; rp->nr = sel.id;
        if (kind == 1) {
            rp->res_item = false; rp->n_res = nu; rp->n_trav = ni; rp->B = Bt;
            rp->n_runs = (int)run_ent.size();
            // enough waves to fill the chip at the kernel's residency; the grid size is free in this mode
            int occ_t = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_t, st.fn, 64, lds_t) != hipSuccess) occ_t = 0;
            rp->waves = (occ_t > 0 ? occ_t : 8) * cus;
            const int64_t tickets = ceil_div((int64_t)rp->n_runs, (int64_t)Bt);
            if (rp->waves > tickets) rp->waves = (int)tickets;
            rp->se_slots = (int64_t)rp->n_runs;
        } else {
            rp->res_item = res_item; rp->n_res = n_res; rp->n_trav = n_trav; rp->B = Bs; rp->waves = waves_s;
            rp->n_runs = n_res;                              // every entity is a run (possibly empty); run = entity id
            rp->se_slots = (int64_t)rp->waves * rp->B;
        }
        if (ralloc(rp, &rp->d_recs, (size_t)nnz) || ralloc(rp, &rp->d_lo, (size_t)rp->n_res) || ralloc(rp, &rp->d_hi, (size_t)rp->n_res) ||
            (kind == 1 && ralloc(rp, &rp->d_run_ent, run_ent.size())) ||
            ralloc(rp, &rp->d_hdr, (size_t)rp->n_trav) || ralloc(rp, &rp->d_ctl, (size_t)kCtlWords) || ralloc(rp, &rp->d_se_run, (size_t)rp->se_slots))
            return bail(last_code());
        if ((kind == 1 && hipMemcpy(rp->d_run_ent, run_ent.data(), run_ent.size() * 4, hipMemcpyHostToDevice) != hipSuccess) ||
            hipMemset(rp->d_ctl, 0, kCtlWords * sizeof(unsigned)) != hipSuccess ||
            hipMemset(rp->d_se_run, 0, (size_t)(rp->se_slots ? rp->se_slots : 1) * sizeof(double)) != hipSuccess)
            return bail(fail(MFX_E_HIP, "mfx_sgd_plan_create: upload of the resident-side tables failed: %s", hipGetErrorString(hipGetLastError())));
        // the records, grouped by resident entity and ranked in the travelling chains: built in HBM from the level-major records
        int rc = res_build_device(d_recs, nnz, rp->n_res, rp->n_trav, rp->res_item, rp->d_recs, rp->d_lo, rp->d_hi);
        if (rc) return bail(rc);
    }
    rp->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = rp;
    return MFX_OK;
}

int res_kind(const ResPlan *rp) { return rp ? rp->kind : 0; }
size_t res_dev_bytes(const ResPlan *rp) { return rp ? rp->dev_bytes : 0; }
int64_t res_se_slots(const ResPlan *rp) { return rp->se_slots; }
const double *res_se_partials(const ResPlan *rp) { return rp->d_se_run; }
bool res_variants_ok(const ResPlan *rp) { return rp && rp->nr != 13; }
int res_describe(const ResPlan *rp, int mode, char *buf, int buflen, int variant)
{
    static const char *const mode_name[] = {"linear", "logistic"};
    snprintf(buf, (size_t)buflen, "k1_res<%d lanes/row,%d slots,%s%s,%s: %s rows resident,%d waves>", rp->kp, rp->B,
             mode_name[mode & 1], variant == 1 ? ",butterfly-dot+fma" : (variant == 2 ? ",f32 rows" : ""), rp->kind == 1 ? "ticket" : (rp->nr == 13 ? "static+bursts" : "static"), rp->res_item ? "item" : "user", rp->waves);
    return MFX_OK;
}

// clear the sticky status word: once per batch of epochs
int res_begin_batch(ResPlan *rp)
{
    MFX_HIP(hipMemsetAsync(rp->d_ctl + kCtlStatus, 0, 2 * sizeof(unsigned), stream()));      // status + applied-launch count
    return MFX_OK;
}

// enqueue ONE epoch on the library stream (capturable): counters, granules, the persistent launch, biases back
int res_enqueue_epoch(ResPlan *rp, int mode, double *U_nk, double *V_nk, double *ib, double *ub, double lr, double Ku,
                      double Ki, double Kb, int variant)
{
    ResParams P;
    memset(&P, 0, sizeof(P));
    P.recs = rp->d_recs; P.lo = rp->d_lo; P.hi = rp->d_hi; P.run_ent = rp->d_run_ent;
    P.Tm = rp->res_item ? V_nk : U_nk;
    P.Rm = rp->res_item ? U_nk : V_nk;
    double *tb = rp->res_item ? ub : ib;
    P.rb = rp->res_item ? ib : ub;
    P.hdr = rp->d_hdr; P.ctl = rp->d_ctl; P.se_run = rp->d_se_run;
    P.n_runs = rp->n_runs; P.k = rp->k; P.res_item = rp->res_item ? 1 : 0;
    P.lr = lr; P.Kres = rp->res_item ? Ki : Ku; P.Ktrav = rp->res_item ? Ku : Ki; P.Kb = Kb;
    P.reg_extra = test_reg_extra();
    // A burst pays two more round trips before its rows than the ordinary step does (the window's records, then their granules): worth
    // it for an entity whose own run is the chain, a loss for everybody else -- above all for a heavy TRAVELLING entity, whose every
    // rating is a hand-off between two waves (MovieLens-10M shape: the epoch was the heaviest user's 7 369 hand-offs x 8.7 us with
    // every item bursting, tools/critical_path.c; 64.7 -> 45.5 ms with the rule below, profiles/r05_k1_burst_run.txt).  A slot
    // bursts only while it has more ratings ahead than an average resident entity has in the whole epoch.  (Measured and dropped
    // beside it: windows per call, bursts only when one slot is ready, polling a not-yet-ready window again inside the burst, a
    // first window made of the slot's head rating out of the caller's registers -- none better, same file.)
    P.burst_run = (int)(rp->nnz / (rp->n_res > 0 ? rp->n_res : 1));
    if (P.burst_run < 3) P.burst_run = 3;
    if (const char *e = getenv("MFX_K1_BURST_RUN")) P.burst_run = atoi(e);                       // tuning, tests (3 = round 4's rule)
    const Sel s = select(rp->k, mode, rp->nr, variant);
    MFX_HIP(hipMemsetAsync(rp->d_ctl, 0, sizeof(unsigned), stream()));                          // ticket
    MFX_HIP(hipMemsetAsync(rp->d_ctl + kCtlReg, 0, sizeof(unsigned), stream()));                // registration
    hipLaunchKernelGGL(k_hdr_reset, dim3(ceil_div(rp->n_trav, 256)), dim3(256), 0, stream(), rp->d_hdr, (const double *)tb, rp->n_trav);
    void *args[] = {&P};
    MFX_HIP(hipLaunchKernel(s.fn, dim3(rp->waves), dim3(64), args, res_lds_bytes(rp->kp, rp->nr), stream()));   // (the dot mode parks its per-lane partials in the same tile: tree_sum)
    hipLaunchKernelGGL(k_hdr_finish, dim3(ceil_div(rp->n_trav, 256)), dim3(256), 0, stream(), (const THdr *)rp->d_hdr, tb, rp->n_trav);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

// after a stream sync: 0 = fine, 1 = a static grid was not resident -- *applied launches of the batch ran in full before it,
// that one and every later one modified NOTHING (the word is sticky) --, 3 = a spin timed out mid-epoch (state undefined)
int res_status(ResPlan *rp, unsigned *status, unsigned *applied)
{
    unsigned w[2] = {0, 0};
    MFX_HIP(hipMemcpy(w, rp->d_ctl + kCtlStatus, sizeof w, hipMemcpyDeviceToHost));
    *status = w[0];
    if (applied) *applied = w[1];
    return MFX_OK;
}

// ================================================================================================
// Funk per-feature pass (estimator_loop_without_bias / _with_bias / _with_bias_dev, gd_estimator.pyx:691-779, :489-582,
// :588-685) in the same resident-side form, scalar edition.  ONE LANE owns one item: U[f,item] lives in a register, the
// lane walks the item's ratings in input order.  The user side travels, and because its payload is ONE scalar it fits the
// hand-off granule itself -- {V[f,user], chain counter} in 16 bytes: the poll RETURNS the value (no separate load), the
// publish is ONE 16-byte sc1 store that carries value and counter together (nothing to drain first).  A hop costs one
// granule round trip (~1.9 us under load) instead of a level hand-off, and the pass runs on every CU of the chip.
// Static owners: needs every lane resident (registration barrier as above; ni <= waves x 64).
// ================================================================================================
struct __attribute__((aligned(16))) HURec { int32_t pos; int32_t item; double r; };   // grouped position (cache word, reply mailbox), item, rating
struct Res2Params {
    const URec *recs;           // grouped by item (stable); trav = user
    const int32_t *lo, *hi;     // [ni]
    double *cache;              // residual cache in the same (grouped) order
    const double *const *dyn;   // -> {uf, vf, trail} of the feature (K2Dyn layout: two pointers + a double)
    const double *ib, *ub;      // nullable (without_bias)
    THdr *hdr;                  // [nu] {V[f,user], chain counter}                (granule form)
    double *box;                // [nnz + 2] one 8-byte mailbox per grouped record  (mailbox form; recs[].seq = NEXT position)
    const int32_t *first_pos;   // [nu] grouped position of the user's first rating, -1: none
    const int32_t *order;       // hot-wave form: [ni] items, the n_hot hot ones first (a WAVE each), then the others (a lane each); else null
    int n_hot;
    // hot USERS (their V[f,user] resident in a wave of their own; recs[].seq <= -2 marks such a rating: mailbox -2 - seq of ubox)
    const struct HURec *urec;   // [hu_off[n_hu]] a hot user's ratings in input order
    const int32_t *hu_ids, *hu_off;
    double *ubox;               // item -> hot user: U[f,item] as it stands in front of the rating
    double *se_user;            // [n_hu]
    int n_hu;
    unsigned *ctl;
    double *se_item;
    int ni, nu, upd_u, upd_i, window, hot_deg, reg_extra;
    double lr, K, mu;
};

namespace {

__device__ __forceinline__ double clamp15r(double x)        // gd_estimator.pyx:26-35
{
    if (x > 5.0) x = 5.0;
    if (x < 1.0) x = 1.0;
    return x;
}
// the same function of every input, NaN included (a NaN passes both comparisons of clamp15r unchanged), in three dependent
// instructions instead of four compare-select pairs: the walks of the hot-wave roles are chains of dependent operations
__device__ __forceinline__ double clamp15w(double x)
{
    const double r = __builtin_fmax(__builtin_fmin(x, 5.0), 1.0);
    return (x != x) ? x : r;
}
__device__ __forceinline__ double funk_est(double uf, double vf, double cache, double base, double trail, bool trailing)
{
    double sum = (cache > 0) ? cache : base;                  // gd_estimator.pyx:38-73
    sum += uf * vf;
    sum = clamp15r(sum);
    if (trailing) {
        sum += trail;
        sum = clamp15r(sum);
    }
    return sum;
}

struct Dyn2 { double *uf; double *vf; double trail; };

__global__ __launch_bounds__(64) void k2_res(Res2Params P)
{
    const int lane = threadIdx.x;
    // ---- registration: nothing is modified unless every wave of the grid is running ----------------
    int ok = 1;
    if (lane == 0) ok = register_wave(P.ctl, P.reg_extra);
    if (!__builtin_amdgcn_readfirstlane(ok)) return;
    if (blockIdx.x == 0 && lane == 0) atomicAdd(P.ctl + kCtlDone, 1u);           // this pass of the batch is being applied
    const Dyn2 *dyn = reinterpret_cast<const Dyn2 *>(P.dyn);
    double *const uf = dyn->uf;
    const double trail = dyn->trail;
    const int item = blockIdx.x * 64 + lane;
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(P.hdr, 0, 0x7fffffff, 0x27000);
    int pos = 0, end = 0;
    double x = 0.0, se = 0.0, bi = 0.0;
    if (item < P.ni) { pos = P.lo[item]; end = P.hi[item]; x = uf[item]; if (P.ib) bi = P.ib[item]; }
    URec rec{0, 0, 0.0};
    double c = 0.0;
    if (pos < end) { rec = P.recs[pos]; c = P.cache[pos]; }
    int spins = 0;
    // one rating: the estimate with the residual cache, the error, both values from the OLD ones (gd_estimator.pyx:750-761)
    auto apply = [&](const URec &q, double cq, const THdr &g) {
        const double mf = x, cf = g.bias;                     // U[f,item], V[f,user]
        double base;
        if (P.ib) base = P.mu + bi + P.ub[q.trav];            // gd_estimator.pyx:553-554
        else base = P.mu + 0.0 + 0.0;
        const double p = funk_est(mf, cf, cq, base, trail, true);
        const double err = q.r - p;
        se += err * err;
        const double un = mf + P.lr * (err * cf - P.K * mf), vn = cf + P.lr * (err * mf - P.K * cf);
        THdr n; n.bias = P.upd_u ? vn : cf; n.cnt = q.seq + 1; n.pad = 0;
        st16c(__builtin_bit_cast(v4u, n), rH, (unsigned)q.trav * 16u);
        if (P.upd_i) x = un;
    };
    // A HOT item (eight times the average degree or more: the chains that bound a pass on skewed data; its raters are mostly
    // idle) on a streak -- its last two heads were ready at the first poll -- looks three ratings further: their records and
    // granules travel together with the head's poll, and every one whose counter already matches is applied in the same
    // iteration (a counter that matches can only be advanced by this lane, so an early answer stays valid).  Ordinary items
    // never do: the pass is bound by the request rate of the polls, and on uniform data the look-ahead costs more requests than
    // it saves round trips (measured: 1.26 -> 1.77 ms per pass at cfg 2 with the window on every lane).
    const bool window = P.window != 0 && end - pos >= P.hot_deg;
    int streak = 0;
    while (__any(pos < end)) {
        const bool act = pos < end;
        const bool wide = window && act && streak >= 2 && pos + 1 < end;
        THdr g0; g0.bias = 0.0; g0.cnt = -1; g0.pad = 0;
        if (act) g0 = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)rec.trav * 16u));
        URec q1{0, -2, 0.0}, q2{0, -2, 0.0}, q3{0, -2, 0.0};
        double c1 = 0.0, c2 = 0.0, c3 = 0.0;
        THdr g1 = g0, g2 = g0, g3 = g0;
        g1.cnt = g2.cnt = g3.cnt = -1;
        if (wide) {
            q1 = P.recs[pos + 1]; c1 = P.cache[pos + 1];
            if (pos + 2 < end) { q2 = P.recs[pos + 2]; c2 = P.cache[pos + 2]; }
            if (pos + 3 < end) { q3 = P.recs[pos + 3]; c3 = P.cache[pos + 3]; }
            g1 = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)q1.trav * 16u));
            if (pos + 2 < end) g2 = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)q2.trav * 16u));
            if (pos + 3 < end) g3 = __builtin_bit_cast(THdr, ld16c(rH, (unsigned)q3.trav * 16u));
        }
        const bool ready = act && g0.cnt == rec.seq;
        if (ready) {
            apply(rec, c, g0);
            ++pos;
            int n = 1;
            if (wide) {
                if (g1.cnt == q1.seq) {
                    apply(q1, c1, g1); ++pos; ++n;
                    if (g2.cnt == q2.seq) {
                        apply(q2, c2, g2); ++pos; ++n;
                        if (g3.cnt == q3.seq) { apply(q3, c3, g3); ++pos; ++n; }
                    }
                }
            }
            streak = (!wide || n >= 2) ? streak + 1 : 0;
            if (pos < end) { rec = P.recs[pos]; c = P.cache[pos]; }
        } else {
            streak = 0;
        }
        if (!__any(ready)) {
            if (++spins > kSpinRun || ((spins & 1023) == 0 && ldc_u32(P.ctl + kCtlStatus))) {
                if (lane == 0) atomicOr(P.ctl + kCtlStatus, 3u);
                break;
            }
        } else spins = 0;
    }
    if (item < P.ni) { uf[item] = x; P.se_item[item] = se; }
}

// ---- mailbox form ---------------------------------------------------------------------------------------------------------
// The travelling value does not go back to a per-user granule that the next rating's lane has to find and poll: it is sent
// straight to the user's NEXT rating.  Every grouped record has an 8-byte mailbox; a lane polls the mailbox of its own head
// record -- and the mailboxes of one item are contiguous in the order the lane walks them, so ONE 16-byte request shows the head
// and the rating after it.  A mailbox holds kBoxEmpty (a quiet-NaN payload no arithmetic produces from other inputs: the launch
// refuses inputs that contain it) until the previous rating of that user -- or, for a user's first rating, the seeding kernel --
// stores V[f,user] there with ONE 8-byte sc1 store: the data is the flag, nothing to drain first.  A user's last rating writes
// V[f,user] itself.  Against the granule form: a ready rating costs half a poll request instead of one or more, a waiting lane
// polls ITS OWN line (no crowd on a hot user's granule), and the look-ahead is free.
constexpr unsigned long long kBoxEmpty = 0x7ff85eedb0c50001ull;

__global__ void k2_box_fill(unsigned long long *box, int64_t n)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) box[s] = kBoxEmpty;
}
__global__ void k2_box_seed(unsigned long long *box, const int32_t *__restrict__ first_pos, const double *const *dyn, int nu, int ni, unsigned *ctl)
{
    const Dyn2 *d = reinterpret_cast<const Dyn2 *>(dyn);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool clash = false;
    if (i < nu) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, d->vf[i]);
        clash = b == kBoxEmpty;
        const int32_t fp = first_pos[i];
        if (fp >= 0) box[fp] = b;
    }
    if (i < ni) clash = clash || __builtin_bit_cast(unsigned long long, d->uf[i]) == kBoxEmpty;
    if (clash) atomicOr(ctl + kCtlStatus, 1u);              // the pass is not applied (same path as a failed registration)
}


// ---- hot-wave role (skewed data) ------------------------------------------------------------------------------------------
// On power-law data a pass is as long as the hottest item's chain (MovieLens-10M shape: 39 K ratings, Zipf-0.8 over cfg 2's shape:
// 254 K), and a LANE walks its chain at one poll round trip per one or two ratings (~0.5 us per rating).  Here a hot item owns a
// whole WAVE: lane j holds the record, cache word and mailbox of the item's rating pos + j -- contiguous, one coalesced request
// each --, the leading run of filled mailboxes is walked by ONE uniform loop (values by v_readlane, U[f,item] in a register of
// every lane: the reference's operations in the reference's order, gd_estimator.pyx:750-761), lane i keeps rating i's new
// V[f,user] and the 64 results leave in one store instruction.  The next window's records, cache words and mailboxes are requested
// BEFORE the walk (a filled mailbox stays filled; an empty one ends the run and is polled again), so a streak costs no round trip
// at all: ~0.1 us per rating.  Same bits as the lane form.
__device__ __forceinline__ unsigned long long ld_box8(__amdgpu_buffer_rsrc_t rB, unsigned off)
{
    const v2u t = __builtin_amdgcn_raw_buffer_load_b64(rB, off, 0, 16);
    return ((unsigned long long)t[1] << 32) | t[0];
}
__device__ __forceinline__ void k2_hot_role(const Res2Params &P, int item, __amdgpu_buffer_rsrc_t rB, __amdgpu_buffer_rsrc_t rU, double *uf, double *vf,
                                            double trail)
{
    const int lane = threadIdx.x;
    int pos = P.lo[item];
    const int end = P.hi[item];
    double x = uf[item], se = 0.0;
    const double bi = P.ib ? P.ib[item] : 0.0;
    URec rec{0, -1, 0.0};
    double c = 0.0, ubv = 0.0;
    unsigned long long m = kBoxEmpty;
    if (pos + lane < end) {
        rec = P.recs[pos + lane];
        c = P.cache[pos + lane];
        m = ld_box8(rB, (unsigned)(pos + lane) * 8u);
        if (P.ib) ubv = P.ub[rec.trav];
    }
    int spins = 0, sent_pos = -1;
    while (pos < end) {
        const unsigned long long filled = __ballot(pos + lane < end && m != kBoxEmpty);
        const unsigned long long rem = __ballot(pos + lane < end && rec.seq <= -2);
        if (rem & 1ull) {
            // the head is a HOT USER's rating: U[f,item] goes to that user's wave and comes back updated (two hand-offs on this
            // chain instead of one on the user's -- the user's chain is the longer one)
            if (sent_pos != pos) {
                if (lane == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), rU, (unsigned)(-2 - rec.seq) * 8u, 0, 16);
                sent_pos = pos;
            }
            if (filled & 1ull) {
                x = readlane64(__builtin_bit_cast(double, m), 0);
                pos += 1;
                spins = 0;
                rec = URec{0, -1, 0.0}; c = 0.0; ubv = 0.0; m = kBoxEmpty;
                if (pos + lane < end) {
                    rec = P.recs[pos + lane];
                    c = P.cache[pos + lane];
                    m = ld_box8(rB, (unsigned)(pos + lane) * 8u);
                    if (P.ib) ubv = P.ub[rec.trav];
                }
                continue;
            }
            if (++spins > kSpinRun || ((spins & 1023) == 0 && ldc_u32(P.ctl + kCtlStatus))) {
                if (lane == 0) atomicOr(P.ctl + kCtlStatus, 3u);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
            if (pos + lane < end) m = ld_box8(rB, (unsigned)(pos + lane) * 8u);
            continue;
        }
        const unsigned long long okm = filled & ~rem;            // (a hot user's rating further on ends the run)
        const int depth = (okm == ~0ull) ? 64 : __builtin_ctzll(~okm);      // leading filled mailboxes of the window
        if (depth == 0) {
            if (++spins > kSpinRun || ((spins & 1023) == 0 && ldc_u32(P.ctl + kCtlStatus))) {
                if (lane == 0) atomicOr(P.ctl + kCtlStatus, 3u);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
            if (pos + lane < end) m = ld_box8(rB, (unsigned)(pos + lane) * 8u);
            continue;
        }
        spins = 0;
        // the next window travels under the walk
        const int np = pos + depth;
        URec recn{0, -1, 0.0};
        double cn = 0.0, ubn = 0.0;
        unsigned long long mn = kBoxEmpty;
        if (np + lane < end) {
            recn = P.recs[np + lane];
            cn = P.cache[np + lane];
            mn = ld_box8(rB, (unsigned)(np + lane) * 8u);
        }
        // what an estimate starts from (gd_estimator.pyx:38-73, :553-554) does not depend on the chain: formed by every lane for its
        // own rating before the walk
        const double base = P.ib ? (P.mu + bi + ubv) : (P.mu + 0.0 + 0.0);
        const double start = (c > 0) ? c : base;
        const double mv = __builtin_bit_cast(double, m);
        double out = 0.0;
        auto walk = [&](auto both) {
            constexpr bool BOTH = decltype(both)::value;         // both update gates on: the common case without selects
            auto step = [&](int i) {
                const double cf = readlane64(mv, i), s0 = readlane64(start, i), rr = readlane64(rec.r, i);
                const double mf = x;
                double sum = s0;                                 // funk_est(), trailing features included
                sum += mf * cf;
                sum = clamp15w(sum);
                sum += trail;
                sum = clamp15w(sum);
                const double err = rr - sum;
                se += err * err;
                const double un = mf + P.lr * (err * cf - P.K * mf), vn = cf + P.lr * (err * mf - P.K * cf);
                if (lane == i) out = (BOTH || P.upd_u) ? vn : cf;
                if (BOTH || P.upd_i) x = un;
            };
            int i = 0;
            for (; i + 1 < depth; i += 2) { step(i); step(i + 1); }      // (pairs: the second rating's values are fetched under the first's chain)
            if (i < depth) step(i);
        };
        if (P.upd_u && P.upd_i) walk(std::true_type{});
        else walk(std::false_type{});
        if (lane < depth) {
            if (rec.seq >= 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, out), rB, (unsigned)rec.seq * 8u, 0, 16);
            else vf[rec.trav] = out;                            // the user's last rating of the pass
        }
        if (P.ib && np + lane < end) ubn = P.ub[recn.trav];
        rec = recn; c = cn; m = mn; ubv = ubn;
        pos = np;
    }
    if (lane == 0) { uf[item] = x; P.se_item[item] = se; }
}


// ---- hot-USER role ----------------------------------------------------------------------------------------------------------
// A user's value hops from item to item -- one hand-off between waves (~1.5 us) per rating, so the hottest user's chain bounds a pass
// (MovieLens-10M shape: 7 369 ratings = 11 ms).  A hot user owns a wave instead: V[f,user] stays in a register of every lane, and
// for ITS ratings the item value travels -- the item's owner stores U[f,item] into the rating's mailbox in `ubox` when its chain
// gets there and waits for the updated value in the rating's own mailbox (two hand-offs on the item's chain, none on the user's).
// The wave walks the user's ratings exactly as k2_hot_role walks an item's: 64 mailboxes per poll, the leading filled run in one
// uniform loop, the next window requested under the walk.  Same operations in the same order (gd_estimator.pyx:750-761).
__device__ __forceinline__ void k2_hot_user_role(const Res2Params &P, int h, __amdgpu_buffer_rsrc_t rB, __amdgpu_buffer_rsrc_t rU, double *vf, double trail)
{
    const int lane = threadIdx.x;
    const int u = P.hu_ids[h];
    int pos = P.hu_off[h];
    const int end = P.hu_off[h + 1];
    double v = vf[u], se = 0.0;
    const double bu = P.ib ? P.ub[u] : 0.0;
    HURec rec{-1, 0, 0.0};
    double c = 0.0, ibv = 0.0;
    unsigned long long m = kBoxEmpty;
    if (pos + lane < end) {
        rec = P.urec[pos + lane];
        m = ld_box8(rU, (unsigned)(pos + lane) * 8u);
        c = P.cache[rec.pos];
        if (P.ib) ibv = P.ib[rec.item];
    }
    int spins = 0;
    while (pos < end) {
        const unsigned long long okm = __ballot(pos + lane < end && m != kBoxEmpty);
        const int depth = (okm == ~0ull) ? 64 : __builtin_ctzll(~okm);
        if (depth == 0) {
            if (++spins > kSpinRun || ((spins & 1023) == 0 && ldc_u32(P.ctl + kCtlStatus))) {
                if (lane == 0) atomicOr(P.ctl + kCtlStatus, 3u);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
            if (pos + lane < end) m = ld_box8(rU, (unsigned)(pos + lane) * 8u);
            continue;
        }
        spins = 0;
        const int np = pos + depth;
        HURec recn{-1, 0, 0.0};
        double cn = 0.0, ibn = 0.0;
        unsigned long long mn = kBoxEmpty;
        if (np + lane < end) {
            recn = P.urec[np + lane];
            mn = ld_box8(rU, (unsigned)(np + lane) * 8u);
        }
        const double base = P.ib ? (P.mu + ibv + bu) : (P.mu + 0.0 + 0.0);      // gd_estimator.pyx:553-554: item bias, then user bias
        const double start = (c > 0) ? c : base;
        const double mv = __builtin_bit_cast(double, m);
        double out = 0.0;
        auto walk = [&](auto both) {
            constexpr bool BOTH = decltype(both)::value;
            auto step = [&](int i) {
                const double mf = readlane64(mv, i), s0 = readlane64(start, i), rr = readlane64(rec.r, i);
                const double cf = v;
                double sum = s0;
                sum += mf * cf;
                sum = clamp15w(sum);
                sum += trail;
                sum = clamp15w(sum);
                const double err = rr - sum;
                se += err * err;
                const double un = mf + P.lr * (err * cf - P.K * mf), vn = cf + P.lr * (err * mf - P.K * cf);
                if (lane == i) out = (BOTH || P.upd_i) ? un : mf;
                if (BOTH || P.upd_u) v = vn;
            };
            int i = 0;
            for (; i + 1 < depth; i += 2) { step(i); step(i + 1); }
            if (i < depth) step(i);
        };
        if (P.upd_u && P.upd_i) walk(std::true_type{});
        else walk(std::false_type{});
        if (lane < depth) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, out), rB, (unsigned)rec.pos * 8u, 0, 16);
        if (np + lane < end) {
            cn = P.cache[recn.pos];
            if (P.ib) ibn = P.ib[recn.item];
        }
        rec = recn; c = cn; m = mn; ibv = ibn;
        pos = np;
    }
    if (lane == 0) { vf[u] = v; P.se_user[h] = se; }
}

template <int W>   // mailboxes seen per poll: 2 (one request); 4, 8: hot items look further (two / four requests)
__global__ __launch_bounds__(64) void k2_res_box(Res2Params P)
{
    const int lane = threadIdx.x;
    int ok = 1;
    if (lane == 0) ok = register_wave(P.ctl, P.reg_extra);
    if (!__builtin_amdgcn_readfirstlane(ok)) return;
    if (blockIdx.x == 0 && lane == 0) atomicAdd(P.ctl + kCtlDone, 1u);
    const Dyn2 *dyn = reinterpret_cast<const Dyn2 *>(P.dyn);
    double *const uf = dyn->uf;
    double *const vf = dyn->vf;
    const double trail = dyn->trail;
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(P.box, 0, 0x7fffffff, 0x27000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(P.ubox ? P.ubox : P.box, 0, 0x7fffffff, 0x27000);
    int item = blockIdx.x * 64 + lane;
    if (P.order) {                                              // hot-wave form: the first n_hot waves own one hot item each,
        if ((int)blockIdx.x < P.n_hot) {                        // the next n_hu one hot user each, the rest 64 items each
            k2_hot_role(P, P.order[blockIdx.x], rB, rU, uf, vf, trail);
            return;
        }
        if ((int)blockIdx.x < P.n_hot + P.n_hu) {
            k2_hot_user_role(P, (int)blockIdx.x - P.n_hot, rB, rU, vf, trail);
            return;
        }
        const int64_t o = (int64_t)P.n_hot + ((int64_t)blockIdx.x - P.n_hot - P.n_hu) * 64 + lane;
        item = o < P.ni ? P.order[o] : P.ni;
    }
    int pos = 0, end = 0;
    double x = 0.0, se = 0.0, bi = 0.0;
    if (item < P.ni) { pos = P.lo[item]; end = P.hi[item]; x = uf[item]; if (P.ib) bi = P.ib[item]; }
    URec rec[W];
    double c[W];
#pragma unroll
    for (int i = 0; i < W; ++i) { rec[i] = URec{0, -1, 0.0}; c[i] = 0.0; }
    auto load_from = [&](int i0) {                              // records i0.. of the window (plain cached loads: sequential per lane)
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (i >= i0 && pos + i < end) { rec[i] = P.recs[pos + i]; c[i] = P.cache[pos + i]; }
    };
    load_from(0);
    int spins = 0, sent_pos = -1;
    // W = 4: only a HOT item (eight times the average degree or more: the chains that bound a pass on skewed data) pays the second
    // request; the others look at their head's pair like the W = 2 kernel
    const bool wide = W > 2 && end - pos >= P.hot_deg;
    auto apply = [&](const URec &q, double cq, double cf) {     // gd_estimator.pyx:750-761
        const double mf = x;
        double base;
        if (P.ib) base = P.mu + bi + P.ub[q.trav];
        else base = P.mu + 0.0 + 0.0;
        const double p = funk_est(mf, cf, cq, base, trail, true);
        const double err = q.r - p;
        se += err * err;
        const double un = mf + P.lr * (err * cf - P.K * mf), vn = cf + P.lr * (err * mf - P.K * cf);
        const double out = P.upd_u ? vn : cf;
        if (q.seq >= 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, out), rB, (unsigned)q.seq * 8u, 0, 16);
        else vf[q.trav] = out;                                  // the user's last rating of the pass
        if (P.upd_i) x = un;
    };
    while (__any(pos < end)) {
        const bool act = pos < end;
        // the aligned group of W mailboxes that holds the head
        unsigned long long m[W];
#pragma unroll
        for (int i = 0; i < W; ++i) m[i] = kBoxEmpty;
        const int g0 = wide ? (pos & ~(W - 1)) : (pos & ~1), sh = pos - g0;
        if (act) {
#pragma unroll
            for (int h = 0; h < W / 2; ++h) {
                if (h > 0 && !wide) break;
                const v4u t = ld16c(rB, (unsigned)(g0 + 2 * h) * 8u);
                m[2 * h] = ((unsigned long long)t[1] << 32) | t[0];
                m[2 * h + 1] = ((unsigned long long)t[3] << 32) | t[2];
            }
        }
        int n = 0;
        if (act) {
#pragma unroll
            for (int i = 0; i < W; ++i) {                       // window slot i is mailbox sh + i of the group
                unsigned long long b = kBoxEmpty;
#pragma unroll
                for (int w = 0; w < W; ++w) if (w == sh + i) b = m[w];
                if (n == i && pos + i < end) {
                    if (rec[i].seq <= -2) {
                        // a HOT USER's rating: U[f,item] goes to that user's wave once, the updated value comes back in this mailbox
                        if (sent_pos != pos + i) {
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), rU, (unsigned)(-2 - rec[i].seq) * 8u, 0, 16);
                            sent_pos = pos + i;
                        }
                        if (b != kBoxEmpty) { x = __builtin_bit_cast(double, b); ++n; }
                    } else if (b != kBoxEmpty) { apply(rec[i], c[i], __builtin_bit_cast(double, b)); ++n; }
                }
            }
        }
        const bool ready = n > 0;
        if (ready) {
            pos += n;
#pragma unroll
            for (int i = 0; i < W; ++i) {                       // shift the window by n
                URec r2 = URec{0, -1, 0.0}; double c2 = 0.0;
#pragma unroll
                for (int w = 0; w < W; ++w) if (w == i + n) { r2 = rec[w]; c2 = c[w]; }
                rec[i] = r2; c[i] = c2;
            }
            load_from(W - n);
        }
        if (!__any(ready)) {
            if (++spins > kSpinRun || ((spins & 1023) == 0 && ldc_u32(P.ctl + kCtlStatus))) {
                if (lane == 0) atomicOr(P.ctl + kCtlStatus, 3u);
                break;
            }
        } else spins = 0;
    }
    if (item < P.ni) { uf[item] = x; P.se_item[item] = se; }
}

__global__ void k2_hdr_reset(THdr *hdr, const double *const *dyn, int n)
{
    const double *vf = reinterpret_cast<const Dyn2 *>(dyn)->vf;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { THdr h; h.bias = vf[i]; h.cnt = 0; h.pad = 0; hdr[i] = h; }
}
// values back -- unless the pass never ran (registration failed: the granules are stale, vf is untouched and stays)
__global__ void k2_hdr_finish(const THdr *hdr, const double *const *dyn, int n, const unsigned *ctl)
{
    if (ctl[kCtlStatus] == 1u) return;
    double *vf = reinterpret_cast<const Dyn2 *>(dyn)->vf;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vf[i] = hdr[i].bias;
}
// cache-commit pass (gd_estimator.pyx:771-777) over the grouped records: embarrassingly parallel, one thread per record
__global__ __launch_bounds__(256) void k2_commit_res(Res2Params P, const int32_t *__restrict__ ent_of, int64_t nnz)
{
    const Dyn2 *dyn = reinterpret_cast<const Dyn2 *>(P.dyn);
    const double *uf = dyn->uf, *vf = dyn->vf;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * 256) {
        const URec rec = P.recs[s];
        const int item = ent_of[s];
        double base;
        if (P.ib) base = P.mu + P.ib[item] + P.ub[rec.trav];
        else base = P.mu + 0.0 + 0.0;
        P.cache[s] = funk_est(uf[item], vf[rec.trav], P.cache[s], base, 0.0, false);
    }
}
// hot users: degrees, then each hot user's chain is walked once (the records are linked through seq = next grouped position) into
// its own list; the ratings are marked (seq = -2 - list position) and the user is taken out of the seeding (first_pos = -1)
__global__ void k2_user_degrees(const URec *__restrict__ recs, int64_t nnz, int *deg)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x) atomicAdd(deg + recs[s].trav, 1);
}
__global__ void k2_build_hot_users(URec *recs, const int32_t *__restrict__ ent_of, int32_t *first_pos, const int32_t *__restrict__ hu_ids,
                                   const int32_t *__restrict__ hu_off, int n_hu, HURec *urec)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_hu) return;
    const int u = hu_ids[t];
    int p = first_pos[u], j = hu_off[t];
    const int jend = hu_off[t + 1];
    while (p >= 0 && j < jend) {
        const URec r = recs[p];
        HURec h; h.pos = p; h.item = ent_of[p]; h.r = r.r;
        urec[j] = h;
        recs[p].seq = -2 - j;
        p = r.seq;
        ++j;
    }
    first_pos[u] = -1;
}
// grouped cache -> the level-major cache of the other Funk kernels (a fallback in the middle of a run)
__global__ void k2_set_next(URec *recs, const int32_t *__restrict__ next, int64_t nnz)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x) recs[s].seq = next[s];
}
__global__ void k2_cache_scatter(const double *src, const int32_t *pos_of, int64_t nnz, double *dst)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < nnz; s += (int64_t)gridDim.x * blockDim.x) dst[pos_of[s]] = src[s];
}

}  // namespace

constexpr int kMaxHotUsers = 256;
struct Res2Plan {
    int nu = 0, ni = 0, waves = 0;
    int64_t nnz = 0;
    URec *d_recs = nullptr;
    int32_t *d_lo = nullptr, *d_hi = nullptr, *d_pos_of = nullptr, *d_ent_of = nullptr;
    double *d_cache = nullptr, *d_se = nullptr;
    THdr *d_hdr = nullptr;
    double *d_box = nullptr;            // mailbox form (box_w != 0): one mailbox per grouped record
    int32_t *d_first = nullptr;
    int32_t *d_order = nullptr;         // hot-wave form: the items, hot ones first
    int n_hot = 0, hot_deg = 0;
    HURec *d_urec = nullptr;            // hot users: their lists, ids, offsets, item -> user mailboxes
    int32_t *d_hu_ids = nullptr, *d_hu_off = nullptr;
    double *d_ubox = nullptr;
    int n_hu = 0, hu_deg = 0;
    int64_t n_urec = 0;
    int box_w = 0;
    unsigned *d_ctl = nullptr;
    size_t dev_bytes = 0;
};

void res2_destroy(Res2Plan *rp)
{
    if (!rp) return;
    void *ptrs[] = {rp->d_recs, rp->d_lo, rp->d_hi, rp->d_pos_of, rp->d_ent_of, rp->d_cache, rp->d_se, rp->d_hdr, rp->d_box, rp->d_first, rp->d_order, rp->d_urec,
                    rp->d_hu_ids, rp->d_hu_off, rp->d_ubox, rp->d_ctl};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete rp;
}

// *out stays nullptr when the pass does not apply (more items than resident lanes, tiny problems unless forced)
int res2_prepare(Res2Plan **out, const Rec *d_recs, int64_t nnz, int nu, int ni, bool force)
{
    *out = nullptr;
    if (nnz <= 0 || nnz >= ((int64_t)1 << 31) - 1 || (int64_t)nu * 16 >= ((int64_t)1 << 31)) return MFX_OK;
    if (!force && nnz < (1 << 16)) return MFX_OK;
    hipDeviceProp_t prop;
    int dev = 0, occ = 0;
    MFX_HIP(hipGetDevice(&dev));
    MFX_HIP(hipGetDeviceProperties(&prop, dev));
    // MFX_K2_BOX: 0 = granule form, 2 / 4 = mailbox form with that many mailboxes per poll
    int box_w = env_i("MFX_K2_BOX", -1);
    const bool box_auto = box_w != 0 && box_w != 2 && box_w != 4 && box_w != 8;
    if (box_auto) {          // default: the wide kernel only where some item is hot (it costs 3 % on uniform data)
        int mu_deg = 0, mi_deg = 0;
        int rc0 = res_max_degrees(d_recs, nnz, nu, ni, &mu_deg, &mi_deg);
        if (rc0) return rc0;
        box_w = (int64_t)mi_deg >= 8 * (nnz / (ni > 0 ? ni : 1)) && mi_deg >= 64 ? 8 : 2;
    }
    if (nnz >= ((int64_t)1 << 28) - 8) box_w = 0;           // (mailbox offsets are 32-bit byte offsets)
    const void *fn = box_w == 8 ? (const void *)k2_res_box<8> : box_w == 4 ? (const void *)k2_res_box<4> : box_w == 2 ? (const void *)k2_res_box<2> : (const void *)k2_res;
    MFX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64, 0));
    const int waves = ceil_div(ni, 64);
    const int wave_budget = occ > 0 ? occ * prop.multiProcessorCount * 7 / 8 : 0;
    if (occ <= 0 || waves > wave_budget) return MFX_OK;
    Res2Plan *rp = new (std::nothrow) Res2Plan();
    if (!rp) return fail(MFX_E_NOMEM, "mfx_sgd_run_funk: out of host memory");
    rp->nu = nu; rp->ni = ni; rp->nnz = nnz; rp->waves = waves;
    auto bail = [&](int rc) { res2_destroy(rp); return rc; };
    auto al = [&](void **q, size_t bytes) {
        if (hipMalloc(q, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return fail(MFX_E_NOMEM, "hipMalloc(%zu bytes)", bytes); }
        rp->dev_bytes += bytes;
        return (int)MFX_OK;
    };
    if (al((void **)&rp->d_recs, (size_t)nnz * sizeof(URec)) || al((void **)&rp->d_lo, (size_t)ni * 4) || al((void **)&rp->d_hi, (size_t)ni * 4) ||
        al((void **)&rp->d_pos_of, (size_t)nnz * 4) || al((void **)&rp->d_ent_of, (size_t)nnz * 4) || al((void **)&rp->d_cache, (size_t)nnz * 8) || al((void **)&rp->d_se, ((size_t)waves * 64 + kMaxHotUsers) * 8) ||
        al((void **)&rp->d_hdr, (size_t)nu * sizeof(THdr)) || al((void **)&rp->d_ctl, kCtlWords * sizeof(unsigned)))
        return bail(last_code());
    if (hipMemset(rp->d_ctl, 0, kCtlWords * sizeof(unsigned)) != hipSuccess || hipMemset(rp->d_se, 0, ((size_t)waves * 64 + kMaxHotUsers) * 8) != hipSuccess)
        return bail(fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError())));
    rp->box_w = box_w;
    int32_t *d_next = nullptr;
    if (box_w) {
        if (al((void **)&rp->d_box, (size_t)(nnz + 8) * 8) || al((void **)&rp->d_first, (size_t)nu * 4)) return bail(last_code());
        if (hipMalloc((void **)&d_next, (size_t)nnz * 4) != hipSuccess) { (void)hipGetLastError(); return bail(fail(MFX_E_NOMEM, "hipMalloc(%zu bytes)", (size_t)nnz * 4)); }
    }
    int rc = res_build_device(d_recs, nnz, ni, nu, true, rp->d_recs, rp->d_lo, rp->d_hi, rp->d_pos_of, rp->d_ent_of, d_next, rp->d_first);
    if (!rc && box_w) {
        hipLaunchKernelGGL(k2_set_next, dim3(2048), dim3(256), 0, stream(), rp->d_recs, (const int32_t *)d_next, nnz);
        if (hipStreamSynchronize(stream()) != hipSuccess) rc = fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError()));
    }
    if (d_next) (void)hipFree(d_next);
    if (rc) return bail(rc);
    // ---- hot-wave form (mailbox form, skewed data): the items whose chains would bound the pass get a wave each ----------------
    // MFX_K2_HOT: 0 = off, n > 0 = every item with n ratings or more (tests, tuning); default: items with at least an eighth of
    // the largest degree (and 1 024 ratings), when that degree is 4 096 or more and eight times the average
    if (box_w) {
        const int henv = env_i("MFX_K2_HOT", -1);
        std::vector<int32_t> lo((size_t)ni), hi((size_t)ni);
        if (hipMemcpy(lo.data(), rp->d_lo, (size_t)ni * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hi.data(), rp->d_hi, (size_t)ni * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return bail(fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError())));
        int thr = 0;
        if (henv > 0) thr = henv;
        else if (henv < 0) {
            int maxd = 0;
            for (int i = 0; i < ni; ++i) maxd = std::max(maxd, hi[(size_t)i] - lo[(size_t)i]);
            if (maxd >= 4096 && (int64_t)maxd >= 8 * (nnz / ni)) thr = std::max(1024, maxd / 8);
        }
        if (thr > 0) {
            std::vector<int32_t> hot;
            for (int i = 0; i < ni; ++i)
                if (hi[(size_t)i] - lo[(size_t)i] >= thr) hot.push_back(i);
            std::stable_sort(hot.begin(), hot.end(), [&](int32_t a, int32_t b) { return hi[(size_t)a] - lo[(size_t)a] > hi[(size_t)b] - lo[(size_t)b]; });
            size_t cap = 2048;                                // (a wave per hot item: the hottest first)
            while (cap > 0 && (int64_t)cap + ceil_div((int64_t)ni - (int64_t)std::min(cap, hot.size()), (int64_t)64) > wave_budget) cap /= 2;
            if (hot.size() > cap) hot.resize(cap);
            if (!hot.empty()) {
                std::vector<unsigned char> is_hot((size_t)ni, 0);
                for (int32_t h : hot) is_hot[(size_t)h] = 1;
                std::vector<int32_t> order(hot);
                for (int i = 0; i < ni; ++i)
                    if (!is_hot[(size_t)i]) order.push_back(i);
                if (al((void **)&rp->d_order, (size_t)ni * 4)) return bail(last_code());
                if (hipMemcpy(rp->d_order, order.data(), (size_t)ni * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return bail(fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError())));
                rp->n_hot = (int)hot.size();
                rp->hot_deg = thr;
                // with the long chains on waves of their own the lanes need no look-ahead, and an iteration of the narrow kernel is
                // shorter: a hot USER's value hops from lane to lane faster (MovieLens-10M shape: 13.9 -> 11.3 ms per pass)
                if (box_auto) rp->box_w = 2;
                rp->waves = rp->n_hot + ceil_div(ni - rp->n_hot, 64);
            }
        }
        // ---- hot users: a wave each, their V[f,user] resident; for their ratings the ITEM value travels (there and back) -------
        // MFX_K2_HOTU: 0 = off, n > 0 = every user with n ratings or more (tests, tuning; at most 256, the hottest first); default:
        // the h hottest users that a cost model picks (below) -- few enough that a hot ITEM's chain, which pays two hand-offs per
        // such rating, does not become the longest (MovieLens-10M shape, measured: 9 users 6.07 ms per pass, 31 users 6.49, 153
        // users 8.4; the model picks 8-9)
        const int uenv = env_i("MFX_K2_HOTU", -1);
        if (uenv != 0) {
            int *d_deg = nullptr;
            if (hipMalloc((void **)&d_deg, (size_t)nu * 4) != hipSuccess) { (void)hipGetLastError(); return bail(fail(MFX_E_NOMEM, "hipMalloc(%zu bytes)", (size_t)nu * 4)); }
            std::vector<int32_t> deg((size_t)nu);
            bool okc = hipMemsetAsync(d_deg, 0, (size_t)nu * 4, stream()) == hipSuccess;
            if (okc) {
                hipLaunchKernelGGL(k2_user_degrees, dim3(2048), dim3(256), 0, stream(), (const URec *)rp->d_recs, nnz, d_deg);
                okc = hipStreamSynchronize(stream()) == hipSuccess && hipMemcpy(deg.data(), d_deg, (size_t)nu * 4, hipMemcpyDeviceToHost) == hipSuccess;
            }
            (void)hipFree(d_deg);
            if (!okc) return bail(fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError())));
            // the candidates: the (at most) 257 largest degrees, descending (ties by ascending id) -- only looked for when some user
            // has a chain worth a wave (or the override asks)
            int max_udeg = 0;
            for (int u = 0; u < nu; ++u) max_udeg = std::max(max_udeg, deg[(size_t)u]);
            std::vector<int32_t> top;
            if (uenv > 0 || max_udeg >= 2048) {
                top.resize((size_t)nu);
                for (int u = 0; u < nu; ++u) top[(size_t)u] = u;
            }
            const size_t ntop = std::min(top.size(), (size_t)kMaxHotUsers + 1);
            auto by_deg = [&](int32_t a, int32_t b) { return deg[(size_t)a] != deg[(size_t)b] ? deg[(size_t)a] > deg[(size_t)b] : a < b; };
            std::partial_sort(top.begin(), top.begin() + (std::ptrdiff_t)ntop, top.end(), by_deg);
            top.resize(ntop);
            int uthr = 0;
            size_t n_take = 0;
            if (uenv > 0) {
                uthr = uenv;
                while (n_take < top.size() && n_take < (size_t)kMaxHotUsers && deg[(size_t)top[n_take]] >= uthr) ++n_take;
            } else if (!top.empty() && max_udeg >= 2048) {
                // cost model (measured constants, DESIGN.md K2 hot-wave roles): with the h hottest users on waves the pass is as long
                // as the longer of (a) the hottest REMAINING user's chain of hand-offs, 1.5 us each, and (b) the hottest item's
                // chain: 0.115 us per rating on a wave (0.5 on a lane) plus two hand-offs (3 us) for every rating by one of the
                // h users (their share of all ratings, assumed the same on that item).  Taken when it saves a tenth or more.
                int max_item = 0;
                for (int i = 0; i < ni; ++i) max_item = std::max(max_item, hi[(size_t)i] - lo[(size_t)i]);
                const double c_item = rp->n_hot ? 0.115 : 0.5;
                double best = 0.0, cost0 = 0.0, share = 0.0;
                size_t best_h = 0;
                for (size_t h = 0; h <= top.size() && h <= (size_t)kMaxHotUsers; ++h) {     // h users taken: top[h] is the hottest one left
                    const double user_chain = h < top.size() ? deg[(size_t)top[h]] * 1.5 : 0.0;      // (h = nu: every user on a wave)
                    const double item_chain = max_item * (c_item + 3.0 * share / (double)nnz);
                    const double cost = std::max(user_chain, item_chain);
                    if (h == 0) cost0 = best = cost;
                    else if (cost < best) { best = cost; best_h = h; }
                    if (h < top.size()) share += deg[(size_t)top[h]];
                }
                if (best > 0.9 * cost0) best_h = 0;
                n_take = best_h;
                if (n_take) uthr = deg[(size_t)top[n_take - 1]];
            }
            std::vector<int32_t> hu(top.begin(), top.begin() + (std::ptrdiff_t)n_take);
            size_t cap = kMaxHotUsers;
            while (cap > 0 && (int64_t)rp->waves + (int64_t)std::min(cap, hu.size()) > wave_budget) cap /= 2;
            if (hu.size() > cap) hu.resize(cap);
            if (!hu.empty()) {
                std::vector<int32_t> off(hu.size() + 1, 0);
                for (size_t t = 0; t < hu.size(); ++t) off[t + 1] = off[t] + deg[(size_t)hu[t]];
                const int64_t M = off.back();
                if (!rp->d_order) {                             // (the role dispatch goes through order[]: identity when no item is hot)
                    std::vector<int32_t> order((size_t)ni);
                    for (int i = 0; i < ni; ++i) order[(size_t)i] = i;
                    if (al((void **)&rp->d_order, (size_t)ni * 4)) return bail(last_code());
                    if (hipMemcpy(rp->d_order, order.data(), (size_t)ni * 4, hipMemcpyHostToDevice) != hipSuccess)
                        return bail(fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError())));
                }
                if (al((void **)&rp->d_urec, (size_t)M * sizeof(HURec)) || al((void **)&rp->d_hu_ids, hu.size() * 4) || al((void **)&rp->d_hu_off, off.size() * 4) ||
                    al((void **)&rp->d_ubox, (size_t)(M + 8) * 8))
                    return bail(last_code());
                if (hipMemcpy(rp->d_hu_ids, hu.data(), hu.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(rp->d_hu_off, off.data(), off.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return bail(fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError())));
                hipLaunchKernelGGL(k2_build_hot_users, dim3(ceil_div((int)hu.size(), 64)), dim3(64), 0, stream(), rp->d_recs, (const int32_t *)rp->d_ent_of,
                                   rp->d_first, (const int32_t *)rp->d_hu_ids, (const int32_t *)rp->d_hu_off, (int)hu.size(), rp->d_urec);
                if (hipStreamSynchronize(stream()) != hipSuccess) return bail(fail(MFX_E_HIP, "mfx_sgd_run_funk: %s", hipGetErrorString(hipGetLastError())));
                rp->n_hu = (int)hu.size();
                rp->hu_deg = uthr;
                if (box_auto) rp->box_w = 2;                    // (as with hot items: the lanes need no look-ahead)
                rp->n_urec = M;
                rp->waves += rp->n_hu;
            }
        }
    }
    *out = rp;
    return MFX_OK;
}

size_t res2_dev_bytes(const Res2Plan *rp) { return rp ? rp->dev_bytes : 0; }
int res2_box_width(const Res2Plan *rp) { return rp ? rp->box_w : 0; }
int res2_waves(const Res2Plan *rp) { return rp ? rp->waves : 0; }
int res2_hot_waves(const Res2Plan *rp) { return rp ? rp->n_hot : 0; }
int res2_hot_user_waves(const Res2Plan *rp) { return rp ? rp->n_hu : 0; }
int64_t res2_se_slots(const Res2Plan *rp) { return (int64_t)ceil_div(rp->ni, 64) * 64 + rp->n_hu; }      // indexed by item, then by hot user
const double *res2_se_partials(const Res2Plan *rp) { return rp->d_se; }
int res2_reset_cache(Res2Plan *rp)
{
    MFX_HIP(hipMemsetAsync(rp->d_cache, 0, (size_t)rp->nnz * sizeof(double), stream()));
    return MFX_OK;
}
int res2_begin_batch(Res2Plan *rp)
{
    MFX_HIP(hipMemsetAsync(rp->d_ctl + kCtlStatus, 0, 2 * sizeof(unsigned), stream()));
    return MFX_OK;
}
static Res2Params res2_params(Res2Plan *rp, const void *d_dyn, const double *ib, const double *ub, double mu, double lr, double K, int upd_u, int upd_i)
{
    Res2Params P;
    memset(&P, 0, sizeof(P));
    P.recs = rp->d_recs; P.lo = rp->d_lo; P.hi = rp->d_hi; P.cache = rp->d_cache; P.dyn = (const double *const *)d_dyn;
    P.ib = ib; P.ub = ub; P.hdr = rp->d_hdr; P.box = rp->d_box; P.first_pos = rp->d_first; P.order = rp->d_order; P.n_hot = rp->n_hot;
    P.urec = rp->d_urec; P.hu_ids = rp->d_hu_ids; P.hu_off = rp->d_hu_off; P.ubox = rp->d_ubox; P.n_hu = rp->n_hu;
    P.se_user = rp->d_se + (size_t)ceil_div(rp->ni, 64) * 64;
    P.ctl = rp->d_ctl; P.se_item = rp->d_se; P.ni = rp->ni; P.nu = rp->nu;
    P.upd_u = upd_u; P.upd_i = upd_i; P.lr = lr; P.K = K; P.mu = mu;
    P.window = env_i("MFX_K2_WIN", 1);                     // 0: no look-ahead (A/B hook)
    P.reg_extra = test_reg_extra();
    { const int64_t h = 8 * (rp->nnz / (rp->ni > 0 ? rp->ni : 1)); P.hot_deg = (int)(h < 64 ? 64 : (h > 0x7fffffff ? 0x7fffffff : h)); }
    return P;
}
// one (feature, epoch) pass: granules from V[f,:], the persistent launch, values back
int res2_enqueue_pass(Res2Plan *rp, const void *d_dyn, const double *ib, const double *ub, double mu, double lr, double K, int upd_u, int upd_i)
{
    Res2Params P = res2_params(rp, d_dyn, ib, ub, mu, lr, K, upd_u, upd_i);
    MFX_HIP(hipMemsetAsync(rp->d_ctl + kCtlReg, 0, sizeof(unsigned), stream()));
    if (rp->box_w) {
        const int n = rp->nu > rp->ni ? rp->nu : rp->ni;
        hipLaunchKernelGGL(k2_box_fill, dim3(2048), dim3(256), 0, stream(), (unsigned long long *)rp->d_box, rp->nnz + 8);
        if (rp->n_hu) hipLaunchKernelGGL(k2_box_fill, dim3(256), dim3(256), 0, stream(), (unsigned long long *)rp->d_ubox, rp->n_urec + 8);
        hipLaunchKernelGGL(k2_box_seed, dim3(ceil_div(n, 256)), dim3(256), 0, stream(), (unsigned long long *)rp->d_box, (const int32_t *)rp->d_first,
                           P.dyn, rp->nu, rp->ni, rp->d_ctl);
        if (rp->box_w == 8) hipLaunchKernelGGL(k2_res_box<8>, dim3(rp->waves), dim3(64), 0, stream(), P);
        else if (rp->box_w == 4) hipLaunchKernelGGL(k2_res_box<4>, dim3(rp->waves), dim3(64), 0, stream(), P);
        else hipLaunchKernelGGL(k2_res_box<2>, dim3(rp->waves), dim3(64), 0, stream(), P);
        MFX_HIP(hipGetLastError());
        return MFX_OK;
    }
    hipLaunchKernelGGL(k2_hdr_reset, dim3(ceil_div(rp->nu, 256)), dim3(256), 0, stream(), rp->d_hdr, P.dyn, rp->nu);
    hipLaunchKernelGGL(k2_res, dim3(rp->waves), dim3(64), 0, stream(), P);
    hipLaunchKernelGGL(k2_hdr_finish, dim3(ceil_div(rp->nu, 256)), dim3(256), 0, stream(), (const THdr *)rp->d_hdr, P.dyn, rp->nu, (const unsigned *)rp->d_ctl);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}
int res2_enqueue_commit(Res2Plan *rp, const void *d_dyn, const double *ib, const double *ub, double mu)
{
    Res2Params P = res2_params(rp, d_dyn, ib, ub, mu, 0.0, 0.0, 0, 0);
    const int blocks = (int)(ceil_div(rp->nnz, (int64_t)256) < 4096 ? ceil_div(rp->nnz, (int64_t)256) : 4096);
    hipLaunchKernelGGL(k2_commit_res, dim3(blocks), dim3(256), 0, stream(), P, (const int32_t *)rp->d_ent_of, rp->nnz);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}
int res2_status(Res2Plan *rp, unsigned *status, unsigned *applied)
{
    unsigned w[2] = {0, 0};
    MFX_HIP(hipMemcpy(w, rp->d_ctl + kCtlStatus, sizeof w, hipMemcpyDeviceToHost));
    *status = w[0];
    if (applied) *applied = w[1];
    return MFX_OK;
}
// hand the residual cache to the level-major kernels (fallback in the middle of a run)
int res2_cache_to_levels(Res2Plan *rp, double *d_cache_levels)
{
    hipLaunchKernelGGL(k2_cache_scatter, dim3(2048), dim3(256), 0, stream(), (const double *)rp->d_cache, (const int32_t *)rp->d_pos_of, rp->nnz, d_cache_levels);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

}  // namespace mfx

#ifdef MFX_RES_STAMPS
extern "C" int mfx_debug_res_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mfx::g_res_stamps), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : -4;
}
#endif

#ifdef MFX_BURST_STATS
extern "C" int mfx_debug_burst_stats(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mfx::g_burst_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -4;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mfx::g_burst_stats), z, sizeof z) != hipSuccess) return -4;
    }
    return 0;
}
#endif
