// Host side of the SGD families: plan (schedule + HBM-resident data), launches, drop-ins.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>
#include "mfx_sgd_kernels.h"
#include "mfx_sgd_internal.h"

using namespace mfx;

namespace mfx {
int levels_group_sort(Rec *d_recs, int64_t nnz, const std::vector<int64_t> &level_off, const int64_t *d_level_off,
                      int64_t n_levels, int nu, int ni, int n_groups, int32_t *d_seg_off,
                      int32_t *d_seg_mid);                                                        // mfx_schedule_dev.hip
int schedule_device(const int32_t *h_idx, const double *h_r, int64_t nnz, int nu, int ni, Rec *d_recs,
                    int32_t *h_level, std::vector<int64_t> &level_off, int64_t *n_levels,
                    int64_t level_budget, bool *over_budget, bool compact, int64_t *h_order);   // mfx_schedule_dev.hip
}

// ------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------
struct LaunchSeg {      // one kernel launch of a full-k epoch
    int64_t l0, l1;     // levels [l0, l1)
    bool tail;          // run of small levels handled by the single-workgroup tail kernel
};

struct mfx_sgd_plan {
    int64_t nnz = 0;
    std::vector<LaunchSeg> segs, segs2;   // full-k / Funk launch sequences
    int64_t *d_level_off = nullptr, *d_k1_wave_off = nullptr, *d_k2_wave_off = nullptr;
    int nu = 0, ni = 0, k = 0;
    int64_t n_levels = 0, max_level = 0;
    int batch = 16;                      // full-k kernel: ratings per wave
    int f32 = 0;                         // row storage: 0 = f64 (exact), 1 = f32, 2 = bf16 (measured options, not exact)
    int dot = 0;                         // 1: butterfly k-dot + FMA (tolerance mode, not bit-exact)
    std::vector<int64_t> level_off;      // n_levels + 1
    std::vector<int64_t> k1_wave_off;    // per level: first se_partial slot (full-k kernels)
    std::vector<int64_t> k2_wave_off;    // per level (Funk kernels)
    int64_t k1_waves = 0, k2_waves = 0;
    int64_t sched_us = 0;
    bool sched_device = false;   // schedule built by mfx_schedule_dev.hip
    size_t dev_bytes = 0;

    Rec *d_recs = nullptr;               // level-major
    double *d_U_km = nullptr, *d_V_km = nullptr;   // feature-major [k][n] (host layout, Funk)
    double *d_U_nk = nullptr, *d_V_nk = nullptr;   // row-major [n][k] (full-k kernels)
    double *d_ib = nullptr, *d_ub = nullptr;
    double *d_cache = nullptr;           // Funk residual cache (lazy)
    double *d_se_partial = nullptr;
    double *d_se_tmp = nullptr;          // stage-1 sums of the SE reduction
    double *d_se_epoch = nullptr;        // per-epoch sums (capacity se_cap)
    int *d_counter = nullptr;
    int64_t se_cap = 0;
    double *d_U_start = nullptr, *d_ib_start = nullptr;   // sharded mode (lazy)
    bool have_factors = false;
    bool nk_current = false, km_current = false;
    bool owns_users = true;              // false: d_V_*, d_ub belong to another plan (plan_alias_users)

    // XCD-resident Funk pass (k2_xcd_pass): 0 = not decided, 1 = ready, -1 = not applicable / disabled
    int k2x_state = 0;
    int32_t *d_seg_off = nullptr;        // [n_levels * kXGroups + 1] after levels_group_sort
    int32_t *d_seg_mid = nullptr;        // [n_levels * kXGroups]: first EARLY rating of every segment
    unsigned *d_xctl = nullptr;          // registration counters, status, barrier flags
    double *d_se_x = nullptr;            // its squared-error partials (sparse slots, zero elsewhere)
    int64_t k2x_slots = 0;
    unsigned x_gen = 1;                  // barrier generation base of the next pass
    int x_lds_doubles = 0;
    K2Dyn *d_k2dyn = nullptr;            // Funk per-feature record (lazy)
    hipGraphExec_t graph2 = nullptr;     // cached hipGraph of one Funk epoch
    K2Params graph2_params{};
    // cached hipGraph of one full-k epoch
    hipGraphExec_t graph = nullptr;
    K1Params graph_params{};
    int graph_mode = -1;
    bool graph_res = false;              // the cached epoch is the resident-side launch
    // resident-side epoch (mfx_sgd_res.hip): prepared at creation when its cost model beats the level schedule
    mfx::ResPlan *res = nullptr;
    int res_state = 0;                   // 1 = ready, -1 = registration failed once: stay on the level schedule
    // resident-side Funk pass (lane = item, user values inside their granules): 0 = not decided, 1 = ready, -1 = no
    mfx::Res2Plan *res2 = nullptr;
    int k2r_state = 0;
};

namespace {

template <typename T>
int dev_alloc(mfx_sgd_plan *p, T **ptr, size_t n)
{
    *ptr = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)ptr, bytes);
    if (e != hipSuccess) {
        return fail(MFX_E_NOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    if (p) p->dev_bytes += bytes;
    return MFX_OK;
}

void drop_graph(mfx_sgd_plan *p);

int ensure_se_cap(mfx_sgd_plan *p, int64_t epochs)
{
    if (epochs <= p->se_cap) return MFX_OK;
    drop_graph(p);                     // the captured epochs hold the old array's address
    if (p->d_se_epoch) (void)hipFree(p->d_se_epoch);
    p->d_se_epoch = nullptr;
    int64_t cap = epochs < 64 ? 64 : epochs;
    int rc = dev_alloc(p, &p->d_se_epoch, (size_t)cap);
    if (rc) return rc;
    p->se_cap = cap;
    return MFX_OK;
}

void drop_graph(mfx_sgd_plan *p)
{
    if (p->graph) (void)hipGraphExecDestroy(p->graph);
    p->graph = nullptr;
    p->graph_mode = -1;
    p->graph_res = false;
    if (p->graph2) (void)hipGraphExecDestroy(p->graph2);
    p->graph2 = nullptr;
}

template <typename TI, typename TO>
int transpose(const TI *in, TO *out, int rows, int cols)
{
    if (rows == 0 || cols == 0) return MFX_OK;
    const int64_t tiles = (int64_t)ceil_div(cols, 32) * ceil_div(rows, 32);
    if (tiles > 0x7fffffff) return fail(MFX_E_ARG, "transpose: %d x %d is too large", rows, cols);
    hipLaunchKernelGGL((k_transpose<TI, TO>), dim3((unsigned)tiles), dim3(256), 0, stream(), in, out, rows, cols);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

// [k][n] f64 (host layout) -> [n][k] rows in the plan's storage type, and back
int to_nk(mfx_sgd_plan *p)
{
    if (p->nk_current) return MFX_OK;
    int rc;
    if (p->f32 == 2) {
        rc = transpose(p->d_U_km, (__bf16 *)p->d_U_nk, p->k, p->ni);
        if (!rc) rc = transpose(p->d_V_km, (__bf16 *)p->d_V_nk, p->k, p->nu);
    } else if (p->f32) {
        rc = transpose(p->d_U_km, (float *)p->d_U_nk, p->k, p->ni);
        if (!rc) rc = transpose(p->d_V_km, (float *)p->d_V_nk, p->k, p->nu);
    } else {
        rc = transpose(p->d_U_km, p->d_U_nk, p->k, p->ni);
        if (!rc) rc = transpose(p->d_V_km, p->d_V_nk, p->k, p->nu);
    }
    if (!rc) p->nk_current = true;
    return rc;
}

int to_km(mfx_sgd_plan *p)
{
    if (p->km_current) return MFX_OK;
    int rc;
    if (p->f32 == 2) {
        rc = transpose((const __bf16 *)p->d_U_nk, p->d_U_km, p->ni, p->k);
        if (!rc) rc = transpose((const __bf16 *)p->d_V_nk, p->d_V_km, p->nu, p->k);
    } else if (p->f32) {
        rc = transpose((const float *)p->d_U_nk, p->d_U_km, p->ni, p->k);
        if (!rc) rc = transpose((const float *)p->d_V_nk, p->d_V_km, p->nu, p->k);
    } else {
        rc = transpose((const double *)p->d_U_nk, p->d_U_km, p->ni, p->k);
        if (!rc) rc = transpose((const double *)p->d_V_nk, p->d_V_km, p->nu, p->k);
    }
    if (!rc) p->km_current = true;
    return rc;
}

size_t k1_lds_bytes(int k, int batch)
{
    return (size_t)kWaves * batch * (size_t)k1_stride(k) * sizeof(double);
}

int k1_vec_mode()
{
    const char *e = getenv("MFX_K1_VEC");      // tuning hook: 0 = 8 bytes per lane, 1 / 2 see k1_pick
    return (e && *e) ? atoi(e) : 2;
}

// pick the kernel instantiation for (k, mode, batch, storage type); `tag` names the template
// arguments that were chosen (mfx_sgd_plan_kernel_name: bench.py labels its roofline with it)
struct K1Sel {
    void *fn;
    const char *tag;
};

template <int MODE, typename TS, bool TAIL, int DOT = 0>
K1Sel k1_pick(int k, int batch)
{
#define MFX_K1(KP, FPL, B) K1Sel{TAIL ? (void *)k1_tail<KP, FPL, B, MODE, 0, TS> : (void *)k1_level<KP, FPL, B, MODE, 0, TS>, #KP "," #FPL "," #B ",8B/lane"}
    // (the tail kernel runs several levels in one launch: it keeps plain stores, which the compiler
    // drains before the workgroup barrier between levels -- an asm store would slip past it)
    // 16 bytes per lane (contiguous feature pairs), f64 rows, even k > 4: vec = 1 plain
    // stores, vec = 2 (default) write-through stores -- no dirty lines left for the end-of-kernel
    // write-back.  cfg 2: 1.68 (8 B per lane) / 1.75 / 1.88 G updates/s; 16 or 4 ratings per wave,
    // and write-through bias stores, measured slower.
#define MFX_K1VF(KP, FPL, B, NTF) K1Sel{TAIL ? (void *)k1_tail<KP, FPL, B, MODE, 0, TS, true, DOT> : (void *)k1_level<KP, FPL, B, MODE, NTF, TS, true, DOT>, (NTF == 2 && !TAIL) ? (DOT ? #KP "," #FPL "," #B ",vec16+wt,butterfly-dot" : #KP "," #FPL "," #B ",vec16+wt") : (DOT ? #KP "," #FPL "," #B ",vec16,butterfly-dot" : #KP "," #FPL "," #B ",vec16")}
    {
        // the learned-bias family stores ONE feature per row: 16-byte loads only, plain stores
        const int vec = (MODE == K1_LEARNED && k1_vec_mode()) ? 1 : k1_vec_mode();   // (write-through scalars: no gain)
        if constexpr (sizeof(TS) == 8) {
            if (vec && k % 2 == 0 && k > 4) {
#define MFX_K1V(KP, B, NTF) MFX_K1VF(KP, 2, B, NTF)
                if (k <= 16 && batch == 16) return vec == 2 ? MFX_K1V(8, 16, 2) : MFX_K1V(8, 16, 0);
                if (k > 16 && k <= 32 && batch == 16) return vec == 2 ? MFX_K1V(16, 16, 2) : MFX_K1V(16, 16, 0);
                if (k > 32 && k <= 64 && batch == 8) return vec == 2 ? MFX_K1V(32, 8, 2) : MFX_K1V(32, 8, 0);
                if (k > 64 && k <= 128 && batch == 8) return vec == 2 ? MFX_K1V(64, 8, 2) : MFX_K1V(64, 8, 0);
                if (k > 128 && k <= 256 && batch == 4) return vec == 2 ? MFX_K1VF(64, 4, 4, 2) : MFX_K1VF(64, 4, 4, 0);
                if (k > 256 && batch == 2) return vec == 2 ? MFX_K1VF(64, 8, 2, 2) : MFX_K1VF(64, 8, 2, 0);
#undef MFX_K1V
            }
        } else {
            // f32 storage: four features per 16-byte chunk, k a multiple of 4
            if (vec && k % 4 == 0 && k > 8) {
                if (k <= 32 && batch == 16) return vec == 2 ? MFX_K1VF(8, 4, 16, 2) : MFX_K1VF(8, 4, 16, 0);
                if (k > 32 && k <= 64 && batch == 8) return vec == 2 ? MFX_K1VF(16, 4, 8, 2) : MFX_K1VF(16, 4, 8, 0);
                if (k > 64 && k <= 128 && batch == 8) return vec == 2 ? MFX_K1VF(32, 4, 8, 2) : MFX_K1VF(32, 4, 8, 0);
                if (k > 128 && k <= 256 && batch == 4) return vec == 2 ? MFX_K1VF(64, 4, 4, 2) : MFX_K1VF(64, 4, 4, 0);
                if (k > 256 && batch == 2) return vec == 2 ? MFX_K1VF(64, 8, 2, 2) : MFX_K1VF(64, 8, 2, 0);
            }
        }
    }
#undef MFX_K1VF
    if (k <= 8) return MFX_K1(8, 1, 16);
    if (k <= 16) return MFX_K1(16, 1, 16);
    if (k <= 32) return MFX_K1(32, 1, 16);
    if (k <= 64) {
        if (batch == 16) return MFX_K1(64, 1, 16);
        if (batch == 4) return MFX_K1(64, 1, 4);
        return MFX_K1(64, 1, 8);
    }
    if (k <= 128) return MFX_K1(64, 2, 8);
    if (k <= 256) return MFX_K1(64, 4, 4);
    return MFX_K1(64, 8, 2);
#undef MFX_K1
}

// bf16 row storage: 16-byte chunks of EIGHT features, k a multiple of 8 and >= 16, linear / logistic only
bool k1_bf16_eligible(int k) { return k % 8 == 0 && k >= 16 && k1_vec_mode() != 0; }
template <int MODE, bool TAIL>
K1Sel k1_pick_bf16(int k, int batch)
{
    const int vec = k1_vec_mode();
#define MFX_K1B(KP, B) K1Sel{TAIL ? (void *)k1_tail<KP, 8, B, MODE, 0, __bf16, true> : (vec == 2 ? (void *)k1_level<KP, 8, B, MODE, 2, __bf16, true> : (void *)k1_level<KP, 8, B, MODE, 0, __bf16, true>), #KP ",8," #B ",vec16,bf16"}
    if (k <= 64 && batch == 8) return MFX_K1B(8, 8);
    if (k <= 64 && batch == 16) return MFX_K1B(8, 16);
    if (k <= 128) return MFX_K1B(16, 8);
    if (k <= 256) return MFX_K1B(32, 4);
    return MFX_K1B(64, 2);
#undef MFX_K1B
}

// the butterfly-dot (tolerance) mode exists for f64 rows, even k > 4, linear / logistic
bool k1_dot_eligible(int k, int f32) { return !f32 && k % 2 == 0 && k > 4 && k1_vec_mode() != 0; }

template <bool TAIL>
K1Sel k1_select(int k, int mode, int batch, int f32, int dot = 0)
{
    if (dot && k1_dot_eligible(k, f32)) {
        if (mode == 0) return k1_pick<K1_LINEAR, double, TAIL, 1>(k, batch);
        if (mode == 1) return k1_pick<K1_LOGISTIC, double, TAIL, 1>(k, batch);
    }
    if (f32 == 2) {
        if (mode == 1) return k1_pick_bf16<K1_LOGISTIC, TAIL>(k, batch);
        return k1_pick_bf16<K1_LINEAR, TAIL>(k, batch);          // (learned bias: refused by its entry point)
    }
    switch (mode * 2 + (f32 ? 1 : 0)) {
    case 0: return k1_pick<K1_LINEAR, double, TAIL>(k, batch);
    case 1: return k1_pick<K1_LINEAR, float, TAIL>(k, batch);
    case 2: return k1_pick<K1_LOGISTIC, double, TAIL>(k, batch);
    case 3: return k1_pick<K1_LOGISTIC, float, TAIL>(k, batch);
    case 4: return k1_pick<K1_LEARNED, double, TAIL>(k, batch);
    default: return k1_pick<K1_LEARNED, float, TAIL>(k, batch);
    }
}

void *k1_kernel(int k, int mode, int batch, int f32, int dot) { return k1_select<false>(k, mode, batch, f32, dot).fn; }
void *k1_tail_kernel(int k, int mode, int batch, int f32, int dot) { return k1_select<true>(k, mode, batch, f32, dot).fn; }

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// ratings per wave: k1_batch(k), or the MFX_K1_BATCH tuning override (k in 33..64 only)
int plan_batch(int k)
{
    int b = k1_batch(k);
    if (k > 32 && k <= 64) {
        const int o = env_int("MFX_K1_BATCH", b);
        if (o == 4 || o == 8 || o == 16) b = o;
    }
    return b;
}

constexpr int kMaxK = 512;
constexpr size_t kMaxGraphNodes = 65536;

int enqueue_reduce(mfx_sgd_plan *p, int64_t n_partials, const double *partials = nullptr)
{
    hipLaunchKernelGGL(k_reduce_se1, dim3(kReduceBlocks), dim3(256), 0, stream(), partials ? partials : p->d_se_partial,
                       n_partials, p->d_se_tmp);
    hipLaunchKernelGGL(k_reduce_se2, dim3(1), dim3(64), 0, stream(), p->d_se_tmp, p->d_se_epoch,
                       p->d_counter);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

// enqueue one full-k epoch (all levels + the SE reduction) on the library stream.
// ev != nullptr: record an event pair around every level launch (profiling epoch).
int enqueue_k1_epoch(mfx_sgd_plan *p, int mode, const K1Params &P, std::vector<hipEvent_t> *ev)
{
    const int batch = p->batch;
    void *fn = k1_kernel(p->k, mode, batch, p->f32, p->dot);
    void *fn_tail = k1_tail_kernel(p->k, mode, batch, p->f32, p->dot);
    // (the butterfly dot needs no product tile)
    const size_t lds = (p->dot && mode != K1_LEARNED && k1_dot_eligible(p->k, p->f32)) ? 0 : k1_lds_bytes(p->k, batch);
    const size_t lds_tail = lds / kWaves * kTailWaves;
    if (lds_tail > 64 * 1024)
        (void)hipFuncSetAttribute(fn_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tail);
    size_t e = 0;
    for (const LaunchSeg &sg : p->segs) {
        K1Params Pl = P;
        if (ev) MFX_HIP(hipEventRecord((*ev)[e++], stream()));
        if (sg.l1 - sg.l0 == 1 && !sg.tail) {
            const int64_t l = sg.l0;
            int64_t st = p->level_off[l];
            int cnt = (int)(p->level_off[l + 1] - st);
            const int waves = ceil_div(cnt, batch);
            const int blocks = ceil_div(waves, kWaves);
            int64_t woff = p->k1_wave_off[l];
            void *args[] = {&Pl, &st, &cnt, &woff};
            MFX_HIP(hipLaunchKernel(fn, dim3(blocks), dim3(kWaves * 64), args, lds, stream()));
        } else {
            const int64_t *lo = p->d_level_off, *wo = p->d_k1_wave_off;
            int l0 = (int)sg.l0, l1 = (int)sg.l1;
            void *args[] = {&Pl, &lo, &wo, &l0, &l1};
            MFX_HIP(hipLaunchKernel(fn_tail, dim3(1), dim3(kTailWaves * 64), args, lds_tail, stream()));
        }
        if (ev) MFX_HIP(hipEventRecord((*ev)[e++], stream()));
    }
    return enqueue_reduce(p, p->k1_waves);
}

bool same_params(const K1Params &a, const K1Params &b)
{
    return memcmp(&a, &b, sizeof(K1Params)) == 0;
}

int use_graph()
{
    const char *e = getenv("MFX_GRAPH");
    return !(e && e[0] == '0');
}

int check_plan(const mfx_sgd_plan *p, const char *who)
{
    if (!p) {
        return fail(MFX_E_ARG, "%s: NULL plan", who);
    }
    if (!p->have_factors) {
        return fail(MFX_E_STATE, "%s: mfx_sgd_set_factors has not been called", who);
    }
    return MFX_OK;
}

K1Params k1_params(mfx_sgd_plan *p, double lr, double Ku, double Ki, double Kb, int upd_u,
                   int upd_i)
{
    K1Params P;
    memset(&P, 0, sizeof(P));
    P.recs = p->d_recs;
    P.U = p->d_U_nk;
    P.V = p->d_V_nk;
    P.ib = p->d_ib;
    P.ub = p->d_ub;
    P.se_partial = p->d_se_partial;
    P.k = p->k;
    P.upd_u = upd_u ? 1 : 0;
    P.upd_i = upd_i ? 1 : 0;
    P.lr = lr;
    P.Ku = Ku;
    P.Ki = Ki;
    P.Kb = Kb;
    P.f = -1;
    return P;
}

// run `epochs` full-k epochs; hook (nullable) is called after each epoch's launches have
// been enqueued (sharded mode enqueues its all-reduce there).
// one resident-side epoch + its squared-error reduction (mfx_sgd_res.hip)
// 0: f64 rows + ordered dot; 1: butterfly dot (f64 rows); 2: f32 rows (ordered dot); -1: not served there (bf16 rows)
int res_variant(const mfx_sgd_plan *p) { return p->f32 == 2 ? -1 : (p->f32 == 1 ? (p->dot ? -1 : 2) : (p->dot ? 1 : 0)); }
int enqueue_res_epoch(mfx_sgd_plan *p, int mode, const K1Params &P)
{
    int rc = res_enqueue_epoch(p->res, mode, (double *)P.U, (double *)P.V, P.ib, P.ub, P.lr, P.Ku, P.Ki, P.Kb, res_variant(p));
    if (rc) return rc;
    return enqueue_reduce(p, res_se_slots(p->res), res_se_partials(p->res));
}

// the resident-side epoch serves the plain linear / logistic run: f64 rows, both sides trained; the ordered dot, or -- round 4 --
// the butterfly-dot tolerance mode (every binding but the burst variant)
bool res_on(const mfx_sgd_plan *p)
{
    const int v = res_variant(p);
    return p->res && p->res_state == 1 && v >= 0 && (v == 0 || res_variants_ok(p->res));
}
bool res_applies(const mfx_sgd_plan *p, int mode, const K1Params &P)
{
    return res_on(p) && (mode == K1_LINEAR || mode == K1_LOGISTIC) && P.upd_u && P.upd_i;
}

template <typename Hook>
int run_k1_epochs(mfx_sgd_plan *p, int mode, int epochs, const K1Params &P, Hook &&hook, bool reset_se = true,
                  bool allow_res = false, bool defer_status = false)
{
    int rc = reset_se ? ensure_se_cap(p, epochs) : MFX_OK;
    if (rc) return rc;
    rc = to_nk(p);
    if (rc) return rc;
    p->km_current = false;
    if (reset_se) MFX_HIP(hipMemsetAsync(p->d_counter, 0, sizeof(int), stream()));
    const bool res = allow_res && res_applies(p, mode, P);
    if (res && !defer_status && (rc = res_begin_batch(p->res))) return rc;
    // very long launch sequences (power-law data without the tail kernel) are not worth a graph
    const bool graph = use_graph() && (res || p->segs.size() <= kMaxGraphNodes);
    if (graph && !(p->graph && p->graph_mode == mode && p->graph_res == res && same_params(p->graph_params, P))) {
        drop_graph(p);
        hipGraph_t g = nullptr;
        MFX_HIP(hipStreamBeginCapture(stream(), hipStreamCaptureModeThreadLocal));
        rc = res ? enqueue_res_epoch(p, mode, P) : enqueue_k1_epoch(p, mode, P, nullptr);
        hipError_t e = hipStreamEndCapture(stream(), &g);
        if (rc) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess) {
            return fail(MFX_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        }
        e = hipGraphInstantiate(&p->graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) {
            p->graph = nullptr;
            return fail(MFX_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        }
        p->graph_mode = mode;
        p->graph_res = res;
        p->graph_params = P;
    }
    for (int ep = 0; ep < epochs; ++ep) {
        if (graph) MFX_HIP(hipGraphLaunch(p->graph, stream()));
        else {
            rc = res ? enqueue_res_epoch(p, mode, P) : enqueue_k1_epoch(p, mode, P, nullptr);
            if (rc) return rc;
        }
        rc = hook(ep);
        if (rc) return rc;
    }
    if (res && !defer_status) {
        // the resident-side launch reports through a status word: 1 = the static grid was not co-resident and
        // NOTHING was modified by any epoch of this batch (the word is sticky) -> the level schedule, from now on
        MFX_HIP(hipStreamSynchronize(stream()));
        unsigned status = 0, applied = 0;
        if ((rc = res_status(p->res, &status, &applied))) return rc;
        if (status == 1) {
            // the first `applied` epochs of the batch ran in full, the one that failed its check and every later one did nothing:
            // the rest goes to the level schedule, its squared errors behind the ones already there
            p->res_state = -1;
            if ((int)applied >= epochs) return MFX_OK;
            const int done = (int)applied;
            MFX_HIP(hipMemcpy(p->d_counter, &done, sizeof(int), hipMemcpyHostToDevice));
            return run_k1_epochs(p, mode, epochs - done, P, hook, false, false);
        }
        if (status) return fail(MFX_E_HIP, "mfx_sgd_run_kmf: the resident-side epoch timed out waiting for a chain counter");
    }
    return MFX_OK;
}

int fetch_rmse(mfx_sgd_plan *p, int64_t n, double *rmse_out, double *se_out)
{
    MFX_HIP(hipStreamSynchronize(stream()));
    if ((!rmse_out && !se_out) || n <= 0) return MFX_OK;
    std::vector<double> se((size_t)n);
    MFX_HIP(hipMemcpy(se.data(), p->d_se_epoch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
        if (se_out) se_out[i] = se[i];
        if (rmse_out) rmse_out[i] = sqrt(se[i] / (double)p->nnz);
    }
    return MFX_OK;
}

}  // namespace

// exposed to mfx_comm.hip / mfx_dsgd.hip (mfx_sgd_internal.h)
namespace mfx {
int sgd_run_k1_hooked(mfx_sgd_plan *p, int mode, int epochs, double lr, double Ku, double Ki,
                      double Kb, int (*hook)(mfx_sgd_plan *, int, void *), void *ctx,
                      double *rmse_out, double *se_out)
{
    int rc = check_plan(p, "mfx_sgd_run_kmf_sharded");
    if (rc) return rc;
    if (p->f32) return fail(MFX_E_STATE, "mfx_sgd_run_kmf_sharded: f32 row storage is single-GPU only");
    K1Params P = k1_params(p, lr, Ku, Ki, Kb, 1, 1);
    rc = run_k1_epochs(p, mode, epochs, P, [&](int ep) { return hook ? hook(p, ep, ctx) : MFX_OK; });
    if (rc) return rc;
    return fetch_rmse(p, epochs, rmse_out, se_out);
}
int sgd_reset_se(mfx_sgd_plan *p, int64_t epochs)
{
    int rc = ensure_se_cap(p, epochs);
    if (rc) return rc;
    MFX_HIP(hipMemsetAsync(p->d_counter, 0, sizeof(int), stream()));
    if (p->res && p->res_state == 1) return res_begin_batch(p->res);
    return MFX_OK;
}
int sgd_enqueue_k1(mfx_sgd_plan *p, int mode, double lr, double Ku, double Ki, double Kb)
{
    int rc = check_plan(p, "sgd_enqueue_k1");
    if (rc) return rc;
    K1Params P = k1_params(p, lr, Ku, Ki, Kb, 1, 1);
    return run_k1_epochs(p, mode, 1, P, [](int) { return MFX_OK; }, false, true, true);
}
int sgd_res_check(mfx_sgd_plan *p)
{
    if (!p->res || p->res_state != 1) return MFX_OK;
    unsigned status = 0;
    int rc = res_status(p->res, &status);
    if (rc) return rc;
    if (status == 1) {
        p->res_state = -1;
        return fail(MFX_E_HIP, "the static resident-side epoch of a block found its grid not co-resident (is another process using this "
                               "GPU?); the block was skipped and the run is invalid -- set MFX_K1_RES=levels and run again");
    }
    if (status) return fail(MFX_E_HIP, "a resident-side block epoch timed out waiting for a chain counter");
    return MFX_OK;
}
int sgd_res_poll(mfx_sgd_plan *p, unsigned *status)
{
    *status = 0;
    if (!p->res || p->res_state != 1) return MFX_OK;
    return res_status(p->res, status);
}
void plan_force_levels(mfx_sgd_plan *p)
{
    if (p->res && p->res_state == 1) p->res_state = -1;
}
int plan_epoch_kernel(const mfx_sgd_plan *p)
{
    return res_on(p) ? res_kind(p->res) + 1 : MFX_EPOCH_LEVELS;
}
int sgd_fetch_se(mfx_sgd_plan *p, int64_t n, double *se_out) { return fetch_rmse(p, n, nullptr, se_out); }
double *plan_U(mfx_sgd_plan *p) { return p->d_U_nk; }
double *plan_V(mfx_sgd_plan *p) { return p->d_V_nk; }
double *plan_ib(mfx_sgd_plan *p) { return p->d_ib; }
double *plan_ub(mfx_sgd_plan *p) { return p->d_ub; }
int plan_ni(mfx_sgd_plan *p) { return p->ni; }
int plan_nu(mfx_sgd_plan *p) { return p->nu; }
int plan_k(mfx_sgd_plan *p) { return p->k; }
int64_t plan_nnz(mfx_sgd_plan *p) { return p->nnz; }
int plan_start_buffers(mfx_sgd_plan *p, double **U_start, double **ib_start)
{
    if (!p->d_U_start) {
        int rc = dev_alloc(p, &p->d_U_start, (size_t)p->ni * p->k);
        if (!rc) rc = dev_alloc(p, &p->d_ib_start, (size_t)p->ni);
        if (rc) return rc;
    }
    *U_start = p->d_U_start;
    *ib_start = p->d_ib_start;
    return MFX_OK;
}
int plan_prepare_nk(mfx_sgd_plan *p) { return to_nk(p); }
int plan_alias_users(mfx_sgd_plan *p, mfx_sgd_plan *master)
{
    if (!p || !master || p == master || p->nu != master->nu || p->k != master->k || p->f32 || master->f32)
        return fail(MFX_E_ARG, "plan_alias_users: plans disagree on the user side");
    if (p->have_factors) return fail(MFX_E_STATE, "plan_alias_users: call before factors are set");
    drop_graph(p);
    if (p->owns_users) {
        if (p->d_V_km) (void)hipFree(p->d_V_km);
        if (p->d_V_nk) (void)hipFree(p->d_V_nk);
        if (p->d_ub) (void)hipFree(p->d_ub);
    }
    p->owns_users = false;
    p->d_V_km = master->d_V_km;
    p->d_V_nk = master->d_V_nk;
    p->d_ub = master->d_ub;
    return MFX_OK;
}
int plan_upload_items(mfx_sgd_plan *p, const double *u_km, const double *ib)
{
    if (p->f32) return fail(MFX_E_STATE, "plan_upload_items: f64 rows only");
    MFX_HIP(hipStreamSynchronize(stream()));
    MFX_HIP(hipMemcpy(p->d_U_km, u_km, (size_t)p->k * p->ni * sizeof(double), hipMemcpyHostToDevice));
    if (ib) MFX_HIP(hipMemcpy(p->d_ib, ib, (size_t)p->ni * sizeof(double), hipMemcpyHostToDevice));
    else MFX_HIP(hipMemset(p->d_ib, 0, (size_t)p->ni * sizeof(double)));
    return transpose(p->d_U_km, p->d_U_nk, p->k, p->ni);
}
int plan_upload_users(mfx_sgd_plan *p, const double *v_km, const double *ub)
{
    if (p->f32) return fail(MFX_E_STATE, "plan_upload_users: f64 rows only");
    MFX_HIP(hipStreamSynchronize(stream()));
    MFX_HIP(hipMemcpy(p->d_V_km, v_km, (size_t)p->k * p->nu * sizeof(double), hipMemcpyHostToDevice));
    if (ub) MFX_HIP(hipMemcpy(p->d_ub, ub, (size_t)p->nu * sizeof(double), hipMemcpyHostToDevice));
    else MFX_HIP(hipMemset(p->d_ub, 0, (size_t)p->nu * sizeof(double)));
    return transpose(p->d_V_km, p->d_V_nk, p->k, p->nu);
}
int plan_download_items(mfx_sgd_plan *p, double *u_km, double *ib)
{
    int rc = transpose((const double *)p->d_U_nk, p->d_U_km, p->ni, p->k);
    if (rc) return rc;
    MFX_HIP(hipStreamSynchronize(stream()));
    if (u_km) MFX_HIP(hipMemcpy(u_km, p->d_U_km, (size_t)p->k * p->ni * sizeof(double), hipMemcpyDeviceToHost));
    if (ib) MFX_HIP(hipMemcpy(ib, p->d_ib, (size_t)p->ni * sizeof(double), hipMemcpyDeviceToHost));
    return MFX_OK;
}
int plan_download_users(mfx_sgd_plan *p, double *v_km, double *ub)
{
    int rc = transpose((const double *)p->d_V_nk, p->d_V_km, p->nu, p->k);
    if (rc) return rc;
    MFX_HIP(hipStreamSynchronize(stream()));
    if (v_km) MFX_HIP(hipMemcpy(v_km, p->d_V_km, (size_t)p->k * p->nu * sizeof(double), hipMemcpyDeviceToHost));
    if (ub) MFX_HIP(hipMemcpy(ub, p->d_ub, (size_t)p->nu * sizeof(double), hipMemcpyDeviceToHost));
    return MFX_OK;
}
void plan_mark_ready(mfx_sgd_plan *p)
{
    p->have_factors = true;
    p->nk_current = true;        // the row-major copies are the live ones from here on
    p->km_current = false;
}
}  // namespace mfx

extern "C" {

mfx_sgd_plan *mfx_sgd_plan_create(const int32_t *idx, const double *r, int64_t nnz, int nu, int ni,
                                  int dim)
{
    return mfx_sgd_plan_create2(idx, r, nnz, nu, ni, dim, MFX_SCHED_AUTO);
}

mfx_sgd_plan *mfx_sgd_plan_create2(const int32_t *idx, const double *r, int64_t nnz, int nu, int ni,
                                   int dim, int sched)
{
    return mfx::plan_create_ex(idx, r, nnz, nu, ni, dim, sched, MFX_EPOCH_AUTO);
}

mfx_sgd_plan *mfx_sgd_plan_create3(const int32_t *idx, const double *r, int64_t nnz, int nu, int ni,
                                   int dim, int sched, int epoch)
{
    return mfx::plan_create_ex(idx, r, nnz, nu, ni, dim, sched, epoch);
}

}  // extern "C"

mfx_sgd_plan *mfx::plan_create_ex(const int32_t *idx, const double *r, int64_t nnz, int nu, int ni,
                                  int dim, int sched, int epoch)
{
    mfx::TraceRange trace_("mfx:plan_create (schedule + upload)");
    if (!idx || !r || nnz < 0 || nu <= 0 || ni <= 0 || dim < 1 || dim > kMaxK ||
        sched < MFX_SCHED_AUTO || sched > MFX_SCHED_DEVICE || (epoch & 0xff) < MFX_EPOCH_AUTO || (epoch & 0xff) > MFX_EPOCH_STATIC) {
        fail(MFX_E_ARG, "mfx_sgd_plan_create: bad argument (nnz=%lld nu=%d ni=%d k=%d sched=%d; 1 <= k <= %d)",
             (long long)nnz, nu, ni, dim, sched, kMaxK);
        return nullptr;
    }
    if (const char *e = getenv("MFX_SCHED")) {               // tuning / test hook
        if (e[0] == 'h') sched = MFX_SCHED_HOST;
        else if (e[0] == 'd') sched = MFX_SCHED_DEVICE;
    }
    if (ensure_init()) return nullptr;
    mfx_sgd_plan *p = new (std::nothrow) mfx_sgd_plan();
    if (!p) {
        fail(MFX_E_NOMEM, "mfx_sgd_plan_create: out of host memory");
        return nullptr;
    }
    p->nnz = nnz;
    p->nu = nu;
    p->ni = ni;
    p->k = dim;
    p->batch = plan_batch(dim);
    auto bail = [&]() -> mfx_sgd_plan * {
        mfx_sgd_plan_destroy(p);
        return nullptr;
    };
    try {
        const auto t0 = std::chrono::steady_clock::now();
        static_assert(sizeof(Rec) == 16, "Rec is the 16-byte record of mfx_schedule_pack");
        if (dev_alloc(p, &p->d_recs, (size_t)nnz)) return bail();
        // Schedule builder.  Device: O(nnz) + ~3 us per level, gives up beyond a level budget (heavily
        // skewed data) -> host.  Auto: device from 2^20 ratings on, budget nnz/1000 levels (the point
        // where the host's 3 ns per rating is cheaper).
        bool on_device = false;
        if (sched == MFX_SCHED_DEVICE || (sched == MFX_SCHED_AUTO && nnz >= (1 << 20) && nnz < ((int64_t)1 << 31) - 1)) {
            const int64_t budget = sched == MFX_SCHED_DEVICE ? ((int64_t)1 << 20) : nnz / 1000;
            bool over = false;
            if (schedule_device(idx, r, nnz, nu, ni, p->d_recs, nullptr, p->level_off, &p->n_levels, budget, &over, false, nullptr)) {
                if (sched == MFX_SCHED_DEVICE || last_code() == MFX_E_RANGE) return bail();
                over = true;                                   // e.g. scratch did not fit: host path
            }
            if (over && sched == MFX_SCHED_DEVICE) {
                fail(MFX_E_STATE, "mfx_sgd_plan_create: more than %lld levels, use the host schedule builder",
                     (long long)budget);
                return bail();
            }
            on_device = !over;
        }
        p->sched_device = on_device;
        if (!on_device) {
            std::vector<int32_t> level((size_t)(nnz ? nnz : 1));
            if (mfx_schedule_levels(idx, nnz, nu, ni, level.data(), &p->n_levels)) return bail();
            p->level_off.assign((size_t)p->n_levels + 1, 0);
            std::vector<Rec> recs((size_t)(nnz ? nnz : 1));
            if (mfx_schedule_pack(idx, r, level.data(), nnz, p->n_levels, recs.data(), p->level_off.data()))
                return bail();
            // inside a level: ascending item, like the device builder (an item occurs once per level)
            {
                unsigned hw = std::thread::hardware_concurrency();
                const int T = nnz < (1 << 18) ? 1 : (int)std::min<unsigned>(hw ? hw : 1, 16);
                auto work = [&](int i) {
                    for (int64_t l = i; l < p->n_levels; l += T)
                        std::sort(recs.begin() + p->level_off[(size_t)l], recs.begin() + p->level_off[(size_t)l + 1],
                                  [](const Rec &a, const Rec &b) { return a.item < b.item; });
                };
                if (T == 1) {
                    work(0);
                } else {
                    std::vector<std::thread> th;
                    for (int i = 0; i < T; ++i) th.emplace_back(work, i);
                    for (auto &x : th) x.join();
                }
            }
            if (hipMemcpy(p->d_recs, recs.data(), (size_t)nnz * sizeof(Rec), hipMemcpyHostToDevice) != hipSuccess) {
                fail(MFX_E_HIP, "mfx_sgd_plan_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
                return bail();
            }
        }
        p->k1_wave_off.assign((size_t)p->n_levels + 1, 0);
        p->k2_wave_off.assign((size_t)p->n_levels + 1, 0);
        for (int64_t l = 0; l < p->n_levels; ++l) {
            const int64_t c = p->level_off[l + 1] - p->level_off[l];
            if (c > p->max_level) p->max_level = c;
            // slots are per launched wave (blocks are whole)
            p->k1_wave_off[l + 1] = p->k1_wave_off[l] +
                                    ceil_div(ceil_div(c, (int64_t)p->batch), (int64_t)kWaves) * kWaves;
            p->k2_wave_off[l + 1] = p->k2_wave_off[l] + ceil_div(c, (int64_t)kK2Block) * (kK2Block / 64);
        }
        // launch segments: big levels one launch each; runs of small levels -> tail kernel
        {
            const int64_t T = (int64_t)kTailWaves * p->batch;
            const char *e = getenv("MFX_K1_TAIL");
            const bool tail_on = !(e && e[0] == '0');
            int64_t l = 0;
            while (l < p->n_levels) {
                const int64_t c = p->level_off[l + 1] - p->level_off[l];
                if (c <= 0) { ++l; continue; }
                if (tail_on && c <= T) {
                    int64_t m = l;
                    while (m < p->n_levels && p->level_off[m + 1] - p->level_off[m] <= T) ++m;
                    p->segs.push_back({l, m, true});
                    l = m;
                } else {
                    p->segs.push_back({l, l + 1, false});
                    ++l;
                }
            }
            l = 0;
            while (l < p->n_levels) {                      // same for the Funk pass
                const int64_t c = p->level_off[l + 1] - p->level_off[l];
                if (c <= 0) { ++l; continue; }
                if (tail_on && c <= kTail2) {
                    int64_t m = l;
                    while (m < p->n_levels && p->level_off[m + 1] - p->level_off[m] <= kTail2) ++m;
                    p->segs2.push_back({l, m, true});
                    l = m;
                } else {
                    p->segs2.push_back({l, l + 1, false});
                    ++l;
                }
            }
        }
        p->k1_waves = p->k1_wave_off[p->n_levels];
        p->k2_waves = p->k2_wave_off[p->n_levels];
        p->sched_us = std::chrono::duration_cast<std::chrono::microseconds>(
                          std::chrono::steady_clock::now() - t0).count();

        if (dev_alloc(p, &p->d_level_off, (size_t)p->n_levels + 1) ||
            dev_alloc(p, &p->d_k1_wave_off, (size_t)p->n_levels + 1) ||
            dev_alloc(p, &p->d_k2_wave_off, (size_t)p->n_levels + 1))
            return bail();
        if (hipMemcpy(p->d_level_off, p->level_off.data(), ((size_t)p->n_levels + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_k1_wave_off, p->k1_wave_off.data(), ((size_t)p->n_levels + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_k2_wave_off, p->k2_wave_off.data(), ((size_t)p->n_levels + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) {
            fail(MFX_E_HIP, "mfx_sgd_plan_create: upload of level offsets failed");
            return bail();
        }
        if (dev_alloc(p, &p->d_U_km, (size_t)ni * dim) || dev_alloc(p, &p->d_V_km, (size_t)nu * dim) ||
            dev_alloc(p, &p->d_U_nk, (size_t)ni * dim) || dev_alloc(p, &p->d_V_nk, (size_t)nu * dim) ||
            dev_alloc(p, &p->d_ib, (size_t)ni) || dev_alloc(p, &p->d_ub, (size_t)nu) ||
            dev_alloc(p, &p->d_counter, 1) || dev_alloc(p, &p->d_se_tmp, (size_t)kReduceBlocks))
            return bail();
        const int64_t np = p->k1_waves > p->k2_waves ? p->k1_waves : p->k2_waves;
        if (dev_alloc(p, &p->d_se_partial, (size_t)np)) return bail();
        // slots of waves that exit early (block padding) must read as 0
        if (hipMemset(p->d_se_partial, 0, (size_t)(np ? np : 1) * sizeof(double)) != hipSuccess) {
            fail(MFX_E_HIP, "mfx_sgd_plan_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
            return bail();
        }
        if (ensure_se_cap(p, 64)) return bail();
        if ((epoch & 0xff) != MFX_EPOCH_LEVELS) {
            if (res_prepare(&p->res, idx, p->d_recs, nnz, nu, ni, dim, p->n_levels, epoch & 0xff, (epoch & kEpochHeadroom) != 0)) return bail();
            if (p->res) { p->res_state = 1; p->dev_bytes += res_dev_bytes(p->res); }
        }
    } catch (const std::bad_alloc &) {
        mfx::fail(MFX_E_NOMEM, "mfx_sgd_plan_create: out of host memory");
        return bail();
    }
    return p;
}

extern "C" {

void mfx_sgd_plan_destroy(mfx_sgd_plan *p)
{
    if (!p) return;
    if (ready()) (void)hipStreamSynchronize(stream());
    drop_graph(p);
    if (!p->owns_users) p->d_V_km = p->d_V_nk = p->d_ub = nullptr;
    void *ptrs[] = {p->d_recs, p->d_U_km, p->d_V_km, p->d_U_nk, p->d_V_nk, p->d_ib, p->d_ub,
                    p->d_cache, p->d_k2dyn, p->d_seg_off, p->d_seg_mid, p->d_xctl, p->d_se_x, p->d_level_off, p->d_k1_wave_off, p->d_k2_wave_off, p->d_se_partial, p->d_se_tmp, p->d_se_epoch, p->d_counter, p->d_U_start,
                    p->d_ib_start};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    res_destroy(p->res);
    res2_destroy(p->res2);
    delete p;
}

int mfx_sgd_plan_schedule_builder(const mfx_sgd_plan *p)
{
    if (!p) return MFX_E_ARG;
    return p->sched_device ? MFX_SCHED_DEVICE : MFX_SCHED_HOST;
}

int mfx_sgd_plan_info(const mfx_sgd_plan *p, int64_t info[8])
{
    if (!p || !info) return MFX_E_ARG;
    int64_t launches = 0, tails = 0, flaunches = 0;
    for (int64_t l = 0; l < p->n_levels; ++l)
        if (p->level_off[l + 1] > p->level_off[l]) ++flaunches;
    for (const LaunchSeg &sg : p->segs) { ++launches; if (sg.tail) ++tails; }
    info[0] = p->n_levels;
    info[1] = p->max_level;
    info[2] = launches;
    info[3] = tails;
    info[4] = p->batch;
    info[5] = (int64_t)p->dev_bytes;
    info[6] = p->sched_us;
    info[7] = (int64_t)p->segs2.size();
    (void)flaunches;
    return MFX_OK;
}

int mfx_sgd_plan_res_info(const mfx_sgd_plan *p, int64_t info[4])
{
    if (!p || !info) return MFX_E_ARG;
    const bool on = res_on(p);                                             // (bf16 rows: level kernels)
    info[0] = on ? res_kind(p->res) + 1 : MFX_EPOCH_LEVELS;          // (kind 1 = ticket, 2 = static)
    info[1] = p->res ? res_kind(p->res) + 1 : MFX_EPOCH_LEVELS;      // prepared (switched off by a failed registration?)
    info[2] = p->res ? (int64_t)res_dev_bytes(p->res) : 0;
    info[3] = p->res ? res_se_slots(p->res) : 0;
    return MFX_OK;
}

int mfx_sgd_plan_kernel_name(const mfx_sgd_plan *p, int mode, char *buf, int buflen)
{
    if (!p || !buf || buflen < 1 || mode < 0 || mode > 3) return fail(MFX_E_ARG, "mfx_sgd_plan_kernel_name: bad argument");
    if (mode == 3) {                      // the Funk pass: known once a Funk run has decided
        if (p->k2r_state == 1 && res2_box_width(p->res2) && (res2_hot_waves(p->res2) || res2_hot_user_waves(p->res2)))
            snprintf(buf, (size_t)buflen, "k2_res<lane = item, mailboxes, %d per poll, %d waves: %d own ONE hot item, %d ONE hot user>", res2_box_width(p->res2),
                     res2_waves(p->res2), res2_hot_waves(p->res2), res2_hot_user_waves(p->res2));
        else if (p->k2r_state == 1 && res2_box_width(p->res2))
            snprintf(buf, (size_t)buflen, "k2_res<lane = item, user values sent to the next rating's mailbox, %d per poll, %d waves>", res2_box_width(p->res2), (p->ni + 63) / 64);
        else if (p->k2r_state == 1) snprintf(buf, (size_t)buflen, "k2_res<lane = item, user values in 16-byte granules, %d waves>", (p->ni + 63) / 64);
        else if (p->k2x_state == 1) snprintf(buf, (size_t)buflen, "k2_xcd_pass<%d worker waves,%d user groups>", kXWorkers, kXGroups);
        else snprintf(buf, (size_t)buflen, "k2_level<%d threads>%s", kK2Block, p->k2x_state == 0 ? " (undecided: no Funk run yet)" : "");
        return MFX_OK;
    }
    if (mode < 2 && res_on(p)) return res_describe(p->res, mode, buf, buflen, res_variant(p));
    static const char *const mode_name[] = {"linear", "logistic", "learned"};
    snprintf(buf, (size_t)buflen, "k1_level<%s,%s%s>", k1_select<false>(p->k, mode, p->batch, p->f32, p->dot).tag,
             mode_name[mode], p->f32 == 1 ? ",f32" : "");
    return MFX_OK;
}

int mfx_sgd_plan_set_storage(mfx_sgd_plan *p, int bits)
{
    if (!p || (bits != 64 && bits != 32 && bits != 16)) return fail(MFX_E_ARG, "mfx_sgd_plan_set_storage: bits must be 64, 32 or 16");
    if (p->have_factors) return fail(MFX_E_STATE, "mfx_sgd_plan_set_storage: call before mfx_sgd_set_factors");
    if (bits == 16 && !k1_bf16_eligible(p->k))
        return fail(MFX_E_ARG, "mfx_sgd_plan_set_storage: bf16 rows need k a multiple of 8, k >= 16 (k = %d)", p->k);
    p->f32 = bits == 32 ? 1 : (bits == 16 ? 2 : 0);
    drop_graph(p);
    return MFX_OK;
}

int mfx_sgd_plan_set_dot(mfx_sgd_plan *p, int dot)
{
    if (!p || (dot != MFX_DOT_ORDERED && dot != MFX_DOT_BUTTERFLY)) return fail(MFX_E_ARG, "mfx_sgd_plan_set_dot: bad argument");
    if (dot == MFX_DOT_BUTTERFLY && !k1_dot_eligible(p->k, p->f32))
        return fail(MFX_E_ARG, "mfx_sgd_plan_set_dot: the butterfly dot needs f64 rows and an even k > 4 (k = %d)", p->k);
    if (p->dot != dot) drop_graph(p);
    p->dot = dot;
    return MFX_OK;
}

int mfx_sgd_set_factors(mfx_sgd_plan *p, const double *u, const double *v, const double *ib,
                        const double *ub)
{
    mfx::TraceRange trace_("mfx:set_factors (H2D + transpose)");
    if (!p || !u || !v) {
        return fail(MFX_E_ARG, "mfx_sgd_set_factors: NULL argument");
    }
    const size_t ku = (size_t)p->k * p->ni * sizeof(double), kv = (size_t)p->k * p->nu * sizeof(double);
    MFX_HIP(hipStreamSynchronize(stream()));
    MFX_HIP(hipMemcpy(p->d_U_km, u, ku, hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(p->d_V_km, v, kv, hipMemcpyHostToDevice));
    if (ib) MFX_HIP(hipMemcpy(p->d_ib, ib, (size_t)p->ni * sizeof(double), hipMemcpyHostToDevice));
    else MFX_HIP(hipMemset(p->d_ib, 0, (size_t)p->ni * sizeof(double)));
    if (ub) MFX_HIP(hipMemcpy(p->d_ub, ub, (size_t)p->nu * sizeof(double), hipMemcpyHostToDevice));
    else MFX_HIP(hipMemset(p->d_ub, 0, (size_t)p->nu * sizeof(double)));
    p->km_current = true;
    p->nk_current = false;
    p->have_factors = true;
    return MFX_OK;
}

int mfx_sgd_get_factors(mfx_sgd_plan *p, double *u, double *v, double *ib, double *ub)
{
    mfx::TraceRange trace_("mfx:get_factors (transpose + D2H)");
    int rc = check_plan(p, "mfx_sgd_get_factors");
    if (rc) return rc;
    rc = to_km(p);
    if (rc) return rc;
    MFX_HIP(hipStreamSynchronize(stream()));
    if (u) MFX_HIP(hipMemcpy(u, p->d_U_km, (size_t)p->k * p->ni * sizeof(double), hipMemcpyDeviceToHost));
    if (v) MFX_HIP(hipMemcpy(v, p->d_V_km, (size_t)p->k * p->nu * sizeof(double), hipMemcpyDeviceToHost));
    if (ib) MFX_HIP(hipMemcpy(ib, p->d_ib, (size_t)p->ni * sizeof(double), hipMemcpyDeviceToHost));
    if (ub) MFX_HIP(hipMemcpy(ub, p->d_ub, (size_t)p->nu * sizeof(double), hipMemcpyDeviceToHost));
    return MFX_OK;
}

int mfx_sgd_run_kmf(mfx_sgd_plan *p, int mode, int nbr_epochs, double lr, double Ku, double Ki,
                    double Kb, int upd_u, int upd_i, double *rmse_out)
{
    mfx::TraceRange trace_("mfx:run_kmf (full-k SGD epochs)");
    int rc = check_plan(p, "mfx_sgd_run_kmf");
    if (rc) return rc;
    if (mode != MFX_MODE_LINEAR && mode != MFX_MODE_LOGISTIC) {
        return fail(MFX_E_ARG, "mfx_sgd_run_kmf: mode %d", mode);
    }
    if (nbr_epochs <= 0) return MFX_OK;
    K1Params P = k1_params(p, lr, Ku, Ki, Kb, upd_u, upd_i);
    rc = run_k1_epochs(p, mode, nbr_epochs, P, [](int) { return MFX_OK; }, true, true);
    if (rc) return rc;
    return fetch_rmse(p, nbr_epochs, rmse_out, nullptr);
}

int mfx_sgd_profile_kmf(mfx_sgd_plan *p, int mode, double lr, double Ku, double Ki, double Kb,
                        double *kernel_ms, int64_t *n_launches)
{
    int rc = check_plan(p, "mfx_sgd_profile_kmf");
    if (rc) return rc;
    rc = ensure_se_cap(p, 1);
    if (!rc) rc = to_nk(p);
    if (rc) return rc;
    p->km_current = false;
    K1Params P = k1_params(p, lr, Ku, Ki, Kb, 1, 1);
    const int64_t launches = (int64_t)p->segs.size();
    std::vector<hipEvent_t> ev((size_t)launches * 2, nullptr);
    struct EvGuard {
        std::vector<hipEvent_t> &v;
        ~EvGuard() { for (auto &e : v) if (e) (void)hipEventDestroy(e); }
    } guard{ev};
    for (auto &e : ev) MFX_HIP(hipEventCreate(&e));
    MFX_HIP(hipMemsetAsync(p->d_counter, 0, sizeof(int), stream()));
    rc = enqueue_k1_epoch(p, mode, P, &ev);
    if (rc) return rc;
    MFX_HIP(hipStreamSynchronize(stream()));
    double tot = 0.0;
    for (int64_t i = 0; i < launches; ++i) {
        float ms = 0.f;
        MFX_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms;
    }
    if (kernel_ms) *kernel_ms = tot;
    if (n_launches) *n_launches = launches;
    return MFX_OK;
}

// ---- Funk: per-feature passes with the data-dependent stopping rule (host-driven) ----
static int enqueue_k2_epoch(mfx_sgd_plan *p, const K2Params &P)
{
    for (const LaunchSeg &sg : p->segs2) {
        if (!sg.tail) {
            const int64_t start = p->level_off[sg.l0];
            const int count = (int)(p->level_off[sg.l0 + 1] - start);
            if (count <= kK2WriteThroughMax)
                hipLaunchKernelGGL(k2_level<true>, dim3(ceil_div(count, kK2Block)), dim3(kK2Block), 0, stream(), P,
                                   start, count, p->k2_wave_off[sg.l0]);
            else
                hipLaunchKernelGGL(k2_level<false>, dim3(ceil_div(count, kK2Block)), dim3(kK2Block), 0, stream(), P,
                                   start, count, p->k2_wave_off[sg.l0]);
        } else {
            hipLaunchKernelGGL(k2_tail, dim3(1), dim3(kTail2), 0, stream(), P,
                               (const int64_t *)p->d_level_off, (const int64_t *)p->d_k2_wave_off,
                               (int)sg.l0, (int)sg.l1);
        }
    }
    MFX_HIP(hipGetLastError());
    return enqueue_reduce(p, p->k2_waves);
}

}  // extern "C"

namespace mfx {
// (FunkHooks: mfx_sgd_internal.h -- snapshot / combine the item-side feature vector around every
// epoch and make the squared error global, so that every rank takes the same decision in the
// data-dependent stopping rule)

// Decide once per plan whether the XCD-resident pass applies, and prepare it: ratings re-ordered inside
// their levels by (user group, item), segment table, control block, sparse squared-error slots.
static int ensure_k2x(mfx_sgd_plan *p)
{
    if (p->k2x_state != 0) return MFX_OK;
    p->k2x_state = -1;
    const int mode = env_int("MFX_K2_XCD", -1);          // 0 = off, 1 = forced, default: by size
    if (mode == 0 || p->nnz <= 0 || p->n_levels <= 0) return MFX_OK;
    if (p->nnz >= ((int64_t)1 << 31) - 1 || p->n_levels >= ((int64_t)1 << 24) / 2) return MFX_OK;
    if (mode != 1 && p->nnz / p->n_levels < 4096) return MFX_OK;    // small levels: the tail kernel's territory
    // LDS: a workgroup holds one user group (32 workgroups on the XCD) -- room for two when it fits, so
    // that an XCD that received fewer workgroups still works
    const int64_t per_group = ceil_div((int64_t)p->nu, (int64_t)kXGroups) + 1;
    const int64_t cap_doubles = (150 * 1024) / 8;
    if (per_group > cap_doubles) return MFX_OK;
    p->x_lds_doubles = (int)(2 * per_group <= cap_doubles ? 2 * per_group : per_group);
    const int64_t nseg = p->n_levels * kXGroups;
    p->k2x_slots = (int64_t)kXGroups * kXWorkers * 64;
    if (dev_alloc(p, &p->d_seg_off, (size_t)nseg + 1) || dev_alloc(p, &p->d_seg_mid, (size_t)nseg + 1) || dev_alloc(p, &p->d_xctl, 4096) ||
        dev_alloc(p, &p->d_se_x, (size_t)p->k2x_slots))
        return last_code();
    MFX_HIP(hipMemsetAsync(p->d_xctl, 0, 4096 * sizeof(unsigned), stream()));
    MFX_HIP(hipMemsetAsync(p->d_se_x, 0, (size_t)p->k2x_slots * sizeof(double), stream()));
    int rc = levels_group_sort(p->d_recs, p->nnz, p->level_off, p->d_level_off, p->n_levels, p->nu, p->ni, kXGroups,
                               p->d_seg_off, p->d_seg_mid);
    if (rc) return rc;
    const size_t lds = (size_t)p->x_lds_doubles * sizeof(double);
    if (lds > 64 * 1024)
        MFX_HIP(hipFuncSetAttribute((const void *)k2_xcd_pass, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    p->k2x_state = 1;
    return MFX_OK;
}

static int enqueue_k2x_pass(mfx_sgd_plan *p, const K2Params &P)
{
    K2XParams X;
    memset(&X, 0, sizeof(X));
    X.recs = P.recs; X.dyn = P.dyn; X.cache = P.cache; X.ib = P.ib; X.ub = P.ub;
    X.seg_off = p->d_seg_off; X.seg_mid = p->d_seg_mid; X.se_partial = p->d_se_x; X.ctl = p->d_xctl;
    X.n_levels = (int)p->n_levels; X.nu = p->nu; X.lds_doubles = p->x_lds_doubles;
    X.gen_base = p->x_gen;
    X.tune = env_int("MFX_K2_TUNE", 0);
    X.lr = P.lr; X.K = P.K; X.mu = P.mu; X.upd_u = P.upd_u; X.upd_i = P.upd_i;
    p->x_gen += (unsigned)p->n_levels + 1u;
    // the registration counters are cleared per pass (the status word per batch, by the caller); the flags
    // keep counting
    MFX_HIP(hipMemsetAsync(p->d_xctl, 0, 256 * sizeof(unsigned), stream()));
    void *args[] = {&X};
    // one workgroup per CU: 256 launched, those on the chosen XCD stay
    MFX_HIP(hipLaunchKernel((const void *)k2_xcd_pass, dim3(256), dim3((kXWorkers + 1) * 64), args,
                            (size_t)p->x_lds_doubles * sizeof(double), stream()));
    return enqueue_reduce(p, p->k2x_slots, p->d_se_x);
}

// ---- the Funk run of ONE plan, in the four steps funk_core and funk_blocks_core are made of ------------------------------------
struct FunkRun {
    mfx_sgd_plan *p = nullptr;
    const FunkHooks *hk = nullptr;
    K2Params P;
    K2Dyn dyn;
    bool k2r = false, graph = false;
    int commit_blocks = 1;
};

// everything in front of the feature loop: layout, residual cache, the choice of the pass kernel, the captured epoch
static int funk_setup(FunkRun &R, mfx_sgd_plan *p, double lr, double K, double mu, int use_bias, int upd_u, int upd_i, const FunkHooks *hk,
                      bool users_km_done = false)
{
    R.p = p;
    R.hk = hk;
    int rc = check_plan(p, "mfx_sgd_run_funk");
    if (rc) return rc;
    if (users_km_done && !p->km_current) {                    // (a block that shares its user side with a plan already converted)
        rc = transpose((const double *)p->d_U_nk, p->d_U_km, p->ni, p->k);
        if (rc) return rc;
        p->km_current = true;
    }
    rc = to_km(p);
    if (rc) return rc;
    p->nk_current = false;
    if (!p->d_cache) {
        rc = dev_alloc(p, &p->d_cache, (size_t)p->nnz);
        if (!rc) rc = dev_alloc(p, &p->d_k2dyn, 1);
        if (rc) return rc;
    }
    // Which kernel runs a pass, decided once per plan: the resident-side pass (mfx_sgd_res.hip: every CU, a hop = one granule
    // round trip) where every item gets a resident lane; else the single-XCD persistent pass (large levels); else one launch
    // per level.  Both persistent forms check their residency first and leave everything untouched when it fails.
    if (p->k2r_state == 0 && !hk) {
        p->k2r_state = -1;
        const int want = env_int("MFX_K2_RES", -1);          // 0 = off, 1 = forced (also tiny problems), default: by size
        // measured at cfg 2, mailbox form: shuffled order 0.98 ms per pass against 1.35 on the single-XCD pass,
        // balanced order (levels of ~37 K) 0.76 against 1.03 -- the resident-side pass wherever every item gets a lane
        if (want != 0) {
            if (res2_prepare(&p->res2, p->d_recs, p->nnz, p->nu, p->ni, want == 1) != MFX_OK) (void)hipGetLastError();
            if (p->res2) { p->k2r_state = 1; p->dev_bytes += res2_dev_bytes(p->res2); }
        }
    }
    R.k2r = !hk && p->k2r_state == 1;
    // may re-order the ratings inside their levels: before the cache is used.  The XCD-resident pass is an optional
    // acceleration: if its preparation fails (scratch memory, attribute), the records are either untouched or fully
    // re-ordered inside their levels -- both are valid inputs of the launch path -- so the run goes on there
    if (!R.k2r && ensure_k2x(p) != MFX_OK) {
        (void)hipGetLastError();
        p->k2x_state = -1;
    }
    MFX_HIP(hipMemsetAsync(p->d_cache, 0, (size_t)(p->nnz ? p->nnz : 1) * sizeof(double), stream()));
    if (R.k2r && (rc = res2_reset_cache(p->res2))) return rc;
    K2Params &P = R.P;
    memset(&P, 0, sizeof(P));
    P.recs = p->d_recs;
    P.dyn = p->d_k2dyn;
    P.cache = p->d_cache;
    P.ib = use_bias ? p->d_ib : nullptr;
    P.ub = use_bias ? p->d_ub : nullptr;
    P.se_partial = p->d_se_partial;
    P.lr = lr;
    P.K = K;
    P.mu = mu;
    P.upd_u = upd_u ? 1 : 0;
    P.upd_i = upd_i ? 1 : 0;
    R.commit_blocks = (int)(p->nnz ? (ceil_div(p->nnz, (int64_t)256) < 4096 ? ceil_div(p->nnz, (int64_t)256) : 4096) : 1);

    // one epoch = all levels + the SE reduction, captured once and replayed for every
    // (feature, epoch): the per-feature pointers live behind P.dyn
    R.graph = use_graph() && p->segs2.size() <= kMaxGraphNodes;
    if (R.graph && !R.k2r && !(p->graph2 && memcmp(&p->graph2_params, &P, sizeof(P)) == 0)) {
        if (p->graph2) (void)hipGraphExecDestroy(p->graph2);
        p->graph2 = nullptr;
        hipGraph_t g = nullptr;
        MFX_HIP(hipStreamBeginCapture(stream(), hipStreamCaptureModeThreadLocal));
        rc = enqueue_k2_epoch(p, P);
        hipError_t e = hipStreamEndCapture(stream(), &g);
        if (rc || e != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            return rc ? rc : fail(MFX_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        }
        e = hipGraphInstantiate(&p->graph2, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) {
            p->graph2 = nullptr;
            return fail(MFX_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        }
        p->graph2_params = P;
    }
    return MFX_OK;
}

// the per-feature pointers behind P.dyn (enqueue + sync: `dyn` is a host object)
static int funk_set_feature(FunkRun &R, int f, double f_init)
{
    mfx_sgd_plan *p = R.p;
    R.dyn.uf = p->d_U_km + (size_t)f * p->ni;
    R.dyn.vf = p->d_V_km + (size_t)f * p->nu;
    R.dyn.trail = (p->k - f - 1) * f_init * f_init;        // gd_estimator.pyx:70
    MFX_HIP(hipMemcpyAsync(p->d_k2dyn, &R.dyn, sizeof(R.dyn), hipMemcpyHostToDevice, stream()));
    MFX_HIP(hipStreamSynchronize(stream()));
    return MFX_OK;
}

// `batch` (<= 32) passes of the current feature, enqueued back to back; their squared errors in se_b.  A persistent form that loses its
// residency check has modified nothing from that pass on: the plan moves to the next form for good and the rest of the batch is redone.
static int funk_passes(FunkRun &R, int batch, double *se_b)
{
    mfx_sgd_plan *p = R.p;
    const FunkHooks *hk = R.hk;
    K2Params &P = R.P;
    int rc;
    int b0 = 0;                 // passes of this batch already applied (by the resident-side pass, before a fall-back)
    for (;;) {
        if (R.k2r) {
            // ---- resident-side pass: granules, one persistent launch over every CU, values back, reduction ----
            MFX_HIP(hipMemsetAsync(p->d_counter, 0, sizeof(int), stream()));
            if ((rc = res2_begin_batch(p->res2))) return rc;
            for (int b = 0; b < batch; ++b) {
                if ((rc = res2_enqueue_pass(p->res2, p->d_k2dyn, P.ib, P.ub, P.mu, P.lr, P.K, P.upd_u, P.upd_i))) return rc;
                if ((rc = enqueue_reduce(p, res2_se_slots(p->res2), res2_se_partials(p->res2)))) return rc;
            }
            MFX_HIP(hipMemcpyAsync(se_b, p->d_se_epoch, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost, stream()));
            MFX_HIP(hipStreamSynchronize(stream()));
            unsigned status = 0, applied = 0;
            if ((rc = res2_status(p->res2, &status, &applied))) return rc;
            if (status == 0) break;
            if (status != 1) return fail(MFX_E_HIP, "mfx_sgd_run_funk: the resident-side pass timed out waiting for a granule");
            // a pass found its grid not co-resident: the first `applied` passes of the batch ran in full, that one and the
            // later ones did NOTHING.  The residual cache moves to the level-major layout and the run goes on, for good, with
            // one launch per level -- from the first pass that was not applied.
            if ((rc = res2_cache_to_levels(p->res2, p->d_cache))) return rc;
            p->k2r_state = -1;
            R.k2r = false;
            if (p->k2x_state == 0) p->k2x_state = -1;       // (its preparation would re-order records the cache now follows)
            b0 = (int)applied < batch ? (int)applied : batch;
            if (b0 == batch) break;
            continue;
        }
        const bool xcd = p->k2x_state == 1;
        MFX_HIP(hipMemcpyAsync(p->d_counter, &b0, sizeof(int), hipMemcpyHostToDevice, stream()));   // squared errors b0 .. batch-1
        if (xcd) MFX_HIP(hipMemsetAsync(p->d_xctl + 512, 0, sizeof(unsigned), stream()));   // status: sticky inside a batch
        for (int b = b0; b < batch; ++b) {
            if (hk && (rc = hk->pre(hk->ctx, R.dyn.uf, p->ni))) return rc;
            if (xcd) rc = enqueue_k2x_pass(p, P);
            else if (R.graph && p->graph2) {
                MFX_HIP(hipGraphLaunch(p->graph2, stream()));
                rc = MFX_OK;
            } else rc = enqueue_k2_epoch(p, P);
            if (rc) return rc;
        }
        unsigned status = 0;
        if (xcd) MFX_HIP(hipMemcpyAsync(&status, p->d_xctl + 512, sizeof(unsigned), hipMemcpyDeviceToHost, stream()));
        if (!xcd && hk && (rc = hk->post(hk->ctx, R.dyn.uf, p->ni))) return rc;
        MFX_HIP(hipMemcpyAsync(se_b, p->d_se_epoch, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MFX_HIP(hipStreamSynchronize(stream()));
        if (!xcd) break;
        if (status == 0) {
            if (hk) {
                if ((rc = hk->post(hk->ctx, R.dyn.uf, p->ni))) return rc;
                MFX_HIP(hipStreamSynchronize(stream()));
            }
            break;
        }
        if (status == 3)
            return fail(MFX_E_HIP, "mfx_sgd_run_funk: the XCD-resident pass lost a workgroup mid-pass");
        // registration failed or LDS too small: NOTHING was modified (the status is sticky: the later
        // passes of the batch left at once) -- redo the batch on the launch path, and stay there
        p->k2x_state = -1;
    }
    return MFX_OK;
}

// the cache-commit pass of the finished feature (gd_estimator.pyx:771-777)
static int funk_commit(FunkRun &R)
{
    mfx_sgd_plan *p = R.p;
    if (R.k2r) return res2_enqueue_commit(p->res2, p->d_k2dyn, R.P.ib, R.P.ub, R.P.mu);
    hipLaunchKernelGGL(k2_commit, dim3(R.commit_blocks), dim3(256), 0, stream(), R.P, p->nnz);
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

int funk_core(mfx_sgd_plan *p, int min_epochs, double min_improvement, double f_init, double lr,
              double K, double mu, int use_bias, int upd_u, int upd_i, int epoch_cap,
              double *rmse_hist, int64_t hist_cap, int64_t *n_hist, const FunkHooks *hk)
{
    FunkRun R;
    int rc = funk_setup(R, p, lr, K, mu, use_bias, upd_u, upd_i, hk);
    if (rc) return rc;
    double rmse = 2.0, rmse_last = 0.0;
    int64_t nh = 0;
    for (int f = 0; f < p->k; ++f) {
        if ((rc = funk_set_feature(R, f, f_init))) return rc;
        int epoch = 0;
        while (epoch < min_epochs || rmse <= rmse_last - min_improvement) {
            if (epoch_cap > 0 && epoch >= epoch_cap) break;
            // Epochs below min_epochs run whatever their rmse turns out to be: they are enqueued back to
            // back and their squared errors read together (one host round trip per batch, not per epoch).
            // The sharded mode (hooks) combines and decides after every epoch.
            int batch = 1;
            if (!hk && epoch < min_epochs) {
                batch = min_epochs - epoch;
                if (epoch_cap > 0 && batch > epoch_cap - epoch) batch = epoch_cap - epoch;
                if (batch > 32) batch = 32;
            }
            double se_b[32];
            double nn = (double)p->nnz;
            if ((rc = funk_passes(R, batch, se_b))) return rc;
            for (int b = 0; b < batch; ++b) {
                double se = se_b[b];
                if (hk && (rc = hk->global_se(hk->ctx, &se, &nn))) return rc;
                rmse_last = rmse;
                rmse = sqrt(se / nn);
                if (rmse_hist && nh < hist_cap) rmse_hist[nh] = rmse;
                ++nh;
                ++epoch;
            }
        }
        if ((rc = funk_commit(R))) return rc;
    }
    MFX_HIP(hipStreamSynchronize(stream()));
    if (n_hist) *n_hist = nh;
    return MFX_OK;
}

// The Funk run on an ITEM-BLOCK order (round 5): plans[j] holds the ratings of item block j (item ids local to the block) and all of
// them share ONE user side (plan_alias_users: mfx_dsgd_create with world = 1).  A pass of feature f is the sequential loop over
// block 0's ratings, then block 1's, ...: block j's pass reads and writes the SAME V[f,:] the block before left -- i.e. the
// reference's loop (gd_estimator.pyx:691-779) on the order "blocks one after the other, every block in its own order", with ONE
// stopping rule over the whole pass (the squared errors of the blocks are added in block order).  Every block is small enough for the
// resident-side pass (a lane per item: k2_res_box) where the whole catalogue is not (more than ~450 K items: BASELINE cfg 5's 1 M).
int funk_blocks_core(mfx_sgd_plan *const *plans, int B, int min_epochs, double min_improvement, double f_init, double lr, double K,
                     double mu, int use_bias, int upd_u, int upd_i, int epoch_cap, double *rmse_hist, int64_t hist_cap, int64_t *n_hist)
{
    if (!plans || B < 1) return fail(MFX_E_ARG, "mfx_dsgd_run_funk: no blocks");
    for (int j = 0; j < B; ++j) {
        int rc = check_plan(plans[j], "mfx_dsgd_run_funk");
        if (rc) return rc;
        if (plans[j]->k != plans[0]->k || plans[j]->nu != plans[0]->nu || plans[j]->f32 ||
            (j > 0 && (plans[j]->d_V_km != plans[0]->d_V_km || plans[j]->d_V_nk != plans[0]->d_V_nk)))
            return fail(MFX_E_ARG, "mfx_dsgd_run_funk: the blocks must share one f64 user side");
    }
    std::vector<FunkRun> R((size_t)B);
    int rc;
    int64_t nnz = 0;
    for (int j = 0; j < B; ++j) {
        // the shared user side is converted to the feature-major layout ONCE, by block 0
        if ((rc = funk_setup(R[(size_t)j], plans[j], lr, K, mu, use_bias, upd_u, upd_i, nullptr, j > 0))) return rc;
        nnz += plans[j]->nnz;
    }
    const int k = plans[0]->k;
    double rmse = 2.0, rmse_last = 0.0;
    int64_t nh = 0;
    for (int f = 0; f < k; ++f) {
        for (int j = 0; j < B; ++j)
            if ((rc = funk_set_feature(R[(size_t)j], f, f_init))) return rc;
        int epoch = 0;
        while (epoch < min_epochs || rmse <= rmse_last - min_improvement) {
            if (epoch_cap > 0 && epoch >= epoch_cap) break;
            double se = 0.0;
            for (int j = 0; j < B; ++j) {                     // one pass = the blocks one after the other on the library stream
                double sj = 0.0;
                if ((rc = funk_passes(R[(size_t)j], 1, &sj))) return rc;
                se += sj;
            }
            rmse_last = rmse;
            rmse = sqrt(se / (double)nnz);
            if (rmse_hist && nh < hist_cap) rmse_hist[nh] = rmse;
            ++nh;
            ++epoch;
        }
        for (int j = 0; j < B; ++j)
            if ((rc = funk_commit(R[(size_t)j]))) return rc;
    }
    // back to the row-major copies the block object keeps live (plan_mark_ready): every block's items, the shared user side once
    for (int j = 0; j < B; ++j) {
        if ((rc = transpose(plans[j]->d_U_km, plans[j]->d_U_nk, k, plans[j]->ni))) return rc;
        plans[j]->nk_current = true;
        plans[j]->km_current = false;
    }
    if ((rc = transpose(plans[0]->d_V_km, plans[0]->d_V_nk, k, plans[0]->nu))) return rc;
    MFX_HIP(hipStreamSynchronize(stream()));
    if (n_hist) *n_hist = nh;
    return MFX_OK;
}

}  // namespace mfx

extern "C" {

int mfx_sgd_run_funk(mfx_sgd_plan *p, int min_epochs, double min_improvement, double f_init,
                     double lr, double K, double mu, int use_bias, int upd_u, int upd_i,
                     int epoch_cap, double *rmse_hist, int64_t hist_cap, int64_t *n_hist)
{
    mfx::TraceRange trace_("mfx:run_funk (per-feature SGD)");
    return mfx::funk_core(p, min_epochs, min_improvement, f_init, lr, K, mu, use_bias, upd_u, upd_i,
                          epoch_cap, rmse_hist, hist_cap, n_hist, nullptr);
}

int mfx_sgd_run_learned_bias(mfx_sgd_plan *p, int min_epochs, double min_improvement,
                             double f_init, double lr, double lr_users, double lr_items,
                             double K_feature, double K_bias, double mu, int epoch_cap,
                             double *rmse_hist, int64_t hist_cap, int64_t *n_hist)
{
    mfx::TraceRange trace_("mfx:run_learned_bias");
    int rc = check_plan(p, "mfx_sgd_run_learned_bias");
    if (rc) return rc;
    if (p->f32 == 2) return fail(MFX_E_STATE, "mfx_sgd_run_learned_bias: bf16 row storage covers the linear / logistic kernels only");
    rc = to_nk(p);
    if (rc) return rc;
    p->km_current = false;
    double rmse = 2.0, rmse_last = 0.0;
    int64_t nh = 0;
    K1Params P = k1_params(p, lr, K_feature, K_feature, K_bias, 1, 1);
    P.lr_u = lr_users;
    P.lr_i = lr_items;
    P.mu = mu;
    for (int f = 0; f < p->k; ++f) {
        P.f = f;
        P.trail = (p->k - f - 1) * f_init * f_init;
        int epoch = 0;
        while (epoch < min_epochs || rmse <= rmse_last - min_improvement) {
            if (epoch_cap > 0 && epoch >= epoch_cap) break;
            rmse_last = rmse;
            MFX_HIP(hipMemsetAsync(p->d_counter, 0, sizeof(int), stream()));
            rc = enqueue_k1_epoch(p, K1_LEARNED, P, nullptr);
            if (rc) return rc;
            double se = 0.0;
            MFX_HIP(hipMemcpyAsync(&se, p->d_se_epoch, sizeof(double), hipMemcpyDeviceToHost, stream()));
            MFX_HIP(hipStreamSynchronize(stream()));
            rmse = sqrt(se / (double)p->nnz);
            if (rmse_hist && nh < hist_cap) rmse_hist[nh] = rmse;
            ++nh;
            ++epoch;
        }
    }
    if (n_hist) *n_hist = nh;
    return MFX_OK;
}

// ------------------------------------------------------------------------------------
// one-shot drop-ins: create plan, upload, run, download, destroy
// ------------------------------------------------------------------------------------
int mfx_kmf_train(int mode, int nbr_epochs, int dim, double lr, double Ku, double Ki, double Kb,
                  double *u, double *v, const int32_t *idx, const double *r, double *ib, double *ub,
                  int64_t nnz, int nu, int ni, int upd_u, int upd_i, double *rmse_out)
{
    if (!u || !v || !ib || !ub || (nnz > 0 && (!idx || !r))) {
        return fail(MFX_E_ARG, "mfx_kmf_train: NULL argument");
    }
    if (nnz == 0 || nbr_epochs <= 0) {
        // the reference divides 0/0 for the rmse and changes nothing
        if (rmse_out) for (int e = 0; e < nbr_epochs; ++e) rmse_out[e] = NAN;
        return MFX_OK;
    }
    // (the resident-side preparation pays for itself over a few epochs with both sides trained)
    mfx_sgd_plan *p = mfx::plan_create_ex(idx, r, nnz, nu, ni, dim, MFX_SCHED_AUTO, (upd_u && upd_i && nbr_epochs >= 3) ? MFX_EPOCH_AUTO : MFX_EPOCH_LEVELS);
    if (!p) return last_code();
    int rc = mfx_sgd_set_factors(p, u, v, ib, ub);
    if (!rc) rc = mfx_sgd_run_kmf(p, mode, nbr_epochs, lr, Ku, Ki, Kb, upd_u, upd_i, rmse_out);
    if (!rc) rc = mfx_sgd_get_factors(p, u, v, ib, ub);
    mfx_sgd_plan_destroy(p);
    return rc;
}

// No ratings: the reference's loops run min_epochs empty epochs per feature (rmse = sqrt(0/0) is NaN and
// `NaN <= x` is false, gd_estimator.pyx:742) and leave every array as it was -- e.g. retrain_user for
// a user without ratings (gradient_descent.py:879-905) is a no-op, not an error.
static int empty_history(int min_epochs, int dim, int epoch_cap, double *rmse_hist, int64_t hist_cap,
                         int64_t *n_hist)
{
    int per_feature = min_epochs > 0 ? min_epochs : 0;
    if (epoch_cap > 0 && per_feature > epoch_cap) per_feature = epoch_cap;
    const int64_t n = (int64_t)per_feature * (dim > 0 ? dim : 0);
    if (rmse_hist)
        for (int64_t i = 0; i < n && i < hist_cap; ++i) rmse_hist[i] = NAN;
    if (n_hist) *n_hist = n;
    return MFX_OK;
}

int mfx_funk(int min_epochs, double min_improvement, int dim, double f_init, double lr, double K,
             double mu, double *u, double *v, const int32_t *idx, const double *r, const double *ib,
             const double *ub, int64_t nnz, int nu, int ni, int upd_u, int upd_i, int epoch_cap,
             double *rmse_hist, int64_t hist_cap, int64_t *n_hist)
{
    if (!u || !v || (nnz > 0 && (!idx || !r)) || ((ib == nullptr) != (ub == nullptr))) {
        return fail(MFX_E_ARG, "mfx_funk: NULL argument");
    }
    if (n_hist) *n_hist = 0;
    if (nnz == 0) return empty_history(min_epochs, dim, epoch_cap, rmse_hist, hist_cap, n_hist);
    mfx_sgd_plan *p = mfx::plan_create_ex(idx, r, nnz, nu, ni, dim, MFX_SCHED_AUTO, MFX_EPOCH_LEVELS);
    if (!p) return last_code();
    int rc = mfx_sgd_set_factors(p, u, v, ib, ub);
    if (!rc) rc = mfx_sgd_run_funk(p, min_epochs, min_improvement, f_init, lr, K, mu, ib != nullptr,
                                   upd_u, upd_i, epoch_cap, rmse_hist, hist_cap, n_hist);
    if (!rc) rc = mfx_sgd_get_factors(p, u, v, nullptr, nullptr);
    mfx_sgd_plan_destroy(p);
    return rc;
}

int mfx_learned_bias(int min_epochs, double min_improvement, int dim, double f_init, double lr,
                     double lr_users, double lr_items, double K_feature, double K_bias, double mu,
                     double *u, double *v, const int32_t *idx, const double *r, double *ib,
                     double *ub, int64_t nnz, int nu, int ni, int epoch_cap, double *rmse_hist,
                     int64_t hist_cap, int64_t *n_hist)
{
    if (!u || !v || !ib || !ub || (nnz > 0 && (!idx || !r))) {
        return fail(MFX_E_ARG, "mfx_learned_bias: NULL argument");
    }
    if (n_hist) *n_hist = 0;
    if (nnz == 0) return empty_history(min_epochs, dim, epoch_cap, rmse_hist, hist_cap, n_hist);
    mfx_sgd_plan *p = mfx::plan_create_ex(idx, r, nnz, nu, ni, dim, MFX_SCHED_AUTO, MFX_EPOCH_LEVELS);
    if (!p) return last_code();
    int rc = mfx_sgd_set_factors(p, u, v, ib, ub);
    if (!rc) rc = mfx_sgd_run_learned_bias(p, min_epochs, min_improvement, f_init, lr, lr_users,
                                           lr_items, K_feature, K_bias, mu, epoch_cap, rmse_hist,
                                           hist_cap, n_hist);
    if (!rc) rc = mfx_sgd_get_factors(p, u, v, ib, ub);
    mfx_sgd_plan_destroy(p);
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// Dense-cache members of the Funk family: estimator_loop (gd_estimator.pyx:210-303),
// estimator_loop2 (:308-395), estimator_subloop (:903-962), predictor_subloop (:967-995).
// The reference indexes an f64[nu*ni] cache by user + item*nu, so ratings with the same
// (user, item) share a cell.  Here the cell of a pair is the device cache slot of its first
// occurrence (no O(nu*ni) device memory); everything else is the K2 machinery.
// ------------------------------------------------------------------------------------
namespace {

struct DenseCtx {
    mfx_sgd_plan *p = nullptr;
    std::vector<Rec> recs;          // level-major copy of the device records
    std::vector<int32_t> slot;      // canonical position per position
    int32_t *d_slot = nullptr;
    K2Params P{};
    ~DenseCtx()
    {
        if (d_slot) (void)hipFree(d_slot);
        if (p) mfx_sgd_plan_destroy(p);
    }
};

int dense_setup(DenseCtx &c, double *u, double *v, const int32_t *idx, const double *r, int64_t nnz,
                int nu, int ni, int dim, double lr, double K)
{
    if (nnz > 0x7fffffff) return fail(MFX_E_ARG, "dense-cache variants: nnz too large");
    c.p = mfx::plan_create_ex(idx, r, nnz, nu, ni, dim, MFX_SCHED_AUTO, MFX_EPOCH_LEVELS);
    if (!c.p) return last_code();
    mfx_sgd_plan *p = c.p;
    int rc = mfx_sgd_set_factors(p, u, v, nullptr, nullptr);
    if (rc) return rc;
    rc = dev_alloc(p, &p->d_cache, (size_t)nnz);
    if (!rc) rc = dev_alloc(p, &p->d_k2dyn, 1);
    if (rc) return rc;
    c.recs.resize((size_t)nnz);
    c.slot.resize((size_t)nnz);
    MFX_HIP(hipMemcpy(c.recs.data(), p->d_recs, (size_t)nnz * sizeof(Rec), hipMemcpyDeviceToHost));
    std::unordered_map<int64_t, int32_t> first;
    first.reserve((size_t)nnz * 2);
    for (int64_t j = 0; j < nnz; ++j) {
        const int64_t key = (int64_t)c.recs[j].user + (int64_t)c.recs[j].item * nu;
        auto it = first.find(key);
        if (it == first.end()) {
            first.emplace(key, (int32_t)j);
            c.slot[j] = (int32_t)j;
        } else {
            c.slot[j] = it->second;
        }
    }
    if (hipMalloc((void **)&c.d_slot, (size_t)nnz * sizeof(int32_t)) != hipSuccess)
        return fail(MFX_E_NOMEM, "dense-cache variants: hipMalloc failed");
    MFX_HIP(hipMemcpy(c.d_slot, c.slot.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    MFX_HIP(hipMemset(p->d_cache, 0, (size_t)nnz * sizeof(double)));
    K2Params &P = c.P;
    memset(&P, 0, sizeof(P));
    P.recs = p->d_recs;
    P.dyn = p->d_k2dyn;
    P.cache = p->d_cache;
    P.slot = c.d_slot;
    P.se_partial = p->d_se_partial;
    P.lr = lr;
    P.K = K;
    P.mu = 1.0;                      // estimator() default overall_avg, gd_estimator.pyx:47
    P.upd_u = P.upd_i = 1;
    rc = to_km(p);
    if (rc) return rc;
    p->nk_current = false;
    return MFX_OK;
}

int dense_set_feature(DenseCtx &c, int f, double f_init)
{
    mfx_sgd_plan *p = c.p;
    K2Dyn dyn;
    dyn.uf = p->d_U_km + (size_t)f * p->ni;
    dyn.vf = p->d_V_km + (size_t)f * p->nu;
    dyn.trail = (p->k - f - 1) * f_init * f_init;
    MFX_HIP(hipMemcpy(p->d_k2dyn, &dyn, sizeof(dyn), hipMemcpyHostToDevice));
    return MFX_OK;
}

int dense_epoch(DenseCtx &c, double *rmse)
{
    mfx_sgd_plan *p = c.p;
    MFX_HIP(hipMemsetAsync(p->d_counter, 0, sizeof(int), stream()));
    int rc = enqueue_k2_epoch(p, c.P);
    if (rc) return rc;
    double se = 0.0;
    MFX_HIP(hipMemcpyAsync(&se, p->d_se_epoch, sizeof(double), hipMemcpyDeviceToHost, stream()));
    MFX_HIP(hipStreamSynchronize(stream()));
    *rmse = sqrt(se / (double)p->nnz);
    return MFX_OK;
}

int dense_commit(DenseCtx &c)
{
    mfx_sgd_plan *p = c.p;
    for (int64_t l = 0; l < p->n_levels; ++l) {
        const int64_t start = p->level_off[l];
        const int count = (int)(p->level_off[l + 1] - start);
        if (count <= 0) continue;
        hipLaunchKernelGGL(k2_commit_level, dim3(ceil_div(count, 256)), dim3(256), 0, stream(), c.P,
                           start, count);
    }
    MFX_HIP(hipGetLastError());
    return MFX_OK;
}

// caller-owned dense cache <-> device slots
int dense_cache_in(DenseCtx &c, const double *dense, int nu)
{
    std::vector<double> h(c.recs.size(), 0.0);
    for (size_t j = 0; j < c.recs.size(); ++j)
        if (c.slot[j] == (int32_t)j) h[j] = dense[(size_t)c.recs[j].user + (size_t)c.recs[j].item * nu];
    MFX_HIP(hipMemcpy(c.p->d_cache, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return MFX_OK;
}

int dense_cache_out(DenseCtx &c, double *dense, int nu)
{
    std::vector<double> h(c.recs.size());
    MFX_HIP(hipStreamSynchronize(stream()));
    MFX_HIP(hipMemcpy(h.data(), c.p->d_cache, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t j = 0; j < c.recs.size(); ++j)
        if (c.slot[j] == (int32_t)j) dense[(size_t)c.recs[j].user + (size_t)c.recs[j].item * nu] = h[j];
    return MFX_OK;
}

}  // namespace

extern "C" {

int mfx_funk_dense(int variant, int min_epochs, int max_epochs, double min_improvement, int dim,
                   double f_init, double lr, double K, double *u, double *v, const int32_t *idx,
                   const double *r, int batch, double *rmse_hist, int64_t nnz, int nu, int ni)
{
    if (!u || !v || nnz <= 0 || !idx || !r || (variant != 1 && variant != 2) ||
        (variant == 1 && (!rmse_hist || max_epochs < 0 || batch < 0)))
        return fail(MFX_E_ARG, "mfx_funk_dense: bad argument");
    DenseCtx c;
    int rc = dense_setup(c, u, v, idx, r, nnz, nu, ni, dim, lr, K);
    if (rc) return rc;
    double rmse = 2.0, rmse_last = 0.0, improvement = 0.0;
    for (int f = 0; f < dim; ++f) {
        rc = dense_set_feature(c, f, f_init);
        if (rc) return rc;
        int epoch = 0;
        for (;;) {
            if (variant == 2) {                              // gd_estimator.pyx:360
                if (!(epoch < min_epochs || rmse <= rmse_last - min_improvement)) break;
            } else {                                         // gd_estimator.pyx:264
                if (!((epoch < min_epochs || improvement >= min_improvement) && epoch < max_epochs)) break;
            }
            rmse_last = rmse;
            rc = dense_epoch(c, &rmse);
            if (rc) return rc;
            if (variant == 1) {
                rmse_hist[epoch + f * max_epochs + batch * max_epochs * dim] = rmse;   // :285
                improvement = rmse_last - rmse;
            }
            ++epoch;
        }
        rc = dense_commit(c);
        if (rc) return rc;
    }
    return mfx_sgd_get_factors(c.p, u, v, nullptr, nullptr);
}

int mfx_funk_subloop(int f, int dim, double f_init, double lr, double K, double *u, double *v,
                     const int32_t *idx, const double *r, double *rating_cache, int64_t nnz, int nu,
                     int ni, double *rmse_out)
{
    if (!u || !v || nnz <= 0 || !idx || !r || !rating_cache || f < 0 || f >= dim)
        return fail(MFX_E_ARG, "mfx_funk_subloop: bad argument");
    DenseCtx c;
    int rc = dense_setup(c, u, v, idx, r, nnz, nu, ni, dim, lr, K);
    if (!rc) rc = dense_cache_in(c, rating_cache, nu);
    if (!rc) rc = dense_set_feature(c, f, f_init);
    double rmse = 0.0;
    if (!rc) rc = dense_epoch(c, &rmse);
    if (rc) return rc;
    if (rmse_out) *rmse_out = rmse;
    return mfx_sgd_get_factors(c.p, u, v, nullptr, nullptr);
}

int mfx_funk_predictor_subloop(int f, int dim, double f_init, const double *u, const double *v,
                               const int32_t *idx, const double *r, double *rating_cache,
                               int64_t nnz, int nu, int ni)
{
    if (!u || !v || nnz <= 0 || !idx || !r || !rating_cache || f < 0 || f >= dim)
        return fail(MFX_E_ARG, "mfx_funk_predictor_subloop: bad argument");
    DenseCtx c;
    int rc = dense_setup(c, const_cast<double *>(u), const_cast<double *>(v), idx, r, nnz, nu, ni,
                         dim, 0.0, 0.0);
    if (!rc) rc = dense_cache_in(c, rating_cache, nu);
    if (!rc) rc = dense_set_feature(c, f, f_init);
    if (!rc) rc = dense_commit(c);
    if (!rc) rc = dense_cache_out(c, rating_cache, nu);
    return rc;
}

}  // extern "C"
