// Internals of the SGD plan shared between mfx_sgd.hip (owner), mfx_comm.hip (user-sharded
// averaging mode) and mfx_dsgd.hip (exact block-rotation mode).  Not part of the ABI.
#pragma once
#include "mfx_common.h"

struct mfx_sgd_plan;

namespace mfx {

// ---- full-k epochs -------------------------------------------------------------------------
// `epochs` full-k epochs with `hook` (nullable) called after each epoch's launches have been enqueued;
// ends with a stream sync and fills rmse_out / se_out (nullable, f64[epochs]).
int sgd_run_k1_hooked(mfx_sgd_plan *p, int mode, int epochs, double lr, double Ku, double Ki,
                      double Kb, int (*hook)(mfx_sgd_plan *, int, void *), void *ctx,
                      double *rmse_out, double *se_out);
// Asynchronous building block: enqueue ONE full-k epoch (hipGraph replay) on the library stream.
// The epoch's squared-error sum lands in the plan's per-epoch array at the plan's running counter;
// sgd_reset_se() rewinds that counter (and sizes the array), sgd_fetch_se() syncs and reads it.
int sgd_reset_se(mfx_sgd_plan *p, int64_t epochs);
int sgd_enqueue_k1(mfx_sgd_plan *p, int mode, double lr, double Ku, double Ki, double Kb);
int sgd_fetch_se(mfx_sgd_plan *p, int64_t n, double *se_out);

// ---- plan state ----------------------------------------------------------------------------
double *plan_U(mfx_sgd_plan *p);          // [ni][k] row-major item rows
double *plan_V(mfx_sgd_plan *p);          // [nu][k]
double *plan_ib(mfx_sgd_plan *p);
double *plan_ub(mfx_sgd_plan *p);
int plan_ni(mfx_sgd_plan *p);
int plan_nu(mfx_sgd_plan *p);
int plan_k(mfx_sgd_plan *p);
int64_t plan_nnz(mfx_sgd_plan *p);
int plan_start_buffers(mfx_sgd_plan *p, double **U_start, double **ib_start);
int plan_prepare_nk(mfx_sgd_plan *p);
// Several plans over the SAME users (the item blocks of one DSGD rank): p gives up its own user-side
// buffers and uses master's.  Call before any factor is set.
int plan_alias_users(mfx_sgd_plan *p, mfx_sgd_plan *master);
// Side-by-side factor transfers in the host layout ([k][n] feature-major f64), one side at a time;
// after the first upload of BOTH sides plan_mark_ready() declares the row-major copies current.
int plan_upload_items(mfx_sgd_plan *p, const double *u_km, const double *ib);
int plan_upload_users(mfx_sgd_plan *p, const double *v_km, const double *ub);
int plan_download_items(mfx_sgd_plan *p, double *u_km, double *ib);
int plan_download_users(mfx_sgd_plan *p, double *v_km, double *ub);
void plan_mark_ready(mfx_sgd_plan *p);

// ---- resident-side full-k epoch (mfx_sgd_res.hip) -------------------------------------------
struct ResPlan;
struct __attribute__((aligned(16))) ResRec { int32_t trav; int32_t seq; double r; };   // travelling entity, its chain rank, rating
// d_recs: the plan's level-major records (they keep every user's and every item's ratings in input order)
int res_prepare(ResPlan **out, const int32_t *idx, const Rec *d_recs, int64_t nnz, int nu, int ni, int k, int64_t n_levels, int want,
                bool headroom);
// mfx_schedule_dev.hip: records grouped by resident entity (stable: each entity's list is its chain), every rating ranked in its
// travelling entity's chain; lo / hi [n_res]: the entity's record range (0, 0 when it has none)
int res_max_degrees(const Rec *d_recs, int64_t nnz, int nu, int ni, int *max_user_deg, int *max_item_deg);
int res_build_device(const Rec *d_recs, int64_t nnz, int n_res, int n_trav, bool res_item, ResRec *d_out, int32_t *d_lo, int32_t *d_hi,
                     int32_t *d_pos_of = nullptr, int32_t *d_ent_of = nullptr, int32_t *d_next = nullptr, int32_t *d_first_pos = nullptr);
// ---- resident-side Funk pass (mfx_sgd_res.hip): lane = item, user values travel inside their 16-byte granules
struct Res2Plan;
int res2_prepare(Res2Plan **out, const Rec *d_recs, int64_t nnz, int nu, int ni, bool force);
void res2_destroy(Res2Plan *rp);
size_t res2_dev_bytes(const Res2Plan *rp);
int res2_box_width(const Res2Plan *rp);
int res2_waves(const Res2Plan *rp);
int res2_hot_waves(const Res2Plan *rp);
int res2_hot_user_waves(const Res2Plan *rp);
int64_t res2_se_slots(const Res2Plan *rp);
const double *res2_se_partials(const Res2Plan *rp);
int res2_reset_cache(Res2Plan *rp);
int res2_begin_batch(Res2Plan *rp);
int res2_enqueue_pass(Res2Plan *rp, const void *d_dyn, const double *ib, const double *ub, double mu, double lr, double K, int upd_u, int upd_i);
int res2_enqueue_commit(Res2Plan *rp, const void *d_dyn, const double *ib, const double *ub, double mu);
int res2_status(Res2Plan *rp, unsigned *status, unsigned *applied = nullptr);
int res2_cache_to_levels(Res2Plan *rp, double *d_cache_levels);
void res_destroy(ResPlan *rp);
int res_kind(const ResPlan *rp);                 // 0 none, 1 ticket (users resident, runs by ticket), 2 static owners
size_t res_dev_bytes(const ResPlan *rp);
int64_t res_se_slots(const ResPlan *rp);
const double *res_se_partials(const ResPlan *rp);
// variant: 0 = f64 rows, ordered dot (exact); 1 = butterfly dot + FMA (tolerance mode, f64 rows); 2 = f32 row storage (ordered dot)
int res_describe(const ResPlan *rp, int mode, char *buf, int buflen, int variant = 0);
bool res_variants_ok(const ResPlan *rp);       // variants 1 and 2 are served by this plan's resident-side kernel (all but the burst one)
int res_begin_batch(ResPlan *rp);
int res_enqueue_epoch(ResPlan *rp, int mode, double *U_nk, double *V_nk, double *ib, double *ub, double lr, double Ku,
                      double Ki, double Kb, int variant = 0);
int res_status(ResPlan *rp, unsigned *status, unsigned *applied = nullptr);
// plan creation with the epoch kernel chosen (MFX_EPOCH_*); the Funk / learned-bias / dense-cache drop-ins pass
// MFX_EPOCH_LEVELS: they never run a plain full-k epoch.  | kEpochHeadroom (DSGD sub-plans): a static grid leaves an eighth
// of the chip's wave slots free, for the transfer kernels that run beside it.
constexpr int kEpochHeadroom = 0x100;
// DSGD enqueues block epochs asynchronously (sgd_enqueue_k1): the resident-side launch's status word is cleared by
// sgd_reset_se and read by sgd_res_check after the caller's own stream sync -- non-zero: MFX_E_HIP (no fallback mid-run).
int sgd_res_check(mfx_sgd_plan *p);
// the same word, raw: 0 fine, 1 = a static launch of the batch found its grid not co-resident (it and every later launch of the plan
// did nothing), 3 = a spin timed out (state undefined).  plan_force_levels: the plan runs on the level schedule from now on.
int sgd_res_poll(mfx_sgd_plan *p, unsigned *status);
void plan_force_levels(mfx_sgd_plan *p);
int plan_epoch_kernel(const mfx_sgd_plan *p);      // MFX_EPOCH_LEVELS / _TICKET / _STATIC in use
mfx_sgd_plan *plan_create_ex(const int32_t *idx, const double *r, int64_t nnz, int nu, int ni, int dim, int sched, int epoch);

// ---- Funk hooks (user-sharded Funk pass) ---------------------------------------------------
struct FunkHooks {
    int (*pre)(void *ctx, double *uf, int ni);
    int (*post)(void *ctx, double *uf, int ni);
    int (*global_se)(void *ctx, double *se, double *nnz);
    void *ctx;
};
int funk_core(mfx_sgd_plan *p, int min_epochs, double min_improvement, double f_init, double lr,
              double K, double mu, int use_bias, int upd_u, int upd_i, int epoch_cap,
              double *rmse_hist, int64_t hist_cap, int64_t *n_hist, const FunkHooks *hk);
// the same run over item blocks that share one user side, a pass = the blocks one after the other (mfx_dsgd_run_funk)
int funk_blocks_core(mfx_sgd_plan *const *plans, int B, int min_epochs, double min_improvement, double f_init, double lr, double K,
                     double mu, int use_bias, int upd_u, int upd_i, int epoch_cap, double *rmse_hist, int64_t hist_cap, int64_t *n_hist);

// ---- communicator (mfx_comm.hip) -----------------------------------------------------------
bool comm_active();
int comm_rank();
int comm_world();
int comm_allgather_f64(double *buf, size_t count);          // in place: rank g's part at buf + g*count
int comm_allreduce_sum_f64(double *buf, size_t count);      // in place, library stream
// cur <- start + scale * allreduce(cur - start); start <- cur; on stream `st` (nullptr: the library stream)
int comm_average_f64(double *cur, double *start, size_t n, double scale, hipStream_t st = nullptr);
int comm_agree_max(int *flag);                              // collective: the maximum of every rank's flag (synchronous)
int comm_broadcast_f64(double *buf, size_t count, int root);
// one fused group on stream `st` (nullptr: the library stream): send every (ptr, count) of `out` to `to`,
// receive every (ptr, count) of `in` from `from`
struct CommSeg { double *ptr; size_t count; };
int comm_sendrecv_f64(const CommSeg *out, int n_out, int to, const CommSeg *in, int n_in, int from,
                      hipStream_t st = nullptr);

}  // namespace mfx
