/*
 * mfx.h -- C ABI of libmfx.so: MI355X (gfx950) latent-factor training hot path.
 *
 * This is the drop-in boundary for the three Cython extension modules of
 * mlaprise/mfrec (mfrec/lib/kmf_train.pyx, gd_estimator.pyx, als_implicit.pyx).
 * Every entry point takes plain pointers and sizes; no Python, numpy or torch
 * types appear.  Host arrays use the REFERENCE's layouts:
 *
 *   u  f64 [k, ni]  item factors, feature-major C-contiguous (gd_estimator.pyx:23,
 *   v  f64 [k, nu]  user factors                              gradient_descent.py:524-525)
 *   ratings_index int32 [nnz, 2]  column 0 = user, column 1 = item (base.py:1121-1124)
 *   ratings       f64   [nnz]
 *   items_bias f64 [ni], users_bias f64 [nu]
 *
 * Arrays are caller-owned and updated IN PLACE, as the reference does.  All
 * functions return 0 on success or a negative MFX_E_* code; mfx_last_error()
 * gives the message.  The library is one-call-at-a-time per process (the
 * reference holds the GIL for the whole call); ctypes releases the GIL.
 *
 * There is NO CPU fallback: without a usable HIP device every compute entry
 * point fails with MFX_E_HIP.
 */
#ifndef MFX_H
#define MFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFX_OK 0
#define MFX_E_ARG (-1)      /* bad argument (NULL, negative size, k < 1, ...)            */
#define MFX_E_RANGE (-2)    /* user/item index outside [0,nu) / [0,ni)                   */
#define MFX_E_NOMEM (-3)    /* host or device allocation failed                          */
#define MFX_E_HIP (-4)      /* HIP runtime error / no device                             */
#define MFX_E_RCCL (-5)     /* RCCL error                                                */
#define MFX_E_SINGULAR (-6) /* ALS: k x k system not positive definite                   */
#define MFX_E_STATE (-7)    /* call sequence error (e.g. plan used before set_factors)   */

/* kernel families of the resident (plan) API */
#define MFX_MODE_LINEAR 0   /* train_linear_kernel    kmf_train.pyx:195-277 */
#define MFX_MODE_LOGISTIC 1 /* train_logistic_kernel  kmf_train.pyx:103-189 */

/* ------------------------------------------------------------------ runtime */
int mfx_version(void);
const char *mfx_last_error(void);
int mfx_device_count(void);                 /* hipGetDeviceCount; 0 if none / error        */
int mfx_init(int device);                   /* bind this process to `device`, create stream */
int mfx_sync(void);                         /* hipStreamSynchronize on the library stream  */
int mfx_device_name(char *buf, int buflen); /* gcnArchName of the bound device             */
/* hipMemGetInfo of the bound device, bytes (either pointer may be NULL) -- capacity planning and leak tests */
int mfx_device_mem_info(int64_t *free_bytes, int64_t *total_bytes);
/* HIP-event stopwatch on the library's stream (the stream every kernel is launched on). */
int mfx_timer_start(void);
int mfx_timer_stop(double *elapsed_ms);
/* Device time (HIP events on the library stream) of the compute part -- epochs only, no
 * schedule build, no H2D/D2H -- of the most recent one-shot call that measures it
 * (currently mfx_als_wrmf); 0 if none. */
double mfx_last_compute_ms(void);

/* ----------------------------------------------------- host-side scheduling
 * Dependency-level schedule of a rating sequence (pure host code, no GPU).
 * level[t] = 1 + max(level of the previous rating with the same user,
 *                    level of the previous rating with the same item)
 * Ratings of one level touch pairwise-distinct users and items, so they can be
 * applied in parallel and, levels taken in order, the result is bit-identical
 * to the reference's sequential loop (kmf_train.pyx:243, gd_estimator.pyx:745).
 * perm is the stable level-major permutation; level_off has n_levels+1 entries.
 */
int mfx_schedule_levels(const int32_t *ratings_index, int64_t nnz, int nu, int ni,
                        int32_t *level_out, int64_t *n_levels_out);
/* The same levels computed on the GPU (host arrays in and out): radix-sorted user / item chains +
 * one frontier launch per level (replaces the marshalling loops of
 * base.py:1115-1131 at sizes where the sequential host pass dominates).  nnz < 2^31, at most 2^20
 * levels.  Result identical to mfx_schedule_levels. */
int mfx_schedule_levels_gpu(const int32_t *ratings_index, int64_t nnz, int nu, int ni,
                            int32_t *level_out, int64_t *n_levels_out);
int mfx_schedule_permutation(const int32_t *level, int64_t nnz, int64_t n_levels,
                             int64_t *perm_out, int64_t *level_off_out);
/* The same ordering applied directly: recs_out[j] = 16-byte record {int32 user, int32 item,
 * f64 rating} of the j-th rating in level-major order (threaded stable counting sort; this is
 * what a plan uploads).  level_off_out has n_levels+1 entries. */
int mfx_schedule_pack(const int32_t *ratings_index, const double *ratings, const int32_t *level,
                      int64_t nnz, int64_t n_levels, void *recs_out, int64_t *level_off_out);

/* An input ORDER with close to the minimum number of dependency levels (first-fit edge colouring
 * over a seeded random visiting order; round-major).  The reference shuffles the ratings with an
 * unseeded RNG before every train() (base.py:1127-1129), i.e. leaves the order open; this structured one
 * roughly halves the level count of an i.i.d. shuffle, and the result is exact ON it.  (Model quality on such
 * orders against uniform draws: measured in profiles/r04_block_order_quality.json, not assumed.)  perm_out[j] = index of the j-th
 * rating of the new order.  Host code. */
int mfx_schedule_balanced_order(const int32_t *ratings_index, int64_t nnz, int nu, int ni,
                                uint64_t seed, int64_t *perm_out, int64_t *n_rounds_out);
/* The same kind of order built on the GPU in milliseconds: the dependency levels of the given order
 * (mfx_schedule_levels_gpu) are first-fit recoloured level by level -- inside a level no two ratings
 * share a user or an item, so a level recolours in parallel, deterministically.  Round-major, ascending
 * item inside a round.  (The reference's own order is an unseeded shuffle, base.py:1127-1129.) */
int mfx_schedule_balanced_order_gpu(const int32_t *ratings_index, int64_t nnz, int nu, int ni,
                                    int64_t *perm_out, int64_t *n_rounds_out);

/* ------------------------------------------- one-shot drop-ins (host arrays)
 * Replace, argument for argument, the Cython `def` functions.  Arguments the
 * reference accepts but never reads (f_init, learning_rate_users/items and
 * overall_avg of the kmf kernels; max_epochs of the Funk loops) are dropped
 * here and swallowed by the Python wrappers.
 */

/* train_linear_kernel (kmf_train.pyx:195-213) / train_logistic_kernel (:103-121).
 * rmse_out: NULL or f64[nbr_epochs], the per-epoch online RMSE the reference prints. */
int mfx_kmf_train(int mode, int nbr_epochs, int dim, double learning_rate, double K_users,
                  double K_items, double K_bias, double *u, double *v,
                  const int32_t *ratings_index, const double *ratings, double *items_bias,
                  double *users_bias, int64_t nnz, int nu, int ni, int update_users,
                  int update_items, double *rmse_out);

/* estimator_loop_without_bias (gd_estimator.pyx:691-704; items_bias = users_bias = NULL,
 * overall_avg = 1.0), estimator_loop_with_bias (:489-507), estimator_loop_with_bias_dev
 * (:588-608, update gates).  epoch_cap > 0 bounds the epochs per feature (the reference's
 * `while` at :742 ignores max_epochs); 0 = unbounded.  rmse_hist: NULL or f64[hist_cap];
 * *n_hist (nullable) = epochs run over all features. */
int mfx_funk(int min_epochs, double min_improvement, int dim, double f_init, double learning_rate,
             double K, double overall_avg, double *u, double *v, const int32_t *ratings_index,
             const double *ratings, const double *items_bias, const double *users_bias,
             int64_t nnz, int nu, int ni, int update_users, int update_items, int epoch_cap,
             double *rmse_hist, int64_t hist_cap, int64_t *n_hist);

/* Dense-cache members of the same family.  The reference indexes a dense f64[nu*ni] cache by
 * user + item*nu, so duplicate (user,item) ratings share a cell; reproduced exactly.
 * variant 1 = estimator_loop (gd_estimator.pyx:210-303: honours max_epochs, fills
 * rmse_hist[epoch + f*max_epochs + batch*max_epochs*dim]); variant 2 = estimator_loop2
 * (:308-395: the stopping rule of :360, rmse_hist unused). */
int mfx_funk_dense(int variant, int min_epochs, int max_epochs, double min_improvement, int dim,
                   double f_init, double learning_rate, double K, double *u, double *v,
                   const int32_t *ratings_index, const double *ratings, int batch,
                   double *rmse_hist, int64_t nnz, int nu, int ni);
/* estimator_subloop (gd_estimator.pyx:903-962): one epoch of feature f on the caller's dense
 * cache f64[nu*ni]; *rmse_out = the value the reference returns. */
int mfx_funk_subloop(int f, int dim, double f_init, double learning_rate, double K, double *u,
                     double *v, const int32_t *ratings_index, const double *ratings,
                     double *rating_cache, int64_t nnz, int nu, int ni, double *rmse_out);
/* predictor_subloop (gd_estimator.pyx:967-995): cache-commit of feature f into the dense cache. */
int mfx_funk_predictor_subloop(int f, int dim, double f_init, const double *u, const double *v,
                               const int32_t *ratings_index, const double *ratings,
                               double *rating_cache, int64_t nnz, int nu, int ni);

/* estimator_loop_with_learned_bias (gd_estimator.pyx:401-421). */
int mfx_learned_bias(int min_epochs, double min_improvement, int dim, double f_init,
                     double learning_rate, double learning_rate_users, double learning_rate_items,
                     double K_feature, double K_bias, double overall_avg, double *u, double *v,
                     const int32_t *ratings_index, const double *ratings, double *items_bias,
                     double *users_bias, int64_t nnz, int nu, int ni, int epoch_cap,
                     double *rmse_hist, int64_t hist_cap, int64_t *n_hist);

/* estimator_loop_with_implicit_feedback (gd_estimator.pyx:785-806): the SVD++-like prototype, reproduced as
 * written (feedback_sum overwritten at :873, cache commit without the implicit term).  y f64[k, ni] is the
 * implicit item factor; rated int32[n_rated, 2] = (user, item) pairs sorted by user, pairs distinct;
 * rated_hash int32[nu, 2] = (first row of the user in `rated`, count) -- gradient_descent.py:261-296.
 * Every rating rewrites y[f, j] for all items j of its user, so the loop has no exact parallel schedule over
 * ratings: ONE workgroup walks them in order (the parallelism is inside a rating).  Bit-identical to the
 * reference and 7-10x SLOWER than its C loop on one host core (0.6 us per rating with the three feature
 * vectors in LDS, 2 ni + nu doubles <= 150 KB; more beyond) -- served for completeness, not for throughput. */
int mfx_funk_implicit(int min_epochs, double min_improvement, int dim, double f_init, double learning_rate,
                      double K, double overall_avg, double *u, double *v, double *y,
                      const int32_t *ratings_index, const double *ratings, const int32_t *rated,
                      int64_t n_rated, const int32_t *rated_hash, const double *items_bias,
                      const double *users_bias, int64_t nnz, int nu, int ni, int epoch_cap,
                      double *rmse_hist, int64_t hist_cap, int64_t *n_hist);

/* als_wrmf (als_implicit.pyx:211-225).  *_row are the run-length arrays of
 * mfrec/lib/datasets.py:13-32 ([0, count_0, count_1, ...], n_*row entries). */
int mfx_als_wrmf(int nbr_epochs, int dim, double *u, double *v, const int32_t *ratings_users_row,
                 int64_t n_urow, const int32_t *ratings_users_col,
                 const int32_t *ratings_items_row, int64_t n_irow,
                 const int32_t *ratings_items_col, int nu, int ni, int c_pos, double lambda);

/* Batched predict / RMSE over triples (metrics.py:51-82 with the family's predict(),
 * kind: 0 GD dot+1 (gradient_descent.py:621-631), 1 KMF linear (kmf.py:88-94),
 * 2 KMF logistic (kmf.py:79-85), 3 GD with bias (gradient_descent.py:637-648), 4 WRMF dot).
 * pred_out: NULL or f64[nnz]. */
int mfx_predict_rmse(int kind, int dim, const double *u, const double *v,
                     const int32_t *ratings_index, const double *ratings, const double *items_bias,
                     const double *users_bias, double overall_avg, int64_t nnz, int nu, int ni,
                     double *pred_out, double *rmse_out);

/* Batched top-N: the ranking of find_recommended_items (mf.py:144-193) / find_user_top_match
 * (gradient_descent.py:768-800) for n_sel users in one call.  Candidates are items 0 .. n_cand-1; a user's rated
 * items (CSR: rated_off int64[n_sel+1], rated_items int32[]; both NULL = none), the item whose id equals the
 * user id, NaN and zero scores are excluded exactly as the reference excludes them; order: descending score, ties by
 * ascending item.  kind as in mfx_predict_rmse.  u [k, ni]; v_sel [k, n_sel] = the selected users' columns of v;
 * ub_sel [n_sel]; users [n_sel] = their ids.  out_items int32[n_sel, N] (-1 where fewer than N qualify),
 * out_scores f64[n_sel, N].  The dot product is summed as mfx_predict_rmse sums it: same bits. */
int mfx_topn_batch(int kind, int dim, const double *u, const double *v_sel, const double *items_bias,
                   const double *users_bias_sel, double overall_avg, int ni, const int32_t *users, int n_sel,
                   int n_cand, int N, const int64_t *rated_off, const int32_t *rated_items, int32_t *out_items,
                   double *out_scores);

/* The serving form of the same ranking: the model (u [k, ni], v [k, nu], biases, every user's rated list as a CSR over
 * user ids: rated_off int64[nu+1], rated_items; both NULL = none) is uploaded ONCE; a query names users by id and costs
 * a gather of their columns and the kernel -- no factor upload.  Row j of a query equals row j of mfx_topn_batch for the
 * same users, bit for bit.  The plan is a snapshot: train again -> create again. */
typedef struct mfx_topn_plan mfx_topn_plan;
mfx_topn_plan *mfx_topn_plan_create(int kind, int dim, const double *u, const double *v, const double *items_bias,
                                    const double *users_bias, double overall_avg, int nu, int ni,
                                    const int64_t *rated_off, const int32_t *rated_items);
int mfx_topn_plan_query(mfx_topn_plan *p, const int32_t *users, int n_sel, int n_cand, int N, int32_t *out_items,
                        double *out_scores);
void mfx_topn_plan_destroy(mfx_topn_plan *p);

/* For N <= 16 the ranking is fused into the score pass (no score matrix; a user for which the fused kernel cannot
 * prove its list complete is redone by the sort path).  stats[0] / stats[1]: users answered by the fused kernel /
 * redone, totals of this process.  MFX_TOPN_FUSED=0 forces the sort path. */
int mfx_topn_stats(int64_t stats[2]);

/* ------------------------------------------------ resident (plan) API
 * Same kernels, but ratings, schedule and factors stay in HBM between calls:
 * create (schedule + upload) -> set_factors -> run_* ... -> get_factors.
 * bench.py times run_* only (inputs resident in HBM).
 */
typedef struct mfx_sgd_plan mfx_sgd_plan;

mfx_sgd_plan *mfx_sgd_plan_create(const int32_t *ratings_index, const double *ratings,
                                  int64_t nnz, int nu, int ni, int dim);
/* The same with the schedule builder chosen: MFX_SCHED_HOST (mfx_schedule_levels + _pack, then one
 * upload), MFX_SCHED_DEVICE (raw arrays uploaded, levels + level-major records built in HBM; fails
 * beyond 2^20 levels) or MFX_SCHED_AUTO (device from 2^20 ratings on, falling back to the host when
 * the data needs more than nnz/1000 levels).  Both builders give the same records in the same order. */
enum { MFX_SCHED_AUTO = 0, MFX_SCHED_HOST = 1, MFX_SCHED_DEVICE = 2 };
mfx_sgd_plan *mfx_sgd_plan_create2(const int32_t *ratings_index, const double *ratings,
                                   int64_t nnz, int nu, int ni, int dim, int sched);
/* The same with the kernel of a plain full-k epoch (mfx_sgd_run_kmf, both sides trained, f64 rows) chosen:
 *   MFX_EPOCH_LEVELS  one launch per dependency level (k1_level / k1_tail): every order, every size.
 *   MFX_EPOCH_TICKET  ONE persistent launch per epoch with the USER rows resident in registers: for ratings grouped
 *                     by user (every user one contiguous run: get_ratings(randomize_order=False), base.py:1115-1126);
 *                     a wave takes the next user runs by ticket and keeps their rows and biases for the whole run,
 *                     only the item rows travel (behind per-item chain counters).  Any number of users.
 *   MFX_EPOCH_STATIC  the same with FIXED owners: every row of the smaller side (users or items) lives in the
 *                     registers of one wave for the whole epoch -- any rating order, but the side must fit the
 *                     register file (65 536 rows at k = 64) and the whole grid must be co-resident; if it is not,
 *                     nothing is modified and the plan goes back to MFX_EPOCH_LEVELS by itself.
 *   MFX_EPOCH_AUTO    by the measured cost model (chain depth x step against ratings / rate); what create/create2 use.
 * All three give the level schedule's bits (= the reference loop's).  TICKET / STATIC on a plan that is not eligible:
 * NULL, MFX_E_ARG.  Even k in 18..128 only; other k: AUTO means LEVELS. */
enum { MFX_EPOCH_AUTO = 0, MFX_EPOCH_LEVELS = 1, MFX_EPOCH_TICKET = 2, MFX_EPOCH_STATIC = 3 };
mfx_sgd_plan *mfx_sgd_plan_create3(const int32_t *ratings_index, const double *ratings,
                                   int64_t nnz, int nu, int ni, int dim, int sched, int epoch);
/* info[0] = the epoch kernel in use (MFX_EPOCH_LEVELS / _TICKET / _STATIC), [1] = the one prepared at creation
 * (differs after a failed co-residency check), [2] = device bytes of its records, [3] = squared-error slots */
int mfx_sgd_plan_res_info(const mfx_sgd_plan *p, int64_t info[4]);
/* MFX_SCHED_HOST or MFX_SCHED_DEVICE: which builder produced this plan's schedule */
int mfx_sgd_plan_schedule_builder(const mfx_sgd_plan *p);
void mfx_sgd_plan_destroy(mfx_sgd_plan *p);
/* info[0]=n_levels, [1]=max level size, [2]=launches per epoch (full-k), [3]=tail launches,
 * [4]=ratings per wave batch, [5]=device bytes held, [6]=schedule build microseconds,
 * [7]=launches per epoch (Funk) */
int mfx_sgd_plan_info(const mfx_sgd_plan *p, int64_t info[8]);
/* Name of the kernel this plan launches for `mode` (0 linear, 1 logistic, 2 learned bias): the resident-side epoch
 * "k1_res<lanes per row,slots per wave,mode[,butterfly-dot+fma | ,f32 rows],binding: resident side,waves>" or
 * "k1_level<lanes per rating,features per lane,ratings per wave,access form,mode>"; mode 3: the kernel of the Funk pass
 * ("k2_res<...>": the persistent lane-per-item pass, mailbox or granule form; "k2_xcd_pass<...>": the single-XCD persistent
 * pass; "k2_level<...>": one launch per level). */
int mfx_sgd_plan_kernel_name(const mfx_sgd_plan *p, int mode, char *buf, int buflen);
/* Storage type of the factor rows inside the plan: 64 (default; exact), 32 or 16 (bf16) -- "narrow
 * storage, f64 arithmetic" modes of the full-k kernels: rows are widened on load and rounded once on
 * store.  NOT bit-exact (measured RMSE deltas: DESIGN.md -- f32 stays within the 1e-5 budget, bf16 does
 * not); single-GPU; call before set_factors.  bf16: k a multiple of 8, k >= 16, linear / logistic.
 * f32 rows are served by the resident-side epoch too (same bits as on the level kernels), bf16 by the level kernels. */
int mfx_sgd_plan_set_storage(mfx_sgd_plan *p, int bits);
/* How the full-k kernels (linear / logistic) form the k-term dot product ib + ub + sum_f U[f,i] V[f,u]:
 * MFX_DOT_ORDERED (default): summed strictly left to right by one lane, no FMA -- the reference's
 * operation order (kmf_train.pyx:75-97), bit-identical results.  MFX_DOT_BUTTERFLY: per-lane partial
 * products and a wave butterfly (DPP + cross-row swaps) with a FIXED association, FMA in dot and row
 * update -- deterministic, f64, but NOT bit-identical to the reference (tolerance mode: final RMSE
 * within 1e-5, measured in DESIGN.md).  f64 rows, even k > 4.  Served by the resident-side epoch and by the level kernels
 * with the SAME association: the same bits whichever kernel runs the plan. */
enum { MFX_DOT_ORDERED = 0, MFX_DOT_BUTTERFLY = 1 };
int mfx_sgd_plan_set_dot(mfx_sgd_plan *p, int dot);
int mfx_sgd_set_factors(mfx_sgd_plan *p, const double *u, const double *v,
                        const double *items_bias, const double *users_bias);
int mfx_sgd_get_factors(mfx_sgd_plan *p, double *u, double *v, double *items_bias,
                        double *users_bias);
/* Asynchronous on the library stream; rmse_out (nullable, f64[nbr_epochs]) is filled
 * after an internal sync at the end of the call. */
int mfx_sgd_run_kmf(mfx_sgd_plan *p, int mode, int nbr_epochs, double learning_rate,
                    double K_users, double K_items, double K_bias, int update_users,
                    int update_items, double *rmse_out);
/* One instrumented epoch: a HIP event pair around every kernel launch.
 * kernel_ms = sum of the per-launch durations, n_launches = their number. */
int mfx_sgd_profile_kmf(mfx_sgd_plan *p, int mode, double learning_rate, double K_users,
                        double K_items, double K_bias, double *kernel_ms, int64_t *n_launches);
int mfx_sgd_run_funk(mfx_sgd_plan *p, int min_epochs, double min_improvement, double f_init,
                     double learning_rate, double K, double overall_avg, int use_bias,
                     int update_users, int update_items, int epoch_cap, double *rmse_hist,
                     int64_t hist_cap, int64_t *n_hist);
int mfx_sgd_run_learned_bias(mfx_sgd_plan *p, int min_epochs, double min_improvement,
                             double f_init, double learning_rate, double learning_rate_users,
                             double learning_rate_items, double K_feature, double K_bias,
                             double overall_avg, int epoch_cap, double *rmse_hist,
                             int64_t hist_cap, int64_t *n_hist);

/* ------------------------------------------------ multi-GPU (RCCL over xGMI)
 * One process per GPU.  Rank 0 calls mfx_comm_unique_id and ships the 128
 * bytes to the other ranks by any out-of-band channel; every rank then calls
 * mfx_comm_init.  Users are sharded across ranks; every rank holds a full
 * replica of the item factors.  After each local epoch:
 *     U <- U_start + scale * allreduce_sum(U_local - U_start)   (same for items_bias)
 * scale = 1/world (parameter averaging, the scheme of the reference's own
 * prototype gradient_descent.py:124-211) or 1 (delta sum).
 */
#define MFX_UNIQUE_ID_BYTES 128
int mfx_comm_unique_id(void *id_out);
int mfx_comm_init(int rank, int world, const void *id);
int mfx_comm_destroy(void);
int mfx_comm_barrier(void);
/* *value <- max over ranks (bench.py takes the slowest rank's time with it) */
int mfx_comm_allreduce_max(double *value);
/* full-k SGD with the per-epoch item-delta all-reduce; the plan holds this rank's shard
 * (its users' ratings, user ids local to the shard). */
int mfx_sgd_run_kmf_sharded(mfx_sgd_plan *p, int mode, int nbr_epochs, double learning_rate,
                            double K_users, double K_items, double K_bias, double scale,
                            double *rmse_out /* local shard's online RMSE */,
                            double *se_out /* nullable: f64[nbr_epochs] local sum of squared errors */);

/* Test / rehearsal hook: the arithmetic of mfx_sgd_run_kmf_sharded for `world` shards that all live on
 * THIS GPU, no communicator.  plans[g] = shard g's plan (same item replica set in each); per epoch every
 * shard runs its local epoch, the item deltas are summed in rank order on the device (the stand-in for
 * the all-reduce, whose summation order RCCL does not fix) and every shard receives
 * U_start + scale * sum.  se_out: nullable f64[world * nbr_epochs], shard-major. */
int mfx_sgd_run_kmf_fake_world(mfx_sgd_plan **plans, int world, int mode, int nbr_epochs,
                               double learning_rate, double K_users, double K_items, double K_bias,
                               double scale, double *se_out);

/* Funk passes (estimator_loop_without_bias / _with_bias) on a user shard: after every
 * feature-epoch the item-side vector U[f,:] is combined as for the full-k kernels
 * (all-reduce of [ni] doubles) and the squared error is made global, so every rank takes the
 * same decision in the stopping rule of gd_estimator.pyx:742. */
int mfx_sgd_run_funk_sharded(mfx_sgd_plan *p, int min_epochs, double min_improvement, double f_init,
                             double learning_rate, double K, double overall_avg, int use_bias,
                             int epoch_cap, double scale, double *rmse_hist, int64_t hist_cap,
                             int64_t *n_hist);

/* ALS shards exactly: every rank holds full replicas of u, v and both run-length views,
 * solves a contiguous slice of the rows of each half-epoch and the slices are all-gathered
 * (RCCL ncclAllGather over xGMI) -- same arguments and same result as mfx_als_wrmf
 * (als_implicit.pyx:211-352) on every rank. */
int mfx_als_wrmf_sharded(int nbr_epochs, int dim, double *u, double *v,
                         const int32_t *ratings_users_row, int64_t n_urow,
                         const int32_t *ratings_users_col, const int32_t *ratings_items_row,
                         int64_t n_irow, const int32_t *ratings_items_col, int nu, int ni, int c_pos,
                         double lambda);

/* ------------------------------------------------ exact multi-GPU mode (DSGD block rotation)
 * Users are cut into `world` blocks (one per GPU), items into n_item_blocks = any positive multiple of world blocks (in ONE process --
 * world = 1 or rank = -1 -- the epoch is the sequential loop on the block order, every block small enough on the static resident-side epoch)
 * (stride = n_item_blocks / world).  In step h GPU g trains the ratings of block
 * (g, (stride * g + h) mod n_item_blocks) and then passes that item block (rows + biases) to GPU g-1 over xGMI
 * (ncclSend/ncclRecv), which trains it `stride` steps later: with 2 * world blocks a block rests one whole
 * step in transit, the transfer runs on a second stream under the next step's levels.  The blocks of a step
 * share no user and no item, so the epoch IS the reference's sequential loop (kmf_train.pyx:243-270) on the
 * rating order
 *     for h: for g: ratings of block (g, (stride * g + h) mod n_item_blocks), in block order
 * and the result is bit-identical to it -- the multi-GPU mode that keeps "RMSE within 1e-5 of the
 * reference" (parameter averaging, above, does not).  "Block order" is the given order, or with
 * MFX_DSGD_BALANCED the block's own minimum-level order (first-fit recolouring, as
 * mfx_schedule_balanced_order_gpu); mfx_dsgd_get_order returns the whole order.  The reference's only
 * multi-machine precedent is the averaging prototype gradient_descent.py:124-211; its rating order is an
 * unseeded shuffle (base.py:1127-1129): the order is open, and the result is exact ON the order this object reports
 * (its training quality against uniform shuffles: profiles/r04_block_order_quality.json).
 * ratings_index holds GLOBAL ids; a rank keeps the ratings of its own user block (so passing all ratings
 * or only the rank's own is the same).  rank = -1: all `world` ranks live in this process on this GPU
 * (hand-off = device copy), for tests and rehearsals.  mfx_comm_init first when world > 1 and rank >= 0.
 * v / users_bias are this rank's users ([k, nu_local] / [nu_local]; fake world: all users); u /
 * items_bias are always the full item side.  mfx_dsgd_get_factors is collective. */
typedef struct mfx_dsgd mfx_dsgd;
enum { MFX_DSGD_BALANCED = 1 };
mfx_dsgd *mfx_dsgd_create(const int32_t *ratings_index, const double *ratings, int64_t nnz, int nu,
                          int ni, int dim, const int64_t *user_bounds /* world+1 */,
                          const int64_t *item_bounds /* n_item_blocks+1 */, int n_item_blocks, int rank,
                          int world, int flags);
void mfx_dsgd_destroy(mfx_dsgd *d);
/* info[0] = ratings held by this process, [1] = dependency levels summed over its blocks,
 * [2] / [3] = largest / smallest block (ratings), [4] = kernel launches per epoch, [5] = item blocks,
 * [6] = blocks whose epoch is the resident-side launch (MFX_EPOCH_STATIC chosen by the cost model: every block is
 *       a plan of its own; such a launch needs its whole grid co-resident -- on a GPU shared with another process the
 *       run ends with MFX_E_HIP and a message naming MFX_K1_RES=levels) */
int mfx_dsgd_info(const mfx_dsgd *d, int64_t info[8]);
/* perm_out[p] = index (into the arrays given to mfx_dsgd_create) of the p-th rating of the sequential order
 * this object's epoch equals, restricted to the ratings this process holds (info[0] entries). */
int mfx_dsgd_get_order(const mfx_dsgd *d, int64_t *perm_out);
int mfx_dsgd_set_factors(mfx_dsgd *d, const double *u, const double *v, const double *items_bias,
                         const double *users_bias);
/* se_out: nullable f64[nbr_epochs], squared-error sum of every epoch over all ratings of the job */
int mfx_dsgd_run_kmf(mfx_dsgd *d, int mode, int nbr_epochs, double learning_rate, double K_users,
                     double K_items, double K_bias, double *se_out);
/* The averaging scheme of mfx_sgd_run_kmf_sharded (gradient_descent.py:124-211: user shards, parameter averaging) with the shard's
 * local epoch on the block order: `d` was created with world = 1 over ONE rank's ratings (local user ids) and any number of item
 * blocks; per epoch and item block the block is trained, then its rows and biases become start + scale * allreduce_sum(delta) over
 * the communicator of mfx_comm_init (none / world 1: no combine).  se_out (nullable): f64[nbr_epochs], this rank's squared errors. */
int mfx_dsgd_run_kmf_averaged(mfx_dsgd *d, int mode, int nbr_epochs, double learning_rate, double K_users, double K_items,
                              double K_bias, double scale, double *se_out);
/* The Funk per-feature loops (estimator_loop_without_bias / _with_bias / _with_bias_dev, gd_estimator.pyx:691-779, :489-685; same
 * arguments as mfx_sgd_run_funk) on the ITEM-BLOCK order of a one-process object (world = 1): a pass of a feature = the item blocks
 * one after the other on the one user side, one stopping rule over the whole pass -- the reference's loop on the order
 * mfx_dsgd_get_order reports, bit for bit.  For catalogues with more items than the resident-side pass has lanes (~450 K): every
 * block gets the persistent pass.  use_bias: the biases of mfx_dsgd_set_factors are the (read-only) inputs of _with_bias. */
int mfx_dsgd_run_funk(mfx_dsgd *d, int min_epochs, double min_improvement, double f_init, double learning_rate, double K,
                      double overall_avg, int use_bias, int update_users, int update_items, int epoch_cap, double *rmse_hist,
                      int64_t hist_cap, int64_t *n_hist);
int mfx_dsgd_get_factors(mfx_dsgd *d, double *u, double *v, double *items_bias, double *users_bias);

#ifdef __cplusplus
}
#endif
#endif /* MFX_H */
