/*
 * mfx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the latent-factor training hot
 * path of mlaprise/mfrec (the three Cython modules under mfrec/lib).  It exists
 * only to *check* the HIP path (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing under mfrec_amd/ may import, link or
 * call it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (ALS:
 * to 1e-9 relative, because the reference calls LAPACK through numpy for the
 * k x k inverse) against the reference's own kernels compiled from
 * /root/reference by oracle/build_ref.py -- see oracle/make_golden.py and
 * tests/test_oracle_golden.py (fixtures in tests/golden/).
 *
 * Conventions are the reference's: factors are feature-major C-contiguous
 * f64 [k, n] (u = items, v = users); ratings_index is int32 [nnz, 2] with
 * column 0 = user, column 1 = item; ratings f64 [nnz].
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off (no FMA contraction: the
 * reference is built for baseline x86-64, where a*b+c is two roundings).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define U(f, i) u[(size_t)(f) * (size_t)ni + (size_t)(i)]
#define V(f, j) v[(size_t)(f) * (size_t)nu + (size_t)(j)]

/* mfrec/lib/gd_estimator.pyx:26-35 (same text at kmf_train.pyx:22-31): the
 * min/max arguments are ignored; bounds are hard-coded to [1, 5]. */
static inline double clamp15(double x)
{
    if (x > 5.0) x = 5.0;
    if (x < 1.0) x = 1.0;
    return x;
}

/* mfrec/lib/gd_estimator.pyx:38-73 `estimator`. */
static inline double funk_estimate(double uf, double vf, int dim, int f, double f_init,
                                   double cache, int trailing, double mu, double ib, double ub)
{
    double sum;
    if (cache > 0) sum = cache;
    else sum = mu + ib + ub;
    sum += uf * vf;
    sum = clamp15(sum);
    if (trailing == 1) {
        sum += (dim - f - 1) * f_init * f_init;
        sum = clamp15(sum);
    }
    return sum;
}

/* ------------------------------------------------------------------------
 * a1/a2: mfrec/lib/kmf_train.pyx:195-277 (train_linear_kernel) and :103-189
 * (train_logistic_kernel).  mode 0 = linear, 1 = logistic.
 * rmse_out (may be NULL) receives the online per-epoch RMSE the reference
 * only prints (kmf_train.pyx:273 / :184).
 * ---------------------------------------------------------------------- */
void mfxo_kmf_train(int mode, int nbr_epochs, int dim, double lr, double K_users, double K_items,
                    double K_bias, double *u, double *v, const int32_t *idx, const double *r,
                    double *ib, double *ub, int64_t nnz, int nu, int ni, int update_users,
                    int update_items, double *rmse_out)
{
    for (int epoch = 0; epoch < nbr_epochs; ++epoch) {
        double se = 0.0;
        for (int64_t t = 0; t < nnz; ++t) {
            const int user = idx[2 * t], item = idx[2 * t + 1];
            const double rating = r[t];
            /* full_estimator, kmf_train.pyx:75-97: overall_avg passed as 0.0 */
            double dot = 0.0 + ib[item] + ub[user];
            for (int f = 0; f < dim; ++f) dot += U(f, item) * V(f, user);
            double err, grad;
            if (mode == 0) {
                err = rating - dot;
                grad = err;
            } else {
                const double sig = 1.0 / (1.0 + exp(-dot));
                const double p = 1.0 + sig * 4.0;
                err = rating - p;
                grad = err * sig * (1.0 - sig) * 4.0;
            }
            se += err * err;
            if (mode == 0 || update_users) ub[user] += lr * (grad - K_bias * ub[user]);
            if (mode == 0 || update_items) ib[item] += lr * (grad - K_bias * ib[item]);
            for (int f = 0; f < dim; ++f) {
                const double cf = V(f, user), mf = U(f, item);
                if (update_items) U(f, item) += lr * (grad * cf - K_items * mf);
                if (update_users) V(f, user) += lr * (grad * mf - K_users * cf);
            }
        }
        if (rmse_out) rmse_out[epoch] = sqrt(se / (double)nnz);
    }
}

/* ------------------------------------------------------------------------
 * a3/a4: Funk-style, one feature at a time with a per-rating residual cache.
 *   estimator_loop_without_bias   gd_estimator.pyx:691-779  (ib = ub = NULL, mu = 1.0)
 *   estimator_loop_with_bias      gd_estimator.pyx:489-582
 *   estimator_loop_with_bias_dev  gd_estimator.pyx:588-685  (update gates)
 * The stopping rule (:742) ignores max_epochs; rmse / rmse_last persist over
 * features (init 2.0 / 0.0).  epoch_cap (>0) is OUR safety bound on the total
 * number of epochs per feature; 0 = unbounded as in the reference.
 * rmse_hist (may be NULL) has room for hist_cap entries; *n_hist counts all
 * epochs run (entries beyond hist_cap are dropped).
 * ---------------------------------------------------------------------- */
void mfxo_funk(int min_epochs, double min_improvement, int dim, double f_init, double lr, double K,
               double mu, double *u, double *v, const int32_t *idx, const double *r,
               const double *ib, const double *ub, int64_t nnz, int nu, int ni, int update_users,
               int update_items, int epoch_cap, double *rmse_hist, int64_t hist_cap,
               int64_t *n_hist)
{
    double rmse = 2.0, rmse_last = 0.0;
    int64_t nh = 0;
    double *cache = (double *)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(double));
    for (int f = 0; f < dim; ++f) {
        int epoch = 0;
        while (epoch < min_epochs || rmse <= rmse_last - min_improvement) {
            if (epoch_cap > 0 && epoch >= epoch_cap) break;
            double se = 0.0;
            rmse_last = rmse;
            for (int64_t t = 0; t < nnz; ++t) {
                const int user = idx[2 * t], item = idx[2 * t + 1];
                const double bi = ib ? ib[item] : 0.0, bu = ub ? ub[user] : 0.0;
                const double p = funk_estimate(U(f, item), V(f, user), dim, f, f_init, cache[t], 1,
                                               mu, bi, bu);
                const double err = r[t] - p;
                se += err * err;
                const double cf = V(f, user), mf = U(f, item);
                if (update_items) U(f, item) += lr * (err * cf - K * mf);
                if (update_users) V(f, user) += lr * (err * mf - K * cf);
            }
            rmse = sqrt(se / (double)nnz);
            if (rmse_hist && nh < hist_cap) rmse_hist[nh] = rmse;
            ++nh;
            ++epoch;
        }
        for (int64_t t = 0; t < nnz; ++t) {
            const int user = idx[2 * t], item = idx[2 * t + 1];
            const double bi = ib ? ib[item] : 0.0, bu = ub ? ub[user] : 0.0;
            cache[t] = funk_estimate(U(f, item), V(f, user), dim, f, f_init, cache[t], 0, mu, bi, bu);
        }
    }
    free(cache);
    if (n_hist) *n_hist = nh;
}

/* ------------------------------------------------------------------------
 * a9: estimator_loop_with_implicit_feedback, gd_estimator.pyx:785-898 (the
 * SVD++-like prototype), restated AS WRITTEN, quirks included:
 *   - feedback_sum is summed over the user's rated items (:869-871) and then
 *     overwritten by y[f, item] (:873), so the item update only sees that one
 *     term;
 *   - the inner loops reuse the loop variable `i` (:869, :879); Cython's range
 *     loops iterate over a temporary, so the outer loop is unaffected, and
 *     rating_cache[i] is read before `i` is clobbered (:860-861);
 *   - every rating rewrites y[f, j] for ALL items j its user rated (:879-881);
 *   - the stopping rule is the Funk one (:844; max_epochs ignored), the cache
 *     commit (:891-896) uses v[f, user] alone (no implicit term).
 * rated int32[n_rated, 2] = (user, item) sorted by user; rated_hash
 * int32[nu, 2] = (first row of the user in `rated`, count)
 * (gradient_descent.py:261-296).
 * ---------------------------------------------------------------------- */
void mfxo_implicit_feedback(int min_epochs, double min_improvement, int dim, double f_init, double lr,
                            double K, double mu, double *u, double *v, double *y, const int32_t *idx,
                            const double *r, const int32_t *rated, const int32_t *rated_hash,
                            const double *ib, const double *ub, int64_t nnz, int nu, int ni,
                            int epoch_cap, double *rmse_hist, int64_t hist_cap, int64_t *n_hist)
{
#define Y(f, i) y[(size_t)(f) * (size_t)ni + (size_t)(i)]
    double rmse = 2.0, rmse_last = 0.0;
    int64_t nh = 0;
    double *cache = (double *)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(double));
    for (int f = 0; f < dim; ++f) {
        int epoch = 0;
        while (epoch < min_epochs || rmse <= rmse_last - min_improvement) {
            if (epoch_cap > 0 && epoch >= epoch_cap) break;
            double se = 0.0;
            rmse_last = rmse;
            for (int64_t t = 0; t < nnz; ++t) {
                const int user = idx[2 * t], item = idx[2 * t + 1];
                const long nbr_rated = rated_hash[2 * user + 1];
                const double feedback_norm = 1.0 / sqrt((double)nbr_rated);              /* :853-854 */
                const double v_implicit = V(f, user) + (feedback_norm * Y(f, item));     /* :855 */
                const double p = funk_estimate(U(f, item), v_implicit, dim, f, f_init, cache[t], 1, mu,
                                               ib[item], ub[user]);                      /* :857-858 */
                const double err = 1.0 * r[t] - p;
                se += err * err;
                const double cf = V(f, user);
                double feedback_sum = 0.0;
                for (long j = 0; j < nbr_rated; ++j)                                     /* :869-871 */
                    feedback_sum += Y(f, rated[2 * (rated_hash[2 * user] + j) + 1]);
                feedback_sum = Y(f, item);                                               /* :873 */
                const double mf = U(f, item);
                U(f, item) += lr * (err * (cf + feedback_norm * feedback_sum) - K * mf); /* :876 */
                V(f, user) += lr * (err * mf - K * cf);                                  /* :877 */
                for (long j = 0; j < nbr_rated; ++j) {                                   /* :879-881 */
                    const int sum_j = rated[2 * (rated_hash[2 * user] + j) + 1];
                    Y(f, sum_j) += lr * (err * feedback_norm * mf - K * Y(f, sum_j));
                }
            }
            rmse = sqrt(se / (double)nnz);
            if (rmse_hist && nh < hist_cap) rmse_hist[nh] = rmse;
            ++nh;
            ++epoch;
        }
        for (int64_t t = 0; t < nnz; ++t) {                                              /* :891-896 */
            const int user = idx[2 * t], item = idx[2 * t + 1];
            cache[t] = funk_estimate(U(f, item), V(f, user), dim, f, f_init, cache[t], 0, mu, ib[item],
                                     ub[user]);
        }
    }
    free(cache);
    if (n_hist) *n_hist = nh;
#undef Y
}

/* ------------------------------------------------------------------------
 * a5: estimator_loop_with_learned_bias, gd_estimator.pyx:401-483, with
 * full_estimator (:118-148): full k-dot per rating, but only feature f and
 * both biases are updated.
 * ---------------------------------------------------------------------- */
void mfxo_learned_bias(int min_epochs, double min_improvement, int dim, double f_init, double lr,
                       double lr_users, double lr_items, double K_feature, double K_bias, double mu,
                       double *u, double *v, const int32_t *idx, const double *r, double *ib,
                       double *ub, int64_t nnz, int nu, int ni, int epoch_cap, double *rmse_hist,
                       int64_t hist_cap, int64_t *n_hist)
{
    double rmse = 2.0, rmse_last = 0.0;
    int64_t nh = 0;
    for (int f = 0; f < dim; ++f) {
        int epoch = 0;
        while (epoch < min_epochs || rmse <= rmse_last - min_improvement) {
            if (epoch_cap > 0 && epoch >= epoch_cap) break;
            double se = 0.0;
            rmse_last = rmse;
            for (int64_t t = 0; t < nnz; ++t) {
                const int user = idx[2 * t], item = idx[2 * t + 1];
                double sum = mu + ib[item] + ub[user];
                for (int g = 0; g < dim; ++g) sum += U(g, item) * V(g, user);
                sum = clamp15(sum);
                sum += (dim - f - 1) * f_init * f_init;
                sum = clamp15(sum);
                const double err = r[t] - sum;
                se += err * err;
                const double cf = V(f, user), mf = U(f, item);
                ub[user] += lr_users * (err - K_bias * ub[user]);
                ib[item] += lr_items * (err - K_bias * ib[item]);
                U(f, item) += lr * (err * cf - K_feature * mf);
                V(f, user) += lr * (err * mf - K_feature * cf);
            }
            rmse = sqrt(se / (double)nnz);
            if (rmse_hist && nh < hist_cap) rmse_hist[nh] = rmse;
            ++nh;
            ++epoch;
        }
    }
    if (n_hist) *n_hist = nh;
}

/* ------------------------------------------------------------------------
 * a6: estimator_subloop (gd_estimator.pyx:903-962, returns rmse) and
 * predictor_subloop (:967-995) on a caller-owned DENSE cache f64[nu*ni],
 * indexed user + item*nu.
 * ---------------------------------------------------------------------- */
double mfxo_estimator_subloop(int f, int dim, double f_init, double lr, double K, double *u,
                              double *v, const int32_t *idx, const double *r, double *cache,
                              int64_t nnz, int nu, int ni)
{
    double se = 0.0;
    for (int64_t t = 0; t < nnz; ++t) {
        const int user = idx[2 * t], item = idx[2 * t + 1];
        const double p = funk_estimate(U(f, item), V(f, user), dim, f, f_init,
                                       cache[(size_t)user + (size_t)item * (size_t)nu], 1, 1.0, 0.0, 0.0);
        const double err = 1.0 * r[t] - p;
        se += err * err;
        const double cf = V(f, user), mf = U(f, item);
        V(f, user) += lr * (err * mf - K * cf);
        U(f, item) += lr * (err * cf - K * mf);
    }
    return sqrt(se / (double)nnz);
}

void mfxo_predictor_subloop(int f, int dim, double f_init, const double *u, const double *v,
                            const int32_t *idx, double *cache, int64_t nnz, int nu, int ni)
{
    for (int64_t t = 0; t < nnz; ++t) {
        const int user = idx[2 * t], item = idx[2 * t + 1];
        const size_t c = (size_t)user + (size_t)item * (size_t)nu;
        cache[c] = funk_estimate(U(f, item), V(f, user), dim, f, f_init, cache[c], 0, 1.0, 0.0, 0.0);
    }
}

/* ------------------------------------------------------------------------
 * a7: estimator_loop, gd_estimator.pyx:210-303 -- dense malloc'd cache,
 * honours max_epochs, fills rmse_hist[epoch + f*max_epochs + batch*max_epochs*dim].
 * (estimator_loop2, :308-395, is mfxo_funk's loop rule on the same dense
 * cache; it is reproduced by mode == 2.)
 * ---------------------------------------------------------------------- */
void mfxo_estimator_loop(int mode, int min_epochs, int max_epochs, double min_improvement, int dim,
                         double f_init, double lr, double K, double *u, double *v,
                         const int32_t *idx, const double *r, int batch, double *rmse_hist,
                         int64_t nnz, int nu, int ni)
{
    double rmse = 2.0, rmse_last = 0.0, improvement = 0.0;
    double *cache = (double *)calloc((size_t)nu * (size_t)ni + 1, sizeof(double));
    for (int f = 0; f < dim; ++f) {
        int epoch = 0;
        for (;;) {
            if (mode == 2) {
                if (!(epoch < min_epochs || rmse <= rmse_last - min_improvement)) break;
            } else {
                if (!((epoch < min_epochs || improvement >= min_improvement) && epoch < max_epochs))
                    break;
            }
            rmse_last = rmse;
            rmse = mfxo_estimator_subloop(f, dim, f_init, lr, K, u, v, idx, r, cache, nnz, nu, ni);
            if (mode != 2) {
                rmse_hist[epoch + f * max_epochs + batch * max_epochs * dim] = rmse;
                improvement = rmse_last - rmse;
            }
            ++epoch;
        }
        mfxo_predictor_subloop(f, dim, f_init, u, v, idx, cache, nnz, nu, ni);
    }
    free(cache);
}

/* ------------------------------------------------------------------------
 * k x k inverse by Gauss-Jordan elimination with partial pivoting.  The
 * reference calls np.linalg.inv (LAPACK dgesv on the identity) from inside
 * the Cython loop (als_implicit.pyx:296,346); LAPACK is a third-party
 * dependency, so this is a restatement of the *mathematical* step, compared
 * to the reference's output at 1e-9 relative, not bit-for-bit.
 * a: k*k row-major, destroyed; inv: k*k row-major out.  returns 0 / -1.
 * ---------------------------------------------------------------------- */
static int invert_gj(double *a, double *inv, int k)
{
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) inv[i * k + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < k; ++c) {
        int p = c;
        double best = fabs(a[c * k + c]);
        for (int i = c + 1; i < k; ++i)
            if (fabs(a[i * k + c]) > best) { best = fabs(a[i * k + c]); p = i; }
        if (best == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < k; ++j) {
                double t = a[c * k + j]; a[c * k + j] = a[p * k + j]; a[p * k + j] = t;
                t = inv[c * k + j]; inv[c * k + j] = inv[p * k + j]; inv[p * k + j] = t;
            }
        const double d = 1.0 / a[c * k + c];
        for (int j = 0; j < k; ++j) { a[c * k + j] *= d; inv[c * k + j] *= d; }
        for (int i = 0; i < k; ++i) {
            if (i == c) continue;
            const double m = a[i * k + c];
            if (m == 0.0) continue;
            for (int j = 0; j < k; ++j) {
                a[i * k + j] -= m * a[c * k + j];
                inv[i * k + j] -= m * inv[c * k + j];
            }
        }
    }
    return 0;
}

/* one half-epoch of als_wrmf: solve rows of `x` (f64[k,nx]) from `y` (f64[k,ny]).
 * als_implicit.pyx:257-302 (user pass; item pass :307-352 is the same with
 * roles swapped).  row = [0,c0,c1,...] run-length, consumed as a running sum
 * (:269-270). */
static int als_half(int dim, double *x, int nx, const double *y, int ny, const int32_t *row,
                    int64_t n_active, const int32_t *col, int c_pos, double lambda, double *HH,
                    double *HC, double *HCp, double *m, double *m_inv)
{
    for (int f1 = 0; f1 < dim; ++f1)
        for (int f2 = 0; f2 < dim; ++f2) {
            double d = 0.0;
            for (int i = 0; i < ny; ++i) d += y[(size_t)f1 * ny + i] * y[(size_t)f2 * ny + i];
            HH[f1 + dim * f2] = d;
        }
    int64_t start = 0;
    for (int64_t j = 0; j < n_active; ++j) {
        start += row[j];
        const int64_t span = row[j + 1];
        for (int f1 = 0; f1 < dim; ++f1)
            for (int f2 = 0; f2 < dim; ++f2) {
                double d = 0.0;
                for (int64_t i = 0; i < span; ++i) {
                    const int c = col[start + i];
                    d += y[(size_t)f1 * ny + c] * y[(size_t)f2 * ny + c] * c_pos;
                }
                HC[f1 + dim * f2] = d;
            }
        for (int f = 0; f < dim; ++f) {
            double d = 0.0;
            for (int64_t i = 0; i < span; ++i) d += y[(size_t)f * ny + col[start + i]] * (1 + c_pos);
            HCp[f] = d;
        }
        for (int f1 = 0; f1 < dim; ++f1)
            for (int f2 = 0; f2 < dim; ++f2) {
                double d = HH[f1 + dim * f2] + HC[f1 + dim * f2];
                if (f1 == f2) d += lambda;
                m[f1 * dim + f2] = d;
            }
        if (invert_gj(m, m_inv, dim) != 0) return -1;
        for (int f = 0; f < dim; ++f) {
            double d = 0.0;
            for (int f2 = 0; f2 < dim; ++f2) d += m_inv[f * dim + f2] * HCp[f2];
            x[(size_t)f * nx + j] = d;
        }
    }
    return 0;
}

/* a8: als_wrmf, mfrec/lib/als_implicit.pyx:211-352. */
int mfxo_als_wrmf(int nbr_epochs, int dim, double *u, double *v, const int32_t *urow,
                  int64_t n_urow, const int32_t *ucol, const int32_t *irow, int64_t n_irow,
                  const int32_t *icol, int nu, int ni, int c_pos, double lambda)
{
    double *HH = (double *)malloc(sizeof(double) * dim * dim);
    double *HC = (double *)malloc(sizeof(double) * dim * dim);
    double *HCp = (double *)malloc(sizeof(double) * dim);
    double *m = (double *)malloc(sizeof(double) * dim * dim);
    double *mi = (double *)malloc(sizeof(double) * dim * dim);
    int rc = 0;
    for (int epoch = 0; epoch < nbr_epochs && rc == 0; ++epoch) {
        rc = als_half(dim, v, nu, u, ni, urow, n_urow - 1, ucol, c_pos, lambda, HH, HC, HCp, m, mi);
        if (rc == 0)
            rc = als_half(dim, u, ni, v, nu, irow, n_irow - 1, icol, c_pos, lambda, HH, HC, HCp, m, mi);
    }
    free(HH); free(HC); free(HCp); free(m); free(mi);
    return rc;
}

/* ------------------------------------------------------------------------
 * RMSE over a list of triples with a full-k predictor -- the formula of
 * mfrec/recommendation/metrics.py:69-73 with predict() of
 *   kind 0: GDRecommender.predict_rating       (gradient_descent.py:621-631) dot + 1.0
 *   kind 1: KMF predict_linear                 (kmf.py:88-94)   dot + ib + ub
 *   kind 2: KMF predict_logistic               (kmf.py:79-85)   1 + 4*sigmoid(dot+ib+ub)
 *   kind 3: GD predict_rating_with_bias        (gradient_descent.py:637-648) dot + mu + ib + ub
 *   kind 4: WRMF predict                       (wrmf.py:67-69)  dot
 * np.dot sums pairwise/BLAS-ordered; we sum sequentially -- RMSE agrees to ~1e-15.
 * ---------------------------------------------------------------------- */
double mfxo_rmse(int kind, int dim, const double *u, const double *v, const int32_t *idx,
                 const double *r, const double *ib, const double *ub, double mu, int64_t nnz, int nu,
                 int ni)
{
    double se = 0.0;
    for (int64_t t = 0; t < nnz; ++t) {
        const int user = idx[2 * t], item = idx[2 * t + 1];
        double dot = 0.0;
        for (int f = 0; f < dim; ++f) dot += U(f, item) * V(f, user);
        double p;
        switch (kind) {
        case 0: p = dot + 1.0; break;
        case 1: p = dot + (ib[item] + ub[user]); break;
        case 2: p = 1.0 + (1.0 / (1.0 + exp(-(dot + (ib[item] + ub[user]))))) * 4.0; break;
        case 3: p = dot + mu + (ib[item] + ub[user]); break;
        default: p = dot; break;
        }
        const double e = r[t] - p;
        se += e * e;
    }
    return sqrt(se / (double)nnz);
}
