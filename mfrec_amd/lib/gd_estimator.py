"""MI355X drop-in for mfrec/lib/gd_estimator.pyx (Funk-style per-feature SGD)."""
import ctypes as C

import numpy as np

from .. import _shim as S

# Safety bound on epochs per feature.  The reference's `while` (gd_estimator.pyx:742)
# ignores max_epochs and may spin for a very long time; 0 keeps its behaviour.
EPOCH_CAP = 0


def _hist(min_epochs, dim):
    return np.zeros(max(1, int(min_epochs)) * max(1, int(dim)) * 4 + 1024, dtype=np.float64)


def _print_hist(verbose, dim, hist, n):
    # the reference prints the last RMSE of each feature (gd_estimator.pyx:767-768); without
    # per-feature epoch counts we print the whole history instead
    if verbose:
        for x in hist[:n]:
            print("RMSE: " + str(x) + "\n")


def _funk(min_epochs, min_improvement, dim, f_init, lr, K, mu, u, v, ratings_index, ratings,
          items_bias, users_bias, upd_u, upd_i, verbose):
    pu, pv, nu, ni = S.factor_args(u, v, int(dim))
    pi, pr, nnz = S.ratings_args(ratings_index, ratings)
    pib = None if items_bias is None else S.f64(items_bias, "items_bias", 1, (ni,))
    pub = None if users_bias is None else S.f64(users_bias, "users_bias", 1, (nu,))
    hist = _hist(min_epochs, dim)
    n = C.c_int64(0)
    S.check(S.lib().mfx_funk(int(min_epochs), float(min_improvement), int(dim), float(f_init),
                             float(lr), float(K), float(mu), pu, pv, pi, pr, pib, pub, nnz, nu, ni,
                             int(bool(upd_u)), int(bool(upd_i)), int(EPOCH_CAP),
                             S.f64(hist, "hist", 1), hist.shape[0], C.byref(n)))
    _print_hist(verbose, dim, hist, min(n.value, hist.shape[0]))
    return hist[:min(n.value, hist.shape[0])]


def estimator_loop_without_bias(min_epochs, max_epochs, min_improvement, dim, f_init,
                                learning_rate, K, u, v, ratings_index, ratings, nbr_users,
                                nbr_items, verbose=0):
    """gd_estimator.pyx:691-779 (default GDRecommender.train)."""
    _funk(min_epochs, min_improvement, dim, f_init, learning_rate, K, 1.0, u, v, ratings_index,
          ratings, None, None, 1, 1, verbose)


def estimator_loop_with_bias(min_epochs, max_epochs, min_improvement, dim, f_init, learning_rate,
                             learning_rate_users, learning_rate_items, K, overall_avg, u, v,
                             ratings_index, ratings, items_bias, users_bias, nbr_users, nbr_items,
                             verbose=0):
    """gd_estimator.pyx:489-582: biases are read-only inputs."""
    _funk(min_epochs, min_improvement, dim, f_init, learning_rate, K, overall_avg, u, v,
          ratings_index, ratings, items_bias, users_bias, 1, 1, verbose)


def estimator_loop_with_bias_dev(min_epochs, max_epochs, min_improvement, dim, f_init,
                                 learning_rate, learning_rate_users, learning_rate_items, K,
                                 overall_avg, u, v, ratings_index, ratings, items_bias, users_bias,
                                 nbr_users, nbr_items, update_users=1, update_items=1, verbose=0):
    """gd_estimator.pyx:588-685."""
    _funk(min_epochs, min_improvement, dim, f_init, learning_rate, K, overall_avg, u, v,
          ratings_index, ratings, items_bias, users_bias, update_users, update_items, verbose)


def estimator_loop_with_learned_bias(min_epochs, max_epochs, min_improvement, dim, f_init,
                                     learning_rate, learning_rate_users, learning_rate_items,
                                     K_feature, K_bias, overall_avg, u, v, ratings_index, ratings,
                                     items_bias, users_bias, nbr_users, nbr_items, verbose=0,
                                     learning_mode=0):
    """gd_estimator.pyx:401-483."""
    pu, pv, nu, ni = S.factor_args(u, v, int(dim))
    pi, pr, nnz = S.ratings_args(ratings_index, ratings)
    pib = S.f64(items_bias, "items_bias", 1, (ni,))
    pub = S.f64(users_bias, "users_bias", 1, (nu,))
    hist = _hist(min_epochs, dim)
    n = C.c_int64(0)
    S.check(S.lib().mfx_learned_bias(int(min_epochs), float(min_improvement), int(dim),
                                     float(f_init), float(learning_rate),
                                     float(learning_rate_users), float(learning_rate_items),
                                     float(K_feature), float(K_bias), float(overall_avg), pu, pv,
                                     pi, pr, pib, pub, nnz, nu, ni, int(EPOCH_CAP),
                                     S.f64(hist, "hist", 1), hist.shape[0], C.byref(n)))
    _print_hist(verbose, dim, hist, min(n.value, hist.shape[0]))


def funk_rmse_history(min_epochs, min_improvement, dim, f_init, learning_rate, K, u, v,
                      ratings_index, ratings, overall_avg=1.0, items_bias=None, users_bias=None,
                      update_users=1, update_items=1):
    """Extension: run the Funk loop and return the per-epoch online RMSE history."""
    return _funk(min_epochs, min_improvement, dim, f_init, learning_rate, K, overall_avg, u, v,
                 ratings_index, ratings, items_bias, users_bias, update_users, update_items, 0)


def _dense_args(u, v, dim, ratings_index, ratings):
    pu, pv, nu, ni = S.factor_args(u, v, int(dim))
    pi, pr, nnz = S.ratings_args(ratings_index, ratings)
    return pu, pv, nu, ni, pi, pr, nnz


def estimator_loop(min_epochs, max_epochs, min_improvement, dim, f_init, learning_rate, K, u, v,
                   ratings_index, ratings, batch, rmse_hist, nbr_users, nbr_features, verbose=0):
    """gd_estimator.pyx:210-303: dense cache, honours max_epochs, fills rmse_hist in place."""
    pu, pv, nu, ni, pi, pr, nnz = _dense_args(u, v, dim, ratings_index, ratings)
    ph = S.f64(rmse_hist, "rmse_hist", 1)
    if rmse_hist.shape[0] < (int(batch) + 1) * int(max_epochs) * int(dim):
        raise ValueError("rmse_hist is too small for batch/max_epochs/dim")
    S.check(S.lib().mfx_funk_dense(1, int(min_epochs), int(max_epochs), float(min_improvement),
                                   int(dim), float(f_init), float(learning_rate), float(K), pu, pv,
                                   pi, pr, int(batch), ph, nnz, nu, ni))


def estimator_loop2(min_epochs, max_epochs, min_improvement, dim, f_init, learning_rate, K, u, v,
                    ratings_index, ratings, features_bias, nbr_users, nbr_features, verbose=0):
    """gd_estimator.pyx:308-395 (features_bias is accepted and unused, as in the reference)."""
    pu, pv, nu, ni, pi, pr, nnz = _dense_args(u, v, dim, ratings_index, ratings)
    S.check(S.lib().mfx_funk_dense(2, int(min_epochs), int(max_epochs), float(min_improvement),
                                   int(dim), float(f_init), float(learning_rate), float(K), pu, pv,
                                   pi, pr, 0, None, nnz, nu, ni))


def estimator_subloop(f, epochs, min_improvement, dim, f_init, learning_rate, K, u, v,
                      ratings_index, ratings, rating_cache, nbr_users, nbr_features, verbose=0):
    """gd_estimator.pyx:903-962: one epoch of feature f on the caller's dense cache; returns rmse."""
    pu, pv, nu, ni, pi, pr, nnz = _dense_args(u, v, dim, ratings_index, ratings)
    pc = S.f64(rating_cache, "rating_cache", 1, (nu * ni,))
    out = C.c_double(0.0)
    S.check(S.lib().mfx_funk_subloop(int(f), int(dim), float(f_init), float(learning_rate), float(K),
                                     pu, pv, pi, pr, pc, nnz, nu, ni, C.byref(out)))
    return out.value


def predictor_subloop(f, epochs, dim, f_init, u, v, ratings_index, ratings, rating_cache,
                      nbr_users, nbr_features):
    """gd_estimator.pyx:967-995: cache-commit of feature f into the caller's dense cache."""
    pu, pv, nu, ni, pi, pr, nnz = _dense_args(u, v, dim, ratings_index, ratings)
    pc = S.f64(rating_cache, "rating_cache", 1, (nu * ni,))
    S.check(S.lib().mfx_funk_predictor_subloop(int(f), int(dim), float(f_init), pu, pv, pi, pr, pc,
                                               nnz, nu, ni))


def estimator_loop_with_implicit_feedback(min_epochs, max_epochs, min_improvement, dim, f_init, learning_rate,
                                          learning_rate_users, learning_rate_items, K, overall_avg, u, v, y,
                                          ratings_index, ratings, rated, rated_hash, items_bias, users_bias,
                                          nbr_users, nbr_items, verbose=0):
    """gd_estimator.pyx:785-898: the SVD++-like prototype, as written.  `rated` int32[n, 2] (user, item) sorted by
    user with distinct pairs, `rated_hash` int32[nu, 2] (first row, count) -- gradient_descent.py:261-296.  The
    loop is sequential by nature (every rating rewrites y[f, :] for all items of its user): one workgroup walks the
    ratings in order; exact, not fast."""
    import logging
    logging.getLogger("mfrec.recommender").warning(
        "estimator_loop_with_implicit_feedback: this loop has no exact parallel schedule; the GPU walks it sequentially, "
        "bit-identical to the reference but 6-7x SLOWER than the reference's C loop on one host core (DESIGN.md section 8)")
    pu, pv, nu, ni = S.factor_args(u, v, int(dim))
    py = S.f64(y, "y", 2, (int(dim), ni))
    pi, pr, nnz = S.ratings_args(ratings_index, ratings)
    prated = S.i32(rated, "rated", 2)
    if rated.shape[1] != 2:
        raise ValueError("rated must have shape [n, 2]")
    phash = S.i32(rated_hash, "rated_hash", 2, (nu, 2))
    pib = S.f64(items_bias, "items_bias", 1, (ni,))
    pub = S.f64(users_bias, "users_bias", 1, (nu,))
    hist = _hist(min_epochs, dim)
    n = C.c_int64(0)
    S.check(S.lib().mfx_funk_implicit(int(min_epochs), float(min_improvement), int(dim), float(f_init),
                                      float(learning_rate), float(K), float(overall_avg), pu, pv, py, pi, pr,
                                      prated, rated.shape[0], phash, pib, pub, nnz, nu, ni, int(EPOCH_CAP),
                                      S.f64(hist, "hist", 1), hist.shape[0], C.byref(n)))
    _print_hist(verbose, dim, hist, min(n.value, hist.shape[0]))
    return hist[:min(n.value, hist.shape[0])]
