"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/libmfx_oracle.so.

The functions carry the *reference's* names and positional argument orders
(mfrec/lib/kmf_train.pyx, gd_estimator.pyx, als_implicit.pyx) so that parity
tests read like calls into the reference.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; mfrec_amd/ never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmfx_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "mfx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                               src, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.mfxo_estimator_subloop.restype = C.c_double
        _lib.mfxo_rmse.restype = C.c_double
        _lib.mfxo_als_wrmf.restype = C.c_int
    return _lib


def _d(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _i(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def _kmf(mode, nbr_epochs, dim, lr, Ku, Ki, Kb, u, v, idx, r, ib, ub, upd_u, upd_i, rmse):
    ni, nu = u.shape[1], v.shape[1]
    assert u.shape[0] == dim and v.shape[0] == dim
    lib().mfxo_kmf_train(C.c_int(mode), C.c_int(nbr_epochs), C.c_int(dim), C.c_double(lr),
                         C.c_double(Ku), C.c_double(Ki), C.c_double(Kb), _d(u), _d(v), _i(idx),
                         _d(r), _d(ib), _d(ub), C.c_int64(r.shape[0]), C.c_int(nu), C.c_int(ni),
                         C.c_int(upd_u), C.c_int(upd_i), _d(rmse))


def train_linear_kernel(nbr_epochs, dim, f_init, learning_rate, learning_rate_users,
                        learning_rate_items, K_users, K_items, K_bias, overall_avg, u, v,
                        ratings_index, ratings, items_bias, users_bias, update_users=1,
                        update_items=1, verbose=0, rmse_out=None):
    """kmf_train.pyx:195-213."""
    _kmf(0, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v, ratings_index, ratings,
         items_bias, users_bias, update_users, update_items, rmse_out)


def train_logistic_kernel(nbr_epochs, dim, f_init, learning_rate, learning_rate_users,
                          learning_rate_items, K_users, K_items, K_bias, overall_avg, u, v,
                          ratings_index, ratings, items_bias, users_bias, update_users=1,
                          update_items=1, verbose=0, rmse_out=None):
    """kmf_train.pyx:103-121."""
    _kmf(1, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v, ratings_index, ratings,
         items_bias, users_bias, update_users, update_items, rmse_out)


def _funk(min_epochs, min_improvement, dim, f_init, lr, K, mu, u, v, idx, r, ib, ub, upd_u, upd_i,
          epoch_cap, hist):
    ni, nu = u.shape[1], v.shape[1]
    n = C.c_int64(0)
    cap = 0 if hist is None else hist.shape[0]
    lib().mfxo_funk(C.c_int(min_epochs), C.c_double(min_improvement), C.c_int(dim),
                    C.c_double(f_init), C.c_double(lr), C.c_double(K), C.c_double(mu), _d(u), _d(v),
                    _i(idx), _d(r), _d(ib), _d(ub), C.c_int64(r.shape[0]), C.c_int(nu), C.c_int(ni),
                    C.c_int(upd_u), C.c_int(upd_i), C.c_int(epoch_cap), _d(hist), C.c_int64(cap),
                    C.byref(n))
    return n.value


def estimator_loop_without_bias(min_epochs, max_epochs, min_improvement, dim, f_init,
                                learning_rate, K, u, v, ratings_index, ratings, nbr_users,
                                nbr_items, verbose=0, rmse_hist=None, epoch_cap=0):
    """gd_estimator.pyx:691-704."""
    return _funk(min_epochs, min_improvement, dim, f_init, learning_rate, K, 1.0, u, v,
                 ratings_index, ratings, None, None, 1, 1, epoch_cap, rmse_hist)


def estimator_loop_with_bias(min_epochs, max_epochs, min_improvement, dim, f_init, learning_rate,
                             learning_rate_users, learning_rate_items, K, overall_avg, u, v,
                             ratings_index, ratings, items_bias, users_bias, nbr_users, nbr_items,
                             verbose=0, rmse_hist=None, epoch_cap=0):
    """gd_estimator.pyx:489-507."""
    return _funk(min_epochs, min_improvement, dim, f_init, learning_rate, K, overall_avg, u, v,
                 ratings_index, ratings, items_bias, users_bias, 1, 1, epoch_cap, rmse_hist)


def estimator_loop_with_bias_dev(min_epochs, max_epochs, min_improvement, dim, f_init,
                                 learning_rate, learning_rate_users, learning_rate_items, K,
                                 overall_avg, u, v, ratings_index, ratings, items_bias, users_bias,
                                 nbr_users, nbr_items, update_users=1, update_items=1, verbose=0,
                                 rmse_hist=None, epoch_cap=0):
    """gd_estimator.pyx:588-608."""
    return _funk(min_epochs, min_improvement, dim, f_init, learning_rate, K, overall_avg, u, v,
                 ratings_index, ratings, items_bias, users_bias, update_users, update_items,
                 epoch_cap, rmse_hist)


def estimator_loop_with_learned_bias(min_epochs, max_epochs, min_improvement, dim, f_init,
                                     learning_rate, learning_rate_users, learning_rate_items,
                                     K_feature, K_bias, overall_avg, u, v, ratings_index, ratings,
                                     items_bias, users_bias, nbr_users, nbr_items, verbose=0,
                                     learning_mode=0, rmse_hist=None, epoch_cap=0):
    """gd_estimator.pyx:401-421."""
    ni, nu = u.shape[1], v.shape[1]
    n = C.c_int64(0)
    cap = 0 if rmse_hist is None else rmse_hist.shape[0]
    lib().mfxo_learned_bias(C.c_int(min_epochs), C.c_double(min_improvement), C.c_int(dim),
                            C.c_double(f_init), C.c_double(learning_rate),
                            C.c_double(learning_rate_users), C.c_double(learning_rate_items),
                            C.c_double(K_feature), C.c_double(K_bias), C.c_double(overall_avg),
                            _d(u), _d(v), _i(ratings_index), _d(ratings), _d(items_bias),
                            _d(users_bias), C.c_int64(ratings.shape[0]), C.c_int(nu), C.c_int(ni),
                            C.c_int(epoch_cap), _d(rmse_hist), C.c_int64(cap), C.byref(n))
    return n.value


def estimator_loop_with_implicit_feedback(min_epochs, max_epochs, min_improvement, dim, f_init,
                                          learning_rate, learning_rate_users, learning_rate_items, K,
                                          overall_avg, u, v, y, ratings_index, ratings, rated, rated_hash,
                                          items_bias, users_bias, nbr_users, nbr_items, verbose=0,
                                          rmse_hist=None, epoch_cap=0):
    """gd_estimator.pyx:785-806 (the SVD++-like prototype, quirks included)."""
    ni, nu = u.shape[1], v.shape[1]
    n = C.c_int64(0)
    cap = 0 if rmse_hist is None else rmse_hist.shape[0]
    lib().mfxo_implicit_feedback(C.c_int(min_epochs), C.c_double(min_improvement), C.c_int(dim),
                                 C.c_double(f_init), C.c_double(learning_rate), C.c_double(K),
                                 C.c_double(overall_avg), _d(u), _d(v), _d(y), _i(ratings_index), _d(ratings),
                                 _i(rated), _i(rated_hash), _d(items_bias), _d(users_bias),
                                 C.c_int64(ratings.shape[0]), C.c_int(nu), C.c_int(ni), C.c_int(epoch_cap),
                                 _d(rmse_hist), C.c_int64(cap), C.byref(n))
    return n.value


def estimator_subloop(f, epochs, min_improvement, dim, f_init, learning_rate, K, u, v,
                      ratings_index, ratings, rating_cache, nbr_users, nbr_features, verbose=0):
    """gd_estimator.pyx:903-917; returns the epoch's online RMSE."""
    ni, nu = u.shape[1], v.shape[1]
    return lib().mfxo_estimator_subloop(C.c_int(f), C.c_int(dim), C.c_double(f_init),
                                        C.c_double(learning_rate), C.c_double(K), _d(u), _d(v),
                                        _i(ratings_index), _d(ratings), _d(rating_cache),
                                        C.c_int64(ratings.shape[0]), C.c_int(nu), C.c_int(ni))


def predictor_subloop(f, epochs, dim, f_init, u, v, ratings_index, ratings, rating_cache,
                      nbr_users, nbr_features):
    """gd_estimator.pyx:967-977."""
    ni, nu = u.shape[1], v.shape[1]
    lib().mfxo_predictor_subloop(C.c_int(f), C.c_int(dim), C.c_double(f_init), _d(u), _d(v),
                                 _i(ratings_index), _d(rating_cache), C.c_int64(ratings.shape[0]),
                                 C.c_int(nu), C.c_int(ni))


def estimator_loop(min_epochs, max_epochs, min_improvement, dim, f_init, learning_rate, K, u, v,
                   ratings_index, ratings, batch, rmse_hist, nbr_users, nbr_features, verbose=0):
    """gd_estimator.pyx:210-225."""
    ni, nu = u.shape[1], v.shape[1]
    lib().mfxo_estimator_loop(C.c_int(1), C.c_int(min_epochs), C.c_int(max_epochs),
                              C.c_double(min_improvement), C.c_int(dim), C.c_double(f_init),
                              C.c_double(learning_rate), C.c_double(K), _d(u), _d(v),
                              _i(ratings_index), _d(ratings), C.c_int(batch), _d(rmse_hist),
                              C.c_int64(ratings.shape[0]), C.c_int(nu), C.c_int(ni))


def estimator_loop2(min_epochs, max_epochs, min_improvement, dim, f_init, learning_rate, K, u, v,
                    ratings_index, ratings, features_bias, nbr_users, nbr_features, verbose=0):
    """gd_estimator.pyx:308-322."""
    ni, nu = u.shape[1], v.shape[1]
    lib().mfxo_estimator_loop(C.c_int(2), C.c_int(min_epochs), C.c_int(max_epochs),
                              C.c_double(min_improvement), C.c_int(dim), C.c_double(f_init),
                              C.c_double(learning_rate), C.c_double(K), _d(u), _d(v),
                              _i(ratings_index), _d(ratings), C.c_int(0), None,
                              C.c_int64(ratings.shape[0]), C.c_int(nu), C.c_int(ni))


def als_wrmf(nbr_epochs, dim, u, v, m, m_inv, ratings_users_row, ratings_users_col,
             ratings_items_row, ratings_items_col, nbr_users, nbr_items, c_pos=1, k=0.015,
             verbose=0):
    """als_implicit.pyx:211-225."""
    rc = lib().mfxo_als_wrmf(C.c_int(nbr_epochs), C.c_int(dim), _d(u), _d(v),
                             _i(ratings_users_row), C.c_int64(ratings_users_row.shape[0]),
                             _i(ratings_users_col), _i(ratings_items_row),
                             C.c_int64(ratings_items_row.shape[0]), _i(ratings_items_col),
                             C.c_int(nbr_users), C.c_int(nbr_items), C.c_int(c_pos), C.c_double(k))
    if rc != 0:
        raise np.linalg.LinAlgError("Singular matrix")


RMSE_GD, RMSE_KMF_LINEAR, RMSE_KMF_LOGISTIC, RMSE_GD_BIAS, RMSE_WRMF = range(5)


def rmse(kind, u, v, ratings_index, ratings, items_bias=None, users_bias=None, overall_avg=0.0):
    """metrics.py:69-73 over training triples with the family's predict()."""
    ni, nu = u.shape[1], v.shape[1]
    return lib().mfxo_rmse(C.c_int(kind), C.c_int(u.shape[0]), _d(u), _d(v), _i(ratings_index),
                           _d(ratings), _d(items_bias), _d(users_bias), C.c_double(overall_avg),
                           C.c_int64(ratings.shape[0]), C.c_int(nu), C.c_int(ni))


# ------------------------------------------------------------------------------------------------
# Top-N (evaluation side).  The reference's Python-2 harness (mf.py, gradient_descent.py, metrics.py) cannot be imported as a
# package here; the golden g12_topn.npz is written FROM this restatement (oracle/make_golden.py) and pins the HIP
# path to these rules.  Round 4: the restatement itself is PINNED -- oracle/pin_topn.py cuts the reference's own definitions of
# these functions out of its files, translates them with lib2to3 and executes them on the fixture's inputs: 270 queries + 2
# precision / recall runs, item lists exact, scores to 1e-12 (tests/golden/g12_topn_pin.json).
# ------------------------------------------------------------------------------------------------
def predict_one(kind, u, v, item, user, items_bias=None, users_bias=None, overall_avg=0.0, min_rating=1.0,
                max_rating=5.0):
    """The families' predict(item, user): GD `dot + 1.0` (gradient_descent.py:621-634), GD with the BellKor baseline
    (:637-648), KMF linear / logistic (kmf.py:79-94), WRMF plain dot (wrmf.py:67-69)."""
    s = float(np.dot(u[:, item], v[:, user]))
    if kind == RMSE_GD:
        return s + 1.0
    if kind == RMSE_WRMF:
        return s
    if kind == RMSE_GD_BIAS:
        return s + (overall_avg + (items_bias[item] + users_bias[user]))
    s += (items_bias[item] + users_bias[user])
    if kind == RMSE_KMF_LINEAR:
        return s
    return min_rating + (1.0 / (1.0 + np.exp(-s))) * (max_rating - min_rating)      # RMSE_KMF_LOGISTIC


def find_recommended_items(kind, u, v, user_index, rated_items, neighborhood, nbr_recommendations, items_bias=None,
                           users_bias=None, overall_avg=0.0):
    """MFRecommender.find_recommended_items (mf.py:144-193) = GDRecommender.find_user_top_match
    (gradient_descent.py:768-800 with neighborhood = nbr_items), the loop as the reference runs it:

      * candidates are the POSITIONS 0 .. neighborhood-1: the reference draws `test_items = get_items_subset(count)`
        (one np.random.shuffle) and then enumerates it, predicting for the enumeration index, not for the drawn id
        (mf.py:159-167) -- the draw only advances the RNG;
      * already_rated = the user's rated items plus the item whose id equals user_index (`np.r_[already_rated,
        user_index]`, mf.py:161-162): those score 0;
      * NaN scores become 0 (mf.py:176); zero scores are dropped (`nonzero()`, mf.py:179);
      * `sorted(top_results.iteritems(), key=itemgetter(1), reverse=True)` (mf.py:184): descending score; Python's sort
        is stable also with reverse=True, and a CPython-2 dict keyed by small non-negative ints iterates in ascending key
        order, so ties come out by ascending item (for ids beyond the dict's table size the reference's tie order is an
        implementation detail of that dict; ascending is used here throughout).
    -> (items list, scores list), at most nbr_recommendations long."""
    n = int(min(neighborhood, u.shape[1]))
    already = set(int(x) for x in rated_items)
    already.add(int(user_index))
    user_ratings = np.zeros(n)
    for i in range(n):
        if i not in already:
            rating = predict_one(kind, u, v, i, user_index, items_bias, users_bias, overall_avg)
        else:
            rating = 0.0
        user_ratings[i] = rating
    user_ratings[np.where(np.isnan(user_ratings))] = 0.0
    top = [(int(i), float(user_ratings[i])) for i in user_ratings.nonzero()[0]]       # ascending item
    top = sorted(top, key=lambda kv: kv[1], reverse=True)[:nbr_recommendations]       # stable
    return [i for i, _ in top], [s for _, s in top]


def precision_recall(kind, u, v, rated_lists, test_triples, nbr_recommendations, neighborhood, items_bias=None,
                     users_bias=None, overall_avg=0.0):
    """metrics.precision_recall (metrics.py:85-130): test triples (user, item, rating) grouped by user; per user
    hits = |top-N ∩ test items|, precision = hits / N, recall = hits / |test items|; averages over the users, and
    F1 = 2PR / (P + R).  rated_lists[user] = the user's TRAINING items (excluded from the candidates)."""
    by_user = {}
    for row in test_triples:
        by_user.setdefault(int(row[0]), []).append(int(row[1]))
    P = R = 0.0
    for user, items in by_user.items():
        rec, _ = find_recommended_items(kind, u, v, user, rated_lists.get(user, ()), neighborhood, nbr_recommendations,
                                        items_bias, users_bias, overall_avg)
        hit = float(len(set(rec) & set(items)))
        P += hit / nbr_recommendations
        R += hit / len(set(items))
    P, R = P / len(by_user), R / len(by_user)
    return P, R, 2 * (P * R) / (P + R)          # (ZeroDivisionError when nothing was hit, as in the reference)
