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
