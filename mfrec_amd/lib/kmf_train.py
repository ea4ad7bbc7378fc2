"""MI355X drop-in for mfrec/lib/kmf_train.pyx: same function names, same positional
argument order, same in-place semantics (u, v, items_bias, users_bias are updated; None
is returned)."""
import numpy as np

from .. import _shim as S


def _run(mode, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v, ratings_index,
         ratings, items_bias, users_bias, update_users, update_items, verbose):
    pu, pv, nu, ni = S.factor_args(u, v, int(dim))
    pi, pr, nnz = S.ratings_args(ratings_index, ratings)
    pib = S.f64(items_bias, "items_bias", 1, (ni,))
    pub = S.f64(users_bias, "users_bias", 1, (nu,))
    n_ep = max(int(nbr_epochs), 0)
    rmse = np.zeros(max(n_ep, 1), dtype=np.float64)
    S.check(S.lib().mfx_kmf_train(mode, n_ep, int(dim), float(learning_rate), float(K_users),
                                  float(K_items), float(K_bias), pu, pv, pi, pr, pib, pub, nnz, nu,
                                  ni, int(bool(update_users)), int(bool(update_items)),
                                  S.f64(rmse, "rmse", 1)))
    if verbose:
        for e in range(n_ep):
            if mode == 1:
                print("EPOCHS: " + str(e + 1))          # kmf_train.pyx:187-189
            print("RMSE: " + str(rmse[e]) + "\n")       # kmf_train.pyx:276-277
    return rmse[:n_ep]


def train_linear_kernel(nbr_epochs, dim, f_init, learning_rate, learning_rate_users,
                        learning_rate_items, K_users, K_items, K_bias, overall_avg, u, v,
                        ratings_index, ratings, items_bias, users_bias, update_users=1,
                        update_items=1, verbose=0):
    """kmf_train.pyx:195-277.  f_init, learning_rate_users/items and overall_avg are
    accepted and ignored, exactly as the reference does."""
    _run(0, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v, ratings_index, ratings,
         items_bias, users_bias, update_users, update_items, verbose)


def train_logistic_kernel(nbr_epochs, dim, f_init, learning_rate, learning_rate_users,
                          learning_rate_items, K_users, K_items, K_bias, overall_avg, u, v,
                          ratings_index, ratings, items_bias, users_bias, update_users=1,
                          update_items=1, verbose=0):
    """kmf_train.pyx:103-189."""
    _run(1, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v, ratings_index, ratings,
         items_bias, users_bias, update_users, update_items, verbose)


def train_kernel_rmse(mode, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v,
                      ratings_index, ratings, items_bias, users_bias, update_users=1,
                      update_items=1):
    """Extension: same as the two kernels above but returns the per-epoch online RMSE
    array (the reference only prints it)."""
    return _run(mode, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v,
                ratings_index, ratings, items_bias, users_bias, update_users, update_items, 0)
