"""MI355X drop-in for mfrec/lib/als_implicit.pyx (WRMF / implicit-feedback ALS)."""
from .. import _shim as S


def als_wrmf(nbr_epochs, dim, u, v, m, m_inv, ratings_users_row, ratings_users_col,
             ratings_items_row, ratings_items_col, nbr_users, nbr_items, c_pos=1, k=0.015,
             verbose=0, sharded=False):
    """als_implicit.pyx:211-352.  `m` / `m_inv` are the reference's k x k scratch arrays; the
    device keeps its own scratch, so they are accepted and left untouched.  `k` is lambda."""
    pu, pv, nu, ni = S.factor_args(u, v, int(dim))
    if (nu, ni) != (int(nbr_users), int(nbr_items)):
        raise ValueError("u/v shapes disagree with nbr_users/nbr_items")
    ur = S.i32(ratings_users_row, "ratings_users_row", 1)
    uc = S.i32(ratings_users_col, "ratings_users_col", 1)
    ir = S.i32(ratings_items_row, "ratings_items_row", 1)
    ic = S.i32(ratings_items_col, "ratings_items_col", 1)
    if verbose:
        for e in range(int(nbr_epochs)):
            print("Epoch : " + str(e))                       # als_implicit.pyx:254-255
    fn = S.lib().mfx_als_wrmf_sharded if sharded else S.lib().mfx_als_wrmf
    S.check(fn(int(nbr_epochs), int(dim), pu, pv, ur,
                                 ratings_users_row.shape[0], uc, ir, ratings_items_row.shape[0],
                                 ic, nu, ni, int(c_pos), float(k)))


def als_wrmf_sharded(*args, **kw):
    """Multi-GPU variant (one process per GPU, mfx_comm_init done): rows of each half-epoch are
    split over the ranks and all-gathered; same arguments and result as als_wrmf."""
    kw["sharded"] = True
    return als_wrmf(*args, **kw)
