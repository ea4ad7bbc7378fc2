"""TEST INFRASTRUCTURE ONLY -- what a DSGD (block-rotation) run must equal: the oracle's sequential
loop on the block order of mfrec_amd.sharding.dsgd_block_order (sub-epoch major, then GPU, then the
given order inside a block).  No simulation of ranks is needed: the G blocks of a sub-epoch share no
user and no item, so the multi-GPU epoch IS this sequential epoch."""
import numpy as np

from mfrec_amd import sharding as SH
from oracle import oracle as O


def simulate(mode, nbr_epochs, k, lr, Ku, Ki, Kb, u0, v0, ib0, ub0, ratings_index, ratings,
             user_bounds, item_bounds, perm=None):
    """-> (u, v, ib, ub, rmse[nbr_epochs]) of the sequential oracle on the DSGD block order (or on the
    explicit order `perm`, e.g. DsgdKMF.order() of a balanced run)."""
    if perm is None:
        perm = SH.dsgd_block_order(ratings_index, user_bounds, item_bounds)
    idx = np.ascontiguousarray(ratings_index[perm])
    r = np.ascontiguousarray(ratings[perm])
    u, v, ib, ub = u0.copy(), v0.copy(), ib0.copy(), ub0.copy()
    rm = np.zeros(max(nbr_epochs, 1))
    fn = O.train_linear_kernel if mode == 0 else O.train_logistic_kernel
    fn(nbr_epochs, k, 0.1, lr, 0, 0, Ku, Ki, Kb, 0.0, u, v, idx, r, ib, ub, 1, 1, 0, rm)
    return u, v, ib, ub, rm[:nbr_epochs]
