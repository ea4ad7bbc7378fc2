"""TEST INFRASTRUCTURE ONLY -- single-process, G-rank CPU simulation of the user-sharded
SGD scheme (mfrec_amd/sharding.py), with the oracle as the local epoch.  G = 1 is the
sequential oracle itself."""
import numpy as np

from mfrec_amd import sharding as SH
from oracle import oracle as O


def simulate(mode, world, nbr_epochs, k, lr, Ku, Ki, Kb, u0, v0, ib0, ub0, ratings_index,
             ratings, scale=None, force_combine=False):
    """-> (u, v, ib, ub) after nbr_epochs; v/ub assembled from the shards."""
    nu = v0.shape[1]
    scale = 1.0 / world if scale is None else scale
    fn = O.train_linear_kernel if mode == 0 else O.train_logistic_kernel
    bounds = SH.user_blocks(ratings_index, nu, world)
    shards = [SH.shard(ratings_index, ratings, bounds, g) for g in range(world)]
    u, ib = u0.copy(), ib0.copy()
    v, ub = v0.copy(), ub0.copy()
    for _ in range(nbr_epochs):
        du, dib = np.zeros_like(u), np.zeros_like(ib)
        for g, (idx, r, lo, n) in enumerate(shards):
            ug, ibg = u.copy(), ib.copy()
            vg = np.ascontiguousarray(v[:, lo:lo + n]); ubg = np.ascontiguousarray(ub[lo:lo + n])
            if r.shape[0]:
                fn(1, k, 0.1, lr, 0, 0, Ku, Ki, Kb, 0.0, ug, vg, idx, r, ibg, ubg)
            v[:, lo:lo + n] = vg; ub[lo:lo + n] = ubg
            if world == 1 and not force_combine:
                u, ib = ug, ibg
            else:
                du += ug - u
                dib += ibg - ib
        if world > 1 or force_combine:
            u = SH.combine(u, du, scale)
            ib = SH.combine(ib, dib, scale)
    return u, v, ib, ub
