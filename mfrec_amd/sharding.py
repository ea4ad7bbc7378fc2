"""User-sharded multi-GPU SGD: host-side logic.

Users are split into `world` contiguous blocks balanced by rating count; rank g trains on
the ratings of its own users (in their original relative order) against a full replica of
the item factors, and after every epoch the item-side state is combined as

    U <- U_start + scale * sum_g (U_g - U_start)        (same for items_bias)

with scale = 1/world (parameter averaging -- the scheme of the reference's two-"machine"
prototype, gradient_descent.py:124-211) or 1 (delta sum).  world == 1 is the sequential
algorithm itself (no combine step is applied, so it stays bit-identical to the oracle).

On the GPU the epoch loop and the RCCL all-reduce run inside libmfx
(mfx_sgd_run_kmf_sharded); `epoch_loop` below is the same arithmetic with the local epoch
and the all-reduce injected, which is what the CPU/gloo tests drive.
"""
import numpy as np


def user_blocks(ratings_index, nbr_users, world):
    """-> int64[world+1] user boundaries: contiguous blocks with ~equal rating counts."""
    counts = np.bincount(ratings_index[:, 0], minlength=nbr_users).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(counts)])
    total = int(csum[-1])
    bounds = [0]
    for g in range(1, world):
        target = total * g // world
        b = int(np.searchsorted(csum, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), nbr_users))
    bounds.append(nbr_users)
    return np.asarray(bounds, dtype=np.int64)


def shard(ratings_index, ratings, bounds, rank):
    """Rank's ratings (order preserved) with user ids made local to its block."""
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sel = (ratings_index[:, 0] >= lo) & (ratings_index[:, 0] < hi)
    idx = np.ascontiguousarray(ratings_index[sel]).copy()
    idx[:, 0] -= lo
    return idx, np.ascontiguousarray(ratings[sel]), lo, hi - lo


def combine(start, local_sum_of_deltas, scale):
    """start + scale * sum_of_deltas -- elementwise, one rounding per operation."""
    return start + scale * local_sum_of_deltas


def epoch_loop(nbr_epochs, u, items_bias, local_epoch, allreduce_sum, world, scale):
    """Generic driver.  local_epoch() runs one SGD epoch in place on this rank's (u, v, ib, ub);
    allreduce_sum(a) returns the element-wise sum of `a` over ranks.  u / items_bias are the
    [k, ni] / [ni] replicas, updated in place."""
    for _ in range(nbr_epochs):
        if world == 1:
            local_epoch()
            continue
        u_start, ib_start = u.copy(), items_bias.copy()
        local_epoch()
        du = allreduce_sum(u - u_start)
        dib = allreduce_sum(items_bias - ib_start)
        u[...] = combine(u_start, du, scale)
        items_bias[...] = combine(ib_start, dib, scale)


class ShardedKMF(object):
    """One rank of a user-sharded full-k SGD run on its own GPU (RCCL inside libmfx)."""

    def __init__(self, ratings_index, ratings, nbr_users, nbr_items, dim, rank, world,
                 bounds=None):
        from .plan import SgdPlan
        self.rank, self.world, self.k = int(rank), int(world), int(dim)
        self.bounds = user_blocks(ratings_index, nbr_users, world) if bounds is None else bounds
        idx, r, self.user_lo, self.nu_local = shard(ratings_index, ratings, self.bounds, rank)
        self.ni = int(nbr_items)
        self.nnz_local = r.shape[0]
        self.plan = SgdPlan(idx, r, max(self.nu_local, 1), nbr_items, dim)

    def set_factors(self, u, v_full, items_bias, users_bias_full):
        lo, n = self.user_lo, self.nu_local
        self.plan.set_factors(u, np.ascontiguousarray(v_full[:, lo:lo + max(n, 1)]), items_bias,
                              np.ascontiguousarray(users_bias_full[lo:lo + max(n, 1)]))

    def train(self, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale=None):
        scale = 1.0 / self.world if scale is None else scale
        return self.plan.run_kmf_sharded(mode, nbr_epochs, learning_rate, K_users, K_items, K_bias,
                                         scale)

    def get_local(self):
        u = np.zeros((self.k, self.ni)); v = np.zeros((self.k, max(self.nu_local, 1)))
        ib = np.zeros(self.ni); ub = np.zeros(max(self.nu_local, 1))
        self.plan.get_factors(u, v, ib, ub)
        return u, v, ib, ub

    def close(self):
        self.plan.close()
