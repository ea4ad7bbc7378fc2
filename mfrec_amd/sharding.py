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


class FakeWorldKMF(object):
    """All `world` shards of the averaging scheme on ONE GPU (test / rehearsal hook): the same
    delta / combine kernels as the RCCL path, with a rank-ordered on-device sum standing in for the
    all-reduce -- so the result equals oracle/sharded_sim.py bit for bit."""

    def __init__(self, ratings_index, ratings, nbr_users, nbr_items, dim, world):
        from .plan import SgdPlan
        self.world, self.k, self.nu, self.ni = int(world), int(dim), int(nbr_users), int(nbr_items)
        self.bounds = user_blocks(ratings_index, nbr_users, world)
        self.shards = [shard(ratings_index, ratings, self.bounds, g) for g in range(world)]
        self.plans = [SgdPlan(idx, r, max(n, 1), nbr_items, dim) for idx, r, lo, n in self.shards]

    def set_factors(self, u, v_full, items_bias, users_bias_full):
        for p, (_, _, lo, n) in zip(self.plans, self.shards):
            p.set_factors(u, np.ascontiguousarray(v_full[:, lo:lo + max(n, 1)]), items_bias,
                          np.ascontiguousarray(users_bias_full[lo:lo + max(n, 1)]))

    def train(self, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale=None):
        from .plan import run_kmf_fake_world
        scale = 1.0 / self.world if scale is None else scale
        return run_kmf_fake_world(self.plans, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale)

    def get_factors(self):
        u = np.zeros((self.k, self.ni)); ib = np.zeros(self.ni)
        v = np.zeros((self.k, self.nu)); ub = np.zeros(self.nu)
        for g, (p, (_, _, lo, n)) in enumerate(zip(self.plans, self.shards)):
            ug = np.zeros((self.k, self.ni)); vg = np.zeros((self.k, max(n, 1)))
            ibg = np.zeros(self.ni); ubg = np.zeros(max(n, 1))
            p.get_factors(ug, vg, ibg, ubg)
            if g == 0:
                u, ib = ug, ibg
            v[:, lo:lo + n] = vg[:, :n]; ub[lo:lo + n] = ubg[:n]
        return u, v, ib, ub

    def close(self):
        for p in self.plans:
            p.close()


# ------------------------------------------------------------------------------------------
# Exact multi-GPU mode: DSGD block rotation (include/mfx.h "exact multi-GPU mode")
# ------------------------------------------------------------------------------------------
def item_blocks(ratings_index, nbr_items, world):
    """-> int64[world+1] item boundaries: contiguous blocks with ~equal rating counts."""
    sw = np.ascontiguousarray(ratings_index[:, ::-1])
    return user_blocks(sw, nbr_items, world)


def dsgd_block_order(ratings_index, user_bounds, item_bounds):
    """The sequential rating order a DSGD epoch is bit-identical to (given order inside the blocks): step
    h = 0..B-1, inside it GPU g = 0..G-1 with its block (g, (stride g + h) mod B), stride = B / G, inside a
    block the given order.  B = len(item_bounds) - 1 is G or 2 G.  -> int64 perm."""
    G, B = len(user_bounds) - 1, len(item_bounds) - 1
    stride = B // G
    ub = np.searchsorted(user_bounds, ratings_index[:, 0], side="right") - 1
    jb = np.searchsorted(item_bounds, ratings_index[:, 1], side="right") - 1
    # empty blocks repeat a bound: searchsorted(right) lands on the LAST block starting at that id,
    # which is the non-empty one -- the same choice as the library's binary search
    step = (jb - stride * ub) % B
    return np.lexsort((np.arange(ratings_index.shape[0]), ub, step)).astype(np.int64)


def dsgd_epoch_loop(nbr_epochs, rank, world, n_item_blocks, train_block, post_transfer, wait_block):
    """Generic DSGD driver (what mfx_dsgd_run_kmf does inside libmfx).  In step h this rank trains item block
    j = (stride rank + h) mod B (after wait_block(j), which blocks until a transfer into j has landed), then
    post_transfer(j_out, j_in) starts sending j_out to rank-1 and receiving j_in = (stride (rank+1) + h) mod B
    from rank+1 -- asynchronously: with B = 2 world the received block is needed only two steps later."""
    B = n_item_blocks
    stride = B // world
    for _ in range(nbr_epochs):
        for h in range(B):
            j = (stride * rank + h) % B
            wait_block(j)
            train_block(j)
            if world > 1:
                post_transfer(j, (stride * (rank + 1) + h) % B)


class DsgdKMF(object):
    """Exact multi-GPU full-k SGD (mfx_dsgd_*).  rank = -1: every rank in this process on this GPU."""

    def __init__(self, ratings_index, ratings, nbr_users, nbr_items, dim, rank, world,
                 user_bounds=None, item_bounds=None, overlap=False, balanced=False, item_blocks_1gpu=None):
        """overlap: 2 * world item blocks (a block rests one step in transit: hand-offs under the next step's
        levels) instead of world -- twice as many, half as large dependency levels, which at BASELINE cfg 3 costs
        more than the hidden hand-offs save, hence off by default; balanced: every block in
        its own minimum-level order.  Explicit item_bounds decide the block count themselves (world + 1 or
        2 * world + 1 entries)."""
        import ctypes as C
        from . import _shim as S
        self.rank, self.world, self.k = int(rank), int(world), int(dim)
        self.nu, self.ni = int(nbr_users), int(nbr_items)
        self.user_bounds = np.ascontiguousarray(
            user_blocks(ratings_index, nbr_users, world) if user_bounds is None else user_bounds, dtype=np.int64)
        nb = (2 * self.world if (overlap and self.world > 1) else self.world)
        if item_blocks_1gpu is not None:
            # ONE GPU, catalogue larger than the register file: the items in n blocks (balanced by rating count), the epoch = the
            # sequential loop on the item-block order (block 0's ratings in the given order, then block 1's, ...); every block small
            # enough runs the static resident-side epoch (its items resident, the users travel)
            # (with rank = -1 and world = G > 1: G user blocks as well -- a user side whose rows exceed the 2 GB a travelling side may
            #  have -- and any multiple of G item blocks; the order is dsgd_block_order's)
            # (real ranks: the same block counts keep every GPU's blocks small enough for the static epoch)
            nb = int(item_blocks_1gpu)
            if nb % self.world:
                raise ValueError("item_blocks_1gpu must be a multiple of world")
        self.item_bounds = np.ascontiguousarray(
            item_blocks(ratings_index, nbr_items, nb) if item_bounds is None else item_bounds, dtype=np.int64)
        self.n_item_blocks = len(self.item_bounds) - 1
        pi, pr, nnz = S.ratings_args(ratings_index, ratings)
        self._h = S.lib().mfx_dsgd_create(pi, pr, nnz, self.nu, self.ni, self.k,
                                          self.user_bounds.ctypes.data_as(S._lp),
                                          self.item_bounds.ctypes.data_as(S._lp), self.n_item_blocks,
                                          self.rank, self.world, 1 if balanced else 0)
        if not self._h:
            msg = S.last_error()
            raise (ValueError if ("outside" in msg or "bad argument" in msg or "bounds" in msg) else S.MfxError)(msg)
        g = max(self.rank, 0)
        self.user_lo = 0 if self.rank < 0 else int(self.user_bounds[g])
        self.nu_local = self.nu if self.rank < 0 else int(self.user_bounds[g + 1] - self.user_bounds[g])

    def info(self):
        import ctypes as C
        from . import _shim as S
        a = (C.c_int64 * 8)()
        S.check(S.lib().mfx_dsgd_info(self._h, a))
        return dict(zip(("nnz_local", "levels_total", "max_block", "min_block", "launches_per_epoch", "item_blocks",
                         "resident_blocks"), [int(x) for x in a][:7]))

    def order(self):
        """int64 perm: the sequential rating order this object's epoch is bit-identical to (the ratings this
        process holds; indices into the arrays given to the constructor)."""
        from . import _shim as S
        perm = np.empty(self.info()["nnz_local"], dtype=np.int64)
        S.check(S.lib().mfx_dsgd_get_order(self._h, perm.ctypes.data_as(S._lp)))
        return perm

    def _args(self, u, v, items_bias, users_bias):
        from . import _shim as S
        pu = S.f64(u, "u", 2, (self.k, self.ni))
        pv = S.f64(v, "v", 2, (self.k, self.nu_local))
        pib = S.f64(items_bias, "items_bias", 1, (self.ni,), allow_none=True)
        pub = S.f64(users_bias, "users_bias", 1, (self.nu_local,), allow_none=True)
        return pu, pv, pib, pub

    def set_factors(self, u, v, items_bias=None, users_bias=None):
        from . import _shim as S
        S.check(S.lib().mfx_dsgd_set_factors(self._h, *self._args(u, v, items_bias, users_bias)))

    def train(self, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias):
        """-> f64[nbr_epochs] squared-error sums over all ratings of the job (online)."""
        from . import _shim as S
        se = np.zeros(max(int(nbr_epochs), 1))
        S.check(S.lib().mfx_dsgd_run_kmf(self._h, int(mode), int(nbr_epochs), float(learning_rate),
                                         float(K_users), float(K_items), float(K_bias), S.f64(se, "se", 1)))
        return se[:int(nbr_epochs)]

    def train_averaged(self, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale):
        """world = 1 objects only (ONE rank's ratings): the averaging scheme with the local epoch on the block order -- per item block
        an all-reduce of its deltas over the communicator of mfx_comm_init (mfx_dsgd_run_kmf_averaged).  -> this rank's squared-error
        sums per epoch."""
        from . import _shim as S
        se = np.zeros(max(int(nbr_epochs), 1))
        S.check(S.lib().mfx_dsgd_run_kmf_averaged(self._h, int(mode), int(nbr_epochs), float(learning_rate), float(K_users),
                                                  float(K_items), float(K_bias), float(scale), S.f64(se, "se", 1)))
        return se[:int(nbr_epochs)]

    def train_funk(self, min_epochs, min_improvement, f_init, learning_rate, K, overall_avg=1.0, use_bias=False,
                   update_users=True, update_items=True, epoch_cap=0):
        """world = 1 objects only: the Funk per-feature loops (estimator_loop_without_bias / _with_bias[_dev]) on the item-block order
        -- a pass of a feature = the blocks one after the other (mfx_dsgd_run_funk).  -> f64[] RMSE after every (feature, epoch)."""
        import ctypes as C
        from . import _shim as S
        cap = max(1, self.k * max(int(min_epochs), int(epoch_cap), 64))
        for _ in range(2):
            hist = np.zeros(cap)
            n = C.c_int64(0)
            S.check(S.lib().mfx_dsgd_run_funk(self._h, int(min_epochs), float(min_improvement), float(f_init), float(learning_rate),
                                              float(K), float(overall_avg), 1 if use_bias else 0, 1 if update_users else 0,
                                              1 if update_items else 0, int(epoch_cap), S.f64(hist, "rmse_hist", 1), cap, C.byref(n)))
            return hist[:min(int(n.value), cap)]

    def get_factors(self, u, v, items_bias=None, users_bias=None):
        from . import _shim as S
        S.check(S.lib().mfx_dsgd_get_factors(self._h, *self._args(u, v, items_bias, users_bias)))

    def close(self):
        if getattr(self, "_h", None):
            from . import _shim as S
            S.lib().mfx_dsgd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


class ShardedKMFBlocks(object):
    """One rank of the user-sharded averaging scheme (ShardedKMF) with its local epoch on the BLOCK ORDER: the rank's shard holds ALL
    items (200 K at cfg 3: too many for the static resident-side epoch, so ShardedKMF's plan runs on the level schedule); cut into
    item blocks every block runs the static epoch, and the item deltas are all-reduced per block (mfx_dsgd_run_kmf_averaged).  Same
    scheme, same arithmetic per element; the shard's local order is the block order (order())."""

    def __init__(self, ratings_index, ratings, nbr_users, nbr_items, dim, rank, world, bounds=None, balanced=True,
                 item_blocks=None, already_local=False):
        """already_local: ratings_index / ratings are THIS rank's shard already (user ids local, nbr_users = its user count)."""
        self.rank, self.world, self.k = int(rank), int(world), int(dim)
        if already_local:
            self.bounds = None
            idx, r, self.user_lo, self.nu_local = ratings_index, ratings, 0, int(nbr_users)
        else:
            self.bounds = user_blocks(ratings_index, nbr_users, world) if bounds is None else bounds
            idx, r, self.user_lo, self.nu_local = shard(ratings_index, ratings, self.bounds, rank)
        self.ni = int(nbr_items)
        self.nnz_local = r.shape[0]
        ub, nb = one_gpu_block_counts(max(self.nu_local, 1), nbr_items, dim)
        if ub != 1:
            raise ValueError("ShardedKMFBlocks: this rank's user rows exceed 2 GB (%d users): more ranks, or ShardedKMF" % self.nu_local)
        nb = int(item_blocks) if item_blocks else nb
        # the SAME item bounds on every rank (an all-reduce pairs block j with block j): equal id ranges, not balanced by a
        # shard's own rating counts
        item_bounds = (np.arange(nb + 1, dtype=np.int64) * int(nbr_items)) // nb
        self.d = DsgdKMF(idx, r, max(self.nu_local, 1), nbr_items, dim, -1, 1, balanced=balanced, item_bounds=item_bounds,
                         item_blocks_1gpu=nb)

    def set_factors(self, u, v_full, items_bias, users_bias_full):
        lo, n = self.user_lo, self.nu_local
        self.d.set_factors(u, np.ascontiguousarray(v_full[:, lo:lo + max(n, 1)]), items_bias,
                           np.ascontiguousarray(users_bias_full[lo:lo + max(n, 1)]))

    def train(self, mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale=None):
        scale = 1.0 / self.world if scale is None else scale
        return self.d.train_averaged(mode, nbr_epochs, learning_rate, K_users, K_items, K_bias, scale)

    def get_local(self):
        u = np.zeros((self.k, self.ni)); v = np.zeros((self.k, max(self.nu_local, 1)))
        ib = np.zeros(self.ni); ub = np.zeros(max(self.nu_local, 1))
        self.d.get_factors(u, v, ib, ub)
        return u, v, ib, ub

    def info(self):
        return self.d.info()

    def order(self):
        return self.d.order()

    def close(self):
        self.d.close()


# ------------------------------------------------------------------------------------------
# ONE GPU, a problem whose smaller side does not fit the register file: the block order
# ------------------------------------------------------------------------------------------
ROWS_PER_BLOCK = 42000          # what the static resident-side epoch holds at k = 64 with headroom (18 slots x 3 072 waves x 7/8)
ROWS_PER_BLOCK_ALONE = 52000    # ... and with the whole grid (one user block: nothing runs beside a block; 18 x 3 072 = 55 296)
TRAVELLING_BYTES = 1.9e9        # rows of a travelling side stay below 2 GB (32-bit buffer offsets)


def one_gpu_block_counts(nbr_users, nbr_items, dim):
    """-> (user blocks, item blocks) for DsgdKMF(rank=-1): item blocks small enough for the static resident-side epoch (the item
    rows of a block live in registers; fewer fit at larger k), user blocks small enough for a travelling side; the item block count
    is a multiple of the user block count."""
    ub = max(1, int(-(-(int(nbr_users) * int(dim) * 8) // TRAVELLING_BYTES)))
    rows = max(1, int((ROWS_PER_BLOCK_ALONE if ub == 1 and ROWS_PER_BLOCK == 42000 else ROWS_PER_BLOCK) * 64 // max(int(dim), 64)))
    nb = max(1, -(-int(nbr_items) // rows))
    return ub, -(-nb // ub) * ub


FUNK_ITEMS_PER_BLOCK = 400000        # a lane per item of the resident-side Funk pass: ~450 K lanes at 7 / 8 of the chip


def funk_block_order(min_epochs, max_epochs, min_improvement, dim, feature_init, learning_rate, K, u, v, ratings_index, ratings,
                     nbr_users, nbr_items, overall_avg=None, items_bias=None, users_bias=None, update_users=True, update_items=True,
                     item_blocks=None, epoch_cap=0):
    """estimator_loop_without_bias (gd_estimator.pyx:691-779; with items_bias / users_bias given: estimator_loop_with_bias[_dev],
    :489-685 -- same positional meaning, max_epochs ignored as the reference ignores it) on ONE GPU on the ITEM-BLOCK order: the
    given ratings grouped by item block (stable), a pass of a feature = block 0's ratings, then block 1's, ... on the one user side.
    For catalogues with more items than the resident-side pass has lanes (BASELINE cfg 5: 1 M items): every block of <= 400 K
    items gets the persistent pass instead of the whole problem falling to one launch per dependency level.  Bit-identical to the
    sequential loop ON THAT ORDER, which is returned (int64 perm) with the RMSE history: (perm, rmse_hist).  u [k, ni], v [k, nu]
    are updated in place."""
    nb = int(item_blocks) if item_blocks else max(1, -(-int(nbr_items) // FUNK_ITEMS_PER_BLOCK))
    d = DsgdKMF(ratings_index, ratings, nbr_users, nbr_items, dim, -1, 1, item_blocks_1gpu=nb)
    try:
        use_bias = items_bias is not None
        if overall_avg is None:
            if use_bias:
                raise ValueError("funk_block_order: overall_avg is needed with biases (estimator_loop_with_bias)")
            overall_avg = 1.0                                   # gd_estimator.pyx:750: an empty cache counts as 1.0
        d.set_factors(u, v, items_bias, users_bias)
        hist = d.train_funk(min_epochs, min_improvement, feature_init, learning_rate, K, overall_avg, use_bias,
                            update_users, update_items, epoch_cap)
        d.get_factors(u, v)
        return d.order(), hist
    finally:
        d.close()


def kmf_train_block_order(mode, nbr_epochs, dim, learning_rate, K_users, K_items, K_bias, u, v, ratings_index, ratings,
                          items_bias, users_bias, balanced=True):
    """train_linear_kernel / train_logistic_kernel (kmf_train.pyx:195-277, :103-189; mode 0 / 1) on ONE GPU on the block order: the
    given ratings grouped by (user block, item block) (stable; with balanced=True every block in its own minimum-level order), every
    block on the static resident-side epoch.  The result is bit-identical to the sequential loop ON THAT ORDER, which is returned (int64
    perm).  The reference draws its order from an unseeded RNG (base.py:1127-1129), i.e. leaves it open; whether this STRUCTURED order
    trains as well as a uniform draw is measured, not assumed: profiles/r04_block_order_quality.json (cfg 3 / 100, CPU oracle, 10
    epochs: train / held-out RMSE 3.4e-3 BELOW the mean of seven uniform orders, whose own spread is 3.5e-3).
    u [k, ni], v [k, nu], items_bias, users_bias are updated in place.  cfg 3 whole: 42 ms per epoch on the level schedule of the given
    order, 20 ms here; cfg 5 whole (1 B ratings): 0.56 s -> 0.21 s."""
    nu, ni = v.shape[1], u.shape[1]
    ub, nb = one_gpu_block_counts(nu, ni, dim)
    d = DsgdKMF(ratings_index, ratings, nu, ni, dim, -1, ub, balanced=balanced, item_blocks_1gpu=nb)
    try:
        d.set_factors(u, v, items_bias, users_bias)
        d.train(mode, nbr_epochs, learning_rate, K_users, K_items, K_bias)
        d.get_factors(u, v, items_bias, users_bias)
        return d.order()
    finally:
        d.close()
