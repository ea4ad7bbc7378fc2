"""GPU (one device): the G > 1 arithmetic of both multi-GPU modes on the HIP path.

  * averaging mode (mfx_sgd_run_kmf_sharded's scheme): all G shards on this GPU through the same
    k_sub / k_combine kernels, rank-ordered on-device sum in place of the all-reduce ->
    bit-identical to oracle/sharded_sim.py at G = 2, 8 and scale = 1/G, 1.
    (On real hardware ncclAllReduce does NOT sum in rank order: a G > 1 hardware run is compared
    with `assert_close_to_simulation`, a tolerance, not with array_equal -- see DESIGN.md section 5.)
  * exact mode (DSGD block rotation, mfx_dsgd_*): all G ranks on this GPU, rotation = device copy ->
    bit-identical to the sequential oracle on the block order, at G = 1, 2, 3, 8, with empty blocks,
    linear and logistic."""
import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu
HP = (0.01, 0.1, 0.1, 0.007)


@pytest.fixture(scope="module", autouse=True)
def _init():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)


def assert_close_to_simulation(got, sim, world):
    """Tolerance for a HARDWARE G > 1 run of the averaging mode: RCCL's all-reduce adds the G deltas in
    an order of its own choosing, so each combined element may differ from the rank-ordered sum by a few
    ulps of the largest delta; after E epochs the difference stays far below 1e-9 relative."""
    for a, b in zip(got, sim):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12 * world)


@pytest.mark.parametrize("world,scale", [(2, None), (2, 1.0), (8, None), (8, 1.0), (3, 0.5)])
def test_averaging_fake_world_equals_simulation(world, scale):
    from mfrec_amd.sharding import FakeWorldKMF
    from oracle import sharded_sim as SIM
    nu, ni, nnz, k, E = 700, 260, 30000, 64, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=31)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    T = FakeWorldKMF(idx, r, nu, ni, k, world)
    T.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    se = T.train(0, E, *HP, scale=scale)
    got = T.get_factors()
    T.close()
    sim = SIM.simulate(0, world, E, k, *HP, u0, v0, np.zeros(ni), np.zeros(nu), idx, r, scale=scale)
    for a, b in zip(got, sim):
        assert np.array_equal(a, b)
    assert_close_to_simulation(got, sim, world)          # the helper accepts the exact result too
    assert se.shape == (world, E) and np.all(np.isfinite(se)) and np.all(se > 0)


@pytest.mark.parametrize("world,k,mode,overlap,balanced",
                         [(1, 16, 0, True, False), (2, 64, 0, True, False), (2, 64, 0, False, False),
                          (3, 10, 0, True, False), (8, 64, 0, True, False), (8, 64, 0, False, False),
                          (4, 32, 1, True, False), (4, 64, 0, True, True), (3, 24, 0, False, True)])
def test_dsgd_fake_world_equals_sequential_oracle_on_block_order(world, k, mode, overlap, balanced):
    """overlap = 2 * world item blocks (transfers on the second stream, one step of slack) vs world blocks;
    balanced = every block in its own minimum-level order (the library reports the order it used)."""
    from mfrec_amd import sharding as SH
    from mfrec_amd.sharding import DsgdKMF
    from oracle import dsgd_sim as DS
    nu, ni, nnz, E = 900, 400, 40000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=5)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    ib0, ub0 = 0.01 * synth.init_normal(1, ni, 3)[0], 0.01 * synth.init_normal(1, nu, 4)[0]
    T = DsgdKMF(idx, r, nu, ni, k, -1, world, overlap=overlap, balanced=balanced)
    assert T.n_item_blocks == (2 * world if (overlap and world > 1) else world)
    T.set_factors(u0, v0, ib0, ub0)
    se = T.train(mode, E, *HP)
    u, v, ib, ub = np.zeros((k, ni)), np.zeros((k, nu)), np.zeros(ni), np.zeros(nu)
    T.get_factors(u, v, ib, ub)
    info = T.info()
    perm = T.order()
    T.close()
    assert np.array_equal(np.sort(perm), np.arange(nnz))
    if not balanced:                                   # the library's order IS the host-side block order
        assert np.array_equal(perm, SH.dsgd_block_order(idx, T.user_bounds, T.item_bounds))
    su, sv, sib, sub, rm = DS.simulate(mode, E, k, *HP, u0, v0, ib0, ub0, idx, r, T.user_bounds, T.item_bounds, perm=perm)
    if mode == 0:
        assert np.array_equal(u, su) and np.array_equal(v, sv)
        assert np.array_equal(ib, sib) and np.array_equal(ub, sub)
    else:                                   # logistic: device exp() (1 ulp), as on one GPU
        np.testing.assert_allclose(u, su, rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(v, sv, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(np.sqrt(se / nnz), rm, rtol=1e-12)
    assert info["nnz_local"] == nnz and info["max_block"] >= info["min_block"] >= 0
    if balanced:
        # fewer dependency levels than the same blocks in the given order
        T2 = DsgdKMF(idx, r, nu, ni, k, -1, world, overlap=overlap, balanced=False)
        assert info["levels_total"] < T2.info()["levels_total"]
        T2.close()


@pytest.mark.parametrize("world,balanced", [(2, False), (3, True)])
def test_dsgd_blocks_on_the_static_resident_epoch(world, balanced, monkeypatch):
    """Every block a plan of its own on the resident-side epoch (forced: the cost model keeps problems this small on the
    level schedule): still the sequential loop on the order the library reports, bit for bit."""
    from mfrec_amd.sharding import DsgdKMF
    from oracle import dsgd_sim as DS
    monkeypatch.setenv("MFX_K1_RES", "static")
    nu, ni, nnz, E, k = 900, 400, 40000, 3, 32
    idx, r = synth.make_ratings(nu, ni, nnz, seed=6)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    T = DsgdKMF(idx, r, nu, ni, k, -1, world, balanced=balanced)
    info = T.info()
    assert info["resident_blocks"] == world * world, info
    T.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    se = T.train(0, E, *HP)
    u, v, ib, ub = np.zeros((k, ni)), np.zeros((k, nu)), np.zeros(ni), np.zeros(nu)
    T.get_factors(u, v, ib, ub)
    perm = T.order()
    T.close()
    su, sv, sib, sub, rm = DS.simulate(0, E, k, *HP, u0, v0, np.zeros(ni), np.zeros(nu), idx, r, T.user_bounds, T.item_bounds, perm=perm)
    assert np.array_equal(u, su) and np.array_equal(v, sv) and np.array_equal(ib, sib) and np.array_equal(ub, sub)
    np.testing.assert_allclose(np.sqrt(se / nnz), rm, rtol=1e-12)


@pytest.mark.parametrize("fail_at", [1, 4, 9])
def test_dsgd_recovers_when_a_static_block_loses_its_co_residency_check(fail_at, monkeypatch):
    """Two ranks in one process, four blocks, every block on the static resident-side epoch; the `fail_at`-th static launch of the
    process (and every later one) finds its grid "not co-resident" (test hook: what a GPU shared with another process does).  That
    launch modifies nothing and its plan skips the rest of the run while the other blocks train on -- the run as a whole is void:
    the library restores the factors it saved when the call began, moves every block to the level schedule and repeats the call.
    The result is the sequential loop's on the block order, bit for bit; the next call simply continues (on the level schedule)."""
    from mfrec_amd.sharding import DsgdKMF
    from oracle import dsgd_sim as DS
    from mfrec_amd.plan import SgdPlan
    monkeypatch.setenv("MFX_K1_RES", "static")
    monkeypatch.setenv("MFX_TESTING", "1")
    monkeypatch.setenv("MFX_GRAPH", "0")        # eager launches: the hook counts enqueued launches
    nu, ni, nnz, E, k = 900, 400, 40000, 3, 32
    idx, r = synth.make_ratings(nu, ni, nnz, seed=6)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    T = DsgdKMF(idx, r, nu, ni, k, -1, 2, balanced=True)
    assert T.info()["resident_blocks"] == 4
    T.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    monkeypatch.setenv("MFX_TEST_REG_FAIL", "0")
    with SgdPlan(idx, r, nu, ni, 18, epoch="static") as p:                       # (a static launch with the hook off resets its counter)
        p.set_factors(synth.init_normal(18, ni, 1), synth.init_normal(18, nu, 2), np.zeros(ni), np.zeros(nu))
        p.run_kmf(0, 1, *HP)
    monkeypatch.setenv("MFX_TEST_REG_FAIL", str(fail_at))
    se = T.train(0, E, *HP)
    assert T.info()["resident_blocks"] == 0                                     # every block went to the level schedule
    monkeypatch.setenv("MFX_TEST_REG_FAIL", "0")
    se2 = T.train(0, 1, *HP)                                                    # ... and stays there: the next call continues
    u, v, ib, ub = np.zeros((k, ni)), np.zeros((k, nu)), np.zeros(ni), np.zeros(nu)
    T.get_factors(u, v, ib, ub)
    perm = T.order()
    T.close()
    su, sv, sib, sub, rm = DS.simulate(0, E + 1, k, *HP, u0, v0, np.zeros(ni), np.zeros(nu), idx, r, T.user_bounds, T.item_bounds, perm=perm)
    assert np.array_equal(u, su) and np.array_equal(v, sv) and np.array_equal(ib, sib) and np.array_equal(ub, sub)
    np.testing.assert_allclose(np.sqrt(np.concatenate([se, se2]) / nnz), rm, rtol=1e-12)


def test_dsgd_empty_blocks_and_real_single_rank():
    """Hand-made bounds with an empty user block and an empty item block; and the non-fake path with
    world = 1 (rank 0): one block, no rotation -- the plain sequential order."""
    from mfrec_amd.sharding import DsgdKMF
    from oracle import dsgd_sim as DS
    from oracle import oracle as O
    nu, ni, nnz, k, E = 300, 200, 8000, 24, 2
    idx, r = synth.make_ratings(nu, ni, nnz, seed=9)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    ub = np.array([0, 100, 100, 300], dtype=np.int64)
    jb = np.array([0, 0, 120, 200], dtype=np.int64)
    T6 = DsgdKMF(idx, r, nu, ni, k, -1, 3, user_bounds=ub, item_bounds=np.array([0, 0, 50, 50, 120, 199, 200], dtype=np.int64))   # 2 * world blocks
    T6.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    T6.train(0, E, *HP)
    u6, v6, ib6, ub6 = np.zeros((k, ni)), np.zeros((k, nu)), np.zeros(ni), np.zeros(nu)
    T6.get_factors(u6, v6, ib6, ub6)
    s6 = DS.simulate(0, E, k, *HP, u0, v0, np.zeros(ni), np.zeros(nu), idx, r, ub, T6.item_bounds)
    T6.close()
    assert np.array_equal(u6, s6[0]) and np.array_equal(v6, s6[1]) and np.array_equal(ib6, s6[2]) and np.array_equal(ub6, s6[3])
    T = DsgdKMF(idx, r, nu, ni, k, -1, 3, user_bounds=ub, item_bounds=jb)
    T.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    T.train(0, E, *HP)
    u, v, ib, ubias = np.zeros((k, ni)), np.zeros((k, nu)), np.zeros(ni), np.zeros(nu)
    T.get_factors(u, v, ib, ubias)
    T.close()
    su, sv, sib, sub, _ = DS.simulate(0, E, k, *HP, u0, v0, np.zeros(ni), np.zeros(nu), idx, r, ub, jb)
    assert np.array_equal(u, su) and np.array_equal(v, sv) and np.array_equal(ib, sib) and np.array_equal(ubias, sub)

    T = DsgdKMF(idx, r, nu, ni, k, 0, 1)
    T.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    T.train(0, E, *HP)
    T.get_factors(u, v, ib, ubias)
    T.close()
    u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    O.train_linear_kernel(E, k, 0.1, HP[0], 0, 0, HP[1], HP[2], HP[3], 0.0, u2, v2, idx, r, ib2, ub2)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ubias, ub2)


def test_dsgd_argument_errors():
    from mfrec_amd.sharding import DsgdKMF
    idx, r = synth.make_ratings(50, 40, 500, seed=1)
    with pytest.raises(ValueError):
        DsgdKMF(idx, r, 50, 40, 8, -1, 2, user_bounds=np.array([0, 30, 49]), item_bounds=np.array([0, 20, 40]))
    with pytest.raises(ValueError):                     # 3 item blocks for 2 GPUs: neither world nor 2 * world
        DsgdKMF(idx, r, 50, 40, 8, -1, 2, item_bounds=np.array([0, 10, 20, 40]))
    bad = idx.copy(); bad[3, 1] = 40
    with pytest.raises(ValueError, match="outside"):
        DsgdKMF(bad, r, 50, 40, 8, -1, 2)


@pytest.mark.parametrize("nb,balanced", [(3, False), (5, True)])
def test_item_block_order_on_one_gpu_equals_the_oracle(nb, balanced):
    """world = 1 with ANY number of item blocks (a catalogue larger than the register file, block by block on the static resident-side
    epoch): bit-identical to the sequential loop (kmf_train.pyx:243-270) on the order the object reports."""
    from mfrec_amd.sharding import DsgdKMF
    from oracle import oracle as O
    nu, ni, nnz, k = 3000, 1700, 120000, 18
    idx, r = synth.make_ratings(nu, ni, nnz, seed=31)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    d = DsgdKMF(idx, r, nu, ni, k, -1, 1, balanced=balanced, item_blocks_1gpu=nb)
    assert d.info()["item_blocks"] == nb
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    d.set_factors(u, v, ib, ub)
    d.train(0, 2, 0.01, 0.1, 0.1, 0.007)
    d.get_factors(u, v, ib, ub)
    perm = d.order()
    d.close()
    assert sorted(perm.tolist()) == list(range(nnz))
    u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    O.train_linear_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm]), ib2, ub2)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ub, ub2)


@pytest.mark.parametrize("world,nb", [(3, 9), (2, 8)])
def test_many_item_blocks_per_rank_in_one_process(world, nb):
    """rank = -1 with stride = nb / world > 2 (a block rests stride - 1 steps in transit): still the sequential loop on the block order."""
    from mfrec_amd.sharding import DsgdKMF
    from oracle import oracle as O
    nu, ni, nnz, k = 2500, 1900, 100000, 18
    idx, r = synth.make_ratings(nu, ni, nnz, seed=33)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    d = DsgdKMF(idx, r, nu, ni, k, -1, world, item_blocks_1gpu=nb)
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    d.set_factors(u, v, ib, ub)
    d.train(0, 2, *HP)
    d.get_factors(u, v, ib, ub)
    perm = d.order()
    d.close()
    assert sorted(perm.tolist()) == list(range(nnz))
    u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    O.train_linear_kernel(2, k, 0.1, HP[0], 0, 0, HP[1], HP[2], HP[3], 0.0, u2, v2, np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm]), ib2, ub2)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ub, ub2)


def test_recommender_train_on_the_block_order(monkeypatch):
    """KMFRecommender.train(order="blocks"): the reference's RNG calls (N(0, 0.1) init, one shuffle), then the block order; equal to
    the oracle's loop on the order the trainer reports.  Small row budgets force several user and item blocks."""
    from mfrec_amd import sharding
    from mfrec_amd.recommendation import KMFRecommender
    from oracle import oracle as O
    monkeypatch.setattr(sharding, "ROWS_PER_BLOCK", 300)
    monkeypatch.setattr(sharding, "TRAVELLING_BYTES", 1500 * 64 * 8)
    nu, ni, nnz, k = 4000, 1100, 90000, 64
    idx, r = synth.make_ratings(nu, ni, nnz, seed=35)
    assert sharding.one_gpu_block_counts(nu, ni, k) == (3, 6)
    R = KMFRecommender(nu, ni, {"nbr_features": k, "nbr_epochs": 2})
    R.set_ratings(idx, r)
    np.random.seed(5)
    R.train(kernel="train_linear_kernel", order="blocks")
    # replay: same RNG calls, then the oracle on the reported order
    R2 = KMFRecommender(nu, ni, {"nbr_features": k, "nbr_epochs": 2})
    R2.set_ratings(idx, r)
    np.random.seed(5)
    R2.init_feature_normal(0.0, 0.1)
    i2, r2 = R2.get_ratings(randomize_order=True)
    perm = R.block_order
    u2, v2, ib2, ub2 = R2.svd_u.copy(), R2.svd_v.copy(), np.zeros(ni), np.zeros(nu)
    O.train_linear_kernel(2, k, R.feature_init, R.learning_rate, 0, 0, R.K_users, R.K_items, R.K_bias, 0.0, u2, v2,
                          np.ascontiguousarray(i2[perm]), np.ascontiguousarray(r2[perm]), ib2, ub2)
    assert np.array_equal(R.svd_u, u2) and np.array_equal(R.svd_v, v2) and np.array_equal(R.items_bias, ib2) and np.array_equal(R.users_bias, ub2)
