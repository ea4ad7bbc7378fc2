"""GPU: the BASELINE.json configs beyond cfg 2, at FULL size, inside pytest -m gpu.

  * cfg 3 whole (1 M x 200 K, 100 M ratings, k=64) on one GPU: ONE epoch, bit for bit against the
    oracle's sequential loop (the oracle needs about a minute at this size);
  * cfg 4 (same shape, k=128, ALS) at full size: one epoch, checked WITHOUT the oracle (it would need
    half an hour): for >= 1000 sampled rows of each side the k x k system is rebuilt in numpy f64
    straight from its definition (als_implicit.pyx:257-302 / :307-352) and the residual
    |A x - b| / |b| of the GPU's solution must be ~1e-12;
  * cfg 5's per-GPU shard shape (1.25 M x 1 M, 125 M ratings, k=64; 1.15 GB of factors, beyond the
    Infinity Cache): size-independent properties -- schedule idempotence (level-major input with the
    levels internally shuffled gives the same bits), monotone online RMSE, device and host schedule
    builders agree on the level structure;
  * cfg 3 whole in get_ratings(randomize_order=False)'s order (user-major): the resident-side epoch with the user
    rows in registers and runs taken by ticket (MFX_EPOCH_TICKET, chosen by the cost model), ONE epoch, bit for bit
    against the oracle;
  * cfg 5 WHOLE (10 M x 1 M, 1 B ratings, k=64) on one GPU: properties -- device-built schedule and its level
    count, finite and strictly decreasing online RMSE, host and device builders agree on a 100 M-rating prefix; and the same
    billion ratings on the ONE-GPU block order (3 user x 24 item blocks, all 72 on the static resident-side epoch);
  * cfg 3 whole on the ONE-GPU block order (4 item blocks, each in its minimum-level order): one epoch, bit for bit against the
    oracle on the order the trainer reports."""
import ctypes as C
import time

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


@pytest.fixture(scope="module")
def cfg3_data():
    nu, ni, nnz = 1000000, 200000, 100000000
    idx, r = synth.make_ratings(nu, ni, nnz)
    return nu, ni, nnz, idx, r


@pytest.fixture(scope="module")
def cfg3_runs(cfg3_data):
    """cfg 3 whole, ONE epoch, on the three schedules the library has for it -- the given (shuffled) order on the level schedule, the
    user-major order on the ticket epoch, the one-GPU block order on the static epoch -- each on the GPU, and the oracle's sequential
    loop (kmf_train.pyx:243-270) on each of the three orders.  The three oracle runs (about a minute each at this size) run side by
    side on three host threads (the C restatement releases the GIL), which is what keeps this file's wall time down; the tests below
    only compare."""
    import threading
    from mfrec_amd import sharding
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    nu, ni, nnz, idx, r = cfg3_data
    k = 64
    u0, v0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
    out = {}
    # -- given order, level schedule
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    with SgdPlan(idx, r, nu, ni, k) as p:
        info = p.info()
        p.set_factors(u, v, ib, ub)
        rm = p.run_kmf(0, 1, *HP)
        p.get_factors(u, v, ib, ub)
    out["given"] = dict(gpu=(u, v, ib, ub), rm=rm, info=info, order=(idx, r))
    # -- user-major, ascending items, generation order among duplicates: one stable sort of a combined key
    order = np.argsort((idx[:, 0].astype(np.int64) << 32) | idx[:, 1].astype(np.int64), kind="stable")
    iu, ru = np.ascontiguousarray(idx[order]), np.ascontiguousarray(r[order])
    del order
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    with SgdPlan(iu, ru, nu, ni, k) as p:
        info = p.info()
        name = p.kernel_name(0)
        p.set_factors(u, v, ib, ub)
        t0 = time.time()
        rm = p.run_kmf(0, 1, *HP)
        dt = time.time() - t0
        p.get_factors(u, v, ib, ub)
        after = p.info()["epoch_kernel"]
    out["user_major"] = dict(gpu=(u, v, ib, ub), rm=rm, info=info, name=name, dt=dt, after=after, order=(iu, ru))
    # -- the one-GPU block order (sharding.kmf_train_block_order: four item blocks, every block in its own minimum-level order)
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    perm = sharding.kmf_train_block_order(0, 1, k, HP[0], HP[1], HP[2], HP[3], u, v, idx, r, ib, ub)
    pidx, pr = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    out["blocks"] = dict(gpu=(u, v, ib, ub), perm_ok=(perm.shape[0] == nnz and int(perm.sum()) == nnz * (nnz - 1) // 2), order=(pidx, pr))
    del perm

    def oracle(tag):
        oi, orr = out[tag]["order"]
        ou, ov, oib, oub, orm = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu), np.zeros(1)
        O.train_linear_kernel(1, k, 0.1, HP[0], 0, 0, HP[1], HP[2], HP[3], 0.0, ou, ov, oi, orr, oib, oub, 1, 1, 0, orm)
        out[tag]["oracle"] = (ou, ov, oib, oub)
        out[tag]["oracle_rm"] = orm

    th = [threading.Thread(target=oracle, args=(tag,)) for tag in ("given", "user_major", "blocks")]
    for x in th:
        x.start()
    for x in th:
        x.join()
    return out


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_cfg3_whole_one_epoch_bitexact_vs_oracle(cfg3_runs):
    g = cfg3_runs["given"]
    assert _same(g["gpu"], g["oracle"])
    assert abs(g["rm"][0] - g["oracle_rm"][0]) <= 1e-12 * g["oracle_rm"][0]
    assert g["info"]["schedule_builder"] == "device" and 900 < g["info"]["n_levels"] < 1500


def test_cfg3_whole_user_major_resident_epoch_bitexact_vs_oracle(cfg3_runs):
    g = cfg3_runs["user_major"]
    assert g["info"]["epoch_kernel"] == "ticket" and "user rows resident" in g["name"], (g["info"], g["name"])
    assert g["after"] == "ticket"                                  # no fallback happened
    assert _same(g["gpu"], g["oracle"])
    assert abs(g["rm"][0] - g["oracle_rm"][0]) <= 1e-12 * g["oracle_rm"][0]
    assert g["dt"] < 1.0                    # ~30 ms on the device + the first launch; a regression guard


def _sampled_residuals(Y, X, off, col, rows, c_pos, lam):
    """max over `rows` of |A x - b| / |b| with A = G + c * sum_N y y^T + lam I, b = (1 + c) sum_N y
    (als_implicit.pyx:257-302); Y [k, n_other] feature-major, X [k, n_this]."""
    G = Y @ Y.T
    k = Y.shape[0]
    worst = 0.0
    for j in rows:
        nb = col[off[j]:off[j + 1]]
        Yn = Y[:, nb]
        A = G + c_pos * (Yn @ Yn.T) + lam * np.eye(k)
        b = (1.0 + c_pos) * Yn.sum(axis=1)
        x = X[:, j]
        if nb.size == 0:
            assert np.all(x == 0.0)
            continue
        worst = max(worst, np.linalg.norm(A @ x - b) / np.linalg.norm(b))
    return worst


def test_cfg3_whole_block_order_bitexact_vs_oracle(cfg3_data, cfg3_runs):
    """cfg 3 whole on the ONE-GPU block order (sharding.kmf_train_block_order: 200 K items do not fit the static resident-side epoch
    at once, four item blocks do; every block in its own minimum-level order): one epoch, bit for bit against the oracle's
    sequential loop (kmf_train.pyx:243-270) ON THE ORDER THE TRAINER REPORTS."""
    from mfrec_amd import sharding
    nu, ni, nnz, idx, r = cfg3_data
    assert sharding.one_gpu_block_counts(nu, ni, 64) == (1, 4)
    g = cfg3_runs["blocks"]
    assert g["perm_ok"]
    pidx = g["order"][0]
    # the order groups the ratings by item block: block ids never decrease along it
    bounds = sharding.item_blocks(idx, ni, 4)
    jb = np.searchsorted(bounds, pidx[:, 1], side="right") - 1
    assert jb[0] == 0 and jb[-1] == 3 and np.all(np.diff(jb) >= 0)
    assert _same(g["gpu"], g["oracle"])


def test_cfg4_full_size_als_epoch_residuals(cfg3_data):
    import scipy.sparse as sp
    from mfrec_amd import _shim as S
    from mfrec_amd.lib.als_implicit import als_wrmf
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    nu, ni, nnz, idx, _ = cfg3_data
    k, c_pos, lam = 128, 1, 0.015
    M = sp.csr_matrix((np.ones(nnz, dtype=np.int8), (idx[:, 0], idx[:, 1])), shape=(nu, ni))
    M.sum_duplicates()
    urow, ucol = create_bool_sparse_row(M)
    irow, icol = create_bool_sparse_col(M)
    del M
    rs = np.random.RandomState(0)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v = u0.copy(), v0.copy()
    als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni, c_pos, lam)
    dev_s = S.lib().mfx_last_compute_ms() * 1e-3
    assert np.isfinite(u).all() and np.isfinite(v).all()
    uoff = np.concatenate([[0], np.cumsum(urow[1:], dtype=np.int64)])
    ioff = np.concatenate([[0], np.cumsum(irow[1:], dtype=np.int64)])
    # user pass: solved against the INITIAL item factors; item pass: against the NEW user factors
    rows_u = rs.choice(len(uoff) - 1, 1000, replace=False)
    rows_i = rs.choice(len(ioff) - 1, 1000, replace=False)
    ru = _sampled_residuals(u0, v, uoff, ucol, rows_u, c_pos, lam)
    ri = _sampled_residuals(v, u, ioff, icol, rows_i, c_pos, lam)
    assert ru < 1e-11 and ri < 1e-11, (ru, ri)
    assert dev_s < 1.0                      # 0.14 s measured; a regression guard, not a benchmark


def test_cfg4_shaped_row_slices_vs_oracle(cfg3_data):
    """cfg 4's shapes against the ORACLE itself (als_implicit.pyx:211-352 restated), where it is affordable: the user pass over
    the first 2 000 users with the FULL 200 K-item factor table (degree ~100 rows, Gramian over 200 K columns, k = 128), the item
    pass over the first 300 items with the FULL 1 M-user table (degree ~500 rows: the long rows of the item half-pass) -- the
    run-length views simply end there (rows beyond the last listed row stay untouched, als_implicit.pyx:269-270).  About a minute
    of CPU.  1e-9 relative, the ALS tolerance of tests/test_gpu_als.py (Cholesky solve against the reference's explicit inverse)."""
    import scipy.sparse as sp
    from mfrec_amd.lib.als_implicit import als_wrmf
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    from oracle import oracle as O
    nu, ni, nnz, idx, _ = cfg3_data
    k, c_pos, lam, n_users, n_items = 128, 1, 0.015, 2000, 300
    M = sp.csr_matrix((np.ones(nnz, dtype=np.int8), (idx[:, 0], idx[:, 1])), shape=(nu, ni))
    M.sum_duplicates()
    urow, ucol = create_bool_sparse_row(M[:n_users])
    irow, icol = create_bool_sparse_col(M[:, :n_items])
    del M
    assert len(urow) == n_users + 1 and len(irow) == n_items + 1 and 150000 < len(ucol) < 250000 and 100000 < len(icol) < 200000
    rs = np.random.RandomState(1)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v = u0.copy(), v0.copy()
    als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni, c_pos, lam)
    u2, v2 = u0.copy(), v0.copy()
    O.als_wrmf(1, k, u2, v2, None, None, urow, ucol, irow, icol, nu, ni, c_pos, lam)
    assert np.array_equal(v[:, n_users:], v0[:, n_users:]) and np.array_equal(u[:, n_items:], u0[:, n_items:])   # untouched rows
    assert not np.array_equal(v[:, :n_users], v0[:, :n_users])
    np.testing.assert_allclose(v[:, :n_users], v2[:, :n_users], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(u[:, :n_items], u2[:, :n_items], rtol=1e-9, atol=1e-12)


def test_cfg5_shard_shape_properties():
    from mfrec_amd import _shim as S
    from mfrec_amd.plan import SgdPlan
    nu, ni, nnz, k = 1250000, 1000000, 125000000, 64
    idx, r = synth.make_ratings(nu, ni, nnz, seed=synth.SEED_DATA + 5)
    u0, v0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)

    def run(i, rr, epochs, schedule="auto"):
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        with SgdPlan(i, rr, nu, ni, k, schedule=schedule) as p:
            info = p.info()
            p.set_factors(u, v, ib, ub)
            rm = p.run_kmf(0, epochs, *HP)
            p.get_factors(u, v, ib, ub)
        return u, v, ib, ub, rm, info

    u, v, ib, ub, rm, info = run(idx, r, 3)
    assert np.all(np.isfinite(rm)) and np.all(np.diff(rm) < 0)
    assert info["schedule_builder"] == "device"
    # the host builder's levels of the same order: same count, and level-major input with every level
    # internally shuffled must not change a single bit
    lev = np.empty(nnz, dtype=np.int32); n = C.c_int64(0)
    S.check(S.lib().mfx_schedule_levels(S.i32(idx, "idx", 2), nnz, nu, ni, lev.ctypes.data_as(S._ip), C.byref(n)))
    assert n.value == info["n_levels"]
    rs = np.random.RandomState(0)
    order = np.lexsort((rs.rand(nnz).astype(np.float32), lev))
    del lev
    ia, ra = np.ascontiguousarray(idx[order]), np.ascontiguousarray(r[order])
    del order
    ua, va, iba, uba, rma, infoa = run(ia, ra, 3)
    assert infoa["n_levels"] == info["n_levels"]
    assert np.array_equal(ua, u) and np.array_equal(va, v) and np.array_equal(iba, ib) and np.array_equal(uba, ub)
    np.testing.assert_allclose(rma, rm, rtol=1e-12)


def test_cfg5_whole_one_gpu_properties():
    """BASELINE cfg 5 at its full size on ONE GPU (28 GB resident): no oracle at this size (10 minutes of CPU per
    epoch) -- size-independent properties instead."""
    from mfrec_amd import _shim as S
    from mfrec_amd.plan import SgdPlan
    nu, ni, nnz, k = 10000000, 1000000, 1000000000, 64
    idx, r = synth.make_ratings(nu, ni, nnz)
    U0, V0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)         # (640 M normals: drawn once, copied where needed)
    u0, v0 = U0.copy(), V0.copy()
    with SgdPlan(idx, r, nu, ni, k) as p:
        info = p.info()
        p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
        rm = p.run_kmf(0, 2, *HP)
        rm = np.concatenate([rm, p.run_kmf(0, 1, *HP)])
        ib, ub = np.zeros(ni), np.zeros(nu)
        p.get_factors(u0, v0, ib, ub)
    assert info["schedule_builder"] == "device" and info["epoch_kernel"] == "levels"
    # level count >= the largest degree (uniform data: ~1000 + a few sigma for items) and well below the device budget
    assert 1200 < info["n_levels"] < 3000, info
    assert np.all(np.isfinite(rm)) and np.all(np.diff(rm) < 0), rm
    assert np.isfinite(u0).all() and np.isfinite(v0).all() and np.isfinite(ib).all() and np.isfinite(ub).all()
    del u0, v0
    # the same billion ratings on the ONE-GPU block order (3 user x 24 item blocks: a user side above 2 GB is cut too): every block
    # on the static resident-side epoch, the order a permutation, online RMSE finite and strictly decreasing
    from mfrec_amd import sharding
    assert sharding.one_gpu_block_counts(nu, ni, k) == (3, 24)
    d = sharding.DsgdKMF(idx, r, nu, ni, k, -1, 3, item_blocks_1gpu=24)
    dinfo = d.info()
    d.set_factors(U0, V0, np.zeros(ni), np.zeros(nu))
    se = np.concatenate([d.train(0, 2, *HP), d.train(0, 1, *HP)])
    perm = d.order()
    d.close()
    assert dinfo["item_blocks"] == 24 and dinfo["resident_blocks"] == 72, dinfo
    assert perm.shape[0] == nnz and int(perm.sum()) == nnz * (nnz - 1) // 2
    assert np.all(np.isfinite(se)) and np.all(np.diff(se) < 0), se
    # ORACLE check of that path on a prefix of the epoch: the first block of the reported order -- block (user block 0, item block 0),
    # ~14 M ratings against the full 10 M x 1 M row tables -- as a trainer of its own on the SAME bounds (every other block empty),
    # one epoch, bit for bit against the sequential loop (kmf_train.pyx:243-270) on those ratings in the order it reports
    from oracle import oracle as O
    ub_, ib_ = d.user_bounds.copy(), d.item_bounds.copy()
    n0 = int(np.count_nonzero((idx[:, 0] < ub_[1]) & (idx[:, 1] < ib_[1])))
    sub = np.sort(perm[:n0])                                   # the block's ratings, in the GIVEN order
    del perm
    sidx, sr = np.ascontiguousarray(idx[sub]), np.ascontiguousarray(r[sub])
    assert int(sidx[:, 0].max()) < ub_[1] and int(sidx[:, 1].max()) < ib_[1] and 10000000 < n0 < 18000000
    del sub
    pu, pv = U0.copy(), V0.copy()
    pib, pub = np.zeros(ni), np.zeros(nu)
    # (the oracle's run of the same block starts now, on a host thread beside the GPU's: the C restatement releases the GIL)
    import threading
    ou, ov, oib, oub = U0, V0, np.zeros(ni), np.zeros(nu)                     # U0 / V0 are not needed unchanged any more
    oracle_in = {}

    def run_oracle():
        oi, orr = oracle_in["idx"], oracle_in["r"]
        O.train_linear_kernel(1, k, 0.1, HP[0], 0, 0, HP[1], HP[2], HP[3], 0.0, ou, ov, oi, orr, oib, oub)
    oracle_in["idx"], oracle_in["r"] = sidx, sr                      # (one block, given order inside it: the object's order is the identity -- asserted below)
    th = threading.Thread(target=run_oracle)
    th.start()
    d2 = sharding.DsgdKMF(sidx, sr, nu, ni, k, -1, 3, user_bounds=ub_, item_bounds=ib_, item_blocks_1gpu=24)
    assert d2.info()["resident_blocks"] == 1                          # (the 71 other blocks are empty)
    d2.set_factors(pu, pv, pib, pub)
    d2.train(0, 1, *HP)
    d2.get_factors(pu, pv, pib, pub)
    porder = d2.order()
    d2.close()
    assert np.array_equal(porder, np.arange(n0))                      # (one block, given order inside it: the identity)
    th.join()
    assert np.array_equal(pu, ou) and np.array_equal(pv, ov) and np.array_equal(pib, oib) and np.array_equal(pub, oub)
    del pu, pv, ou, ov, U0, V0, sidx, sr
    # host and device builders on the first 100 M ratings of the same tables: same level structure
    n = 100000000
    pidx, pr = np.ascontiguousarray(idx[:n]), np.ascontiguousarray(r[:n])
    del idx, r
    lev = np.empty(n, dtype=np.int32); nl = C.c_int64(0)
    S.check(S.lib().mfx_schedule_levels(S.i32(pidx, "idx", 2), n, nu, ni, lev.ctypes.data_as(S._ip), C.byref(nl)))
    with SgdPlan(pidx, pr, nu, ni, k, schedule="device", epoch="levels") as p:
        pinfo = p.info()
    assert pinfo["schedule_builder"] == "device" and pinfo["n_levels"] == nl.value
    assert pinfo["max_level"] == int(np.bincount(lev).max())


def test_funk_on_a_one_million_item_catalogue_block_order_bitexact():
    """BASELINE cfg 5's item side (1 M items: more than the resident-side Funk pass has lanes) with 2 M users and 40 M ratings, k = 2
    features x 2 epochs: sharding.funk_block_order cuts the catalogue into blocks of <= 400 K items, every block takes the
    persistent pass, a pass of a feature = the blocks one after the other on the one user side -- bit for bit the oracle's
    estimator_loop_without_bias (gd_estimator.pyx:691-779) on the order the object reports, RMSE history included."""
    from mfrec_amd import sharding as SH
    from oracle import oracle as O
    nu, ni, nnz, k = 2000000, 1000000, 40000000, 2
    idx, r = synth.make_ratings(nu, ni, nnz)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    perm, hist = SH.funk_block_order(2, 2, 1e9, k, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni)
    assert len(hist) == 2 * k
    ip, rp = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    del idx, r, perm
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    h2 = np.zeros(16)
    n2 = O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, ip, rp, nu, ni, rmse_hist=h2)
    assert n2 == 2 * k and np.allclose(hist, h2[:n2], rtol=1e-12, atol=0)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
