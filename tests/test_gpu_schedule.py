"""GPU: the device-side schedule builder (mfx_schedule_dev.hip) gives the same
levels as the sequential host pass and plans built either way train to identical factors."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    yield


def _levels(fn, idx, nu, ni):
    from mfrec_amd import _shim as S
    lev = np.zeros(max(len(idx), 1), dtype=np.int32)
    n = C.c_int64(0)
    S.check(fn(S.i32(idx, "idx", 2), len(idx), nu, ni, lev.ctypes.data_as(S._ip), C.byref(n)))
    return lev[:len(idx)], n.value


def _cases():
    rs = np.random.RandomState(5)
    yield "tiny", 3, 2, np.array([[0, 0], [0, 1], [1, 0], [2, 1], [0, 0]], dtype=np.int32)
    yield "single", 1, 1, np.zeros((1, 2), dtype=np.int32)
    yield "one-user-chain", 1, 50, np.stack([np.zeros(300, int), rs.randint(0, 50, 300)], 1).astype(np.int32)
    yield "independent", 1000, 1000, np.stack([np.arange(1000), np.arange(1000)], 1).astype(np.int32)
    for nu, ni, nnz in ((40, 30, 5000), (2000, 1500, 200000), (100, 100000, 300000)):
        yield "uniform-%d" % nnz, nu, ni, np.stack([rs.randint(0, nu, nnz), rs.randint(0, ni, nnz)], 1).astype(np.int32)
    # skewed: Zipf-like users
    w = 1.0 / np.arange(1, 3001) ** 0.9
    us = rs.choice(3000, 150000, p=w / w.sum())
    yield "zipf", 3000, 5000, np.stack([us, rs.randint(0, 5000, 150000)], 1).astype(np.int32)
    # user-major sorted input (get_ratings(randomize_order=False))
    idx = np.stack([rs.randint(0, 500, 60000), rs.randint(0, 700, 60000)], 1).astype(np.int32)
    yield "user-major", 500, 700, idx[np.lexsort((idx[:, 1], idx[:, 0]))]


def test_levels_identical_to_host_pass():
    from mfrec_amd import _shim as S
    for name, nu, ni, idx in _cases():
        a, na = _levels(S.lib().mfx_schedule_levels, idx, nu, ni)
        b, nb = _levels(S.lib().mfx_schedule_levels_gpu, idx, nu, ni)
        assert na == nb, name
        assert np.array_equal(a, b), name


def test_out_of_range_index_is_reported():
    from mfrec_amd import _shim as S
    idx = np.array([[0, 0], [1, 1], [5, 0]], dtype=np.int32)
    with pytest.raises(ValueError, match="rating 2"):
        _levels(S.lib().mfx_schedule_levels_gpu, idx, 3, 2)


@pytest.mark.parametrize("nnz", [1, 777, 250000])
def test_plans_from_both_builders_train_identically(nnz):
    from mfrec_amd.plan import SgdPlan
    rs = np.random.RandomState(nnz)
    nu, ni, k = 900, 700, 24
    idx = np.stack([rs.randint(0, nu, nnz), rs.randint(0, ni, nnz)], 1).astype(np.int32)
    r = rs.randint(1, 6, nnz).astype(np.float64)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    out = {}
    for sched in ("host", "device"):
        with SgdPlan(idx, r, nu, ni, k, schedule=sched) as p:
            info = p.info()
            assert info["schedule_builder"] == sched
            p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
            rm = p.run_kmf(0, 3, 0.02, 0.1, 0.05, 0.007)
            u, v, ib, ub = np.empty_like(u0), np.empty_like(v0), np.empty(ni), np.empty(nu)
            p.get_factors(u, v, ib, ub)
            out[sched] = (info["n_levels"], info["max_level"], u, v, ib, ub, np.asarray(rm))
    h, d = out["host"], out["device"]
    assert h[0] == d[0] and h[1] == d[1]
    for a, b in zip(h[2:], d[2:]):
        assert np.array_equal(a, b)          # the same records in the same order: bit-identical


def test_auto_falls_back_to_host_when_levels_explode():
    # one user owns every rating: nnz levels > nnz/1000 -> the auto builder must hand over to the host
    from mfrec_amd.plan import SgdPlan
    nnz = 1 << 20
    rs = np.random.RandomState(0)
    idx = np.stack([np.zeros(nnz, int), rs.randint(0, 1000, nnz)], 1).astype(np.int32)
    idx[:, 0] = 0
    idx[: nnz // 2, 0] = rs.randint(1, 50000, nnz // 2)         # half of them spread out
    r = np.ones(nnz)
    with SgdPlan(idx, r, 50000, 1000, 4, schedule="auto") as p:
        info = p.info()
    assert info["schedule_builder"] == "host"
    assert info["n_levels"] >= nnz // 2


def test_device_builder_refuses_more_levels_than_its_budget_and_bad_flags():
    from mfrec_amd import _shim as S
    from mfrec_amd.plan import SgdPlan
    # a single user with 2^20 + 1 ratings needs that many levels: explicit "device" must fail loudly
    nnz = (1 << 20) + 1
    idx = np.zeros((nnz, 2), dtype=np.int32)
    idx[:, 1] = np.arange(nnz) % 1000
    with pytest.raises(S.MfxError, match="levels"):
        SgdPlan(idx, np.ones(nnz), 1, 1000, 2, schedule="device")
    with pytest.raises(ValueError, match="schedule"):
        SgdPlan(idx[:10], np.ones(10), 1, 1000, 2, schedule="gpu")
    h = S.lib().mfx_sgd_plan_create2(S.i32(idx[:10], "idx", 2), S.f64(np.ones(10), "r", 1), 10, 1, 1000, 2, 7)
    assert not h and "bad argument" in S.last_error()


def test_balanced_order_gpu_is_a_permutation_with_vertex_disjoint_rounds():
    """mfx_schedule_balanced_order_gpu: a permutation; its level schedule has (about) max-degree levels,
    far fewer than the input order's; training on it equals the oracle's sequential loop over it."""
    import time
    from mfrec_amd import _shim as S
    from mfrec_amd.lib.datasets import balanced_order_gpu
    from mfrec_amd.lib import kmf_train as K
    from oracle import oracle as O
    rs = np.random.RandomState(3)
    nu, ni, nnz, k = 3000, 2000, 300000, 16
    idx = np.stack([rs.randint(0, nu, nnz), rs.randint(0, ni, nnz)], 1).astype(np.int32)
    r = rs.randint(1, 6, nnz).astype(np.float64)
    perm, rounds = balanced_order_gpu(idx, nu, ni)
    assert np.array_equal(np.sort(perm), np.arange(nnz))
    deg = max(np.bincount(idx[:, 0]).max(), np.bincount(idx[:, 1]).max())
    _, n_in = _levels(S.lib().mfx_schedule_levels, idx, nu, ni)
    bidx, br = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    lev, n_bal = _levels(S.lib().mfx_schedule_levels, bidx, nu, ni)
    assert n_bal <= rounds and deg <= rounds <= int(1.25 * deg) + 2 and rounds < 0.7 * n_in
    # round-major with ascending input index inside a round
    # (rounds are the colour classes; the level of a rating never exceeds its round)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    K.train_linear_kernel(2, k, 0.1, 0.02, 0, 0, 0.1, 0.05, 0.007, 0.0, u, v, bidx, br, ib, ub)
    u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    O.train_linear_kernel(2, k, 0.1, 0.02, 0, 0, 0.1, 0.05, 0.007, 0.0, u2, v2, bidx, br, ib2, ub2)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ub, ub2)
    # determinism
    perm2, rounds2 = balanced_order_gpu(idx, nu, ni)
    assert rounds2 == rounds and np.array_equal(perm, perm2)
