"""GPU: the XCD-resident Funk pass (k2_xcd_pass: one persistent launch per feature-epoch on a single XCD,
user slices in LDS, flat L2 barrier between dependency levels) must give the SAME BITS as the oracle's
sequential loops (gd_estimator.pyx:691-779, :489-582, :588-685) -- it is the default path for large levels
(cfg 2: 618 levels of 16 K) and is forced here (MFX_K2_XCD=1) on problems the oracle finishes in seconds."""
import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force(monkeypatch):
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    monkeypatch.setenv("MFX_K2_XCD", "1")


@pytest.mark.parametrize("nu,ni,nnz,k", [(3000, 2000, 200000, 3), (50000, 700, 400000, 2), (37, 29, 600, 4),
                                         (100000, 50000, 1000000, 2)])
def test_without_bias_bitexact(nu, ni, nnz, k):
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    idx, r = synth.make_ratings(nu, ni, nnz, seed=nu + k)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    hist = np.zeros(64)
    G.estimator_loop_without_bias(3, 3, 1e9, k, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni)
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    O.estimator_loop_without_bias(3, 3, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni, rmse_hist=hist)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)


def test_stopping_rule_history_and_second_run_on_the_same_plan():
    """Data-dependent stopping rule (gd_estimator.pyx:742) through the resident API, twice on one plan: the
    barrier generations keep counting across passes and runs."""
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    nu, ni, nnz, k = 4000, 1500, 150000, 4
    idx, r = synth.make_ratings(nu, ni, nnz, seed=2)
    with SgdPlan(idx, r, nu, ni, k) as p:
        for rep in range(2):
            u, v = synth.init_const(k, ni), synth.init_const(k, nu)
            p.set_factors(u, v)
            hist = p.run_funk(2, 1e-4, 0.1, 0.002, 0.05)
            p.get_factors(u, v)
            u2, v2, h2 = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros(4096)
            n = O.estimator_loop_without_bias(2, 0, 1e-4, k, 0.1, 0.002, 0.05, u2, v2, idx, r, nu, ni, rmse_hist=h2)
            assert np.array_equal(u, u2) and np.array_equal(v, v2)
            assert len(hist) == n
            np.testing.assert_allclose(hist, h2[:n], rtol=1e-12)
        # the full-k kernels on the SAME plan after the in-level re-ordering: still the oracle's bits
        u, v, ib, ub = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2), np.zeros(ni), np.zeros(nu)
        u2, v2, ib2, ub2 = u.copy(), v.copy(), ib.copy(), ub.copy()
        p.set_factors(u, v, ib, ub)
        p.run_kmf(0, 2, 0.01, 0.1, 0.1, 0.007)
        p.get_factors(u, v, ib, ub)
        O.train_linear_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2)
        assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ub, ub2)


def test_with_bias_and_gates_bitexact():
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    nu, ni, nnz, k = 5000, 3000, 250000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=8)
    rs = np.random.RandomState(0)
    ib, ub = rs.normal(0, 0.2, ni), rs.normal(0, 0.2, nu)
    for gates in ((1, 1), (1, 0), (0, 1)):
        u, v = synth.init_const(k, ni), synth.init_const(k, nu)
        u2, v2 = u.copy(), v.copy()
        G.estimator_loop_with_bias_dev(2, 2, 1e9, k, 0.1, 0.001, 0.0, 0.0, 0.05, 3.4, u, v, idx, r, ib, ub, nu, ni, *gates)
        O.estimator_loop_with_bias_dev(2, 2, 1e9, k, 0.1, 0.001, 0.0, 0.0, 0.05, 3.4, u2, v2, idx, r, ib, ub, nu, ni, *gates)
        assert np.array_equal(u, u2) and np.array_equal(v, v2), gates


def test_large_levels_take_the_xcd_path_by_default(monkeypatch):
    """cfg-2 sized levels: the path is chosen without the override (avg level >= 4096 ratings)."""
    monkeypatch.delenv("MFX_K2_XCD")
    from mfrec_amd.plan import SgdPlan, Timer
    from oracle import oracle as O
    nu, ni, nnz, k = 60000, 12000, 8000000, 2          # ~1200 levels of ~6.5 K ratings
    idx, r = synth.make_ratings(nu, ni, nnz, seed=4)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(u, v)
        p.run_funk(2, 1e9, 0.1, 0.001, 0.05)
        p.get_factors(u, v)
        assert p.info()["n_levels"] * 4096 <= nnz           # i.e. the size rule selects the XCD pass
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)


@pytest.mark.parametrize("skew", ["zipf_items", "zipf_users", "few_users"])
def test_uneven_load_bitexact(skew):
    """Hand-offs must hold under UNEVEN load: skewed items (long chains of tiny levels next to fat ones), skewed
    users (one workgroup's group carries most of a level, others sit in the barrier), and fewer users than groups
    (most workgroups own empty slices)."""
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    rs = np.random.RandomState(5)
    nu, ni, nnz, k = 6000, 4000, 300000, 2
    if skew == "zipf_items":
        users = rs.randint(0, nu, nnz)
        items = np.minimum((rs.zipf(1.3, nnz) - 1), ni - 1)
    elif skew == "zipf_users":
        users = np.minimum((rs.zipf(1.2, nnz) - 1), nu - 1)
        items = rs.randint(0, ni, nnz)
    else:
        nu = 20
        users = rs.randint(0, nu, nnz)
        items = rs.randint(0, ni, nnz)
    idx = np.ascontiguousarray(np.stack([users, items], 1).astype(np.int32))
    r = rs.randint(1, 6, nnz).astype(np.float64)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    G.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni)
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
