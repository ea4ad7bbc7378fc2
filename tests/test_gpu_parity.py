"""GPU: the HIP path, called through the C-ABI, against the golden vectors and the oracle.

Tolerances: linear full-k SGD, Funk and learned-bias families -> BIT-EXACT factors (the level
schedule + in-order summation reproduce the sequential loop); logistic -> 1e-12 relative
(device exp() differs from libm in the last ulp); online RMSE -> 1e-12 relative (parallel
sum of squared errors)."""
import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)            # raises if the HIP library or the GPU is missing: no fallback
    yield


def eq(a, b):
    assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), \
        "max abs diff %g" % np.abs(a - b).max()


def test_linear_golden_bitexact(golden):
    from mfrec_amd.lib import kmf_train as K
    g = golden("g1_linear")
    E, k = int(g["p_epochs"]), int(g["p_k"])
    for gate in ((1, 1), (1, 0), (0, 1)):
        tag = "%d%d" % gate
        u, v = g["in_u0"].copy(), g["in_v0"].copy()
        ib, ub = np.zeros(40), np.zeros(60)
        rm = K.train_kernel_rmse(0, E, k, float(g["p_lr"]), float(g["p_Ku"]), float(g["p_Ki"]),
                                 float(g["p_Kb"]), u, v, g["in_idx"], g["in_r"], ib, ub, *gate)
        eq(u, g["out_u_" + tag]); eq(v, g["out_v_" + tag])
        eq(ib, g["out_ib_" + tag]); eq(ub, g["out_ub_" + tag])
        np.testing.assert_allclose(rm, g["out_rmse_" + tag], rtol=1e-12)


def test_logistic_golden(golden):
    from mfrec_amd.lib import kmf_train as K
    g = golden("g2_logistic")
    E, k = int(g["p_epochs"]), int(g["p_k"])
    for gate in ((1, 1), (1, 0), (0, 1)):
        tag = "%d%d" % gate
        u, v = g["in_u0"].copy(), g["in_v0"].copy()
        ib, ub = np.zeros(40), np.zeros(60)
        rm = K.train_kernel_rmse(1, E, k, float(g["p_lr"]), float(g["p_Ku"]), float(g["p_Ki"]),
                                 float(g["p_Kb"]), u, v, g["in_idx"], g["in_r"], ib, ub, *gate)
        for a, b in ((u, g["out_u_" + tag]), (v, g["out_v_" + tag]), (ib, g["out_ib_" + tag]),
                     (ub, g["out_ub_" + tag]), (rm, g["out_rmse_" + tag])):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("k", [1, 3, 40, 64, 128])
def test_linear_edges_bitexact(golden, k):
    from mfrec_amd.lib import kmf_train as K
    g = golden("g9_edges")
    if k in (1, 3):      # duplicates, ratings outside [1,5]
        u, v, ib, ub = g["in_u0_k%d" % k].copy(), g["in_v0_k%d" % k].copy(), np.zeros(5), np.zeros(6)
        K.train_linear_kernel(4, k, 0.1, 0.05, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, g["in_idx_a"],
                              g["in_r_a"], ib, ub)
    else:
        u, v, ib, ub = g["in_u0_k%d" % k].copy(), g["in_v0_k%d" % k].copy(), np.zeros(50), np.zeros(80)
        K.train_linear_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, g["in_idx_b"],
                              g["in_r_b"], ib, ub)
    eq(u, g["out_lin_u_k%d" % k]); eq(v, g["out_lin_v_k%d" % k])
    eq(ib, g["out_lin_ib_k%d" % k]); eq(ub, g["out_lin_ub_k%d" % k])


@pytest.mark.parametrize("k", [40, 64, 128])
def test_logistic_edges(golden, k):
    from mfrec_amd.lib import kmf_train as K
    g = golden("g9_edges")
    u, v, ib, ub = g["in_u0_k%d" % k].copy(), g["in_v0_k%d" % k].copy(), np.zeros(50), np.zeros(80)
    K.train_logistic_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, g["in_idx_b"],
                            g["in_r_b"], ib, ub)
    np.testing.assert_allclose(u, g["out_log_u_k%d" % k], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(v, g["out_log_v_k%d" % k], rtol=1e-12, atol=1e-15)


def test_funk_golden_bitexact(golden):
    from mfrec_amd.lib import gd_estimator as G
    g = golden("g3_funk")
    k, E = int(g["p_k"]), int(g["p_epochs"])
    u, v = synth.init_const(k, 40), synth.init_const(k, 60)
    hist = G.funk_rmse_history(E, 1e9, k, float(g["p_init"]), float(g["p_lr"]), float(g["p_K"]),
                               u, v, g["in_idx"], g["in_r"])
    eq(u, g["out_u"]); eq(v, g["out_v"])
    np.testing.assert_allclose(hist, g["out_rmse_hist"], rtol=1e-12)
    # the drop-in name
    u, v = synth.init_const(k, 40), synth.init_const(k, 60)
    G.estimator_loop_without_bias(E, E, 1e9, k, float(g["p_init"]), float(g["p_lr"]),
                                  float(g["p_K"]), u, v, g["in_idx"], g["in_r"], 60, 40, 0)
    eq(u, g["out_u"]); eq(v, g["out_v"])


def test_funk_stopping_rule(golden):
    from mfrec_amd.lib import gd_estimator as G
    g = golden("g3b_funk_stop")
    k = int(g["p_k"])
    u, v = synth.init_const(k, 40), synth.init_const(k, 60)
    hist = G.funk_rmse_history(int(g["p_min_epochs"]), float(g["p_min_improvement"]), k,
                               float(g["p_init"]), float(g["p_lr"]), float(g["p_K"]), u, v,
                               g["in_idx"], g["in_r"])
    assert len(hist) == int(g["out_n_epochs"])
    eq(u, g["out_u"]); eq(v, g["out_v"])


def test_funk_bias_and_gates(golden):
    from mfrec_amd.lib import gd_estimator as G
    g = golden("g4_funk_bias")
    k, E = int(g["p_k"]), int(g["p_epochs"])
    a = (E, E, 1e9, k, float(g["p_init"]), float(g["p_lr"]), 0, 0, float(g["p_K"]), float(g["p_mu"]))
    for tag, gate in (("11", None), ("10", (1, 0)), ("01", (0, 1))):
        u, v = synth.init_const(k, 40), synth.init_const(k, 60)
        ib, ub = g["in_ib"].copy(), g["in_ub"].copy()
        if gate is None:
            G.estimator_loop_with_bias(*a, u, v, g["in_idx"], g["in_r"], ib, ub, 60, 40)
        else:
            G.estimator_loop_with_bias_dev(*a, u, v, g["in_idx"], g["in_r"], ib, ub, 60, 40, *gate)
        eq(u, g["out_u_" + tag]); eq(v, g["out_v_" + tag])
        eq(ib, g["in_ib"]); eq(ub, g["in_ub"])               # biases are inputs only


def test_learned_bias_bitexact(golden):
    from mfrec_amd.lib import gd_estimator as G
    g = golden("g5_learned_bias")
    k, E = int(g["p_k"]), int(g["p_epochs"])
    u, v = synth.init_const(k, 40), synth.init_const(k, 60)
    ib, ub = g["in_ib"].copy(), g["in_ub"].copy()
    G.estimator_loop_with_learned_bias(E, E, 1e9, k, float(g["p_init"]), float(g["p_lr"]),
                                       float(g["p_lr_users"]), float(g["p_lr_items"]),
                                       float(g["p_K"]), float(g["p_Kb"]), float(g["p_mu"]), u, v,
                                       g["in_idx"], g["in_r"], ib, ub, 60, 40)
    eq(u, g["out_u"]); eq(v, g["out_v"]); eq(ib, g["out_ib"]); eq(ub, g["out_ub"])


def test_cfg1_shape_linear_and_funk(golden):
    """BASELINE configs[0] shape (943 x 1682, 100k ratings, k=16, 10 epochs)."""
    from mfrec_amd.lib import gd_estimator as G
    from mfrec_amd.lib import kmf_train as K
    from oracle import oracle as O
    g = golden("g7_cfg1")
    nu, ni, k, E = int(g["p_nu"]), int(g["p_ni"]), int(g["p_k"]), int(g["p_epochs"])
    idx, r = g["in_idx"], g["in_r"].astype(np.float64)
    seed = int(g["p_seed_init"])
    u, v = synth.init_normal(k, ni, seed), synth.init_normal(k, nu, seed + 1)
    ib, ub = np.zeros(ni), np.zeros(nu)
    rm = K.train_kernel_rmse(0, E, k, 0.01, 0.1, 0.1, 0.007, u, v, idx, r, ib, ub)
    eq(u, g["out_lin_u"]); eq(v, g["out_lin_v"]); eq(ib, g["out_lin_ib"]); eq(ub, g["out_lin_ub"])
    np.testing.assert_allclose(rm, g["out_lin_rmse"], rtol=1e-12)
    assert abs(O.rmse(O.RMSE_KMF_LINEAR, u, v, idx, r, ib, ub) - float(g["out_lin_final_rmse"])) <= 1e-5
    fu, fv = synth.init_const(k, ni), synth.init_const(k, nu)
    G.estimator_loop_without_bias(E, E, 1e9, k, 0.1, 0.001, 0.05, fu, fv, idx, r, nu, ni, 0)
    eq(fu, g["out_funk_u"]); eq(fv, g["out_funk_v"])


def test_medium_vs_oracle_k64():
    """A problem big enough for multi-block levels (k = 64, 200k ratings), vs the oracle."""
    from mfrec_amd.lib import kmf_train as K
    from oracle import oracle as O
    nu, ni, nnz, k, E = 4000, 3000, 200000, 64, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=5)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    for mode, fn, ofn in ((0, K.train_linear_kernel, O.train_linear_kernel),
                          (1, K.train_logistic_kernel, O.train_logistic_kernel)):
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        fn(E, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)
        u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        ofn(E, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2)
        if mode == 0:
            eq(u, u2); eq(v, v2); eq(ib, ib2); eq(ub, ub2)
        else:
            np.testing.assert_allclose(u, u2, rtol=1e-11, atol=1e-14)
            np.testing.assert_allclose(v, v2, rtol=1e-11, atol=1e-14)
        kind = O.RMSE_KMF_LINEAR if mode == 0 else O.RMSE_KMF_LOGISTIC
        assert abs(O.rmse(kind, u, v, idx, r, ib, ub) - O.rmse(kind, u2, v2, idx, r, ib2, ub2)) <= 1e-5


def test_empty_and_errors():
    from mfrec_amd.lib import kmf_train as K
    u, v = np.ones((4, 3)), np.ones((4, 5))
    ib, ub = np.zeros(3), np.zeros(5)
    K.train_linear_kernel(2, 4, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v,
                          np.zeros((0, 2), dtype=np.int32), np.zeros(0), ib, ub)
    assert np.all(u == 1.0) and np.all(v == 1.0)          # empty input changes nothing
    with pytest.raises(ValueError, match="outside"):       # the reference would corrupt memory
        K.train_linear_kernel(1, 4, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v,
                              np.array([[0, 3]], dtype=np.int32), np.array([3.0]), ib, ub)


def _zipf_problem(nu, ni, nnz, seed):
    rs = np.random.RandomState(seed)
    pi = 1.0 / np.arange(1, ni + 1) ** 1.1
    pu = 1.0 / np.arange(1, nu + 1) ** 0.8
    idx = np.stack([rs.choice(nu, nnz, p=pu / pu.sum()), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32)
    return idx, rs.randint(1, 6, nnz).astype(np.float64)


@pytest.mark.parametrize("k", [16, 64])
def test_power_law_tail_kernel_bitexact(k, monkeypatch):
    """Skewed degrees: thousands of tiny levels (the chain of the hottest item) go through the
    single-workgroup tail kernel; result must still be the sequential loop's, bit for bit, and
    identical to the launch-per-level path (MFX_K1_TAIL=0)."""
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    nu, ni, nnz = 3000, 2000, 60000
    idx, r = _zipf_problem(nu, ni, nnz, 3)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    res = []
    for tail in ("1", "0"):
        monkeypatch.setenv("MFX_K1_TAIL", tail)
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        with SgdPlan(idx, r, nu, ni, k) as p:
            info = p.info()
            p.set_factors(u, v, ib, ub)
            rm = p.run_kmf(0, 2, 0.01, 0.1, 0.1, 0.007)
            p.get_factors(u, v, ib, ub)
        res.append((u, v, ib, ub, rm, info))
    assert res[0][5]["tail_launches"] >= 1 and res[1][5]["tail_launches"] == 0
    assert res[0][5]["launches_per_epoch"] < res[1][5]["launches_per_epoch"] / 4
    u2, v2, ib2, ub2, rm2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu), np.zeros(2)
    O.train_linear_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2, 1, 1, 0, rm2)
    for u, v, ib, ub, rm, _ in res:
        eq(u, u2); eq(v, v2); eq(ib, ib2); eq(ub, ub2)
        np.testing.assert_allclose(rm, rm2, rtol=1e-12)


def test_power_law_funk_tail_bitexact(monkeypatch):
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    nu, ni, nnz, k = 3000, 2000, 60000, 4
    idx, r = _zipf_problem(nu, ni, nnz, 5)
    fu2, fv2 = synth.init_const(k, ni), synth.init_const(k, nu)
    O.estimator_loop_without_bias(3, 3, 1e9, k, 0.1, 0.001, 0.05, fu2, fv2, idx, r, nu, ni)
    for tail in ("1", "0"):
        monkeypatch.setenv("MFX_K1_TAIL", tail)
        fu, fv = synth.init_const(k, ni), synth.init_const(k, nu)
        G.estimator_loop_without_bias(3, 3, 1e9, k, 0.1, 0.001, 0.05, fu, fv, idx, r, nu, ni)
        eq(fu, fu2); eq(fv, fv2)


def test_funk_dense_cache_family_bitexact(golden):
    """estimator_loop / estimator_loop2 / estimator_subloop / predictor_subloop: the dense-cache
    semantics (duplicates share a cell) against the reference's goldens."""
    from mfrec_amd.lib import gd_estimator as G
    g = golden("g3_funk")
    k, E = int(g["p_k"]), int(g["p_epochs"])
    init, lr, K = float(g["p_init"]), float(g["p_lr"]), float(g["p_K"])
    idx, r = g["in_idx"], g["in_r"]
    assert len(set(map(tuple, idx))) < len(idx)              # the fixture does contain duplicates
    u, v, hist = synth.init_const(k, 40), synth.init_const(k, 60), np.zeros(k * E)
    G.estimator_loop(E, E, 1e9, k, init, lr, K, u, v, idx, r, 0, hist, 60, 40, 0)
    eq(u, g["out_dense_u"]); eq(v, g["out_dense_v"])
    np.testing.assert_allclose(hist, g["out_dense_rmse_hist"], rtol=1e-12)
    u, v = synth.init_const(k, 40), synth.init_const(k, 60)
    G.estimator_loop2(E, E, 1e9, k, init, lr, K, u, v, idx, r, np.zeros(40), 60, 40)
    eq(u, g["out_dense_u"]); eq(v, g["out_dense_v"])
    u, v, c, sub = synth.init_const(k, 40), synth.init_const(k, 60), np.zeros(2400), []
    for f in range(2):                                       # gradient_descent.py:322-328 pattern
        for _ in range(3):
            sub.append(G.estimator_subloop(f, 1, 0.0, k, init, lr, K, u, v, idx, r, c, 60, 40))
        G.predictor_subloop(f, 1, k, init, u, v, idx, r, c, 60, 40)
    np.testing.assert_allclose(sub, g["out_sub_rmse"], rtol=1e-12)
    eq(u, g["out_sub_u"]); eq(v, g["out_sub_v"]); eq(c, g["out_sub_cache"])


def test_plan_reuse_param_change_and_family_switch():
    """One resident plan: linear epochs, then different hyper-parameters (graph re-capture),
    then logistic, then Funk passes on the same factors, then linear again -- every step must
    continue exactly where the oracle continues (layout switches [n,k] <-> [k,n] included)."""
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    nu, ni, nnz, k = 700, 500, 30000, 24
    idx, r = synth.make_ratings(nu, ni, nnz, seed=31)
    u, v, ib, ub = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2), np.zeros(ni), np.zeros(nu)
    u2, v2, ib2, ub2 = u.copy(), v.copy(), ib.copy(), ub.copy()
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(u, v, ib, ub)
        p.run_kmf(0, 2, 0.01, 0.1, 0.1, 0.007)
        p.run_kmf(0, 1, 0.02, 0.05, 0.2, 0.0)            # new parameters -> new graph
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, 1, 0)     # gates
        p.run_funk(2, 1e9, 0.1, 0.001, 0.05)             # [n,k] -> [k,n], Funk on the same factors
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007)           # and back
        p.get_factors(u, v, ib, ub)
    a = (0.1, )
    O.train_linear_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2)
    O.train_linear_kernel(1, k, 0.1, 0.02, 0, 0, 0.05, 0.2, 0.0, 0.0, u2, v2, idx, r, ib2, ub2)
    O.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2, 1, 0)
    O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni)
    O.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2)
    eq(u, u2); eq(v, v2); eq(ib, ib2); eq(ub, ub2)


def test_many_one_shot_calls_do_not_leak_or_drift():
    from mfrec_amd import _shim as S
    from mfrec_amd.lib import kmf_train as K
    nu, ni, nnz, k = 300, 200, 5000, 16
    idx, r = synth.make_ratings(nu, ni, nnz, seed=9)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    first = None
    for _ in range(40):
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        K.train_linear_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)
        if first is None:
            first = (u.copy(), v.copy())
        assert np.array_equal(u, first[0]) and np.array_equal(v, first[1])
    # a failing call leaves a message and does not poison the next one
    with pytest.raises(ValueError):
        K.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v,
                              np.array([[nu, 0]], dtype=np.int32), np.array([1.0]), ib, ub)
    assert "outside" in S.last_error()
    K.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)


def test_f32_storage_fast_mode_rmse_budget(golden):
    """The measured option: factor rows stored as f32, arithmetic in f64.
    Not bit-exact by construction; the contract is the north-star's RMSE budget: final train
    RMSE within 1e-5 of the reference after the fixed epoch count (cfg-1 shape, 10 epochs)."""
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    g = golden("g7_cfg1")
    nu, ni, k, E = int(g["p_nu"]), int(g["p_ni"]), int(g["p_k"]), int(g["p_epochs"])
    idx, r = g["in_idx"], g["in_r"].astype(np.float64)
    seed = int(g["p_seed_init"])
    u, v = synth.init_normal(k, ni, seed), synth.init_normal(k, nu, seed + 1)
    ib, ub = np.zeros(ni), np.zeros(nu)
    with SgdPlan(idx, r, nu, ni, k, storage="f32") as p:
        p.set_factors(u, v, ib, ub)
        rm = p.run_kmf(0, E, 0.01, 0.1, 0.1, 0.007)
        p.get_factors(u, v, ib, ub)
    rmse = O.rmse(O.RMSE_KMF_LINEAR, u, v, idx, r, ib, ub)
    assert abs(rmse - float(g["out_lin_final_rmse"])) <= 1e-5
    assert np.max(np.abs(rm - g["out_lin_rmse"])) <= 1e-5
    assert not np.array_equal(u, g["out_lin_u"])                 # it is NOT the exact mode
    np.testing.assert_allclose(u, g["out_lin_u"], rtol=0, atol=2e-5)
    with pytest.raises(ValueError):
        SgdPlan(idx, r, nu, ni, k, storage="f16")


def test_bf16_storage_is_a_measured_option_that_misses_the_budget(golden):
    # SURVEY 8(f)4 names bf16 row storage next to f32.  Built, measured, NOT recommended: with 8 significant
    # bits the row updates are quantised to ~0.4 % of the value, and after the fixed epoch count the train RMSE is
    # off by far more than north_star's 1e-5 (while f32 storage, above, stays inside it).  The test pins both
    # facts: the mode trains (RMSE falls, finite, deterministic) and it does not meet the budget.
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    g = golden("g7_cfg1")
    nu, ni, k, E = int(g["p_nu"]), int(g["p_ni"]), int(g["p_k"]), int(g["p_epochs"])
    idx, r = g["in_idx"], g["in_r"].astype(np.float64)
    seed = int(g["p_seed_init"])
    res = []
    for rep in range(2):
        u, v = synth.init_normal(k, ni, seed), synth.init_normal(k, nu, seed + 1)
        ib, ub = np.zeros(ni), np.zeros(nu)
        with SgdPlan(idx, r, nu, ni, k, storage="bf16") as p:
            assert "bf16" in p.kernel_name(0)
            p.set_factors(u, v, ib, ub)
            rm = p.run_kmf(0, E, 0.01, 0.1, 0.1, 0.007)
            p.get_factors(u, v, ib, ub)
        res.append((u, v, rm))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][2], res[1][2])      # deterministic
    u, v, rm = res[0]
    rmse = O.rmse(O.RMSE_KMF_LINEAR, u, v, idx, r, ib, ub)
    delta = abs(rmse - float(g["out_lin_final_rmse"]))
    print("bf16 storage: final train RMSE %.6f vs reference %.6f (delta %.2e)" % (rmse, float(g["out_lin_final_rmse"]), delta))
    assert np.all(np.isfinite(rm)) and rm[-1] < rm[0]            # it does train
    assert delta > 1e-5                                          # ... and misses north_star's budget
    assert delta < 0.2                                           # (without diverging)
    with pytest.raises(ValueError):
        SgdPlan(idx, r, nu, ni, 12, storage="bf16")              # k must be a multiple of 8


def test_butterfly_dot_tolerance_mode(golden):
    """North-star's own wording of the kernel ("wavefront __shfl/ds_swizzle reductions for the k-dim dot
    product"): per-lane partial products + a wave butterfly with a FIXED association, FMA in dot and row
    update.  Deterministic and f64, but not the reference's left-to-right sum (kmf_train.pyx:75-97): the
    contract is the tolerance north_star states -- final RMSE within 1e-5 (cfg-1 shape, 10 epochs; measured:
    ~1e-13) -- plus run-to-run bit-reproducibility.  Never the default."""
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    g = golden("g7_cfg1")
    nu, ni, k, E = int(g["p_nu"]), int(g["p_ni"]), int(g["p_k"]), int(g["p_epochs"])
    idx, r = g["in_idx"], g["in_r"].astype(np.float64)
    seed = int(g["p_seed_init"])
    outs = []
    for rep in range(2):
        u, v = synth.init_normal(k, ni, seed), synth.init_normal(k, nu, seed + 1)
        ib, ub = np.zeros(ni), np.zeros(nu)
        with SgdPlan(idx, r, nu, ni, k, dot="butterfly") as p:
            assert "butterfly-dot" in p.kernel_name(0)
            p.set_factors(u, v, ib, ub)
            rm = p.run_kmf(0, E, 0.01, 0.1, 0.1, 0.007)
            p.get_factors(u, v, ib, ub)
        outs.append((u, v, ib, ub, rm))
    u, v, ib, ub, rm = outs[0]
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)                                # deterministic
    rmse = O.rmse(O.RMSE_KMF_LINEAR, u, v, idx, r, ib, ub)
    assert abs(rmse - float(g["out_lin_final_rmse"])) <= 1e-5      # north_star's bar
    assert abs(rmse - float(g["out_lin_final_rmse"])) <= 1e-10     # what is actually measured
    assert np.max(np.abs(rm - g["out_lin_rmse"])) <= 1e-10
    assert not np.array_equal(u, g["out_lin_u"])                   # it is NOT the exact mode
    np.testing.assert_allclose(u, g["out_lin_u"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(v, g["out_lin_v"], rtol=0, atol=1e-10)
    # several k (lanes per rating 8 / 16 / 32 / 64, partial batches, logistic) against the oracle
    for kk, mode in ((6, 0), (16, 0), (24, 1), (64, 0), (100, 0), (128, 1), (200, 0)):
        nu2, ni2, nnz2 = 211, 97, 5003
        i2, r2 = synth.make_ratings(nu2, ni2, nnz2, seed=kk)
        u0, v0 = synth.init_normal(kk, ni2, 1), synth.init_normal(kk, nu2, 2)
        ua, va, iba, uba = u0.copy(), v0.copy(), np.zeros(ni2), np.zeros(nu2)
        with SgdPlan(i2, r2, nu2, ni2, kk, dot="butterfly") as p:
            p.set_factors(ua, va, iba, uba)
            p.run_kmf(mode, 3, 0.01, 0.1, 0.1, 0.007)
            p.get_factors(ua, va, iba, uba)
        ub_, vb, ibb, ubb = u0.copy(), v0.copy(), np.zeros(ni2), np.zeros(nu2)
        fn = O.train_linear_kernel if mode == 0 else O.train_logistic_kernel
        fn(3, kk, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, ub_, vb, i2, r2, ibb, ubb)
        np.testing.assert_allclose(ua, ub_, rtol=0, atol=1e-11)
        np.testing.assert_allclose(va, vb, rtol=0, atol=1e-11)
        np.testing.assert_allclose(iba, ibb, rtol=0, atol=1e-11)
    with pytest.raises(ValueError):
        SgdPlan(idx, r, nu, ni, 7, dot="butterfly")                 # odd k: no 16-byte feature pairs
