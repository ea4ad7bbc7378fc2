"""GPU: estimator_loop_with_implicit_feedback (gd_estimator.pyx:785-898, the SVD++-like prototype) through the
C-ABI -- bit for bit against the reference's own outputs (golden g11) and against the oracle on a larger problem,
with the three feature vectors in LDS and (MovieLens-1M-like item count) in global memory; recommender level too."""
import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)


def _feedback(idx, nu):
    pairs = np.array(sorted(set(map(tuple, idx.tolist()))), dtype=np.int32)
    counts = np.bincount(pairs[:, 0], minlength=nu)
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    h = np.stack([np.where(counts > 0, first, -1), np.where(counts > 0, counts, -1)], axis=1).astype(np.int32)
    return pairs, np.ascontiguousarray(h)


def test_golden_fixed_epochs_and_stopping_rule(golden):
    from mfrec_amd.lib import gd_estimator as G
    g = golden("g11_implicit")
    k, E = int(g["p_k"]), int(g["p_epochs"])
    idx, r = g["in_idx"], g["in_r"]
    u, v, y = synth.init_const(k, 40), synth.init_const(k, 60), np.zeros((k, 40))
    hist = G.estimator_loop_with_implicit_feedback(E, E, 1e9, k, float(g["p_init"]), float(g["p_lr"]), 0, 0, float(g["p_K"]),
                                                   float(g["p_mu"]), u, v, y, idx, r, g["in_rated"], g["in_rated_hash"],
                                                   g["in_ib"], g["in_ub"], 60, 40)
    assert np.array_equal(u, g["out_u"]) and np.array_equal(v, g["out_v"]) and np.array_equal(y, g["out_y"])
    assert np.array_equal(hist, g["out_rmse_hist"])
    u, v, y = synth.init_const(k, 40), synth.init_const(k, 60), np.zeros((k, 40))
    hist = G.estimator_loop_with_implicit_feedback(int(g["p_b_min_epochs"]), 0, float(g["p_b_min_improvement"]), k,
                                                   float(g["p_init"]), float(g["p_b_lr"]), 0, 0, float(g["p_K"]),
                                                   float(g["p_mu"]), u, v, y, idx, r, g["in_rated"], g["in_rated_hash"],
                                                   g["in_ib"], g["in_ub"], 60, 40)
    assert np.array_equal(u, g["out_b_u"]) and np.array_equal(v, g["out_b_v"]) and np.array_equal(y, g["out_b_y"])
    assert np.array_equal(hist, g["out_b_rmse_hist"])


@pytest.mark.parametrize("nu,ni,nnz", [(943, 1682, 20000), (300, 12000, 6000)])     # LDS-resident / global (2 ni + nu > 19 K)
def test_vs_oracle(nu, ni, nnz):
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    k = 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=nu)
    rated, h = _feedback(idx, nu)
    rs = np.random.RandomState(1)
    ib, ub = rs.normal(0, 0.2, ni), rs.normal(0, 0.2, nu)
    u, v, y = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros((k, ni))
    hist = G.estimator_loop_with_implicit_feedback(2, 2, 1e9, k, 0.1, 0.01, 0, 0, 0.05, 3.4, u, v, y, idx, r, rated, h, ib, ub, nu, ni)
    u2, v2, y2, h2 = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros((k, ni)), np.zeros(2 * k)
    O.estimator_loop_with_implicit_feedback(2, 2, 1e9, k, 0.1, 0.01, 0, 0, 0.05, 3.4, u2, v2, y2, idx, r, rated, h, ib, ub, nu, ni, 0, h2)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(y, y2)
    assert np.array_equal(hist, h2)


def test_recommender_level_and_errors():
    from mfrec_amd.lib import gd_estimator as G
    from mfrec_amd.recommendation import GDRecommender
    from oracle import oracle as O
    nu, ni, k = 80, 60, 3
    R = GDRecommender(nu, ni, {"min_epochs": 2, "max_epochs": 2, "min_improvement": 1e9, "nbr_features": k})
    idx, r = synth.make_ratings(nu, ni, 1500, seed=3)
    R.set_ratings(idx, r)
    np.random.seed(4)
    R.feature_training_implicit()
    np.random.seed(4)
    i2, r2 = R.get_ratings(randomize_order=True)
    rated, h = R.get_feedback()
    assert np.array_equal(rated, _feedback(i2, nu)[0]) and np.array_equal(h, _feedback(i2, nu)[1])
    u, v, y = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros((k, ni))
    O.estimator_loop_with_implicit_feedback(2, 2, 1e9, k, 0.1, 0.001, 0.001, 0.001, 0.05, R.overall_bias, u, v, y, i2, r2,
                                            rated, h, R.items_bias, R.users_bias, nu, ni)
    assert np.array_equal(R.svd_u, u) and np.array_equal(R.svd_v, v) and np.array_equal(R.items_feedback, y)
    # the paired predictor (gradient_descent.py:651-667), restated term by term
    user, item = int(i2[0, 0]), 7
    items_of_user = rated[h[user, 0]:h[user, 0] + h[user, 1], 1]
    p_u = R.svd_v[:, user] + (1.0 / np.sqrt(h[user, 1])) * R.items_feedback[:, items_of_user].sum(axis=1)
    want = np.dot(R.svd_u[:, item], p_u) + R.overall_bias + (R.items_bias[item] + R.users_bias[user])
    assert R.predict_rating_implicit(item, user) == want
    bad = h.copy(); bad[int(i2[0, 0])] = [5, 10 ** 6]
    with pytest.raises(ValueError, match="does not index"):
        G.estimator_loop_with_implicit_feedback(1, 1, 1e9, k, 0.1, 0.001, 0, 0, 0.05, 3.0, u, v, y, i2, r2, rated, bad,
                                                R.items_bias, R.users_bias, nu, ni)
