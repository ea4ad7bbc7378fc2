"""GPU: the recommender classes (train / fit / predict surface) against the oracle run on
the same injected permutation and initial factors (the reference shuffles and initialises
with the unseeded global numpy RNG: base.py:1128-1129, mf.py:132-133)."""
import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    yield


def _fill(R, nu, ni, nnz, seed):
    idx, r = synth.make_ratings(nu, ni, nnz, seed=seed)
    R.set_ratings(idx, r)
    return R


def test_kmf_recommender_matches_oracle():
    from mfrec_amd.recommendation import KMFRecommender
    from oracle import oracle as O
    nu, ni, k = 200, 150, 12
    for kernel, ofn, pred in (("train_linear_kernel", O.train_linear_kernel, "predict_linear"),
                              ("train_logistic_kernel", O.train_logistic_kernel, "predict_logistic")):
        R = _fill(KMFRecommender(nu, ni, {"nbr_epochs": 4, "nbr_features": k}), nu, ni, 5000, 9)
        np.random.seed(123)
        R.train(kernel=kernel)
        # replay the same RNG stream for the oracle
        np.random.seed(123)
        u = np.random.normal(0.0, 0.1, [k, ni]); v = np.random.normal(0.0, 0.1, [k, nu])
        idx, r = R.get_ratings(randomize_order=True)
        ib, ub = np.zeros(ni), np.zeros(nu)
        ofn(4, k, 0.1, 0.01, 0.01, 0.01, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)
        if kernel == "train_linear_kernel":
            assert np.array_equal(R.svd_u, u) and np.array_equal(R.svd_v, v)
            assert np.array_equal(R.items_bias, ib) and np.array_equal(R.users_bias, ub)
        else:
            np.testing.assert_allclose(R.svd_u, u, rtol=1e-11, atol=1e-14)
        kind = O.RMSE_KMF_LINEAR if pred == "predict_linear" else O.RMSE_KMF_LOGISTIC
        assert abs(R.train_rmse(pred) - O.rmse(kind, u, v, idx, r, ib, ub)) <= 1e-5
        # scalar predict(item, user) agrees with the batched one
        p = getattr(R, pred)(int(idx[0, 1]), int(idx[0, 0]))
        from mfrec_amd.recommendation.metrics import predict_batch
        assert abs(predict_batch(R, idx[:1], pred)[0][0] - p) < 1e-12
    assert KMFRecommender.fit is KMFRecommender.train


def test_kmf_balanced_order_is_a_replayable_exact_order():
    """train(order="balanced"): shuffle (global numpy RNG) + GPU recolouring; replaying the RNG gives the
    same order, and the oracle's sequential loop over it gives the same factors bit for bit."""
    from mfrec_amd.recommendation import KMFRecommender
    from oracle import oracle as O
    nu, ni, k = 300, 200, 10
    R = _fill(KMFRecommender(nu, ni, {"nbr_epochs": 3, "nbr_features": k}), nu, ni, 20000, 4)
    np.random.seed(7)
    R.train(kernel="train_linear_kernel", order="balanced")
    np.random.seed(7)
    u = np.random.normal(0.0, 0.1, [k, ni]); v = np.random.normal(0.0, 0.1, [k, nu])
    idx, r = R.get_ratings(randomize_order="balanced")
    plain, _ = R.get_ratings(randomize_order=False)
    assert sorted(map(tuple, idx.tolist())) == sorted(map(tuple, plain.tolist()))      # a permutation
    ib, ub = np.zeros(ni), np.zeros(nu)
    O.train_linear_kernel(3, k, 0.1, 0.01, 0.01, 0.01, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)
    assert np.array_equal(R.svd_u, u) and np.array_equal(R.svd_v, v)
    assert np.array_equal(R.items_bias, ib) and np.array_equal(R.users_bias, ub)


def test_gd_recommender_matches_oracle():
    from mfrec_amd.recommendation import GDRecommender
    from oracle import oracle as O
    nu, ni, k = 150, 100, 6
    params = {"min_epochs": 4, "max_epochs": 4, "min_improvement": 1e9, "nbr_features": k}
    for handle_bias in (False, True):
        R = _fill(GDRecommender(nu, ni, params), nu, ni, 4000, 4)
        np.random.seed(5)
        R.train(handle_bias=handle_bias)
        np.random.seed(5)
        idx, r = R.get_ratings(randomize_order=True)
        u, v = synth.init_const(k, ni), synth.init_const(k, nu)
        if handle_bias:
            O.estimator_loop_with_bias(4, 4, 1e9, k, 0.1, 0.001, 0, 0, 0.05, R.overall_bias, u, v,
                                       idx, r, R.items_bias, R.users_bias, nu, ni)
            pred, kind = "predict_rating_with_bias", O.RMSE_GD_BIAS
        else:
            O.estimator_loop_without_bias(4, 4, 1e9, k, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni)
            pred, kind = "predict_rating", O.RMSE_GD
        assert np.array_equal(R.svd_u, u) and np.array_equal(R.svd_v, v)
        ref = O.rmse(kind, u, v, idx, r, R.items_bias, R.users_bias, R.overall_bias or 0.0)
        assert abs(R.train_rmse(pred) - ref) <= 1e-5
    # learned-bias trainer (gradient_descent.py:472-503)
    R = _fill(GDRecommender(nu, ni, params), nu, ni, 4000, 4)
    np.random.seed(6)
    R.feature_training_bias()
    assert R.svd_u.shape == (k, ni) and np.isfinite(R.svd_u).all()


def test_wrmf_recommender_matches_oracle():
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    from mfrec_amd.recommendation import WRMFRecommender
    from oracle import oracle as O
    nu, ni, k = 120, 80, 10
    R = _fill(WRMFRecommender(nu, ni, {"nbr_epochs": 1, "nbr_features": k}), nu, ni, 3000, 8)
    R.train()
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    a, b = create_bool_sparse_row(R.relationship_matrix)
    c, d = create_bool_sparse_col(R.relationship_matrix)
    O.als_wrmf(1, k, u, v, None, None, a, b, c, d, nu, ni, 1, 0.015)
    np.testing.assert_allclose(R.svd_u, u, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(R.svd_v, v, rtol=1e-8, atol=1e-10)
    assert abs(R.predict(3, 5) - np.dot(u[:, 3], v[:, 5])) < 1e-9


def test_unknown_parameter_raises():
    from mfrec_amd.recommendation import Error, GDRecommender
    with pytest.raises(Error, match="Wrong parameters"):
        GDRecommender(4, 4, {"no_such_parameter": 1})


def test_folding_in_and_top_n():
    """retrain_user / add_user (update gates) and the batched top-N."""
    from mfrec_amd.recommendation import KMFRecommender
    nu, ni, k = 120, 90, 8
    R = _fill(KMFRecommender(nu, ni, {"nbr_epochs": 5, "nbr_features": k}), nu, ni, 3000, 2)
    np.random.seed(1)
    R.train(kernel="train_linear_kernel")
    u_before, v_before = R.svd_u.copy(), R.svd_v.copy()
    items = np.array([3, 10, 20, 33], dtype=np.int32)
    new = R.add_user("someone", items, np.array([5.0, 4.0, 1.0, 2.0]))
    assert new == nu and R.svd_v.shape == (k, nu + 1) and R.nbr_users == nu + 1
    assert R.users_index["someone"] == nu and R.users_label[nu] == "someone"     # kmf.py:156-158
    assert np.array_equal(R.svd_u, u_before)                       # items untouched (gate 0)
    assert np.array_equal(R.svd_v[:, :nu], v_before)               # other users untouched
    assert np.any(R.svd_v[:, nu] != 0.0)
    top, top_scores = R.find_recommended_items(new, nbr_recommendations=5, predictor="predict_linear")
    scores = np.array([R.predict_linear(i, new) for i in range(ni)])
    assert top == list(np.argsort(-scores, kind="stable")[:5])
    assert np.allclose(top_scores, scores[top], rtol=1e-12)
    # an existing user: the reference's rules (mf.py:144-193) restated as the loop it runs -- rated items and the
    # item whose id equals the user id score 0, zero scores are dropped, the global RNG is advanced by one shuffle
    user = 7
    R.neighborhood = 60
    np.random.seed(5)
    got_items, got_scores = R.find_recommended_items(user_index=user, nbr_recommendations=8, predictor="predict_linear")
    after = np.random.randint(0, 1 << 30)
    np.random.seed(5)
    np.random.shuffle(np.arange(R.nbr_items))
    assert after == np.random.randint(0, 1 << 30)
    rated = set(R.relationship_matrix[user, :].tocoo().col.tolist()) | {user}
    want = {i: R.predict_linear(i, user) for i in range(60) if i not in rated}
    ranked = sorted(want.items(), key=lambda kv: (-kv[1], kv[0]))[:8]
    assert got_items == [i for i, _ in ranked] and np.allclose(got_scores, [x for _, x in ranked], rtol=1e-12)
    R.items_label = ["item-%d" % i for i in range(R.nbr_items)]
    R.set_user_label(user, "u7")
    lab, _ = R.find_recommended_items(user_label="u7", nbr_recommendations=3, output_label=True, predictor="predict_linear")
    assert lab == ["item-%d" % i for i in got_items[:3]]
    new_item = R.add_item("thing", np.array([1, 2, 3], dtype=np.int32), np.array([4.0, 4.0, 5.0]))
    assert new_item == ni and R.svd_u.shape == (k, ni + 1)
    assert np.array_equal(R.svd_v[:, :nu], v_before)


def test_gd_learned_bias_trainer_matches_oracle():
    """GDRecommender.feature_training_bias (gradient_descent.py:472-503) at recommender level: replay the RNG,
    rebuild the bias pre-computes, run the oracle's estimator_loop_with_learned_bias -- bit for bit."""
    from mfrec_amd.recommendation import GDRecommender
    from oracle import oracle as O
    nu, ni, k = 150, 100, 5
    params = {"min_epochs": 3, "max_epochs": 3, "min_improvement": 1e9, "nbr_features": k}
    R = _fill(GDRecommender(nu, ni, params), nu, ni, 4000, 4)
    np.random.seed(6)
    R.feature_training_bias()
    Q = _fill(GDRecommender(nu, ni, params), nu, ni, 4000, 4)
    np.random.seed(6)
    idx, r = Q.get_ratings(randomize_order=True)
    Q.compute_overall_avg(); Q.compute_items_bias_bk(); Q.compute_users_bias_bk()
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    ib, ub = Q.items_bias.copy(), Q.users_bias.copy()
    O.estimator_loop_with_learned_bias(3, 3, 1e9, k, 0.1, 0.001, 0.001, 0.001, 0.05, 0.01, Q.overall_bias,
                                       u, v, idx, r, ib, ub, nu, ni)
    assert np.array_equal(R.svd_u, u) and np.array_equal(R.svd_v, v)
    assert np.array_equal(R.items_bias, ib) and np.array_equal(R.users_bias, ub)


def test_folding_in_matches_oracle_and_empty_fold_in_is_a_noop():
    """retrain_user / retrain_item (gradient_descent.py:879-905, kmf.py:120-146) against the oracle with the
    same update gates; a fold-in WITHOUT ratings runs the reference's empty epochs and changes nothing but the
    re-initialised row (it used to raise after the row had been re-initialised)."""
    from mfrec_amd.recommendation import GDRecommender, KMFRecommender
    from oracle import oracle as O
    nu, ni, k = 120, 90, 6
    # --- GD family ---------------------------------------------------------------------------
    params = {"min_epochs": 3, "max_epochs": 3, "min_improvement": 1e9, "nbr_features": k}
    R = _fill(GDRecommender(nu, ni, params), nu, ni, 3000, 2)
    np.random.seed(2)
    R.train(handle_bias=True)
    idx, r = R.get_ratings(randomize_order=False)
    user = int(idx[7, 0])
    u0, v0 = R.svd_u.copy(), R.svd_v.copy()
    np.random.seed(9)
    R.retrain_user(user, idx, r)
    np.random.seed(9)
    v0[:, user] = np.random.normal(0.0, 0.1, k)                      # init_user_features (mf.py:136-137)
    sel = np.where(idx[:, 0] == user)[0]
    O.estimator_loop_with_bias_dev(3, 3, 1e9, k, 0.1, 0.001, 0.001, 0.001, 0.05, R.overall_bias, u0, v0,
                                   np.ascontiguousarray(idx[sel]), np.ascontiguousarray(r[sel]), R.items_bias,
                                   R.users_bias, nu, ni, 1, 0)
    assert np.array_equal(R.svd_u, u0) and np.array_equal(R.svd_v, v0)
    item = int(idx[11, 1])
    np.random.seed(10)
    R.retrain_item(item, idx, r)
    np.random.seed(10)
    u0[:, item] = np.random.normal(0.0, 0.1, k)
    sel = np.where(idx[:, 1] == item)[0]
    O.estimator_loop_with_bias_dev(3, 3, 1e9, k, 0.1, 0.001, 0.001, 0.001, 0.05, R.overall_bias, u0, v0,
                                   np.ascontiguousarray(idx[sel]), np.ascontiguousarray(r[sel]), R.items_bias,
                                   R.users_bias, nu, ni, 0, 1)
    assert np.array_equal(R.svd_u, u0) and np.array_equal(R.svd_v, v0)
    # a user nobody rated for: no ratings selected -> only the re-initialised row differs
    lonely = [x for x in range(nu) if x not in set(idx[:, 0].tolist())]
    if not lonely:
        idx2 = idx[idx[:, 0] != 0]; r2 = r[idx[:, 0] != 0]; lonely = [0]
    else:
        idx2, r2 = idx, r
    before_u, before_v = R.svd_u.copy(), R.svd_v.copy()
    np.random.seed(11)
    R.retrain_user(lonely[0], idx2, r2)                                # must not raise
    np.random.seed(11)
    before_v[:, lonely[0]] = np.random.normal(0.0, 0.1, k)
    assert np.array_equal(R.svd_u, before_u) and np.array_equal(R.svd_v, before_v)
    # --- KMF family --------------------------------------------------------------------------
    K = _fill(KMFRecommender(nu, ni, {"nbr_epochs": 4, "nbr_features": k}), nu, ni, 3000, 2)
    np.random.seed(1)
    K.train(kernel="train_linear_kernel")
    idx, r = K.get_ratings(randomize_order=False)
    user = int(idx[5, 0])
    u0, v0, ib0, ub0 = K.svd_u.copy(), K.svd_v.copy(), K.items_bias.copy(), K.users_bias.copy()
    np.random.seed(3)
    K.retrain_user(user, idx, r, kernel="train_linear_kernel")
    np.random.seed(3)
    v0[:, user] = np.random.normal(0.0, 0.1, k)
    sel = np.where(idx[:, 0] == user)[0]
    O.train_linear_kernel(4, k, 0.1, 0.01, 0.01, 0.01, 0.1, 0.1, 0.007, K.overall_bias, u0, v0,
                          np.ascontiguousarray(idx[sel]), np.ascontiguousarray(r[sel]), ib0, ub0, 1, 0)
    assert np.array_equal(K.svd_u, u0) and np.array_equal(K.svd_v, v0)
    assert np.array_equal(K.items_bias, ib0) and np.array_equal(K.users_bias, ub0)


def test_gd_find_user_top_match_follows_the_reference_rules():
    """gradient_descent.py:768-800 restated as the loop it runs: every item scored with predict_rating, the
    user's rated items and the item whose id equals the user id excluded, zero scores dropped, descending."""
    from mfrec_amd.recommendation import GDRecommender
    nu, ni, k = 80, 70, 6
    R = _fill(GDRecommender(nu, ni, {"min_epochs": 2, "max_epochs": 2, "nbr_features": k}), nu, ni, 1500, 4)
    np.random.seed(2)
    R.train()
    for user in (0, 11, 69, 79):
        items, scores = R.find_user_top_match(user, nbr_recommendations=6)
        rated = set(R.relationship_matrix[user, :].tocoo().col.tolist()) | {user}
        want = {i: R.predict_rating(i, user) for i in range(ni) if i not in rated}
        ranked = sorted(((i, x) for i, x in want.items() if x != 0.0), key=lambda kv: (-kv[1], kv[0]))[:6]
        assert items == [i for i, _ in ranked]
        assert np.allclose(scores, [x for _, x in ranked], rtol=1e-12)


def test_gd_secondary_trainers_match_oracle():
    """feature_training2 (per-epoch subloop on a dense cache, gradient_descent.py:299-328), feature_training_dev
    (estimator_loop, :577-599, returns the rmse history) and feature_training_folding (:450-470) against the
    oracle's loops on the same order and initial factors -- bit for bit."""
    from mfrec_amd.recommendation import GDRecommender
    from oracle import oracle as O
    nu, ni, k = 60, 45, 5
    P = {"min_epochs": 3, "max_epochs": 4, "nbr_features": k, "min_improvement": 1e9}
    R = _fill(GDRecommender(nu, ni, P), nu, ni, 900, 6)
    idx, r = R.get_ratings(False)

    R.feature_training2()
    u, v = np.full((k, ni), 0.1), np.full((k, nu), 0.1)
    cache = np.zeros(nu * ni)
    for f in range(k):
        for e in range(3):
            O.estimator_subloop(f, e, 1e9, k, 0.1, R.learning_rate, R.K, u, v, idx, r, cache, nu, ni)
        O.predictor_subloop(f, 3, k, 0.1, u, v, idx, r, cache, nu, ni)
    assert np.array_equal(R.svd_u, u) and np.array_equal(R.svd_v, v)

    np.random.seed(9)
    hist = R.feature_training_dev()
    u, v = np.full((k, ni), 0.1), np.full((k, nu), 0.1)
    h2 = np.zeros(4 * k)
    O.estimator_loop(3, 4, 1e9, k, 0.1, R.learning_rate, R.K, u, v, idx, r, 0, h2, nu, ni)
    assert np.array_equal(R.svd_u, u) and np.array_equal(R.svd_v, v)
    assert np.allclose(hist, h2, rtol=1e-12, atol=0) and hist.shape == (4 * k,)

    # folding: continues from the model as it stands, shuffled order from the global RNG, then the means
    u0, v0 = R.svd_u.copy(), R.svd_v.copy()
    np.random.seed(4)
    R.feature_training_folding()
    np.random.seed(4)
    sidx, sr = R.get_ratings(True)
    O.estimator_loop_without_bias(3, 4, 1e9, k, 0.1, R.learning_rate, R.K, u0, v0, sidx, sr, nu, ni)
    assert np.array_equal(R.svd_u, u0) and np.array_equal(R.svd_v, v0)
    assert np.array_equal(R.components_mean, R.svd_u.mean(axis=1))
    with pytest.raises(TypeError):
        R.feature_training_folding(handle_bias=True)          # the reference's call is two arguments short
    assert R.estimate_rating(3, 2, 1, trailing=True) == R.clamping(
        R.clamping(1.0 + R.svd_u[1, 3] * R.svd_v[1, 2]) + (k - 2) * 0.1 * 0.1)


def test_metrics_module_follows_the_reference():
    """metrics.py:51-82 (rmse, errors over the first nbr_samples rows, NaN errors dropped) and :85-130
    (precision / recall / F over find_recommended_items), restated as the loops the reference runs."""
    from mfrec_amd.recommendation import KMFRecommender, metrics
    nu, ni, k = 90, 70, 6
    R = _fill(KMFRecommender(nu, ni, {"nbr_epochs": 4, "nbr_features": k}), nu, ni, 2500, 8)
    np.random.seed(3)
    R.train(kernel="train_linear_kernel")
    tidx, tr = synth.make_ratings(nu, ni, 400, seed=77)
    u_test = np.column_stack([tidx, tr])
    rmse, errors = metrics.test_predict_rating(R, u_test, nbr_samples=300, predictor="predict_linear")
    want = np.array([row[2] - R.predict_linear(int(row[1]), int(row[0])) for row in u_test[:300]])
    assert errors.shape == (300,) and np.allclose(errors, want, rtol=1e-12, atol=1e-13)
    assert abs(rmse - np.sqrt((want ** 2).mean())) < 1e-12
    R.svd_v[:, 5] = np.nan                                    # a user without a model: its rows drop out
    rmse2, errors2 = metrics.test_predict_rating(R, u_test, nbr_samples=300, predictor="predict_linear")
    assert errors2.shape[0] == 300 - int((u_test[:300, 0] == 5).sum()) and np.isfinite(rmse2)
    R.svd_v[:, 5] = 0.0

    R.neighborhood = ni
    np.random.seed(1)
    p, r, f = metrics.precision_recall(R, u_test, nbr_recommendations=5, predictor="predict_linear")
    np.random.seed(1)
    by_user = {}
    for row in u_test:
        by_user.setdefault(int(row[0]), []).append(int(row[1]))
    P = Rc = 0.0
    for user, items in by_user.items():
        rec = set(R.find_recommended_items(user_index=user, nbr_recommendations=5, predictor="predict_linear")[0])
        hit = float(len(rec & set(items)))
        P += hit / 5
        Rc += hit / len(set(items))
    P, Rc = P / len(by_user), Rc / len(by_user)
    assert (p, r) == (P, Rc) and f == 2 * P * Rc / (P + Rc)


def test_batched_top_n_equals_the_single_user_ranking():
    """find_recommended_items_batch (mfx_topn_batch: score kernel + segmented radix sort + zero-dropping take):
    row j is find_recommended_items(users[j]) -- items AND score bits -- for every predictor family, including a
    folded-in user, a user whose id is a candidate item, exact zeros, ties, and N larger than what qualifies."""
    from mfrec_amd.recommendation import GDRecommender, KMFRecommender, metrics
    nu, ni, k = 150, 110, 8
    K = _fill(KMFRecommender(nu, ni, {"nbr_epochs": 3, "nbr_features": k}), nu, ni, 4000, 12)
    np.random.seed(2)
    K.train(kernel="train_logistic_kernel")
    K.add_user("late", np.array([1, 2, 3], dtype=np.int32), np.array([5.0, 1.0, 3.0]))
    K.svd_u[:, 40] = 0.0; K.items_bias[40] = 0.0                    # ties / exact zeros for the linear predictor
    K.svd_u[:, 41] = K.svd_u[:, 42]; K.items_bias[41] = K.items_bias[42]
    K.svd_u[:, 50] = np.nan                                          # NaN scores -> 0 -> dropped
    users = [0, 7, 149, 150, 33, 109, 110, 5, 5]
    for pred in ("predict_linear", "predict_logistic"):
        for n_cand, N in ((ni, 6), (64, 70), (17, 3)):
            K.neighborhood = n_cand
            items, scores = K.find_recommended_items_batch(users, N, pred)
            assert items.shape == (len(users), N) and scores.shape == (len(users), N)
            for j, user in enumerate(users):
                K.neighborhood = n_cand
                want_i, want_s = K.find_recommended_items(user_index=user, nbr_recommendations=N, predictor=pred)
                got = [int(x) for x in items[j] if x >= 0]
                assert got == want_i, (pred, n_cand, N, user)
                assert np.array_equal(scores[j, :len(want_s)], np.array(want_s))
                assert np.all(items[j, len(want_i):] == -1)
    G = _fill(GDRecommender(nu, ni, {"min_epochs": 2, "max_epochs": 2, "nbr_features": 5}), nu, ni, 3000, 13)
    np.random.seed(3)
    G.train()
    G.neighborhood = ni
    items, scores = G.find_recommended_items_batch(list(range(0, nu, 7)), 5)
    for j, user in enumerate(range(0, nu, 7)):
        want_i, want_s = G.find_user_top_match(user, 5)
        assert list(items[j]) == want_i and np.array_equal(scores[j], np.array(want_s))
    # the batched precision / recall is the per-user loop's
    tidx, tr = synth.make_ratings(nu, ni, 300, seed=5)
    u_test = np.column_stack([tidx, tr])
    a = metrics.precision_recall(G, u_test, 5, "predict_rating")
    b = metrics.precision_recall(G, u_test, 5, "predict_rating", batched=True)
    assert a == b
    with pytest.raises(IndexError):
        G.find_recommended_items_batch([nu], 3)


@pytest.mark.parametrize("kernel", ["mfma", "tiled", "strided"])
def test_fused_top_n_proves_or_falls_back(monkeypatch, kernel):
    """N <= 16 goes through the fused kernel (no score matrix).  It must (a) equal the sort path and the single-user
    ranking, (b) notice when a thread visits more than its three kept slots' worth of the list -- four of the best
    items sharing the stride-256 lane -- and hand that user to the sort path, (c) survive every user being sent the
    long way (test hook)."""
    import ctypes as C
    from mfrec_amd import _shim as S
    from mfrec_amd.recommendation import KMFRecommender

    # matrix-core kernel (the default: approximate scores select, k_topn_exactify restores the reference's bits or flags) /
    # LDS-tiled vector kernel / column-strided fused kernel
    monkeypatch.setenv("MFX_TOPN_KERNEL", kernel)

    def stats():
        a = (C.c_int64 * 2)()
        S.check(S.lib().mfx_topn_stats(a))
        return a[0], a[1]

    nu, ni, k = 64, 1500, 6
    rs = np.random.RandomState(8)
    R = KMFRecommender(nu, ni, {"nbr_features": k})
    R.set_ratings(np.stack([rs.randint(0, nu, 3000), rs.randint(0, ni, 3000)], 1), rs.randint(1, 6, 3000).astype(float))
    R.svd_u, R.svd_v = rs.normal(0, 0.3, (k, ni)), rs.normal(0, 0.3, (k, nu))
    R.items_bias, R.users_bias, R.overall_bias = rs.normal(0, 0.1, ni), rs.normal(0, 0.1, nu), 3.0
    R.neighborhood = ni
    users = list(range(nu))
    for N in (1, 5, 16):
        f0, r0 = stats()
        fi, fs = R.find_recommended_items_batch(users, N, "predict_linear")
        f1, r1 = stats()
        assert (f1 - f0) + (r1 - r0) == nu and (f1 - f0) >= nu - 8          # fused, almost always proven
        monkeypatch.setenv("MFX_TOPN_FUSED", "0")
        si, ss = R.find_recommended_items_batch(users, N, "predict_linear")
        monkeypatch.delenv("MFX_TOPN_FUSED")
        assert stats() == (f1, r1)                                          # the sort path does not count
        assert np.array_equal(fi, si) and np.array_equal(fs, ss)
        monkeypatch.setenv("MFX_TOPN_FORCE_UNSAFE", "1")
        ui, us = R.find_recommended_items_batch(users, N, "predict_linear")
        monkeypatch.delenv("MFX_TOPN_FORCE_UNSAFE")
        assert stats()[1] - r1 >= (nu + 2) // 3
        assert np.array_equal(fi, ui) and np.array_equal(fs, us)
    # (b): items 5, 261, 517, 773 are visited by the same thread (stride 256) / lane (stride 64) / holder of the matrix-core kernel
    # (blocks 0, 8, 16, 24 = wave 0, column pair 2); make them user 9's four best by far
    for it in (5, 261, 517, 773):
        R.svd_u[:, it] = 0.0
        R.items_bias[it] = 50.0 + it / 1000.0
        R.relationship_matrix[9, it] = 0
    monkeypatch.setenv("MFX_TOPN_SPLITS", "1")     # one workgroup walks all candidates: the four items meet in one lane
    f0, r0 = stats()
    gi, gs = R.find_recommended_items_batch([9, 10], 6, "predict_linear")
    f1, r1 = stats()
    monkeypatch.delenv("MFX_TOPN_SPLITS")
    assert r1 - r0 >= 1                                                     # noticed, redone
    si, ss = R.find_recommended_items_batch([9, 10], 6, "predict_linear")  # candidates split over workgroups: same rows
    assert np.array_equal(si, gi) and np.array_equal(ss, gs)
    for row, user in zip(range(2), (9, 10)):
        wi, wsc = R.find_recommended_items(user_index=user, nbr_recommendations=6, predictor="predict_linear")
        assert list(gi[row]) == wi and np.array_equal(gs[row], np.array(wsc))
    assert list(gi[0][:4]) == [773, 517, 261, 5]
    # N = 17 is the sort path's
    f0, r0 = stats()
    R.find_recommended_items_batch(users, 17, "predict_linear")
    assert stats() == (f0, r0)


@pytest.mark.parametrize("k", [64, 80, 130])
def test_fused_top_n_with_more_than_one_feature_chunk(k, monkeypatch):
    """k > 64: the tiled kernel walks the features in chunks of 64 with the sums carried across chunks -- same bits
    as the sort path and the single-user query (k = 80: a ragged second chunk; k = 130: three chunks)."""
    from mfrec_amd.recommendation import KMFRecommender
    nu, ni = 70, 333
    rs = np.random.RandomState(k)
    R = KMFRecommender(nu, ni, {"nbr_features": k})
    R.set_ratings(np.stack([rs.randint(0, nu, 2000), rs.randint(0, ni, 2000)], 1), rs.randint(1, 6, 2000).astype(float))
    R.svd_u, R.svd_v = rs.normal(0, 0.2, (k, ni)), rs.normal(0, 0.2, (k, nu))
    R.items_bias, R.users_bias, R.overall_bias = rs.normal(0, 0.1, ni), rs.normal(0, 0.1, nu), 3.0
    R.neighborhood = ni
    users = list(range(nu))
    fi, fs = R.find_recommended_items_batch(users, 7, "predict_logistic")
    monkeypatch.setenv("MFX_TOPN_FUSED", "0")
    si, ss = R.find_recommended_items_batch(users, 7, "predict_logistic")
    monkeypatch.delenv("MFX_TOPN_FUSED")
    assert np.array_equal(fi, si) and np.array_equal(fs, ss)
    for user in (0, 33, 69):
        wi, ws = R.find_recommended_items(user_index=user, nbr_recommendations=7, predictor="predict_logistic")
        assert list(fi[user]) == wi and np.array_equal(fs[user], np.array(ws))


def test_topn_batch_argument_checks_and_unsorted_rated_lists():
    """mfx_topn_batch refuses malformed arguments with MFX_E_ARG, and rated lists that are not ascending (the fused
    kernel searches them) silently take the sort path -- same answer."""
    import ctypes as C
    from mfrec_amd import _shim as S
    k, ni, nsel, N = 4, 50, 3, 5
    rs = np.random.RandomState(1)
    u, v = rs.normal(0, 1, (k, ni)), rs.normal(0, 1, (k, nsel))
    users = np.array([0, 1, 2], dtype=np.int32)
    off = np.array([0, 3, 3, 5], dtype=np.int64)
    rated_sorted = np.array([4, 9, 30, 1, 2], dtype=np.int32)
    rated_shuffled = np.array([30, 4, 9, 2, 1], dtype=np.int32)

    def call(kind=4, n_cand=ni, n=N, rated=rated_sorted, offs=off):
        items = np.full((nsel, max(n, 1)), -7, dtype=np.int32)
        scores = np.zeros((nsel, max(n, 1)))
        rc = S.lib().mfx_topn_batch(kind, k, S.f64(u, "u", 2), S.f64(v, "v", 2), None, None, 0.0, ni,
                                    S.i32(users, "users", 1), nsel, n_cand, n, offs.ctypes.data_as(S._lp),
                                    S.i32(rated, "rated", 1), S.i32(items, "items", 2), S.f64(scores, "scores", 2))
        return rc, items, scores

    rc, a_items, a_scores = call()
    assert rc == 0
    st0 = (C.c_int64 * 2)(); S.check(S.lib().mfx_topn_stats(st0))
    rc, b_items, b_scores = call(rated=rated_shuffled)
    st1 = (C.c_int64 * 2)(); S.check(S.lib().mfx_topn_stats(st1))
    assert rc == 0 and np.array_equal(a_items, b_items) and np.array_equal(a_scores, b_scores)
    assert (st1[0], st1[1]) == (st0[0], st0[1])                       # not the fused kernel's business
    for j, banned in enumerate(({4, 9, 30, 0}, {1}, {1, 2})):         # rated items and the item named like the user
        assert not (set(a_items[j].tolist()) & banned)
        full = u.T @ v[:, j]
        want = [i for i in np.argsort(-full, kind="stable") if i not in banned and full[i] != 0.0][:N]
        assert a_items[j].tolist() == want
    for bad in (dict(kind=5), dict(kind=1), dict(n_cand=ni + 1), dict(n_cand=0), dict(n=0),
                dict(offs=np.array([1, 3, 3, 5], dtype=np.int64)), dict(offs=np.array([0, 3, 2, 5], dtype=np.int64))):
        rc, _, _ = call(**bad)
        assert rc != 0 and "mfx_topn_batch" in S.last_error()


def test_reference_example_flow_runs_end_to_end():
    """examples/example1b_synthetic_wrmf.py = the reference's own example script (mfrec/examples/
    example1b_movielens_100k_wrmf.py) with synthetic MovieLens-100K-shaped data: same calls, same order."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "example1b_synthetic_wrmf.py")],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Precision @ 5" in out.stdout and "F-Measure" in out.stdout and "Epoch : 29" in out.stdout
    lines = [l for l in out.stdout.splitlines() if l.startswith("(")]
    assert len(lines) == 2 and lines[0] == lines[1]                  # per-user loop == batched evaluation


def test_serving_plan_answers_like_the_batch_call():
    """metrics.TopNPlan (mfx_topn_plan_*): model uploaded once, users queried by id, several queries -- every row equal
    to find_recommended_items_batch's (items and score bits), folded-in user and sort path (N > 16) included."""
    import ctypes as C
    from mfrec_amd import _shim as S
    from mfrec_amd.recommendation import KMFRecommender, metrics
    nu, ni, k = 200, 700, 10
    R = _fill(KMFRecommender(nu, ni, {"nbr_epochs": 3, "nbr_features": k}), nu, ni, 6000, 31)
    np.random.seed(6)
    R.train(kernel="train_linear_kernel")
    R.add_user("late", np.array([4, 5, 6], dtype=np.int32), np.array([5.0, 2.0, 3.0]))
    R.neighborhood = ni
    with metrics.TopNPlan(R, "predict_linear") as plan:
        for users, N in (([0, 5, 199, 200, 7, 7], 6), (list(range(0, 201, 3)), 16), ([200, 1], 20), ([], 3)):
            a_i, a_s = plan.query(users, N)
            b_i, b_s = R.find_recommended_items_batch(users, N, "predict_linear")
            assert np.array_equal(a_i, b_i) and np.array_equal(a_s, b_s)
        a_i, _ = plan.query([3, 4], 5, n_candidates=100)
        R.neighborhood = 100
        b_i, _ = R.find_recommended_items_batch([3, 4], 5, "predict_linear")
        assert np.array_equal(a_i, b_i)
        with pytest.raises(IndexError):
            plan.query([201], 3)
    assert S.lib().mfx_topn_plan_query(None, None, 0, 1, 1, None, None) != 0


def test_matrix_core_top_n_with_long_rated_lists_and_every_predictor(monkeypatch):
    """k_topn_mfma keeps a user's rated list in LDS up to 176 items and searches global memory beyond that; its scores are
    approximate and only select -- the rows returned carry the bits of the sort path and of the single-user query for every
    predictor, also when the best-scored items are exactly the rated (excluded) ones."""
    from mfrec_amd.recommendation import KMFRecommender
    nu, ni, k = 40, 2500, 20
    rs = np.random.RandomState(3)
    R = KMFRecommender(nu, ni, {"nbr_features": k})
    users_r = np.concatenate([rs.randint(0, nu, 3000), np.full(400, 7), np.full(177, 8), np.full(176, 9)])
    items_r = np.concatenate([rs.randint(0, ni, 3000), rs.choice(ni, 400, replace=False), rs.choice(ni, 177, replace=False),
                              rs.choice(ni, 176, replace=False)])
    R.set_ratings(np.stack([users_r, items_r], 1), rs.randint(1, 6, users_r.shape[0]).astype(float))
    R.svd_u, R.svd_v = rs.normal(0, 0.3, (k, ni)), rs.normal(0, 0.3, (k, nu))
    R.items_bias, R.users_bias, R.overall_bias = rs.normal(0, 0.1, ni), rs.normal(0, 0.1, nu), 3.0
    for it in np.unique(items_r[users_r == 7])[:60]:          # user 7's best items by far are rated ones: all must be skipped
        R.items_bias[it] += 30.0
    R.neighborhood = ni
    users = list(range(nu))
    for pred in ("predict_linear", "predict_logistic"):
        fi, fs = R.find_recommended_items_batch(users, 12, pred)
        monkeypatch.setenv("MFX_TOPN_FUSED", "0")
        si, ss = R.find_recommended_items_batch(users, 12, pred)
        monkeypatch.delenv("MFX_TOPN_FUSED")
        assert np.array_equal(fi, si) and np.array_equal(fs, ss)
        rated7 = set(int(x) for x in items_r[users_r == 7])
        assert not (set(int(x) for x in fi[7]) & rated7)
        for user in (7, 8, 9, 0):
            wi, ws = R.find_recommended_items(user_index=user, nbr_recommendations=12, predictor=pred)
            assert list(fi[user]) == wi and np.array_equal(fs[user], np.array(ws))


@pytest.mark.parametrize("shape", ["normal", "ascending", "descending", "ties", "around_zero", "huge_biases", "saturated"])
def test_matrix_core_top_n_user_bars(monkeypatch, shape):
    """k_topn_mfma raises every holder's reject bar to (the N-th largest of the user's 64 holders' best scores - a margin) after
    4, 8, 16, ... blocks per wave.  One workgroup walks a 20 000-item catalogue here (157 blocks per wave: six raises), on score
    distributions chosen against the bar: the best items LAST (the bar never binds), FIRST (it binds at once), every score
    equal (ties go by item id), scores straddling zero (an exact 0.0 is never listed), item biases far larger than the margin's
    magnitude estimate (flagged and redone, never wrong), a saturated logistic predictor (ties at the cap).  The rows must be
    the sort path's bit for bit, and outside the adversarial shapes the fused path must prove nearly every user itself."""
    import ctypes as C
    from mfrec_amd import _shim as S
    from mfrec_amd.recommendation import KMFRecommender
    nu, ni, k = 48, 20000, 12
    rs = np.random.RandomState(11)
    R = KMFRecommender(nu, ni, {"nbr_features": k})
    R.set_ratings(np.stack([rs.randint(0, nu, 4000), rs.randint(0, ni, 4000)], 1), rs.randint(1, 6, 4000).astype(float))
    R.svd_u, R.svd_v = rs.normal(0, 0.3, (k, ni)), rs.normal(0, 0.3, (k, nu))
    R.items_bias, R.users_bias, R.overall_bias = rs.normal(0, 0.1, ni), rs.normal(0, 0.1, nu), 3.0
    pred = "predict_linear"
    ramp = np.arange(ni, dtype=np.float64) / ni
    if shape == "ascending":
        R.items_bias = R.items_bias + 4.0 * ramp
    elif shape == "descending":
        R.items_bias = R.items_bias - 4.0 * ramp
    elif shape == "ties":
        R.svd_u[:] = 0.0; R.items_bias[:] = 0.25
        R.items_bias[rs.choice(ni, 300, replace=False)] = 0.5          # 300 items tie at the top, the rest one step below
    elif shape == "around_zero":
        R.svd_u *= 1e-3; R.items_bias = rs.normal(0, 1e-3, ni); R.users_bias[:] = 0.0; R.overall_bias = 0.0
        R.svd_u[:, ::7] = 0.0; R.items_bias[::7] = 0.0                  # exact zeros: never listed
    elif shape == "huge_biases":
        big = rs.choice(ni, 200, replace=False)
        R.items_bias[big] = rs.choice([-1.0, 1.0], 200) * 1e9
    elif shape == "saturated":
        R.svd_u *= 40.0; pred = "predict_logistic"
    R.neighborhood = ni
    users = list(range(nu))
    monkeypatch.setenv("MFX_TOPN_SPLITS", "1")

    def stats():
        a = (C.c_int64 * 2)()
        S.check(S.lib().mfx_topn_stats(a))
        return a[0], a[1]

    for N in (1, 10, 16):
        f0, r0 = stats()
        fi, fs = R.find_recommended_items_batch(users, N, pred)
        f1, r1 = stats()
        monkeypatch.setenv("MFX_TOPN_FUSED", "0")
        si, ss = R.find_recommended_items_batch(users, N, pred)
        monkeypatch.delenv("MFX_TOPN_FUSED")
        assert np.array_equal(fi, si) and np.array_equal(fs, ss), (shape, N)
        assert (f1 - f0) + (r1 - r0) == nu
        if shape in ("normal", "ascending", "descending"):
            assert f1 - f0 >= nu - 4, (shape, N, f1 - f0)
    for user in (0, 17, 47):
        wi, ws = R.find_recommended_items(user_index=user, nbr_recommendations=16, predictor=pred)
        got = [int(x) for x in fi[user] if x >= 0]
        assert got == wi and np.array_equal(fs[user, :len(ws)], np.array(ws))


def test_matrix_core_top_n_odd_catalogue_and_the_pad_behind_u(monkeypatch):
    """k_topn_mfma loads two items per lane (16 bytes) through a bounds-checked buffer: an ODD catalogue's last item shares its load
    with 8 pad bytes behind `u`, and the padded feature steps (k not a multiple of 16) address those bytes as item 0 of feature row k
    -- times an A operand of 0.  Left-over NaNs there made item 0 silently unlisted (caught by the golden test of round 5); the pad
    is zeroed now.  Here: a model full of NaN is scored and freed first, so that the next (slightly smaller) model's pad lies where
    NaNs were; its best items are the FIRST and the LAST of the odd catalogue."""
    from mfrec_amd.recommendation import KMFRecommender
    nu, k = 40, 6
    rs = np.random.RandomState(21)
    for ni_nan, ni in ((1502, 1501), (1501, 1499), (3001, 2999)):
        P = KMFRecommender(nu, ni_nan, {"nbr_features": k})
        P.set_ratings(np.stack([rs.randint(0, nu, 200), rs.randint(0, ni_nan, 200)], 1), np.ones(200))
        P.svd_u, P.svd_v = np.full((k, ni_nan), np.nan), rs.normal(0, 0.3, (k, nu))
        P.items_bias, P.users_bias, P.overall_bias = np.zeros(ni_nan), np.zeros(nu), 3.0
        P.neighborhood = ni_nan
        pi, _ = P.find_recommended_items_batch(list(range(nu)), 5, "predict_linear")
        assert np.all(pi == -1)                                            # NaN scores: nothing is ever listed
        R = KMFRecommender(nu, ni, {"nbr_features": k})
        R.set_ratings(np.stack([rs.randint(0, nu, 900), rs.randint(1, ni - 1, 900)], 1), rs.randint(1, 6, 900).astype(float))
        R.svd_u, R.svd_v = rs.normal(0, 0.3, (k, ni)), rs.normal(0, 0.3, (k, nu))
        R.items_bias, R.users_bias, R.overall_bias = rs.normal(0, 0.1, ni), rs.normal(0, 0.1, nu), 3.0
        R.items_bias[0] += 40.0; R.items_bias[ni - 1] += 50.0
        R.neighborhood = ni
        users = list(range(nu))
        for splits in ("1", None):
            if splits: monkeypatch.setenv("MFX_TOPN_SPLITS", splits)
            fi, fs = R.find_recommended_items_batch(users, 4, "predict_linear")
            if splits: monkeypatch.delenv("MFX_TOPN_SPLITS")
            monkeypatch.setenv("MFX_TOPN_FUSED", "0")
            si, ss = R.find_recommended_items_batch(users, 4, "predict_linear")
            monkeypatch.delenv("MFX_TOPN_FUSED")
            assert np.array_equal(fi, si) and np.array_equal(fs, ss), (ni, splits)
            assert np.all(fi[:, 0] == ni - 1) and np.all(fi[1:, 1] == 0)        # (item 0 is user 0's own index: mf.py skips it)


def test_matrix_core_top_n_partial_last_round_with_split_candidates(monkeypatch):
    """More user groups than the chip has workgroup slots (2 per CU): whole rounds of workgroups go first, the partial last round's groups
    get their candidates split over the idle slots and merged (100 000 users: 3 125 groups on 512 slots used to cost 7 rounds for 6.1).
    The rows must be those of the single launch and of the sort path, the padded last group included."""
    from mfrec_amd.recommendation import KMFRecommender
    nu, ni, k = 512 * 32 + 41 * 32 + 5, 1500, 6
    rs = np.random.RandomState(31)
    R = KMFRecommender(nu, ni, {"nbr_features": k})
    R.set_ratings(np.stack([rs.randint(0, nu, 60000), rs.randint(0, ni, 60000)], 1), rs.randint(1, 6, 60000).astype(float))
    R.svd_u, R.svd_v = rs.normal(0, 0.3, (k, ni)), rs.normal(0, 0.3, (k, nu))
    R.items_bias, R.users_bias, R.overall_bias = rs.normal(0, 0.1, ni), rs.normal(0, 0.1, nu), 3.0
    R.neighborhood = ni
    users = list(range(nu))
    fi, fs = R.find_recommended_items_batch(users, 5, "predict_linear")
    monkeypatch.setenv("MFX_TOPN_TAIL", "0")
    oi, os_ = R.find_recommended_items_batch(users, 5, "predict_linear")
    monkeypatch.delenv("MFX_TOPN_TAIL")
    assert np.array_equal(fi, oi) and np.array_equal(fs, os_)
    monkeypatch.setenv("MFX_TOPN_FUSED", "0")
    tail = list(range(512 * 32 - 40, nu))                                   # the sort path on the users around and in the tail
    si, ss = R.find_recommended_items_batch(tail, 5, "predict_linear")
    monkeypatch.delenv("MFX_TOPN_FUSED")
    assert np.array_equal(fi[tail], si) and np.array_equal(fs[tail], ss)
    for user in (0, 512 * 32 - 1, 512 * 32, nu - 1):
        wi, ws = R.find_recommended_items(user_index=user, nbr_recommendations=5, predictor="predict_linear")
        assert [int(x) for x in fi[user] if x >= 0] == wi and np.array_equal(fs[user, :len(ws)], np.array(ws))
