"""GPU: batched top-N (mfx_topn_batch, mfx_topn_plan_query) against tests/golden/g12_topn.npz -- the lists the cited
restatement of the reference's ranking rules gives (oracle/oracle.py: find_recommended_items, mf.py:144-193 /
gradient_descent.py:768-800).  The fixture is written from the restatement (oracle/make_golden.py topn_fixture) and, since round 4,
PINNED: oracle/pin_topn.py executes the reference's own definitions of these functions (cut from its files, translated by lib2to3)
on the fixture's inputs and refuses any difference (tests/golden/g12_topn_pin.json; tests/test_oracle_golden.py checks the record).

Items must match exactly (the fixture keeps neighbouring scores either exactly tied or 1e-9 apart); scores to 1e-12
relative (the reference sums the dot with np.dot, whose association is BLAS's; the kernel sums left to right)."""
import ctypes as C

import numpy as np
import pytest

from mfrec_amd import _shim as S

pytestmark = pytest.mark.gpu
KINDS = (0, 1, 2, 3, 4)          # GD, KMF linear, KMF logistic, GD + BellKor baseline, WRMF (oracle.RMSE_*)


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    yield


def _model(g, tag):
    u, v = np.ascontiguousarray(g["in_%s_u" % tag]), np.ascontiguousarray(g["in_%s_v" % tag])
    return (u, v, np.ascontiguousarray(g["in_%s_ib" % tag]), np.ascontiguousarray(g["in_%s_ub" % tag]),
            float(g["p_%s_mu" % tag]), g["in_%s_rated_off" % tag], g["in_%s_rated" % tag], g["in_%s_users" % tag],
            int(g["p_%s_nu" % tag]), int(g["p_%s_ni" % tag]))


def _compare(items, scores, want_items, want_scores, what):
    assert np.array_equal(items, want_items), what
    live = want_items >= 0
    np.testing.assert_allclose(scores[live], want_scores[live], rtol=1e-12, atol=0, err_msg=str(what))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_topn_batch_vs_golden(golden, tag):
    g = golden("g12_topn")
    u, v, ib, ub, mu, roff, rated, users, nu, ni = _model(g, tag)
    k = u.shape[0]
    # the batch call takes the selected users' columns and their rated lists (the folded-in user nu has none)
    v_sel = np.ascontiguousarray(v[:, users])
    ub_sel = np.ascontiguousarray(ub[users])
    lists = [rated[roff[x]:roff[x + 1]] if x < nu else np.zeros(0, np.int32) for x in users]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    flat = np.ascontiguousarray(np.concatenate(lists + [np.zeros(1, np.int32)]), dtype=np.int32)
    for kind in KINDS:
        for ci, (n_cand, N) in enumerate(g["p_%s_cases" % tag]):
            items = np.full((users.shape[0], int(N)), -1, dtype=np.int32)
            scores = np.zeros((users.shape[0], int(N)))
            bias = kind in (1, 2, 3)
            S.check(S.lib().mfx_topn_batch(kind, k, S.f64(u, "u", 2), S.f64(v_sel, "v", 2),
                                           S.f64(ib, "ib", 1) if bias else None, S.f64(ub_sel, "ub", 1) if bias else None,
                                           mu, ni, S.i32(users, "users", 1), users.shape[0], int(n_cand), int(N),
                                           off.ctypes.data_as(S._lp), S.i32(flat, "rated", 1),
                                           S.i32(items, "items", 2), S.f64(scores, "scores", 2)))
            _compare(items, scores, g["out_%s_items_kind%d_case%d" % (tag, kind, ci)],
                     g["out_%s_scores_kind%d_case%d" % (tag, kind, ci)], (tag, kind, ci))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_topn_plan_query_vs_golden(golden, tag):
    g = golden("g12_topn")
    u, v, ib, ub, mu, roff, rated, users, nu, ni = _model(g, tag)
    k = u.shape[0]
    off = np.concatenate([roff, [roff[-1]]]).astype(np.int64)          # + the folded-in user's empty list
    flat = np.ascontiguousarray(np.concatenate([rated, np.zeros(1, np.int32)]), dtype=np.int32)
    for kind in KINDS:
        bias = kind in (1, 2, 3)
        h = S.lib().mfx_topn_plan_create(kind, k, S.f64(u, "u", 2), S.f64(v, "v", 2), S.f64(ib, "ib", 1) if bias else None,
                                         S.f64(ub, "ub", 1) if bias else None, mu, nu + 1, ni,
                                         off.ctypes.data_as(S._lp), S.i32(flat, "rated", 1))
        assert h, S.last_error()
        try:
            for ci, (n_cand, N) in enumerate(g["p_%s_cases" % tag]):
                items = np.full((users.shape[0], int(N)), -1, dtype=np.int32)
                scores = np.zeros((users.shape[0], int(N)))
                S.check(S.lib().mfx_topn_plan_query(h, S.i32(users, "users", 1), users.shape[0], int(n_cand), int(N),
                                                    S.i32(items, "items", 2), S.f64(scores, "scores", 2)))
                _compare(items, scores, g["out_%s_items_kind%d_case%d" % (tag, kind, ci)],
                         g["out_%s_scores_kind%d_case%d" % (tag, kind, ci)], (tag, kind, ci))
        finally:
            S.lib().mfx_topn_plan_destroy(h)


def test_recommender_level_precision_recall_vs_golden(golden):
    """metrics.precision_recall over the recommender (per-user loop and batched) against the restated metric."""
    import scipy.sparse as sp
    from mfrec_amd.recommendation import GDRecommender, metrics
    g = golden("g12_topn")
    u, v, ib, ub, mu, roff, rated, users, nu, ni = _model(g, "a")
    R = GDRecommender(nu, ni, {"nbr_features": u.shape[0]})
    rows = np.repeat(np.arange(nu), np.diff(roff))
    R.set_ratings(np.ascontiguousarray(np.stack([rows, rated], axis=1), dtype=np.int32), np.full(rated.shape[0], 3.0))
    R.svd_u, R.svd_v = u.copy(), np.ascontiguousarray(v[:, :nu])
    R.neighborhood = ni
    want = g["out_a_prf"]
    for batched in (False, True):
        p, r, f = metrics.precision_recall(R, g["in_a_test"], 5, "predict_rating", batched=batched)
        np.testing.assert_allclose([p, r, f], want, rtol=1e-12)


def test_minus_infinity_score_is_listed_like_the_reference_does():
    """A score of -inf is non-zero and not NaN: the reference lists it (last).  The fused kernel cannot tell it from its
    empty-slot sentinel, so it flags the user for the sort path -- either way the list is the restatement's."""
    from oracle import oracle as O
    rs = np.random.RandomState(4)
    k, nu, ni = 8, 5, 40
    u, v = rs.normal(0, 0.5, (k, ni)), np.abs(rs.normal(0, 0.5, (k, nu))) + 0.1
    u[:, 7] = -np.inf
    users = np.arange(nu, dtype=np.int32)
    off = np.zeros(nu + 1, dtype=np.int64)
    flat = np.zeros(1, dtype=np.int32)
    for N in (5, 45):
        items = np.full((nu, N), -1, dtype=np.int32)
        scores = np.zeros((nu, N))
        S.check(S.lib().mfx_topn_batch(4, k, S.f64(u, "u", 2), S.f64(v, "v", 2), None, None, 0.0, ni,
                                       S.i32(users, "users", 1), nu, ni, N, off.ctypes.data_as(S._lp),
                                       S.i32(flat, "rated", 1), S.i32(items, "items", 2), S.f64(scores, "scores", 2)))
        for j in range(nu):
            it, sc = O.find_recommended_items(O.RMSE_WRMF, u, v, j, (), ni, N)
            assert [int(x) for x in items[j] if x >= 0] == it, (N, j)
            if N == 45:
                assert it[-1] == 7 and sc[-1] == -np.inf


def test_small_queries_in_a_loop_stay_golden(golden):
    """Round 5: a variant of k_topn_mfma (asm loads with hand-written wait counts) returned one wrong row in ~170 of these queries and
    passed every single-shot test; the product's loads are the compiler's again (profiles/r05_topn_asm_loads.txt).  The fixture's fused
    cases, 40 times over, every list compared (tools/stress_topn.py is the long form)."""
    g = golden("g12_topn")
    calls = []
    for tag in "ab":
        u, v, ib, ub, mu, roff, rated, users, nu, ni = _model(g, tag)
        v_sel, ub_sel = np.ascontiguousarray(v[:, users]), np.ascontiguousarray(ub[users])
        lists = [rated[roff[x]:roff[x + 1]] if x < nu else np.zeros(0, np.int32) for x in users]
        off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
        flat = np.ascontiguousarray(np.concatenate(lists + [np.zeros(1, np.int32)]), dtype=np.int32)
        for kind in KINDS:
            for ci in (0, 2):                                                   # (case 1 is N = 70: the sort path)
                n_cand, N = (int(x) for x in g["p_%s_cases" % tag][ci])
                calls.append((tag, kind, ci, u, v_sel, ib, ub_sel, mu, ni, users, n_cand, N, off, flat))
    for rep in range(40):
        for (tag, kind, ci, u, v_sel, ib, ub_sel, mu, ni, users, n_cand, N, off, flat) in calls:
            items = np.full((users.shape[0], N), -1, dtype=np.int32)
            scores = np.zeros((users.shape[0], N))
            bias = kind in (1, 2, 3)
            S.check(S.lib().mfx_topn_batch(kind, u.shape[0], S.f64(u, "u", 2), S.f64(v_sel, "v", 2),
                                           S.f64(ib, "ib", 1) if bias else None, S.f64(ub_sel, "ub", 1) if bias else None,
                                           mu, ni, S.i32(users, "users", 1), users.shape[0], n_cand, N,
                                           off.ctypes.data_as(S._lp), S.i32(flat, "rated", 1),
                                           S.i32(items, "items", 2), S.f64(scores, "scores", 2)))
            assert np.array_equal(items, g["out_%s_items_kind%d_case%d" % (tag, kind, ci)]), (rep, tag, kind, ci)
