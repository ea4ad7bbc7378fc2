"""RMSE of a recommender over rating triples -- the parity metric
(reference formula: mfrec/recommendation/metrics.py:58-73), as one batched GPU pass."""
import ctypes as C

import numpy as np

from .. import _shim as S

_KIND = {"predict_rating": 0, "predict_linear": 1, "predict_logistic": 2,
         "predict_rating_with_bias": 3, "predict_wrmf": 4}


def _kind(recommender, predictor):
    if predictor == "predict":
        predictor = {"GDRecommender": "predict_rating", "KMFRecommender": "predict_logistic",
                     "WRMFRecommender": "predict_wrmf"}[type(recommender).__name__]
    return _KIND[predictor]


def predict_batch(recommender, ratings_index, predictor="predict", ratings=None, one_user=None):
    """-> (predictions f64[n], rmse or None).  one_user: every row of ratings_index is about this user --
    only its column of svd_v (and its bias) is sent to the device, as user 0."""
    kind = _kind(recommender, predictor)
    idx = np.ascontiguousarray(ratings_index, dtype=np.int32)
    n = idx.shape[0]
    r = np.zeros(n) if ratings is None else np.ascontiguousarray(ratings, dtype=np.float64)
    pred = np.zeros(n)
    out = C.c_double(0.0)
    u, v = recommender.svd_u, recommender.svd_v
    ib = recommender.items_bias if kind in (1, 2, 3) else None
    ub = recommender.users_bias if kind in (1, 2, 3) else None
    if one_user is not None:
        v = np.ascontiguousarray(v[:, one_user:one_user + 1])
        ub = None if ub is None else np.ascontiguousarray(ub[one_user:one_user + 1])
        idx = idx.copy()
        idx[:, 0] = 0
    mu = float(recommender.overall_bias or 0.0)
    S.check(S.lib().mfx_predict_rmse(kind, u.shape[0], S.f64(u, "svd_u", 2), S.f64(v, "svd_v", 2),
                                     S.i32(idx, "ratings_index", 2), S.f64(r, "ratings", 1),
                                     S.f64(ib, "items_bias", 1, allow_none=True),
                                     S.f64(ub, "users_bias", 1, allow_none=True), mu, n,
                                     v.shape[1], u.shape[1], S.f64(pred, "pred", 1), C.byref(out)))
    return pred, (None if ratings is None else out.value)


def topn_batch(recommender, users, n_candidates, nbr_recommendations, predictor="predict"):
    """Top-N for many users in one device call (mfx_topn_batch): -> (items int32[n, N] with -1 where fewer than N
    items qualify, scores f64[n, N]).  Same candidates, exclusions and order as find_recommended_items."""
    kind = _kind(recommender, predictor)
    users = np.ascontiguousarray(users, dtype=np.int32)
    n = users.shape[0]
    if n and (users.min() < 0 or users.max() >= recommender.svd_v.shape[1]):
        raise IndexError("user outside the model")
    u = recommender.svd_u
    v_sel = np.ascontiguousarray(recommender.svd_v[:, users])
    ib = recommender.items_bias if kind in (1, 2, 3) else None
    ub = np.ascontiguousarray(recommender.users_bias[users]) if kind in (1, 2, 3) else None
    mu = float(recommender.overall_bias or 0.0)
    rel = recommender.relationship_matrix.tocsr()
    rel.eliminate_zeros()
    rel.sort_indices()                                     # ascending lists: the fused kernel searches them
    in_matrix = users < rel.shape[0]                       # folded-in users have no row in the ratings matrix
    counts = np.zeros(n, dtype=np.int64)
    counts[in_matrix] = np.diff(rel.indptr)[users[in_matrix]]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    # users outside the matrix contribute empty lists, so the rows of the others, in order, are the whole array
    rated = np.ascontiguousarray(rel[users[in_matrix]].indices, dtype=np.int32) if in_matrix.any() else np.zeros(0, np.int32)
    if rated.shape[0] == 0:
        rated = np.zeros(1, dtype=np.int32)
    items = np.full((n, nbr_recommendations), -1, dtype=np.int32)
    scores = np.zeros((n, nbr_recommendations))
    S.check(S.lib().mfx_topn_batch(kind, u.shape[0], S.f64(u, "svd_u", 2), S.f64(v_sel, "v_sel", 2),
                                   S.f64(ib, "items_bias", 1, allow_none=True),
                                   S.f64(ub, "users_bias", 1, allow_none=True), mu, u.shape[1],
                                   S.i32(users, "users", 1), n, int(n_candidates), int(nbr_recommendations),
                                   off.ctypes.data_as(S._lp), S.i32(rated, "rated", 1),
                                   S.i32(items, "items", 2), S.f64(scores, "scores", 2)))
    return items, scores


class TopNPlan(object):
    """Serving form of the batched top-N (mfx_topn_plan_*): the recommender's model and every user's rated list are
    uploaded once; query(users, N) then ranks without touching the host copies again.  A snapshot of the model at
    construction time -- after another train() build a new one.  Rows equal find_recommended_items_batch's."""

    def __init__(self, recommender, predictor="predict"):
        kind = _kind(recommender, predictor)
        u, v = recommender.svd_u, recommender.svd_v
        nu, ni = v.shape[1], u.shape[1]
        ib = recommender.items_bias if kind in (1, 2, 3) else None
        ub = recommender.users_bias if kind in (1, 2, 3) else None
        rel = recommender.relationship_matrix.tocsr()
        rel.eliminate_zeros()
        rel.sort_indices()
        counts = np.zeros(nu, dtype=np.int64)                   # folded-in users have no row in the ratings matrix
        m = min(nu, rel.shape[0])
        counts[:m] = np.diff(rel.indptr)[:m]
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        rated = np.ascontiguousarray(rel.indices[:rel.indptr[m]], dtype=np.int32)
        if rated.shape[0] == 0:
            rated = np.zeros(1, dtype=np.int32)
        self.neighborhood = int(min(recommender.neighborhood, recommender.nbr_items))
        self.nu = nu
        self._h = S.lib().mfx_topn_plan_create(kind, u.shape[0], S.f64(u, "svd_u", 2), S.f64(v, "svd_v", 2),
                                               S.f64(ib, "items_bias", 1, allow_none=True),
                                               S.f64(ub, "users_bias", 1, allow_none=True),
                                               float(recommender.overall_bias or 0.0), nu, ni,
                                               off.ctypes.data_as(S._lp), S.i32(rated, "rated", 1))
        if not self._h:
            raise S.MfxError(S.last_error())

    def query(self, users, nbr_recommendations=5, n_candidates=None):
        users = np.ascontiguousarray(users, dtype=np.int32)
        if users.shape[0] and (users.min() < 0 or users.max() >= self.nu):
            raise IndexError("user outside the model")
        n = users.shape[0]
        items = np.full((n, nbr_recommendations), -1, dtype=np.int32)
        scores = np.zeros((n, nbr_recommendations))
        S.check(S.lib().mfx_topn_plan_query(self._h, S.i32(users, "users", 1), n,
                                            int(n_candidates or self.neighborhood), int(nbr_recommendations),
                                            S.i32(items, "items", 2), S.f64(scores, "scores", 2)))
        return items, scores

    def close(self):
        if getattr(self, "_h", None):
            S.lib().mfx_topn_plan_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rmse_on(recommender, ratings_index, ratings, predictor="predict"):
    return predict_batch(recommender, ratings_index, predictor, ratings)[1]


def test_predict_rating(recommender, u_test, nbr_samples=10, verbose=False,
                        predictor="predict_rating"):
    """metrics.py:51-82: u_test rows are (user, item, rating); returns (rmse, errors) over the first
    nbr_samples rows, NaN errors left out (metrics.py:69-73).  The predictions are one batched GPU pass."""
    t = np.asarray(u_test)[:nbr_samples]
    idx = t[:, :2].astype(np.int32)
    real = t[:, 2].astype(np.float64)
    pred, _ = predict_batch(recommender, idx, predictor, real)
    all_errors = real - pred
    errors = all_errors[~np.isnan(all_errors)]
    abs_errors = np.abs(errors)
    rmse = np.sqrt(np.power(abs_errors, 2).mean())
    if verbose:
        print("\nNumber of succesful rating: %d/%d" % (len(abs_errors), nbr_samples))
        print("Mean abs. error: " + str(abs_errors.mean()))
        print("Variance of the error: " + str(abs_errors.var()))
        print("Mean root mean square error (RMSE): " + str(rmse))
        print("NMAE: " + str(abs_errors.mean() / 1.6))
        print("MAE: " + str(abs_errors.mean()) + "\n\n")
    return rmse, errors


def precision_recall(recommender, u_test, nbr_recommendations=5, predictor="predict", verbose=False, batched=False):
    """metrics.py:85-130: precision / recall / F-measure of find_recommended_items against the test pairs
    (u_test rows: user, item, ...), user by user; a user whose lookup raises KeyError recommends nothing and
    is not counted."""
    precision = recall = 0.0
    test_sample_dict = {}
    for rating in u_test:
        test_sample_dict.setdefault(int(rating[0]), []).append(int(rating[1]))
    users_count = 0
    if batched:
        # extension: every user's ranking in one device call (same rankings; the per-user RNG draws of
        # find_recommended_items, which do not influence its result, are not made)
        keys = list(test_sample_dict.keys())
        top, _ = recommender.find_recommended_items_batch(keys, nbr_recommendations, predictor)
        ranked = {k: set(int(i) for i in row if i >= 0) for k, row in zip(keys, top)}
    for user_index in test_sample_dict.keys():
        try:
            if batched:
                recommended_set = ranked[user_index]
            else:
                recommended_set = set(recommender.find_recommended_items(
                    user_index=user_index, nbr_recommendations=nbr_recommendations, output_label=False,
                    predictor=predictor)[0])
            users_count += 1
        except KeyError:
            recommended_set = set()
        already_rated = set(test_sample_dict[user_index])
        intersection = float(len(recommended_set.intersection(already_rated)))
        precision += intersection / nbr_recommendations
        recall += intersection / len(already_rated)
    precision = precision / users_count
    recall = recall / users_count
    f_measure = 2 * (precision * recall) / (precision + recall)
    if verbose:
        print("Precision @ " + str(nbr_recommendations) + " : " + str(precision))
        print("Recall @ " + str(nbr_recommendations) + " : " + str(recall))
        print("F-Measure : " + str(f_measure))
    return precision, recall, f_measure


test_predict_rating.__test__ = False          # not a pytest test, despite the reference's name
