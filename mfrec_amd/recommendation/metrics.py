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


def predict_batch(recommender, ratings_index, predictor="predict", ratings=None):
    """-> (predictions f64[n], rmse or None)."""
    kind = _kind(recommender, predictor)
    idx = np.ascontiguousarray(ratings_index, dtype=np.int32)
    n = idx.shape[0]
    r = np.zeros(n) if ratings is None else np.ascontiguousarray(ratings, dtype=np.float64)
    pred = np.zeros(n)
    out = C.c_double(0.0)
    u, v = recommender.svd_u, recommender.svd_v
    ib = recommender.items_bias if kind in (1, 2, 3) else None
    ub = recommender.users_bias if kind in (1, 2, 3) else None
    mu = float(recommender.overall_bias or 0.0)
    S.check(S.lib().mfx_predict_rmse(kind, u.shape[0], S.f64(u, "svd_u", 2), S.f64(v, "svd_v", 2),
                                     S.i32(idx, "ratings_index", 2), S.f64(r, "ratings", 1),
                                     S.f64(ib, "items_bias", 1, allow_none=True),
                                     S.f64(ub, "users_bias", 1, allow_none=True), mu, n,
                                     v.shape[1], u.shape[1], S.f64(pred, "pred", 1), C.byref(out)))
    return pred, (None if ratings is None else out.value)


def rmse_on(recommender, ratings_index, ratings, predictor="predict"):
    return predict_batch(recommender, ratings_index, predictor, ratings)[1]


def test_predict_rating(recommender, u_test, nbr_samples=10, verbose=False,
                        predictor="predict_rating"):
    """metrics.py:51-82: u_test rows are (user, item, rating); returns (rmse, errors)."""
    t = np.asarray(u_test)[:nbr_samples]
    idx = t[:, :2].astype(np.int32)
    real = t[:, 2].astype(np.float64)
    pred, rmse = predict_batch(recommender, idx, predictor, real)
    errors = real - pred
    if verbose:
        print("Mean root mean square error (RMSE): " + str(rmse))
    return rmse, errors


test_predict_rating.__test__ = False          # not a pytest test, despite the reference's name
