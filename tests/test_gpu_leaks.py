"""GPU: device memory comes back.  Every object of the library (plans with their Funk / XCD state, DSGD objects,
the one-shot calls with their scratch, ALS, the implicit-feedback loop, the device schedule builder) is created,
used and destroyed repeatedly; hipMemGetInfo's free figure must return to where it was (the runtime pools
allocations, so the check is "no growth from the 2nd round to the last", not an exact byte count)."""
import ctypes as C

import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _lib():
    from mfrec_amd import _shim as S, build as B
    B.build(verbose=False)
    S.init(0)
    yield


def _free():
    from mfrec_amd import _shim as S
    S.sync()
    f = C.c_int64(0)
    S.check(S.lib().mfx_device_mem_info(C.byref(f), None))
    return f.value


def _round():
    from mfrec_amd.plan import SgdPlan
    from mfrec_amd.sharding import DsgdKMF
    from mfrec_amd.lib import kmf_train as K, gd_estimator as G, als_implicit as A
    from mfrec_amd.lib.datasets import balanced_order_gpu, create_bool_sparse_col, create_bool_sparse_row
    import scipy.sparse as sp
    nu, ni, nnz, k = 60000, 12000, 3000000, 16
    idx, r = synth.make_ratings(nu, ni, nnz, seed=21)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    # resident plan: full-k epochs, Funk on the XCD-resident pass (levels of ~10 K), narrow storage
    for storage in ("f64", "f32"):
        with SgdPlan(idx, r, nu, ni, k, storage=storage) as p:
            p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
            p.run_kmf(0, 2, 0.01, 0.1, 0.1, 0.007)
            if storage == "f64":
                p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
    # one-shot drop-ins
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    K.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)
    s = slice(0, 200000)
    uu, vv = np.full((4, ni), 0.1), np.full((4, nu), 0.1)
    G.estimator_loop_without_bias(1, 1, 1e9, 4, 0.1, 0.001, 0.05, uu, vv, np.ascontiguousarray(idx[s]),
                                  np.ascontiguousarray(r[s]), nu, ni)
    balanced_order_gpu(idx, nu, ni)
    # DSGD fake world (4 ranks x 4 blocks on this GPU)
    d = DsgdKMF(idx, r, nu, ni, k, -1, 4, balanced=True)
    d.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    d.train(0, 1, 0.01, 0.1, 0.1, 0.007)
    d.close()
    # batched top-N: fused kernel and sort path
    from mfrec_amd.recommendation import KMFRecommender
    R = KMFRecommender(4, 6, {"nbr_features": k})
    R._nbr_users, R._nbr_items = nu, ni
    R.relationship_matrix = sp.csr_matrix((r[s], (idx[s, 0], idx[s, 1])), shape=(nu, ni)).tolil()
    R.svd_u, R.svd_v, R.items_bias, R.users_bias, R.overall_bias = u0, v0, np.zeros(ni), np.zeros(nu), 3.0
    R.neighborhood = ni
    R.find_recommended_items_batch(np.arange(2000), 10, "predict_linear")
    R.find_recommended_items_batch(np.arange(500), 20, "predict_linear")
    from mfrec_amd.recommendation import metrics
    with metrics.TopNPlan(R, "predict_linear") as plan:
        plan.query(np.arange(3), 10)
        plan.query(np.arange(4000), 10)
    # ALS: tuned kernel and the k > 128 scratch kernel
    M = sp.csr_matrix((np.ones(200000), (idx[s, 0] % 5000, idx[s, 1] % 3000)), shape=(5000, 3000))
    M.sum_duplicates()
    urow, ucol = create_bool_sparse_row(M)
    irow, icol = create_bool_sparse_col(M)
    for ka in (32, 136):
        rs = np.random.RandomState(ka)
        ua, va = rs.normal(0, 0.1, (ka, 3000)), rs.normal(0, 0.1, (ka, 5000))
        A.als_wrmf(1, ka, ua, va, None, None, urow, ucol, irow, icol, 5000, 3000)


def test_device_memory_returns_after_every_kind_of_object():
    _round()                       # first round: lazily created library state (streams, graphs, code objects)
    _round()
    base = _free()
    for _ in range(3):
        _round()
    after = _free()
    assert base - after < 64 << 20, "device memory shrank by %.1f MB over three rounds" % ((base - after) / 1e6)
