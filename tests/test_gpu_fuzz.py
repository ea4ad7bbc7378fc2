"""GPU: randomized shapes of every SGD family against the oracle (bit-exact where the family
is exact).  Covers ragged sizes: k not a multiple of the lane group, batches that do not fill
a wave, single-rating inputs, users/items that never occur, all gate combinations."""
import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
# MFX_FUZZ_SEED=n shifts every seed of this file: a soak run over other random shapes (default 0)
SEED = int(os.environ.get("MFX_FUZZ_SEED", "0"))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    yield


def _case(rs):
    k = int(rs.choice([1, 2, 5, 8, 9, 16, 17, 31, 33, 40, 63, 64, 65, 70, 100, 129, 200, 257]))
    nu, ni = int(rs.randint(1, 400)), int(rs.randint(1, 300))
    nnz = int(rs.choice([1, 2, 7, 15, 16, 17, 100, 1000, 5000]))
    idx = np.stack([rs.randint(0, nu, nnz), rs.randint(0, ni, nnz)], 1).astype(np.int32)
    r = rs.randint(1, 6, nnz).astype(np.float64)
    return k, nu, ni, idx, r


def test_full_k_random_shapes():
    from mfrec_amd.lib import kmf_train as K
    from oracle import oracle as O
    rs = np.random.RandomState(2024 + SEED)
    for it in range(40):
        k, nu, ni, idx, r = _case(rs)
        u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
        gates = [(1, 1), (1, 0), (0, 1), (0, 0)][it % 4]
        mode = it % 2
        fn, ofn = ((K.train_linear_kernel, O.train_linear_kernel), (K.train_logistic_kernel, O.train_logistic_kernel))[mode]
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        fn(2, k, 0.1, 0.02, 0, 0, 0.1, 0.05, 0.007, 0.0, u, v, idx, r, ib, ub, gates[0], gates[1])
        u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        ofn(2, k, 0.1, 0.02, 0, 0, 0.1, 0.05, 0.007, 0.0, u2, v2, idx, r, ib2, ub2, gates[0], gates[1])
        msg = "case %d: k=%d nu=%d ni=%d nnz=%d gates=%s mode=%d" % (it, k, nu, ni, len(r), gates, mode)
        if mode == 0:
            assert np.array_equal(u, u2) and np.array_equal(v, v2), msg
            assert np.array_equal(ib, ib2) and np.array_equal(ub, ub2), msg
        else:
            np.testing.assert_allclose(u, u2, rtol=1e-11, atol=1e-14, err_msg=msg)
            np.testing.assert_allclose(v, v2, rtol=1e-11, atol=1e-14, err_msg=msg)
            np.testing.assert_allclose(ub, ub2, rtol=1e-11, atol=1e-14, err_msg=msg)


def test_funk_and_learned_bias_random_shapes():
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    rs = np.random.RandomState(7 + SEED)
    for it in range(16):
        k, nu, ni, idx, r = _case(rs)
        k = min(k, 12)
        init = float(rs.choice([0.1, 0.5]))
        fu, fv = np.zeros((k, ni)) + init, np.zeros((k, nu)) + init
        fu2, fv2 = fu.copy(), fv.copy()
        ib, ub = rs.normal(0, 0.2, ni), rs.normal(0, 0.2, nu)
        if it % 2 == 0:
            G.estimator_loop_with_bias_dev(2, 2, 1e9, k, init, 0.01, 0, 0, 0.05, 3.0, fu, fv, idx, r, ib, ub, nu, ni, it % 3 != 0, it % 3 != 1)
            O.estimator_loop_with_bias_dev(2, 2, 1e9, k, init, 0.01, 0, 0, 0.05, 3.0, fu2, fv2, idx, r, ib, ub, nu, ni, it % 3 != 0, it % 3 != 1)
        else:
            ib1, ub1, ib2, ub2 = ib.copy(), ub.copy(), ib.copy(), ub.copy()
            G.estimator_loop_with_learned_bias(2, 2, 1e9, k, init, 0.01, 0.008, 0.006, 0.05, 0.01, 3.0, fu, fv, idx, r, ib1, ub1, nu, ni)
            O.estimator_loop_with_learned_bias(2, 2, 1e9, k, init, 0.01, 0.008, 0.006, 0.05, 0.01, 3.0, fu2, fv2, idx, r, ib2, ub2, nu, ni)
            assert np.array_equal(ib1, ib2) and np.array_equal(ub1, ub2)
        assert np.array_equal(fu, fu2) and np.array_equal(fv, fv2), "case %d k=%d nnz=%d" % (it, k, len(r))


def test_every_even_k_through_the_level_kernel_and_the_tail():
    """16-byte-lane kernels (VEC): every even k up to 130 plus the wide ones, with levels large enough
    for k1_level (first levels ~1500 ratings) and small enough for k1_tail (the last ones), both
    update-gate settings, linear (bit-exact) -- guards the lane/feature-pair mapping of every KP/FPL."""
    from mfrec_amd.lib import kmf_train as K
    from oracle import oracle as O
    rs = np.random.RandomState(99 + SEED)
    nu, ni, nnz = 2000, 1500, 4000
    idx = np.stack([rs.randint(0, nu, nnz), rs.randint(0, ni, nnz)], 1).astype(np.int32)
    r = rs.randint(1, 6, nnz).astype(np.float64)
    for k in list(range(2, 132, 2)) + [200, 256, 258, 384, 512]:
        u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
        gates = (1, 1) if k % 4 else (1, 0)
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        K.train_linear_kernel(2, k, 0.1, 0.02, 0, 0, 0.1, 0.05, 0.007, 0.0, u, v, idx, r, ib, ub, *gates)
        u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        O.train_linear_kernel(2, k, 0.1, 0.02, 0, 0, 0.1, 0.05, 0.007, 0.0, u2, v2, idx, r, ib2, ub2, *gates)
        assert np.array_equal(u, u2) and np.array_equal(v, v2), "k=%d gates=%s" % (k, gates)
        assert np.array_equal(ib, ib2) and np.array_equal(ub, ub2), "k=%d" % k
