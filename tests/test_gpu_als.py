"""GPU: ALS / WRMF (als_wrmf) through the C-ABI against the golden vectors and the oracle.
Tolerance: 1e-9 relative on the factors (the reference inverts with LAPACK, the device
solves by Cholesky; see oracle/make_golden.py for why the constant init is looser)."""
import numpy as np
import pytest
import scipy.sparse as sp

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    yield


def test_als_golden(golden):
    from mfrec_amd.lib.als_implicit import als_wrmf
    g = golden("g6_als")
    k, nu, ni = int(g["p_k"]), int(g["p_nu"]), int(g["p_ni"])
    m = np.zeros((k, k))
    for tag, u0, v0 in (("rand", g["in_u0_rand"], g["in_v0_rand"]),
                        ("const", synth.init_const(k, ni), synth.init_const(k, nu))):
        for E in (1, 2):
            u, v = u0.copy(), v0.copy()
            als_wrmf(E, k, u, v, m, m, g["in_urow"], g["in_ucol"], g["in_irow"], g["in_icol"],
                     nu, ni, int(g["p_c_pos"]), float(g["p_lambda"]))
            tol = 1e-9 if (tag == "rand" or E == 1) else 1e-4
            np.testing.assert_allclose(u, g["out_u_%s_e%d" % (tag, E)], rtol=tol, atol=tol * 1e-2)
            np.testing.assert_allclose(v, g["out_v_%s_e%d" % (tag, E)], rtol=tol, atol=tol * 1e-2)
            assert np.all(v[:, 7] == 0.0)                  # user without ratings -> 0
            assert np.array_equal(v[:, -1], v0[:, -1])     # trailing unrated ids untouched


def _problem(nu, ni, nnz, seed):
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    rs = np.random.RandomState(seed)
    M = sp.coo_matrix((np.ones(nnz), (rs.randint(0, nu, nnz), rs.randint(0, ni, nnz))),
                      shape=(nu, ni)).tocsr()
    M.data[:] = 1.0
    return create_bool_sparse_row(M) + create_bool_sparse_col(M)


# every 16-column tile count NT = 1..8 (k_als_rows2: NT 1, 2, 4; k_als_rows3: NT 3, 5..8), ragged k
@pytest.mark.parametrize("k", [1, 5, 16, 20, 33, 40, 48, 64, 65, 72, 96, 100, 112, 127, 128])
def test_als_vs_oracle(k):
    from mfrec_amd.lib.als_implicit import als_wrmf
    from oracle import oracle as O
    nu, ni = 300, 200
    urow, ucol, irow, icol = _problem(nu, ni, 6000, 10 + k)
    rs = np.random.RandomState(k)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v = u0.copy(), v0.copy()
    als_wrmf(2, k, u, v, None, None, urow, ucol, irow, icol, nu, ni, 1, 0.015)
    u2, v2 = u0.copy(), v0.copy()
    O.als_wrmf(2, k, u2, v2, None, None, urow, ucol, irow, icol, nu, ni, 1, 0.015)
    np.testing.assert_allclose(u, u2, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(v, v2, rtol=1e-8, atol=1e-11)
    # predictions (wrmf.py:67-69) agree far inside the 1e-5 RMSE budget
    assert np.abs(u.T @ v - u2.T @ v2).max() < 1e-9


# k > 128 (als_implicit.pyx:211-225 takes any dim): the global-scratch kernel k_als_rows_big, NT = 9..32,
# ragged k, several accumulation rounds (k >= 176 needs more than one round of 64 tiles)
@pytest.mark.parametrize("k", [129, 144, 176, 200, 256, 300, 512])
def test_als_big_k_vs_oracle(k):
    from mfrec_amd.lib.als_implicit import als_wrmf
    from oracle import oracle as O
    nu, ni = 70, 50
    urow, ucol, irow, icol = _problem(nu, ni, 1200, 10 + k)
    rs = np.random.RandomState(k)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v = u0.copy(), v0.copy()
    als_wrmf(2, k, u, v, None, None, urow, ucol, irow, icol, nu, ni, 1, 0.015)
    u2, v2 = u0.copy(), v0.copy()
    O.als_wrmf(2, k, u2, v2, None, None, urow, ucol, irow, icol, nu, ni, 1, 0.015)
    np.testing.assert_allclose(u, u2, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(v, v2, rtol=1e-7, atol=1e-10)
    assert np.abs(u.T @ v - u2.T @ v2).max() < 1e-8


def test_als_big_k_slices_singular_and_limit(monkeypatch):
    from mfrec_amd.lib.als_implicit import als_wrmf
    nu, ni, k = 90, 60, 160
    urow, ucol, irow, icol = _problem(nu, ni, 1500, 7)
    rs = np.random.RandomState(1)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v = u0.copy(), v0.copy()
    als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni)
    monkeypatch.setenv("MFX_ALS_FAKE_WORLD", "3")                 # the sharded row slices, one GPU
    u2, v2 = u0.copy(), v0.copy()
    als_wrmf(1, k, u2, v2, None, None, urow, ucol, irow, icol, nu, ni)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
    monkeypatch.delenv("MFX_ALS_FAKE_WORLD")
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite"):
        als_wrmf(1, k, np.zeros((k, ni)), np.zeros((k, nu)), None, None, urow, ucol, irow, icol, nu, ni, 1, 0.0)
    with pytest.raises(ValueError, match="1 <= k <= 512"):
        als_wrmf(1, 513, np.zeros((513, ni)), np.zeros((513, nu)), None, None, urow, ucol, irow, icol, nu, ni)


def test_als_c_pos_and_lambda():
    from mfrec_amd.lib.als_implicit import als_wrmf
    from oracle import oracle as O
    nu, ni, k = 120, 90, 16
    urow, ucol, irow, icol = _problem(nu, ni, 2500, 3)
    rs = np.random.RandomState(0)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    for c_pos, lam in ((3, 0.5), (10, 0.015)):
        u, v, u2, v2 = u0.copy(), v0.copy(), u0.copy(), v0.copy()
        als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni, c_pos, lam)
        O.als_wrmf(1, k, u2, v2, None, None, urow, ucol, irow, icol, nu, ni, c_pos, lam)
        np.testing.assert_allclose(u, u2, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(v, v2, rtol=1e-8, atol=1e-11)


def test_als_ragged_spans_and_empty_rows():
    """Spans 0, 1, 31, 32, 33 (the 32-row gather chunk), a 700-long row, unrated tails, with the
    register-tile kernel (k=80) and the LDS kernel (k=64)."""
    from mfrec_amd.lib.als_implicit import als_wrmf
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    from oracle import oracle as O
    nu, ni = 40, 800
    rs = np.random.RandomState(9)
    spans = [0, 1, 31, 32, 33, 64, 65, 700, 0, 2] + list(rs.randint(0, 50, 20)) + [0] * 10
    rows, cols = [], []
    for uidx, sp_ in enumerate(spans):
        c = rs.choice(ni - 100, sp_, replace=False)          # the last 100 items stay unrated
        rows += [uidx] * sp_
        cols += list(c)
    M = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(nu, ni))
    urow, ucol = create_bool_sparse_row(M)
    irow, icol = create_bool_sparse_col(M)
    for k in (64, 80):
        u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
        u, v, u2, v2 = u0.copy(), v0.copy(), u0.copy(), v0.copy()
        als_wrmf(2, k, u, v, None, None, urow, ucol, irow, icol, nu, ni, 1, 0.015)
        O.als_wrmf(2, k, u2, v2, None, None, urow, ucol, irow, icol, nu, ni, 1, 0.015)
        np.testing.assert_allclose(u, u2, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(v, v2, rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("k", [8, 24, 64, 80, 128])
def test_als_singular_system_raises_like_numpy_inv(k):
    """lambda = 0 with all-zero factors: A = 0, np.linalg.inv in the reference raises LinAlgError
    (als_implicit.pyx:296); every row kernel must report it instead of returning NaNs."""
    from mfrec_amd.lib.als_implicit import als_wrmf
    nu, ni = 50, 40
    urow, ucol, irow, icol = _problem(nu, ni, 600, 1)
    u, v = np.zeros((k, ni)), np.zeros((k, nu))
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite"):
        als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni, 1, 0.0)


def test_als_errors():
    from mfrec_amd.lib.als_implicit import als_wrmf
    u, v = np.ones((4, 3)), np.ones((4, 5))
    row = np.array([0, 1], dtype=np.int32)
    with pytest.raises(ValueError, match="outside"):
        als_wrmf(1, 4, u, v, None, None, row, np.array([7], dtype=np.int32), row,
                 np.array([0], dtype=np.int32), 5, 3)
    with pytest.raises(ValueError, match="dtype"):
        als_wrmf(1, 4, u, v, None, None, row.astype(np.int64), np.array([0], dtype=np.int32), row,
                 np.array([0], dtype=np.int32), 5, 3)


def test_als_row_slicing_arithmetic(monkeypatch):
    """MFX_ALS_FAKE_WORLD=W runs the W rank-slices of the sharded scheme one after the other on
    this GPU: the slice bounds must tile the active rows exactly (bit-identical result)."""
    from mfrec_amd.lib.als_implicit import als_wrmf
    nu, ni, k = 301, 197, 24
    urow, ucol, irow, icol = _problem(nu, ni, 5000, 5)
    rs = np.random.RandomState(1)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v = u0.copy(), v0.copy()
    als_wrmf(2, k, u, v, None, None, urow, ucol, irow, icol, nu, ni)
    for w in ("2", "3", "8"):
        monkeypatch.setenv("MFX_ALS_FAKE_WORLD", w)
        u2, v2 = u0.copy(), v0.copy()
        als_wrmf(2, k, u2, v2, None, None, urow, ucol, irow, icol, nu, ni)
        assert np.array_equal(u, u2) and np.array_equal(v, v2)
