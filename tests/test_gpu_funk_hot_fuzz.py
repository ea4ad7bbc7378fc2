"""GPU: the hot-wave roles of the resident-side Funk pass (mfx_sgd_res.hip: k2_hot_role, k2_hot_user_role) on randomized SKEWED
shapes -- power-law degrees on either or both sides, duplicates, users / items that never occur, random thresholds for both roles
(so that the remote ratings of hot users meet lanes AND hot-item waves, windows end on remote and on empty mailboxes, hot users rate
one item twice in a row), every bias / gate variant -- against the oracle's sequential loops (gd_estimator.pyx:691-779, :588-685),
bit for bit.  MFX_FUZZ_SEED=n shifts the seeds (soak runs)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("MFX_FUZZ_SEED", "0"))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    yield


def _skewed(rs, n, a, size):
    if a == 0.0:
        return rs.randint(0, n, size)
    p = 1.0 / np.arange(1, n + 1) ** a
    return rs.permutation(n)[rs.choice(n, size, p=p / p.sum())]          # (hot ids anywhere in the range)


@pytest.mark.parametrize("case", range(14))
def test_hot_roles_random_skewed_shapes(case, monkeypatch):
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    rs = np.random.RandomState(4100 + 17 * case + SEED)
    nu, ni = int(rs.choice([40, 700, 5000, 30000])), int(rs.choice([30, 500, 3000, 12000]))
    nnz = int(rs.choice([3000, 40000, 250000]))
    au, ai = float(rs.choice([0.0, 0.6, 1.1])), float(rs.choice([0.0, 0.7, 1.2]))
    users, items = _skewed(rs, nu, au, nnz), _skewed(rs, ni, ai, nnz)
    if case % 3 == 0:                                       # runs of one (user, item) pair: a remote rating right behind another
        rep = rs.randint(0, nnz - 4, 50)
        for j in rep:
            users[j + 1], items[j + 1] = users[j], items[j]
    idx = np.ascontiguousarray(np.stack([users, items], 1).astype(np.int32))
    r = rs.randint(1, 6, nnz).astype(np.float64)
    k = int(rs.choice([1, 2, 3]))
    monkeypatch.setenv("MFX_K2_RES", "1")
    monkeypatch.delenv("MFX_K2_BOX", raising=False)
    hot, hotu = rs.choice(["auto", "1", "20", "300"]), rs.choice(["auto", "1", "15", "200"])
    for name, val in (("MFX_K2_HOT", hot), ("MFX_K2_HOTU", hotu)):
        if val == "auto":
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(val))
    init = 0.1
    u, v = np.zeros((k, ni)) + init, np.zeros((k, nu)) + init
    u2, v2 = u.copy(), v.copy()
    msg = "case %d: nu=%d ni=%d nnz=%d a_u=%.1f a_i=%.1f k=%d HOT=%s HOTU=%s" % (case, nu, ni, nnz, au, ai, k, hot, hotu)
    if case % 2 == 0:
        G.estimator_loop_without_bias(2, 2, 1e9, k, init, 0.002, 0.05, u, v, idx, r, nu, ni)
        O.estimator_loop_without_bias(2, 2, 1e9, k, init, 0.002, 0.05, u2, v2, idx, r, nu, ni)
    else:
        ib, ub = rs.normal(0, 0.2, ni), rs.normal(0, 0.2, nu)
        gates = [(1, 1), (1, 0), (0, 1)][(case // 2) % 3]
        G.estimator_loop_with_bias_dev(2, 2, 1e9, k, init, 0.002, 0.0, 0.0, 0.05, 3.2, u, v, idx, r, ib, ub, nu, ni, *gates)
        O.estimator_loop_with_bias_dev(2, 2, 1e9, k, init, 0.002, 0.0, 0.0, 0.05, 3.2, u2, v2, idx, r, ib, ub, nu, ni, *gates)
    assert np.array_equal(u, u2) and np.array_equal(v, v2), msg
