"""GPU (one device): the RCCL plumbing of the sharded mode, exercised with a 1-rank
communicator (MFX_FORCE_COMM=1): dlopen(librccl), unique id, comm init, the all-reduce on
the library stream and the delta/combine kernels.  The result must equal the CPU simulation
of the same arithmetic (U_start + 1.0 * (U - U_start), not bit-identical to U itself)."""
import ctypes as C
import os

import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


def test_forced_single_rank_comm_matches_simulation():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    from mfrec_amd.sharding import ShardedKMF
    from oracle import sharded_sim as SIM
    B.build(verbose=False)
    S.init(0)
    nu, ni, nnz, k, E = 400, 250, 12000, 64, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=21)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    os.environ["MFX_FORCE_COMM"] = "1"
    try:
        buf = C.create_string_buffer(S.UNIQUE_ID_BYTES)
        S.check(S.lib().mfx_comm_unique_id(buf))
        S.check(S.lib().mfx_comm_init(0, 1, buf.raw))
        T = ShardedKMF(idx, r, nu, ni, k, 0, 1)
        T.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
        rm, se = T.train(0, E, 0.01, 0.1, 0.1, 0.007, scale=1.0)
        u, v, ib, ub = T.get_local()
        T.close()
        S.check(S.lib().mfx_comm_barrier())
        w = C.c_double(3.25)                                  # bench.py's max-over-ranks reduction
        S.check(S.lib().mfx_comm_allreduce_max(C.byref(w)))
        assert w.value == 3.25
    finally:
        S.check(S.lib().mfx_comm_destroy())
        del os.environ["MFX_FORCE_COMM"]
    su, sv, sib, sub = SIM.simulate(0, 1, E, k, 0.01, 0.1, 0.1, 0.007, u0, v0, np.zeros(ni),
                                    np.zeros(nu), idx, r, scale=1.0, force_combine=True)
    assert np.array_equal(u, su) and np.array_equal(v, sv)
    assert np.array_equal(ib, sib) and np.array_equal(ub, sub)
    assert np.all(np.isfinite(rm)) and rm[-1] < rm[0]
    np.testing.assert_allclose(rm, np.sqrt(se / nnz), rtol=1e-15)


def test_forced_single_rank_als_allgather():
    """The sharded ALS entry point with a 1-rank communicator (real ncclAllGather on the
    library stream) must reproduce the un-sharded result bit for bit."""
    import scipy.sparse as sp
    from mfrec_amd import _shim as S
    from mfrec_amd.lib.als_implicit import als_wrmf, als_wrmf_sharded
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    S.init(0)
    nu, ni, k = 300, 200, 40
    rs = np.random.RandomState(2)
    M = sp.coo_matrix((np.ones(6000), (rs.randint(0, nu, 6000), rs.randint(0, ni, 6000))), shape=(nu, ni)).tocsr()
    a, b = create_bool_sparse_row(M); c, d = create_bool_sparse_col(M)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    u, v = u0.copy(), v0.copy()
    als_wrmf(2, k, u, v, None, None, a, b, c, d, nu, ni)
    os.environ["MFX_FORCE_COMM"] = "1"
    try:
        buf = C.create_string_buffer(S.UNIQUE_ID_BYTES)
        S.check(S.lib().mfx_comm_unique_id(buf))
        S.check(S.lib().mfx_comm_init(0, 1, buf.raw))
        u2, v2 = u0.copy(), v0.copy()
        als_wrmf_sharded(2, k, u2, v2, None, None, a, b, c, d, nu, ni)
    finally:
        S.check(S.lib().mfx_comm_destroy())
        del os.environ["MFX_FORCE_COMM"]
    assert np.array_equal(u, u2) and np.array_equal(v, v2)


def test_forced_single_rank_funk_sharded_matches_simulation():
    """Sharded Funk pass with a 1-rank communicator: the per-feature-epoch delta / all-reduce /
    combine arithmetic must equal its CPU simulation (oracle epoch + the same combine)."""
    from mfrec_amd import _shim as S
    from mfrec_amd.plan import SgdPlan
    from mfrec_amd.sharding import combine
    from oracle import oracle as O
    S.init(0)
    nu, ni, k, E = 300, 200, 5, 3
    rs = np.random.RandomState(4)
    cells = rs.permutation(nu * ni)[:8000]                       # no duplicate (user, item) pairs
    idx = np.stack([cells % nu, cells // nu], 1).astype(np.int32)
    r = rs.randint(1, 6, len(cells)).astype(np.float64)
    os.environ["MFX_FORCE_COMM"] = "1"
    try:
        buf = C.create_string_buffer(S.UNIQUE_ID_BYTES)
        S.check(S.lib().mfx_comm_unique_id(buf))
        S.check(S.lib().mfx_comm_init(0, 1, buf.raw))
        u, v = synth.init_const(k, ni), synth.init_const(k, nu)
        with SgdPlan(idx, r, nu, ni, k) as p:
            p.set_factors(u, v)
            hist = p.run_funk_sharded(E, 1e9, 0.1, 0.01, 0.05, scale=1.0)
            p.get_factors(u, v)
    finally:
        S.check(S.lib().mfx_comm_destroy())
        del os.environ["MFX_FORCE_COMM"]
    u2, v2, cache, h2 = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros(nu * ni), []
    for f in range(k):
        for _ in range(E):
            start = u2[f].copy()
            h2.append(O.estimator_subloop(f, 1, 0.0, k, 0.1, 0.01, 0.05, u2, v2, idx, r, cache, nu, ni))
            u2[f] = combine(start, u2[f] - start, 1.0)
        O.predictor_subloop(f, 1, k, 0.1, u2, v2, idx, r, cache, nu, ni)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
    np.testing.assert_allclose(hist, h2, rtol=1e-12)
