"""GPU: the Funk per-feature loops on the ITEM-BLOCK order (sharding.funk_block_order / mfx_dsgd_run_funk): a pass of a feature walks the
item blocks one after the other on ONE user side -- catalogues with more items than the resident-side pass has lanes (~450 K) get the
persistent pass block by block.  Bit-identical to the oracle's sequential loops (gd_estimator.pyx:691-779, :489-685) ON THE ORDER THE
OBJECT REPORTS, history of the stopping rule included."""
import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _init():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)


@pytest.mark.parametrize("res", ["1", "0"])          # every block on the resident-side pass (forced: tiny blocks) / on the launch paths
@pytest.mark.parametrize("nu,ni,nnz,k,blocks", [(3000, 2000, 200000, 3, 3), (50000, 700, 300000, 2, 2), (37, 29, 600, 4, 5)])
def test_without_bias_on_the_block_order_bitexact(monkeypatch, res, nu, ni, nnz, k, blocks):
    from mfrec_amd import sharding as SH
    from oracle import oracle as O
    monkeypatch.setenv("MFX_K2_RES", res)
    idx, r = synth.make_ratings(nu, ni, nnz, seed=nu + k)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    perm, hist = SH.funk_block_order(3, 3, 1e9, k, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni, item_blocks=blocks)
    assert sorted(perm.tolist()) == list(range(nnz))
    # the order: grouped by item block, stable
    bounds = SH.item_blocks(idx, ni, blocks)
    blk = np.searchsorted(bounds, idx[perm, 1], side="right") - 1
    assert np.all(np.diff(blk) >= 0)
    for b in range(blocks):
        assert np.all(np.diff(perm[blk == b]) > 0)
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    h2 = np.zeros(64)
    n2 = O.estimator_loop_without_bias(3, 3, 1e9, k, 0.1, 0.001, 0.05, u2, v2, np.ascontiguousarray(idx[perm]),
                                       np.ascontiguousarray(r[perm]), nu, ni, rmse_hist=h2)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
    assert len(hist) == n2 == 3 * k and np.allclose(hist, h2[:n2], rtol=1e-12, atol=0)


def test_stopping_rule_with_biases_and_gates_on_the_block_order(monkeypatch):
    """The data-dependent stopping rule (gd_estimator.pyx:742: rmse / rmse_last persist across features) decides over the WHOLE pass
    -- the blocks' squared errors added in block order --, with read-only biases and both update gates (estimator_loop_with_bias_dev)."""
    from mfrec_amd import sharding as SH
    from oracle import oracle as O
    monkeypatch.setenv("MFX_K2_RES", "1")
    nu, ni, nnz, k = 900, 600, 60000, 4
    idx, r = synth.make_ratings(nu, ni, nnz, seed=5)
    rs = np.random.RandomState(2)
    ib, ub, mu = rs.normal(0, 0.2, ni), rs.normal(0, 0.2, nu), 3.4
    for gates in ((1, 1), (1, 0), (0, 1)):
        u, v = synth.init_const(k, ni), synth.init_const(k, nu)
        d = SH.DsgdKMF(idx, r, nu, ni, k, -1, 1, item_blocks_1gpu=4)
        d.set_factors(u, v, ib, ub)
        hist = d.train_funk(2, 1e-4, 0.1, 0.02, 0.05, mu, True, gates[0], gates[1])
        d.get_factors(u, v)
        perm = d.order()
        d.close()
        u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
        h2 = np.zeros(4096)
        n2 = O.estimator_loop_with_bias_dev(2, 2, 1e-4, k, 0.1, 0.02, 0.0, 0.0, 0.05, mu, u2, v2, np.ascontiguousarray(idx[perm]),
                                            np.ascontiguousarray(r[perm]), ib, ub, nu, ni, gates[0], gates[1], rmse_hist=h2)
        assert n2 > 2 * k                                   # the rule extended some feature beyond min_epochs
        assert len(hist) == n2 and np.allclose(hist, h2[:n2], rtol=1e-12, atol=0)
        assert np.array_equal(u, u2) and np.array_equal(v, v2)


def test_second_run_and_a_full_k_epoch_on_the_same_object():
    """The block object stays usable: a Funk run, then a full-k epoch (mfx_dsgd_run_kmf) on the factors it left, then Funk again."""
    from mfrec_amd import sharding as SH
    from oracle import oracle as O
    nu, ni, nnz, k = 500, 400, 30000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=9)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    ib, ub = np.zeros(ni), np.zeros(nu)
    d = SH.DsgdKMF(idx, r, nu, ni, k, -1, 1, item_blocks_1gpu=2)
    d.set_factors(u, v, ib, ub)
    d.train_funk(2, 1e9, 0.1, 0.001, 0.05)
    d.train(0, 1, 0.01, 0.1, 0.1, 0.007)
    d.train_funk(1, 1e9, 0.1, 0.001, 0.05)
    d.get_factors(u, v, ib, ub)
    perm = d.order()
    d.close()
    ip, rp = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    u2, v2, ib2, ub2 = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros(ni), np.zeros(nu)
    O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, ip, rp, nu, ni)
    O.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, ip, rp, ib2, ub2)
    O.estimator_loop_without_bias(1, 1, 1e9, k, 0.1, 0.001, 0.05, u2, v2, ip, rp, nu, ni)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ub, ub2)
