"""GPU: the resident-side Funk pass (mfx_sgd_res.hip: ONE persistent launch per feature-epoch over every CU, a lane owns an item and
keeps U[f,item] in a register) must give the SAME BITS as the oracle's sequential loops (gd_estimator.pyx:691-779, :489-582,
:588-685) -- in its mailbox form (k2_res_box: V[f,user] is sent straight to the 8-byte mailbox of the user's next rating; 2, 4 or 8
mailboxes per poll) and in its granule form (k2_res: V[f,user] travels inside a per-user 16-byte {value, chain counter} granule).
On skewed data the items whose chains would bound the pass own a whole WAVE each (k2_hot_role: 64 mailboxes per poll, one
uniform walk), and so do the hottest users (k2_hot_user_role: V[f,user] resident, the item value travels there and back); forced
here by MFX_K2_HOT=<degree> / MFX_K2_HOTU=<degree>.
It is the default from 65 536 ratings on; forced here (MFX_K2_RES=1) also on tiny problems, every test in every form."""
import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["mailbox-auto", "mailbox-2", "mailbox-4", "mailbox-8", "granule", "hot-1", "hot-40", "hotu-1", "hotu-25+hot-60"])
def _force(monkeypatch, request):
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)
    monkeypatch.setenv("MFX_K2_RES", "1")
    form = request.param
    monkeypatch.delenv("MFX_K2_HOT", raising=False)
    monkeypatch.delenv("MFX_K2_HOTU", raising=False)
    if form.startswith("hot"):
        # hot-wave roles of the mailbox form.  hot-<d>: every item with d ratings or more owns a whole wave (up to 2 048 of them,
        # the hottest first), the rest stay one lane each -- hot-1: (almost) everything through the wave role, hot-40: a mix.
        # hotu-<d>: every USER with d ratings or more owns a wave (up to 256, the hottest first) and for its ratings the item
        # value travels there and back -- alone (items on lanes, or on waves where the plan finds hot ones) and with hot items
        for part in form.split("+"):
            name, d = part.split("-")
            monkeypatch.setenv("MFX_K2_HOT" if name == "hot" else "MFX_K2_HOTU", d)
        monkeypatch.delenv("MFX_K2_BOX", raising=False)
    elif form == "granule":
        monkeypatch.setenv("MFX_K2_BOX", "0")
    elif form != "mailbox-auto":
        monkeypatch.setenv("MFX_K2_BOX", form.split("-")[1])
    else:
        monkeypatch.delenv("MFX_K2_BOX", raising=False)


@pytest.mark.parametrize("nu,ni,nnz,k", [(3000, 2000, 200000, 3), (50000, 700, 400000, 2), (37, 29, 600, 4),
                                         (100000, 50000, 1000000, 2)])
def test_without_bias_bitexact(nu, ni, nnz, k):
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    idx, r = synth.make_ratings(nu, ni, nnz, seed=nu + k)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    hist = np.zeros(64)
    G.estimator_loop_without_bias(3, 3, 1e9, k, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni)
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    O.estimator_loop_without_bias(3, 3, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni, rmse_hist=hist)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)


def test_stopping_rule_history_and_second_run_on_the_same_plan():
    """Data-dependent stopping rule (gd_estimator.pyx:742) through the resident API, twice on one plan: the
    chain counters are reset per pass, the residual cache per run."""
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    nu, ni, nnz, k = 4000, 1500, 150000, 4
    idx, r = synth.make_ratings(nu, ni, nnz, seed=2)
    with SgdPlan(idx, r, nu, ni, k) as p:
        for rep in range(2):
            u, v = synth.init_const(k, ni), synth.init_const(k, nu)
            p.set_factors(u, v)
            hist = p.run_funk(2, 1e-4, 0.1, 0.002, 0.05)
            p.get_factors(u, v)
            u2, v2, h2 = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros(4096)
            n = O.estimator_loop_without_bias(2, 0, 1e-4, k, 0.1, 0.002, 0.05, u2, v2, idx, r, nu, ni, rmse_hist=h2)
            assert np.array_equal(u, u2) and np.array_equal(v, v2)
            assert len(hist) == n
            np.testing.assert_allclose(hist, h2[:n], rtol=1e-12)
        # the full-k kernels on the SAME plan: still the oracle's bits
        u, v, ib, ub = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2), np.zeros(ni), np.zeros(nu)
        u2, v2, ib2, ub2 = u.copy(), v.copy(), ib.copy(), ub.copy()
        p.set_factors(u, v, ib, ub)
        p.run_kmf(0, 2, 0.01, 0.1, 0.1, 0.007)
        p.get_factors(u, v, ib, ub)
        O.train_linear_kernel(2, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2)
        assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ub, ub2)


def test_with_bias_and_gates_bitexact():
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    nu, ni, nnz, k = 5000, 3000, 250000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=8)
    rs = np.random.RandomState(0)
    ib, ub = rs.normal(0, 0.2, ni), rs.normal(0, 0.2, nu)
    for gates in ((1, 1), (1, 0), (0, 1)):
        u, v = synth.init_const(k, ni), synth.init_const(k, nu)
        u2, v2 = u.copy(), v.copy()
        G.estimator_loop_with_bias_dev(2, 2, 1e9, k, 0.1, 0.001, 0.0, 0.0, 0.05, 3.4, u, v, idx, r, ib, ub, nu, ni, *gates)
        O.estimator_loop_with_bias_dev(2, 2, 1e9, k, 0.1, 0.001, 0.0, 0.0, 0.05, 3.4, u2, v2, idx, r, ib, ub, nu, ni, *gates)
        assert np.array_equal(u, u2) and np.array_equal(v, v2), gates


def test_large_problems_take_the_resident_pass_by_default(monkeypatch):
    """From 65 536 ratings on the pass is chosen without the override."""
    monkeypatch.delenv("MFX_K2_RES")
    from mfrec_amd.plan import SgdPlan, Timer
    from oracle import oracle as O
    nu, ni, nnz, k = 60000, 12000, 8000000, 2          # ~1200 levels of ~6.5 K ratings
    idx, r = synth.make_ratings(nu, ni, nnz, seed=4)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(u, v)
        p.run_funk(2, 1e9, 0.1, 0.001, 0.05)
        p.get_factors(u, v)
        assert p.kernel_name(3).startswith("k2_res<"), p.kernel_name(3)
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)


@pytest.mark.parametrize("skew", ["zipf_items", "zipf_users", "few_users"])
def test_uneven_load_bitexact(skew):
    """Hand-offs must hold under UNEVEN load: skewed items (long chains of tiny levels next to fat ones), skewed
    users (one workgroup's group carries most of a level, others sit in the barrier), and fewer users than groups
    (most workgroups own empty slices)."""
    from mfrec_amd.lib import gd_estimator as G
    from oracle import oracle as O
    rs = np.random.RandomState(5)
    nu, ni, nnz, k = 6000, 4000, 300000, 2
    if skew == "zipf_items":
        users = rs.randint(0, nu, nnz)
        items = np.minimum((rs.zipf(1.3, nnz) - 1), ni - 1)
    elif skew == "zipf_users":
        users = np.minimum((rs.zipf(1.2, nnz) - 1), nu - 1)
        items = rs.randint(0, ni, nnz)
    else:
        nu = 20
        users = rs.randint(0, nu, nnz)
        items = rs.randint(0, ni, nnz)
    idx = np.ascontiguousarray(np.stack([users, items], 1).astype(np.int32))
    r = rs.randint(1, 6, nnz).astype(np.float64)
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    G.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni)
    u2, v2 = synth.init_const(k, ni), synth.init_const(k, nu)
    O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)


def test_goldens_through_the_resident_pass(golden):
    """The reference's own outputs (g3: without bias + duplicates, g3b: stopping rule, g4: BellKor baseline + gates)."""
    from mfrec_amd.lib import gd_estimator as G
    g = golden("g3_funk")
    k, E = int(g["p_k"]), int(g["p_epochs"])
    u, v = synth.init_const(k, 40, float(g["p_init"])), synth.init_const(k, 60, float(g["p_init"]))
    G.estimator_loop_without_bias(E, E, 1e9, k, float(g["p_init"]), float(g["p_lr"]), float(g["p_K"]), u, v, g["in_idx"], g["in_r"], 60, 40)
    assert np.array_equal(u, g["out_u"]) and np.array_equal(v, g["out_v"])
    g = golden("g4_funk_bias")
    k, E = int(g["p_k"]), int(g["p_epochs"])
    for gate, tag in (((1, 1), "11"), ((1, 0), "10"), ((0, 1), "01")):
        u, v = synth.init_const(k, 40, float(g["p_init"])), synth.init_const(k, 60, float(g["p_init"]))
        G.estimator_loop_with_bias_dev(E, E, 1e9, k, float(g["p_init"]), float(g["p_lr"]), 0.0, 0.0, float(g["p_K"]), float(g["p_mu"]),
                                       u, v, g["in_idx"], g["in_r"], g["in_ib"], g["in_ub"], 60, 40, *gate)
        assert np.array_equal(u, g["out_u_" + tag]) and np.array_equal(v, g["out_v_" + tag]), tag


def test_input_holding_the_empty_mailbox_pattern_is_served_by_the_launch_path():
    """The mailbox form marks an empty mailbox by ONE quiet-NaN payload; a factor that holds exactly these bits (here: the
    V column of a user without ratings, so no arithmetic ever sees it) must not be mistaken for it -- the seeding kernel
    notices, the pass is not applied and the run goes on with one launch per level: same bits as the oracle."""
    from mfrec_amd.plan import SgdPlan
    from oracle import oracle as O
    nu, ni, nnz, k = 3000, 900, 120000, 3
    idx, r = synth.make_ratings(nu - 1, ni, nnz, seed=9)           # user nu-1 has no ratings
    u, v = synth.init_const(k, ni), synth.init_const(k, nu)
    v.view(np.uint64)[:, nu - 1] = 0x7ff85eedb0c50001
    u2, v2 = u.copy(), v.copy()
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(u, v)
        p.run_funk(2, 1e9, 0.1, 0.001, 0.05)
        p.get_factors(u, v)
    O.estimator_loop_without_bias(2, 2, 1e9, k, 0.1, 0.001, 0.05, u2, v2, idx, r, nu, ni)
    assert np.array_equal(u.view(np.uint64), u2.view(np.uint64)) and np.array_equal(v.view(np.uint64), v2.view(np.uint64))
