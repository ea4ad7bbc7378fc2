"""GPU parity of the resident-side full-k epoch (mfx_sgd_res.hip: MFX_EPOCH_TICKET / MFX_EPOCH_STATIC)
against the oracle: train_linear_kernel / train_logistic_kernel (kmf_train.pyx:195-277, :103-189).

The resident-side launch must give the level schedule's bits -- i.e. the sequential loop's -- whatever
the binding: user rows resident with runs taken by ticket (ratings grouped by user), or the smaller
side's rows owned by fixed waves (any order)."""
import numpy as np
import pytest

from mfrec_amd import synth
from mfrec_amd.plan import LINEAR, LOGISTIC, SgdPlan
from oracle import oracle as O

pytestmark = pytest.mark.gpu
LR, KU, KI, KB = 0.01, 0.1, 0.1, 0.007


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    B.build(verbose=False)
    S.init(0)            # raises if the HIP library or the GPU is missing: no fallback
    yield


def _run_plan(idx, r, nu, ni, k, u0, v0, epochs, mode, epoch, ib0=None, ub0=None, calls=1):
    u, v = u0.copy(), v0.copy()
    ib = np.zeros(ni) if ib0 is None else ib0.copy()
    ub = np.zeros(nu) if ub0 is None else ub0.copy()
    with SgdPlan(idx, r, nu, ni, k, epoch=epoch) as p:
        info = p.info()
        p.set_factors(u, v, ib, ub)
        rm = []
        for _ in range(calls):
            rm.append(p.run_kmf(mode, epochs, LR, KU, KI, KB))
        p.get_factors(u, v, ib, ub)
        name = p.kernel_name(mode)
        info_after = p.info()
    return u, v, ib, ub, np.concatenate(rm), info, info_after, name


def _oracle(idx, r, nu, ni, k, u0, v0, epochs, mode, ib0=None, ub0=None):
    u, v = u0.copy(), v0.copy()
    ib = np.zeros(ni) if ib0 is None else ib0.copy()
    ub = np.zeros(nu) if ub0 is None else ub0.copy()
    fn = O.train_linear_kernel if mode == LINEAR else O.train_logistic_kernel
    fn(epochs, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u, v, idx, r, ib, ub)
    return u, v, ib, ub


def _check(got, want, mode):
    if mode == LINEAR:
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    else:       # logistic: device exp() is within 1 ulp of libm's (tests/test_gpu_parity.py)
        for a, b in zip(got, want):
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("k", [18, 32, 64, 66, 128])
@pytest.mark.parametrize("mode", [LINEAR, LOGISTIC])
def test_ticket_user_major_vs_oracle(k, mode):
    nu, ni, nnz, E = 700, 300, 30000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=5)
    idx, r = synth.user_major(idx, r)                        # get_ratings(randomize_order=False)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    rs = np.random.RandomState(3)
    ib0, ub0 = rs.normal(0, 0.05, ni), rs.normal(0, 0.05, nu)
    u, v, ib, ub, rm, info, _, name = _run_plan(idx, r, nu, ni, k, u0, v0, E, mode, "ticket", ib0, ub0)
    assert info["epoch_kernel"] == "ticket" and "ticket" in name
    _check((u, v, ib, ub), _oracle(idx, r, nu, ni, k, u0, v0, E, mode, ib0, ub0), mode)
    assert np.all(np.isfinite(rm)) and rm.shape == (E,)


@pytest.mark.parametrize("shape", [(900, 250), (250, 900)])     # items resident / users resident
@pytest.mark.parametrize("k", [20, 40, 64, 128])
@pytest.mark.parametrize("mode", [LINEAR, LOGISTIC])
def test_static_any_order_vs_oracle(shape, k, mode):
    nu, ni = shape
    nnz, E = 40000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=9)          # generation order = a shuffled order, duplicates kept
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    u, v, ib, ub, rm, info, _, name = _run_plan(idx, r, nu, ni, k, u0, v0, E, mode, "static")
    assert info["epoch_kernel"] == "static" and ("item rows" if ni <= nu else "user rows") in name
    _check((u, v, ib, ub), _oracle(idx, r, nu, ni, k, u0, v0, E, mode), mode)


@pytest.mark.parametrize("mode", [LINEAR, LOGISTIC])
def test_static_on_dense_chains_and_repeated_pairs(mode):
    """Few users and few items: long chains on both sides; runs of the SAME (user, item) pair back to back (a rating that depends on its
    predecessor through the resident row AND the travelling row); items with one or two ratings; more slots than entities."""
    k, E = 64, 4
    rs = np.random.RandomState(17)
    nu, ni = 37, 23
    users = rs.randint(0, nu, 9000).astype(np.int32)
    items = rs.randint(0, ni - 3, 9000).astype(np.int32)
    # back-to-back repeats of a pair, and a user walking one item several times in a row between other users' ratings
    for at in range(100, 8900, 400):
        users[at:at + 5] = users[at]
        items[at:at + 5] = items[at]
        users[at + 10:at + 16:2] = users[at + 10]
        items[at + 10:at + 16] = items[at + 10]
    items[-1], items[-2], items[-3] = ni - 1, ni - 2, ni - 2   # items with one / two ratings at the very end
    idx = np.ascontiguousarray(np.stack([users, items], axis=1))
    r = rs.randint(1, 6, users.size).astype(np.float64)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    ib0, ub0 = rs.normal(0, 0.05, ni), rs.normal(0, 0.05, nu)
    got = _run_plan(idx, r, nu, ni, k, u0, v0, E, mode, "static", ib0, ub0)
    _check(got[:4], _oracle(idx, r, nu, ni, k, u0, v0, E, mode, ib0, ub0), mode)
    # the online RMSE is a fixed-association sum per slot: the same bits run to run, whatever the timing
    again = _run_plan(idx, r, nu, ni, k, u0, v0, E, mode, "static", ib0, ub0)
    assert np.array_equal(again[4], got[4])


def test_static_on_user_major_and_repeated_calls():
    """The static binding takes any order, also a grouped one; a second call continues from the first."""
    nu, ni, nnz, k = 400, 600, 25000, 64
    idx, r = synth.user_major(*synth.make_ratings(nu, ni, nnz, seed=2))
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    got = _run_plan(idx, r, nu, ni, k, u0, v0, 2, LINEAR, "static", calls=2)
    _check(got[:4], _oracle(idx, r, nu, ni, k, u0, v0, 4, LINEAR), LINEAR)
    got = _run_plan(idx, r, nu, ni, k, u0, v0, 2, LINEAR, "ticket", calls=2)
    _check(got[:4], _oracle(idx, r, nu, ni, k, u0, v0, 4, LINEAR), LINEAR)


def test_rmse_matches_level_schedule():
    """Per-epoch online RMSE: a deterministic fixed-association sum in every kernel (1e-12 relative)."""
    nu, ni, nnz, k = 500, 400, 30000, 64
    idx, r = synth.user_major(*synth.make_ratings(nu, ni, nnz, seed=4))
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    rms = {e: _run_plan(idx, r, nu, ni, k, u0, v0, 3, LINEAR, e)[4] for e in ("levels", "ticket", "static")}
    np.testing.assert_allclose(rms["ticket"], rms["levels"], rtol=1e-12)
    np.testing.assert_allclose(rms["static"], rms["levels"], rtol=1e-12)
    again = _run_plan(idx, r, nu, ni, k, u0, v0, 3, LINEAR, "ticket")[4]
    assert np.array_equal(again, rms["ticket"])              # same bits run to run, whatever the placement


def test_edge_cases():
    """One user holding every rating (a single run), users without ratings, trailing unrated ids, a run of one."""
    k = 64
    u0 = synth.init_normal(k, 50, 1)
    # one user, 300 ratings over 50 items (each item's chain is this user's own order)
    idx = np.stack([np.zeros(300, dtype=np.int32), (np.arange(300) * 7 % 50).astype(np.int32)], axis=1)
    r = (1.0 + np.arange(300) % 5).astype(np.float64)
    v0 = synth.init_normal(k, 3, 2)
    for e in ("ticket", "static"):
        got = _run_plan(np.ascontiguousarray(idx), r, 3, 50, k, u0, v0, 2, LINEAR, e)
        _check(got[:4], _oracle(idx, r, 3, 50, k, u0, v0, 2, LINEAR), LINEAR)
    # users 1 and 3 unrated, user 9 (last id) unrated, user 5 has one rating
    users = np.array([0] * 40 + [2] * 30 + [4] * 50 + [5] + [7] * 60, dtype=np.int32)
    items = (np.arange(users.size) * 11 % 37).astype(np.int32)
    idx = np.ascontiguousarray(np.stack([users, items], axis=1))
    r = (1.0 + np.arange(users.size) % 5).astype(np.float64)
    u0, v0 = synth.init_normal(k, 40, 1), synth.init_normal(k, 10, 2)
    for e in ("ticket", "static"):
        got = _run_plan(idx, r, 10, 40, k, u0, v0, 3, LINEAR, e)
        _check(got[:4], _oracle(idx, r, 10, 40, k, u0, v0, 3, LINEAR), LINEAR)


def test_not_eligible_is_an_argument_error():
    nu, ni, nnz = 300, 200, 5000
    idx, r = synth.make_ratings(nu, ni, nnz, seed=1)          # not grouped by user
    with pytest.raises(ValueError):
        SgdPlan(idx, r, nu, ni, 64, epoch="ticket")
    with pytest.raises(ValueError):
        SgdPlan(idx, r, nu, ni, 7, epoch="static")            # odd k
    big = 70000                                               # more rows than the register file holds at k = 64
    idx2 = np.stack([np.arange(big, dtype=np.int32) % big, np.arange(big, dtype=np.int32)], axis=1)
    with pytest.raises(ValueError):
        SgdPlan(np.ascontiguousarray(idx2), np.ones(big), big, big, 64, epoch="static")
    # AUTO never fails: it falls back to the level schedule
    with SgdPlan(idx, r, nu, ni, 7) as p:
        assert p.info()["epoch_kernel"] == "levels"


@pytest.mark.parametrize("epoch,order", [("static", "shuffled"), ("static", "user-major"), ("ticket", "user-major")])
@pytest.mark.parametrize("k", [24, 64, 100, 128])
@pytest.mark.parametrize("mode", [LINEAR, LOGISTIC])
def test_butterfly_dot_on_the_resident_side_epoch(epoch, order, k, mode):
    """The tolerance mode (SgdPlan(dot="butterfly"): per-lane FMA partials + a wave butterfly over the lanes of a slot, FMA in the row
    updates -- north_star's wording of the kernel) inside k1_res: no LDS product tile, no chain of k dependent adds.  Deterministic;
    within 1e-11 of the oracle after 3 epochs (the budget is 1e-5 RMSE: tests/test_gpu_parity.py, tests/test_gpu_fullsize.py); for
    the linear kernel the SAME bits as the level kernels' tolerance mode at every k (24 / 64 / 100 / 128: 16 / 32 / 64 / 64 lanes per
    row in both) -- a plan that falls back to the level schedule in the middle of a run keeps its trajectory; logistic: 1e-11."""
    nu, ni, nnz, E = 700, 300, 30000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=6)
    if order == "user-major":
        idx, r = synth.user_major(idx, r)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    rs = np.random.RandomState(3)
    ib0, ub0 = rs.normal(0, 0.05, ni), rs.normal(0, 0.05, nu)

    def run(e):
        u, v, ib, ub = u0.copy(), v0.copy(), ib0.copy(), ub0.copy()
        with SgdPlan(idx, r, nu, ni, k, dot="butterfly", epoch=e) as p:
            name, kern = p.kernel_name(mode), p.info()["epoch_kernel"]
            p.set_factors(u, v, ib, ub)
            rm = p.run_kmf(mode, E, LR, KU, KI, KB)
            p.get_factors(u, v, ib, ub)
        return (u, v, ib, ub, rm), name, kern

    got, name, kern = run(epoch)
    assert kern == epoch and name.startswith("k1_res<") and "butterfly-dot" in name, name
    again, _, _ = run(epoch)
    for a, b in zip(got, again):
        assert np.array_equal(a, b)                                  # deterministic, whatever the timing
    want = _oracle(idx, r, nu, ni, k, u0, v0, E, mode, ib0, ub0)
    for a, b in zip(got[:4], want):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-11)
    assert not np.array_equal(got[0], want[0])                       # it is NOT the exact mode
    lev, lname, _ = run("levels")
    assert lname.startswith("k1_level<") and "butterfly-dot" in lname
    if mode == LINEAR:           # every k: the same lanes-per-row layout and the same pairwise tree in both kernels (include/mfx.h)
        for a, b in zip(got[:4], lev[:4]):
            assert np.array_equal(a, b)
    else:
        for a, b in zip(got[:4], lev[:4]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-11)


@pytest.mark.parametrize("epoch,order", [("static", "shuffled"), ("ticket", "user-major")])
@pytest.mark.parametrize("k", [24, 64, 100, 128])
@pytest.mark.parametrize("mode", [LINEAR, LOGISTIC])
def test_f32_row_storage_on_the_resident_side_epoch(epoch, order, k, mode):
    """SgdPlan(storage="f32") -- "f32 storage, f64 arithmetic": rows widened on load, rounded once per update -- inside k1_res: the
    travelling rows move 8 bytes per lane instead of 16, the resident row in registers is rounded to f32 after every update (what a
    reload would return).  The ordered sum and every other operation are the exact mode's, so the result is the level kernels' f32
    mode bit for bit (linear), whatever their lane layout; NOT the reference's precision (its RMSE budget: tests/test_gpu_parity.py)."""
    nu, ni, nnz, E = 700, 300, 30000, 3
    idx, r = synth.make_ratings(nu, ni, nnz, seed=6)
    if order == "user-major":
        idx, r = synth.user_major(idx, r)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    rs = np.random.RandomState(3)
    ib0, ub0 = rs.normal(0, 0.05, ni), rs.normal(0, 0.05, nu)

    def run(e):
        u, v, ib, ub = u0.copy(), v0.copy(), ib0.copy(), ub0.copy()
        with SgdPlan(idx, r, nu, ni, k, storage="f32", epoch=e) as p:
            name, kern = p.kernel_name(mode), p.info()["epoch_kernel"]
            p.set_factors(u, v, ib, ub)
            rm = p.run_kmf(mode, E, LR, KU, KI, KB)
            p.get_factors(u, v, ib, ub)
        return (u, v, ib, ub, rm), name, kern

    got, name, kern = run(epoch)
    assert kern == epoch and name.startswith("k1_res<") and "f32 rows" in name, name
    lev, lname, _ = run("levels")
    assert lname.startswith("k1_level<") and "f32" in lname
    if mode == LINEAR:
        for a, b in zip(got[:4], lev[:4]):
            assert np.array_equal(a, b)
        np.testing.assert_allclose(got[4], lev[4], rtol=1e-12)
    else:
        for a, b in zip(got[:4], lev[:4]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)      # (f32 rounding amplifies the 1-ulp difference of the device exp)
    want = _oracle(idx, r, nu, ni, k, u0, v0, E, mode, ib0, ub0)
    assert not np.array_equal(got[0], want[0])                       # it is NOT the exact mode ...
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-5)   # ... but close to it


def test_gated_and_narrow_runs_take_the_level_schedule():
    """update_users/items gates (and bf16 storage) are served by the level kernels of the same plan."""
    nu, ni, nnz, k = 300, 500, 20000, 64
    idx, r = synth.user_major(*synth.make_ratings(nu, ni, nnz, seed=8))
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    with SgdPlan(idx, r, nu, ni, k, epoch="ticket") as p:
        p.set_factors(u, v, ib, ub)
        p.run_kmf(LINEAR, 2, LR, KU, KI, KB)                              # resident
        p.run_kmf(LINEAR, 1, LR, KU, KI, KB, update_users=1, update_items=0)   # levels
        p.run_kmf(LINEAR, 1, LR, KU, KI, KB)                              # resident again
        p.get_factors(u, v, ib, ub)
    u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    O.train_linear_kernel(2, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u2, v2, idx, r, ib2, ub2)
    O.train_linear_kernel(1, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u2, v2, idx, r, ib2, ub2, 1, 0)
    O.train_linear_kernel(1, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u2, v2, idx, r, ib2, ub2)
    _check((u, v, ib, ub), (u2, v2, ib2, ub2), LINEAR)


def _zipf(nu, ni, nnz, seed, a_users, a_items):
    rs = np.random.RandomState(seed)
    pu, pi = 1.0 / np.arange(1, nu + 1) ** a_users, 1.0 / np.arange(1, ni + 1) ** a_items
    idx = np.stack([rs.choice(nu, nnz, p=pu / pu.sum()), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32)
    return np.ascontiguousarray(idx), rs.randint(1, 6, nnz).astype(np.float64)


@pytest.mark.parametrize("shape,k", [((3000, 2000), 64), ((2000, 3000), 64), ((3000, 2000), 24), ((2500, 1500), 128)])
@pytest.mark.parametrize("mode", [LINEAR, LOGISTIC])
@pytest.mark.parametrize("burst_run", [None, "3", "400"])
def test_power_law_data_through_the_burst_variant(shape, k, mode, burst_run, monkeypatch):
    """Skewed degrees (both sides): the static binding's burst variant walks a hot resident entity's run eight ratings per
    step -- duplicates of a pair inside a window, runs shorter than a window and windows that stop at a rating that is not
    ready yet included -- and must still give the sequential loop's bits.  A slot bursts only while it has more ratings ahead
    than an average resident entity has in the epoch (round 5; MFX_K1_BURST_RUN overrides: 3 = round 4's rule, every slot)."""
    monkeypatch.setenv("MFX_K1_BURST", "1")
    if burst_run:
        monkeypatch.setenv("MFX_K1_BURST_RUN", burst_run)
    nu, ni = shape
    idx, r = _zipf(nu, ni, 60000, 3, 0.8, 1.1)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    u, v, ib, ub, rm, info, _, name = _run_plan(idx, r, nu, ni, k, u0, v0, 2, mode, "static")
    assert info["epoch_kernel"] == "static" and "bursts" in name, name
    _check((u, v, ib, ub), _oracle(idx, r, nu, ni, k, u0, v0, 2, mode), mode)
    monkeypatch.setenv("MFX_K1_BURST", "0")
    rm0 = _run_plan(idx, r, nu, ni, k, u0, v0, 2, mode, "static")[4]
    np.testing.assert_allclose(rm, rm0, rtol=1e-12)


def test_skew_is_detected_and_served_by_the_burst_variant():
    """AUTO on a catalogue with one very hot item: the cost model finds the chain on the resident side and picks the burst variant."""
    nu, ni, nnz, k = 20000, 8000, 1500000, 64
    idx, r = _zipf(nu, ni, nnz, 5, 0.0, 1.0)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    got = _run_plan(idx, r, nu, ni, k, u0, v0, 1, LINEAR, "auto")
    assert got[5]["epoch_kernel"] == "static" and "bursts" in got[7], (got[5], got[7])
    lev = _run_plan(idx, r, nu, ni, k, u0, v0, 1, LINEAR, "levels")
    _check(got[:4], lev[:4], LINEAR)


def test_failed_co_residency_check_falls_back_untouched(monkeypatch):
    """A static launch whose grid is not co-resident (a GPU shared with another process; forced here by a test hook that makes the
    registration wait for a wave that does not exist) must modify NOTHING and hand the batch to the level schedule -- for the full-k
    epoch and for the Funk pass (whose residual cache moves to the level-major layout on the way)."""
    from mfrec_amd.lib import gd_estimator as G
    monkeypatch.setenv("MFX_TESTING", "1")
    monkeypatch.setenv("MFX_TEST_REG_FAIL", "1")
    nu, ni, nnz, k = 900, 400, 40000, 64
    idx, r = synth.make_ratings(nu, ni, nnz, seed=11)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    u, v, ib, ub, rm, info, info_after, name = _run_plan(idx, r, nu, ni, k, u0, v0, 2, LINEAR, "static", calls=2)
    assert info["epoch_kernel"] == "static" and info_after["epoch_kernel"] == "levels" and info_after["epoch_kernel_prepared"] == "static"
    _check((u, v, ib, ub), _oracle(idx, r, nu, ni, k, u0, v0, 4, LINEAR), LINEAR)
    assert np.all(np.isfinite(rm)) and rm.shape == (4,)
    # Funk: forced onto the resident-side pass, which fails its check at the first batch
    monkeypatch.setenv("MFX_K2_RES", "1")
    kf = 3
    fu, fv = synth.init_const(kf, ni), synth.init_const(kf, nu)
    G.estimator_loop_without_bias(2, 2, 1e9, kf, 0.1, 0.001, 0.05, fu, fv, idx, r, nu, ni)
    fu2, fv2 = synth.init_const(kf, ni), synth.init_const(kf, nu)
    O.estimator_loop_without_bias(2, 2, 1e9, kf, 0.1, 0.001, 0.05, fu2, fv2, idx, r, nu, ni)
    assert np.array_equal(fu, fu2) and np.array_equal(fv, fv2)


def test_funk_fallback_in_the_middle_of_a_run_hands_the_cache_over(monkeypatch):
    """The resident-side Funk pass loses its co-residency check at its 5th launch (3 features x 2 epochs each + commits are
    under way): the residual cache built so far moves to the level-major layout and the run ends on the level schedule with
    the oracle's bits."""
    from mfrec_amd.lib import gd_estimator as G
    monkeypatch.setenv("MFX_TESTING", "1")
    monkeypatch.setenv("MFX_K2_RES", "1")
    nu, ni, nnz, kf = 900, 400, 40000, 4
    idx, r = synth.make_ratings(nu, ni, nnz, seed=12)
    fu2, fv2 = synth.init_const(kf, ni), synth.init_const(kf, nu)
    O.estimator_loop_without_bias(2, 2, 1e9, kf, 0.1, 0.001, 0.05, fu2, fv2, idx, r, nu, ni)
    for n in (3, 5, 6):
        monkeypatch.setenv("MFX_TEST_REG_FAIL", "0")                 # (resets the hook's launch counter)
        fu, fv = synth.init_const(kf, ni), synth.init_const(kf, nu)
        G.estimator_loop_without_bias(1, 1, 1e9, 1, 0.1, 0.001, 0.05, fu[:1].copy(), fv[:1].copy(), idx, r, nu, ni)
        monkeypatch.setenv("MFX_TEST_REG_FAIL", str(n))
        G.estimator_loop_without_bias(2, 2, 1e9, kf, 0.1, 0.001, 0.05, fu, fv, idx, r, nu, ni)
        assert np.array_equal(fu, fu2) and np.array_equal(fv, fv2), n


def test_full_k_fallback_in_the_middle_of_a_batch(monkeypatch):
    """Four epochs enqueued back to back on the static binding; the third launch fails its co-residency check: epochs 1-2 stay
    applied, 3-4 run on the level schedule, and the per-epoch RMSE history is the level schedule's."""
    monkeypatch.setenv("MFX_TESTING", "1")
    monkeypatch.setenv("MFX_GRAPH", "0")        # eager launches: the hook counts enqueued launches (a replayed graph is enqueued once)
    nu, ni, nnz, k = 900, 400, 40000, 64
    idx, r = synth.make_ratings(nu, ni, nnz, seed=13)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    want = _run_plan(idx, r, nu, ni, k, u0, v0, 4, LINEAR, "levels")
    for n in (1, 3, 4):
        monkeypatch.setenv("MFX_TEST_REG_FAIL", "0")
        _run_plan(idx, r, nu, ni, 18, synth.init_normal(18, ni, 1), synth.init_normal(18, nu, 2), 1, LINEAR, "static")   # resets the hook's counter
        monkeypatch.setenv("MFX_TEST_REG_FAIL", str(n))
        got = _run_plan(idx, r, nu, ni, k, u0, v0, 4, LINEAR, "static")
        _check(got[:4], want[:4], LINEAR)
        np.testing.assert_allclose(got[4], want[4], rtol=1e-12)
        assert got[6]["epoch_kernel"] == "levels", n


def test_give_up_on_a_complete_registration_count_is_success_everywhere(monkeypatch):
    """The registration barrier's give-up path raises its flag with a compare-and-swap that only succeeds on an INCOMPLETE count.
    Test hook MFX_TEST_REG_FAIL=-1: wave 0 takes the give-up path at the very moment it sees the count complete (what a time-out
    that fires together with the last arrival looks like) -- every wave must still pass, the epoch runs on the static binding and
    gives the oracle's bits; the same for both forms of the resident-side Funk pass."""
    from mfrec_amd.lib import gd_estimator as G
    monkeypatch.setenv("MFX_TESTING", "1")
    monkeypatch.setenv("MFX_TEST_REG_FAIL", "-1")
    nu, ni, nnz, k = 900, 400, 40000, 64
    idx, r = synth.make_ratings(nu, ni, nnz, seed=21)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    got = _run_plan(idx, r, nu, ni, k, u0, v0, 3, LINEAR, "static")
    assert got[6]["epoch_kernel"] == "static"                       # no fall-back happened
    _check(got[:4], _oracle(idx, r, nu, ni, k, u0, v0, 3, LINEAR), LINEAR)
    monkeypatch.setenv("MFX_K2_RES", "1")
    kf = 3
    fu2, fv2 = synth.init_const(kf, ni), synth.init_const(kf, nu)
    O.estimator_loop_without_bias(2, 2, 1e9, kf, 0.1, 0.001, 0.05, fu2, fv2, idx, r, nu, ni)
    for box in ("2", "0"):                                          # mailbox form (2 per poll) / granule form
        monkeypatch.setenv("MFX_K2_BOX", box)
        fu, fv = synth.init_const(kf, ni), synth.init_const(kf, nu)
        G.estimator_loop_without_bias(2, 2, 1e9, kf, 0.1, 0.001, 0.05, fu, fv, idx, r, nu, ni)
        assert np.array_equal(fu, fu2) and np.array_equal(fv, fv2), box
