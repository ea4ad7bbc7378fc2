"""CPU: host-side marshalling that sits either side of the kernels."""
import numpy as np
import pytest
import scipy.sparse as sp

from mfrec_amd.lib import datasets as D
from mfrec_amd.recommendation import Error, GDRecommender, KMFRecommender, WRMFRecommender


def test_create_bool_sparse_matches_reference_fixture(golden):
    g = golden("g10_datasets")
    M = sp.coo_matrix((g["in_val"], (g["in_row"], g["in_col"])), shape=tuple(g["p_shape"])).tolil()
    a, b = D.create_bool_sparse_row(M)
    c, d = D.create_bool_sparse_col(M)
    for x, y in ((a, "out_urow"), (b, "out_ucol"), (c, "out_irow"), (d, "out_icol")):
        assert x.dtype == np.int32 and np.array_equal(x, g[y])


def test_als_fixture_runlength_matches_reference(golden):
    g = golden("g6_als")
    M = sp.coo_matrix((g["in_coo_val"], (g["in_coo_row"], g["in_coo_col"])),
                      shape=(int(g["p_nu"]), int(g["p_ni"]))).tolil()
    a, b = D.create_bool_sparse_row(M)
    c, d = D.create_bool_sparse_col(M)
    assert np.array_equal(a, g["in_urow"]) and np.array_equal(b, g["in_ucol"])
    assert np.array_equal(c, g["in_irow"]) and np.array_equal(d, g["in_icol"])


def test_get_ratings_order_and_shuffle():
    R = GDRecommender(5, 4, {"nbr_features": 3})
    for a, b, x in ((2, 0, 5), (0, 1, 3), (0, 0, 1), (4, 3, 2)):
        R.set_item_by_id(a, b, x)
    idx, r = R.get_ratings()
    assert idx.dtype == np.int32 and r.dtype == np.float64
    assert idx.tolist() == [[0, 0], [0, 1], [2, 0], [4, 3]] and r.tolist() == [1, 3, 5, 2]
    np.random.seed(3)
    idx2, r2 = R.get_ratings(randomize_order=True)
    np.random.seed(3)
    order = np.arange(4); np.random.shuffle(order)
    assert np.array_equal(idx2, idx[order]) and np.array_equal(r2, r[order])
    R.compute_overall_avg()
    assert R.overall_bias == 2.75 and R.get_nbr_ratings() == 4


def test_bellkor_biases():
    R = GDRecommender(4, 3, {})
    trip = [(0, 0, 5.0), (0, 1, 3.0), (1, 0, 4.0), (2, 2, 1.0), (3, 0, 2.0)]
    for a, b, x in trip:
        R.set_item_by_id(a, b, x)
    R.compute_overall_avg(); R.compute_items_bias_bk(); R.compute_users_bias_bk()
    mu = np.mean([t[2] for t in trip])
    bi = np.zeros(3); bu = np.zeros(4)
    for i in range(3):                       # direct evaluation of mf.py:95
        xs = [x - mu for (a, b, x) in trip if b == i]
        bi[i] = sum(xs) / (R.K3 + len(xs)) if xs else 0.0
    for u in range(4):                       # mf.py:119
        xs = [x - mu - bi[b] for (a, b, x) in trip if a == u]
        bu[u] = sum(xs) / (R.K2 + len(xs)) if xs else 0.0
    np.testing.assert_allclose(R.items_bias, bi, rtol=1e-14)
    np.testing.assert_allclose(R.users_bias, bu, rtol=1e-14)


def test_defaults_and_parameter_maps():
    g, k, w = GDRecommender(), KMFRecommender(), WRMFRecommender()
    assert (g.min_epochs, g.learning_rate, g.K, g.dimensionality, g.feature_init) == (275, 0.001, 0.05, 40, 0.1)
    assert (k.nbr_epochs, k.learning_rate, k.K_users, k.K_items, k.K_bias) == (200, 0.01, 0.1, 0.1, 0.007)
    assert (w.nbr_epochs, w.dimensionality) == (20, 20)
    k2 = KMFRecommender(3, 3, {"regularization_users": 9.0})
    assert k2.K == 9.0 and k2.K_users == 0.1          # the reference's inert mapping (kmf.py:39-41)
    try:
        WRMFRecommender(3, 3, {"bogus": 1})
        assert False
    except Error:
        pass
    assert GDRecommender.fit is GDRecommender.train or GDRecommender.fit == GDRecommender.train


def test_snapshot_roundtrip(tmp_path):
    R = GDRecommender(3, 2, {"nbr_features": 2})
    R.svd_u, R.svd_v = np.arange(4.0).reshape(2, 2), np.arange(6.0).reshape(2, 3)
    R.save_model_snapshot(str(tmp_path / "m"))
    Q = GDRecommender(3, 2, {"nbr_features": 2})
    Q.load_model_snapshot(str(tmp_path / "m"))
    assert np.array_equal(Q.svd_u, R.svd_u) and np.array_equal(Q.svd_v, R.svd_v)
    z = np.load(str(tmp_path / "m") + "_model_snapshot.npz")
    assert sorted(z.files) == ["svd_u", "svd_v"]          # keys of base.py:805-806


def test_save_state_load_state_file_set(tmp_path):
    """base.py:557-596: `_relations.mtx`, `_svd.npz` (svd_s / svd_u / svd_v), five pickled dictionaries."""
    import pickle
    import scipy.io
    from mfrec_amd.recommendation.base import BaseRecommender
    R = BaseRecommender(5, 7)
    R.set_ratings(np.array([[0, 1], [4, 6], [2, 2]]), [3.0, 5.0, 1.0])
    R.svd_u, R.svd_v = np.arange(14.0).reshape(2, 7), np.arange(10.0).reshape(2, 5)
    assert R.users_label == ["user%d" % i for i in range(5)] and R.items_index["item6"] == 6   # base.py:275-281
    R.set_user_label(0, "ann"); R.set_user_label(4, "bob"); R.set_item_label(1, "x")
    assert R.users_index["bob"] == 4 and "user4" not in R.users_index and R.items_label[1] == "x"
    R.items_index_map = {1: 1}
    base = str(tmp_path / "state")
    R.save_state(base)
    for suffix in ("_relations.mtx", "_svd.npz", "_userslabel.dat", "_usersindex.dat", "_itemslabel.dat",
                   "_itemsindex.dat", "_itemsindexmap.dat"):
        assert (tmp_path / ("state" + suffix)).exists()
    z = np.load(base + "_svd.npz", allow_pickle=True)
    assert sorted(z.files) == ["svd_s", "svd_u", "svd_v"]
    assert z["svd_s"].shape == () and z["svd_s"].item() is None           # None as numpy stores it
    assert pickle.load(open(base + "_userslabel.dat", "rb")) == R.users_label
    assert open(base + "_usersindex.dat", "rb").read(1) not in (b"\x80",)    # protocol 0: Python 2 reads it
    Q = BaseRecommender(1, 1)
    Q.load_state(base)
    assert (Q.nbr_users, Q.nbr_items) == (5, 7) and Q.svd_s is None
    assert np.array_equal(Q.svd_u, R.svd_u) and np.array_equal(Q.svd_v, R.svd_v)
    assert (Q.relationship_matrix != R.relationship_matrix).nnz == 0
    assert Q.users_index == R.users_index and Q.items_index_map == R.items_index_map
    a, b = Q.get_ratings()
    assert a.tolist() == [[0, 1], [2, 2], [4, 6]] and b.tolist() == [3.0, 1.0, 5.0]

    # a file set as the REFERENCE writes it: text-mode protocol-0 pickles, svd_s a real array
    scipy.io.mmwrite(base + "2_relations.mtx", R.relationship_matrix)
    np.savez(base + "2_svd.npz", svd_s=np.array([2.0, 1.0]), svd_u=R.svd_u, svd_v=R.svd_v)
    for suffix, obj in (("_userslabel.dat", ["ann"]), ("_usersindex.dat", {"ann": 0}), ("_itemslabel.dat", []),
                        ("_itemsindex.dat", {}), ("_itemsindexmap.dat", {})):
        with open(base + "2" + suffix, "w") as f:
            f.write(pickle.dumps(obj, protocol=0).decode("latin1"))
    Q.load_state(base + "2")
    assert Q.svd_s.tolist() == [2.0, 1.0] and Q.users_label == ["ann"]


def test_save_then_load_state_of_an_untrained_recommender(tmp_path):
    """A state saved BEFORE training holds svd_u = svd_v = None, which np.savez stores as 0-d object arrays; the reference's
    np.load round-trips them (base.py:578-596) and so must load_state."""
    from mfrec_amd.recommendation.base import BaseRecommender
    R = BaseRecommender(4, 3)
    R.set_ratings(np.array([[0, 1], [3, 2]]), [2.0, 4.0])
    assert R.svd_u is None and R.svd_v is None
    base = str(tmp_path / "untrained")
    R.save_state(base)
    Q = BaseRecommender(1, 1)
    Q.svd_u = Q.svd_v = np.zeros((2, 2))
    Q.load_state(base)
    assert Q.svd_u is None and Q.svd_v is None and Q.svd_s is None and (Q.nbr_users, Q.nbr_items) == (4, 3)
    assert (Q.relationship_matrix != R.relationship_matrix).nnz == 0


def test_files_written_by_the_reference_own_statements_load(tmp_path):
    """tests/golden/refwritten_* were written, in the build container, by the reference's OWN statements (base.py:805-806
    save_model_snapshot; base.py:559-560, the mmwrite + np.savez statements of save_state), cut from the file and executed by
    oracle/pin_topn.py (record: tests/golden/snapshot_pin.json).  mfrec_amd must read them: keys, shapes, values, the None that
    np.savez stores for svd_s.  (The five label / index pickles of a state are Python-2 text pickles the reference code cannot write
    under Python 3: hand-made here in that format.)"""
    import json
    import os
    import pickle
    import shutil
    from mfrec_amd.recommendation.base import BaseRecommender
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rec = json.load(open(os.path.join(gold, "snapshot_pin.json")))
    assert "refwritten_model_snapshot.npz" in rec["fixtures_written_by_the_reference_statements"]
    rs = np.random.RandomState(5)
    u, v = rs.normal(0, 0.1, (3, 6)), rs.normal(0, 0.1, (3, 4))
    Q = BaseRecommender(1, 1)
    Q.load_model_snapshot(os.path.join(gold, "refwritten"))
    assert np.array_equal(Q.svd_u, u) and np.array_equal(Q.svd_v, v)
    base = str(tmp_path / "s")
    shutil.copy(os.path.join(gold, "refwritten_state_relations.mtx"), base + "_relations.mtx")
    shutil.copy(os.path.join(gold, "refwritten_state_svd.npz"), base + "_svd.npz")
    for suffix, obj in (("_userslabel.dat", ["u0", "u1", "u2", "u3"]), ("_usersindex.dat", {"u0": 0}), ("_itemslabel.dat", []),
                        ("_itemsindex.dat", {}), ("_itemsindexmap.dat", {})):
        with open(base + suffix, "w") as f:
            f.write(pickle.dumps(obj, protocol=0).decode("latin1"))
    P = BaseRecommender(1, 1)
    P.load_state(base)
    assert (P.nbr_users, P.nbr_items) == (4, 6) and P.svd_s is None
    assert np.array_equal(P.svd_u, u) and np.array_equal(P.svd_v, v)
    assert P.relationship_matrix[3, 5] == 5.0 and P.relationship_matrix[2, 0] == 4.0 and P.relationship_matrix.nnz == 4
    assert P.users_label == ["u0", "u1", "u2", "u3"]


def test_balanced_order_falls_back_when_a_builder_refuses(monkeypatch):
    """get_ratings('balanced') promises GPU builder -> host builder -> plain shuffle; a builder's refusal
    arrives as MfxError (MFX_E_STATE) or MemoryError (MFX_E_NOMEM) after _shim.check()."""
    from mfrec_amd import _shim as S
    from mfrec_amd.lib import datasets as D
    from mfrec_amd.recommendation.base import BaseRecommender
    R = BaseRecommender(6, 5)
    R.set_ratings(np.array([[0, 1], [1, 1], [2, 3], [5, 4], [3, 0]]), [1.0, 2.0, 3.0, 4.0, 5.0])
    calls = []

    def gpu_refuses(*a, **k):
        calls.append("gpu")
        raise MemoryError("libmfx error -3: colour bitsets")

    def host_refuses(*a, **k):
        calls.append("host")
        raise S.MfxError("libmfx error -7: too many levels")

    monkeypatch.setattr(D, "balanced_order_gpu", gpu_refuses)
    monkeypatch.setattr(D, "balanced_order", host_refuses)
    np.random.seed(3)
    idx, r = R.get_ratings(randomize_order="balanced")
    assert calls == ["gpu", "host"]
    assert sorted(r.tolist()) == [1.0, 2.0, 3.0, 4.0, 5.0] and idx.shape == (5, 2)     # the plain shuffle


def test_bulk_set_ratings_equals_the_per_cell_loop():
    """set_ratings (array operations) leaves the matrix a loop of set_item_by_id calls (base.py:823-828) would:
    later duplicates win, cells given again are overwritten, a 0 clears its cell -- checked over several rounds."""
    from mfrec_amd.recommendation.base import BaseRecommender, Error
    rs = np.random.RandomState(3)
    for trial in range(12):
        nu, ni = rs.randint(1, 30), rs.randint(1, 30)
        a, b = BaseRecommender(nu, ni), BaseRecommender(nu, ni)
        for rnd in range(3):
            n = rs.randint(0, 200)
            idx = np.stack([rs.randint(0, nu, n), rs.randint(0, ni, n)], 1).astype(np.int32)
            r = rs.randint(0, 6, n).astype(np.float64)
            a.set_ratings(idx, r)
            for (x, y), v in zip(idx, r):
                b.set_item_by_id(int(x), int(y), v)
            assert np.array_equal(a.relationship_matrix.toarray(), b.relationship_matrix.toarray())
            (ia, ra), (ib, rb) = a.get_ratings(False), b.get_ratings(False)
            assert np.array_equal(ia, ib) and np.array_equal(ra, rb)
            assert a.get_nbr_ratings() == b.get_nbr_ratings()
    with pytest.raises(IndexError):
        BaseRecommender(3, 3).set_ratings(np.array([[3, 0]]), np.array([1.0]))
    with pytest.raises(Error):
        BaseRecommender(3, 3).set_ratings(np.array([[1, 0]]), np.array([1.0, 2.0]))


def test_label_based_ingest_follows_the_reference_container():
    """base.py:815-851,1043-1112: set_item (list of {'label', 'value'} for a user label), set_item_by_label,
    set_user_label / set_item_label keep list and dict in step, build_index rebuilds the dicts."""
    from mfrec_amd.recommendation.base import BaseRecommender
    R = BaseRecommender(3, 4)
    R.set_user_label(1, "ann")
    R.set_item_label(2, "book")
    R.set_item("ann", [{"label": "book", "value": 4}, {"label": "item0", "value": 2}])
    R.set_item_by_label("user2", "item3", 5)
    assert R.relationship_matrix.toarray().tolist() == [[0, 0, 0, 0], [2, 0, 4, 0], [0, 0, 0, 5]]
    with pytest.raises(KeyError):
        R.set_item_by_label("user1", "item0", 1)              # the old label is gone (base.py:1101)
    R.users_index, R.items_index = {}, {}
    R.build_index()
    assert R.users_index == {"user0": 0, "ann": 1, "user2": 2} and R.items_index["book"] == 2
    assert R._get_new_user_id() == 3 and R.users_label[3] == "user3"
    assert R._get_new_item_id() == 4 and R.items_label[4] == "item4"
    R.set_dimensionality(7)
    assert R.dimensionality == 7


def test_bench_self_launch_without_a_gpu_fails_loudly_and_returns_the_child_code():
    """`python bench.py --gpus 2` started plainly (no launcher, no WORLD_SIZE): bench.py starts its own two ranks.  In this
    container there is no HIP device: the ranks fail with libmfx's "no CPU fallback" error, the launcher ends the rank still
    waiting for the RCCL id after the grace period and returns a non-zero code -- and no JSON line is printed."""
    import os
    import subprocess
    import sys
    from mfrec_amd import _shim as S
    if S.device_count() > 0:
        pytest.skip("a HIP device is present: the GPU suite runs the two-rank self-launch for real")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MFX_BENCH_LAUNCH_GRACE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--users", "2000", "--items", "1000", "--nnz", "50000"], capture_output=True, text=True,
                         timeout=300, cwd=root, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "no CPU fallback" in out.stderr and "[bench launcher] child exit codes" in out.stderr


def test_load_state_refuses_a_pickled_object_in_the_svd_file(tmp_path):
    """`_svd.npz` members are never run through a general unpickle: np.savez's image of None is decoded by an unpickler that
    admits numpy's array constructors only; an object member that names any other global is refused, not executed."""
    import pickle
    import scipy.sparse as sp
    from mfrec_amd.recommendation import KMFRecommender
    R = KMFRecommender(4, 6, {"nbr_features": 3})
    R.relationship_matrix = sp.lil_matrix((4, 6))
    R._nbr_users, R._nbr_items = 4, 6
    R.svd_u, R.svd_v, R.svd_s = None, np.ones((3, 4)), None
    base = str(tmp_path / "s")
    R.save_state(base)
    Q = KMFRecommender(4, 6, {"nbr_features": 3})
    Q.load_state(base)
    assert Q.svd_u is None and Q.svd_s is None and np.array_equal(Q.svd_v, np.ones((3, 4)))
    hit = []

    class Evil(object):
        def __reduce__(self):
            return (hit.append, ("executed",))

    np.savez(base + "_svd.npz", svd_s=np.array(Evil(), dtype=object), svd_u=np.ones(2), svd_v=np.ones(2))
    with pytest.raises(pickle.UnpicklingError):
        Q.load_state(base)
    assert not hit
    np.savez(base + "_svd.npz", svd_s=np.array(7, dtype=object), svd_u=np.ones(2), svd_v=np.ones(2))
    with pytest.raises(ValueError):
        Q.load_state(base)
