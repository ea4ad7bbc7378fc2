"""GPU: the N > 1 code paths of libmfx with REAL processes -- two or three ranks, each its own process with its own
HIP context, stream and communicator, all on the one GPU of the box.  RCCL refuses two ranks on one device, so the
ranks talk through tests/mock_rccl (a shared-memory stand-in with the same twelve entry points, found through
MFX_RCCL_LIB; its reductions add in rank order).  What is exercised is everything of ours that a hardware run
exercises and that the single-process tests cannot reach: the file rendezvous of the unique id, mfx_comm_init at
world > 1, the real-rank branch of the block rotation (send to g - 1 / receive from g + 1 inside one group on the
transfer stream, events both ways), its rank-ordered squared-error all-gather and the home-rank broadcasts of
get_factors, the item-delta all-reduce of the averaging mode, row-sharded ALS, and bench.py's whole N > 1 path.
Every result must equal the single-process fake world's bit for bit (which the other tests tie to the oracle).
The stand-in delivers results LATE and asynchronously in stream order (an idling kernel, then an async copy), so a
missing or wrong event between the transfer stream and the compute stream reads stale rows: with the wait on
`arrived[j]` removed from mfx_dsgd_run_kmf two of the three rotation cases fail (checked by hand, round 2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from mfrec_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NU, NI, NNZ, K, E = 3000, 1100, 120000, 16, 3
HP = (0.01, 0.1, 0.1, 0.007)


@pytest.fixture(scope="module")
def mock_lib():
    """tests/mock_rccl/mock_rccl.cpp -> build/libmock_rccl.so (hipcc; a few seconds)."""
    from mfrec_amd import _shim as S, build as B
    B.build(verbose=False)
    S.init(0)
    out = os.path.join(ROOT, "build", "libmock_rccl.so")
    src = os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.run([hipcc, "-O2", "-fPIC", "-shared", "--offload-arch=gfx950", "-o", out, src, "-lrt", "-lpthread"],
                       check=True, capture_output=True, timeout=300)
    return out


WORKER = r"""
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from mfrec_amd import _shim as S, rendezvous, synth
from mfrec_amd import sharding as SH
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode, out = sys.argv[1], sys.argv[2]
NU, NI, NNZ, K, E = %(dims)s
K = int(os.environ.get("MFX_TEST_K", K))                      # (the static resident-side epoch needs k > 16)
HP = %(hp)s
uid = None
if rank != 0:
    uid = rendezvous.exchange_unique_id(rank, world, None, timeout_s=120.0)
S.init(0)                                                     # every rank on the box's one GPU
def make_id():
    b = C.create_string_buffer(S.UNIQUE_ID_BYTES)
    S.check(S.lib().mfx_comm_unique_id(b))
    return b.raw
if rank == 0:
    uid = rendezvous.exchange_unique_id(rank, world, make_id)
S.check(S.lib().mfx_comm_init(rank, world, uid))
idx, r = synth.make_ratings(NU, NI, NNZ, seed=41)
u0, v0 = synth.init_normal(K, NI, 1), synth.init_normal(K, NU, 2)
res = {}
if mode.startswith("dsgd"):
    overlap, balanced = "overlap" in mode, "balanced" in mode
    d = SH.DsgdKMF(idx, r, NU, NI, K, rank, world, overlap=overlap, balanced=balanced,
                   item_blocks_1gpu=3 * world if "stride3" in mode else None)
    lo, hi = int(d.user_bounds[rank]), int(d.user_bounds[rank + 1])
    d.set_factors(u0, np.ascontiguousarray(v0[:, lo:hi]), np.zeros(NI), np.zeros(hi - lo))
    se = d.train(0, E, *HP)
    u, v, ib, ub = np.zeros((K, NI)), np.zeros((K, hi - lo)), np.zeros(NI), np.zeros(hi - lo)
    d.get_factors(u, v, ib, ub)
    res = dict(u=u, v=v, ib=ib, ub=ub, se=np.asarray(se), lo=lo, hi=hi, order=d.order(), resident_blocks_after=d.info()["resident_blocks"])
    d.close()
elif mode == "avg":
    s = SH.ShardedKMF(idx, r, NU, NI, K, rank, world)
    s.set_factors(u0, v0, np.zeros(NI), np.zeros(NU))
    rm, se = s.train(0, E, *HP)
    u, v, ib, ub = s.get_local()
    res = dict(u=u, v=v[:, :s.nu_local], ib=ib, ub=ub[:s.nu_local], se=np.asarray(se), lo=s.user_lo, hi=s.user_lo + s.nu_local)
    s.close()
elif mode == "avgblocks":
    s = SH.ShardedKMFBlocks(idx, r, NU, NI, K, rank, world, item_blocks=3)
    resident_before = s.info()["resident_blocks"]
    s.set_factors(u0, v0, np.zeros(NI), np.zeros(NU))
    se = s.train(0, E, *HP)
    u, v, ib, ub = s.get_local()
    res = dict(u=u, v=v[:, :s.nu_local], ib=ib, ub=ub[:s.nu_local], se=np.asarray(se), lo=s.user_lo, hi=s.user_lo + s.nu_local,
               order=s.order(), resident_blocks_after=s.info()["resident_blocks"], resident_blocks_before=resident_before)
    s.close()
elif mode == "funk":
    from mfrec_amd.plan import SgdPlan
    rs = np.random.RandomState(4)
    cells = rs.permutation(NU * NI)[:60000]                      # no duplicate (user, item) pairs
    fidx = np.stack([cells %% NU, cells // NU], 1).astype(np.int32)
    fr = rs.randint(1, 6, len(cells)).astype(np.float64)
    bounds = SH.user_blocks(fidx, NU, world)
    sidx, sr, lo, n = SH.shard(fidx, fr, bounds, rank)
    u, v = synth.init_const(5, NI), synth.init_const(5, max(n, 1))
    with SgdPlan(sidx, sr, max(n, 1), NI, 5) as p:
        p.set_factors(u, v)
        hist = p.run_funk_sharded(2, 1e9, 0.1, 0.01, 0.05, scale=1.0 / world)
        p.get_factors(u, v)
    res = dict(u=u, v=v[:, :n], hist=np.asarray(hist), lo=lo, hi=lo + n)
elif mode == "als":
    import scipy.sparse as sp
    from mfrec_amd.lib.als_implicit import als_wrmf_sharded
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    M = sp.csr_matrix((np.ones(NNZ), (idx[:, 0], idx[:, 1])), shape=(NU, NI)); M.sum_duplicates()
    views = create_bool_sparse_row(M) + create_bool_sparse_col(M)
    rs = np.random.RandomState(3)
    u, v = rs.normal(0, 0.1, (24, NI)), rs.normal(0, 0.1, (24, NU))
    als_wrmf_sharded(2, 24, u, v, None, None, *views, NU, NI)
    res = dict(u=u, v=v)
np.savez(out + ".rank%%d.npz" %% rank, **res)
S.check(S.lib().mfx_comm_barrier())
S.check(S.lib().mfx_comm_destroy())
rendezvous.cleanup(rank)
"""


def _launch(mock_lib, tmp_path, world, mode, port, env_extra=None):
    """env_extra: a dict for every rank, or a list of dicts (one per rank)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "dims": repr((NU, NI, NNZ, K, E)), "hp": repr(HP)})
    out = str(tmp_path / ("res_" + mode.replace(",", "_")))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MFX_RCCL_LIB=mock_lib, MFX_TESTING="1", MFX_K1_RES="levels")
        env.update((env_extra[rank] if isinstance(env_extra, (list, tuple)) else env_extra) or {})
        procs.append(subprocess.Popen([sys.executable, str(script), mode, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for rank, (p, o) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, "rank %d:\n%s" % (rank, o[-3000:])
    return [np.load(out + ".rank%d.npz" % rank) for rank in range(world)]


def _problem():
    idx, r = synth.make_ratings(NU, NI, NNZ, seed=41)
    return idx, r, synth.init_normal(K, NI, 1), synth.init_normal(K, NU, 2)


@pytest.mark.parametrize("world,mode,port", [(2, "dsgd", 29711), (3, "dsgd,balanced", 29712), (2, "dsgd,overlap", 29713),
                                             (2, "dsgd,stride3,balanced", 29714)])
def test_block_rotation_over_real_processes_equals_the_fake_world(mock_lib, tmp_path, world, mode, port):
    from mfrec_amd import sharding as SH
    idx, r, u0, v0 = _problem()
    res = _launch(mock_lib, tmp_path, world, mode, port)
    d = SH.DsgdKMF(idx, r, NU, NI, K, -1, world, overlap="overlap" in mode, balanced="balanced" in mode,
                   item_blocks_1gpu=3 * world if "stride3" in mode else None)
    d.set_factors(u0, v0, np.zeros(NI), np.zeros(NU))
    se = d.train(0, E, *HP)
    u, v, ib, ub = np.zeros((K, NI)), np.zeros((K, NU)), np.zeros(NI), np.zeros(NU)
    d.get_factors(u, v, ib, ub)
    order = d.order()
    d.close()
    for rank, z in enumerate(res):
        lo, hi = int(z["lo"]), int(z["hi"])
        assert np.array_equal(z["u"], u) and np.array_equal(z["ib"], ib)          # every rank ends with the whole item side
        assert np.array_equal(z["v"], v[:, lo:hi]) and np.array_equal(z["ub"], ub[lo:hi])
        # squared errors: per-rank partials added in rank order -- the same bits on every rank, and the fake world's
        # (step, rank)-ordered total up to the association
        assert np.array_equal(z["se"], res[0]["se"]) and np.allclose(z["se"], np.asarray(se), rtol=1e-13, atol=0)
    # the ranks' local orders, interleaved step by step, are the fake world's order
    assert sorted(np.concatenate([z["order"] for z in res]).tolist()) == sorted(order.tolist())


def test_block_rotation_with_static_blocks_recovers_when_one_rank_loses_a_co_residency_check(mock_lib, tmp_path):
    """Block rotation over two real ranks, every block on the static resident-side epoch (k = 32, forced) with the ring's send / recv
    kernels beside it; rank 1's 4th static launch fails its registration (test hook).  The ranks agree after the run, BOTH restore
    their snapshots, go to the level schedule and repeat the call: the fake world's bits (= the sequential loop's on the block order)."""
    from mfrec_amd import sharding as SH
    idx, r = synth.make_ratings(NU, NI, NNZ, seed=41)
    u0, v0 = synth.init_normal(32, NI, 1), synth.init_normal(32, NU, 2)
    base = {"MFX_K1_RES": "static", "MFX_TEST_K": "32", "MFX_GRAPH": "0"}
    res = _launch(mock_lib, tmp_path, 2, "dsgd,balanced", 29715, [base, dict(base, MFX_TEST_REG_FAIL="4")])
    d = SH.DsgdKMF(idx, r, NU, NI, 32, -1, 2, balanced=True)
    d.set_factors(u0, v0, np.zeros(NI), np.zeros(NU))
    se = d.train(0, E, *HP)
    u, v, ib, ub = np.zeros((32, NI)), np.zeros((32, NU)), np.zeros(NI), np.zeros(NU)
    d.get_factors(u, v, ib, ub)
    d.close()
    for z in res:
        lo, hi = int(z["lo"]), int(z["hi"])
        assert np.array_equal(z["u"], u) and np.array_equal(z["ib"], ib)
        assert np.array_equal(z["v"], v[:, lo:hi]) and np.array_equal(z["ub"], ub[lo:hi])
        assert np.allclose(z["se"], np.asarray(se), rtol=1e-12, atol=0)
        assert int(z["resident_blocks_after"]) == 0


def test_averaging_mode_over_real_processes_equals_the_fake_world(mock_lib, tmp_path):
    from mfrec_amd import sharding as SH
    idx, r, u0, v0 = _problem()
    res = _launch(mock_lib, tmp_path, 2, "avg", 29721)
    f = SH.FakeWorldKMF(idx, r, NU, NI, K, 2)
    f.set_factors(u0, v0, np.zeros(NI), np.zeros(NU))
    f.train(0, E, *HP)
    u, v, ib, ub = f.get_factors()
    f.close()
    for z in res:
        lo, hi = int(z["lo"]), int(z["hi"])
        assert np.array_equal(z["u"], u) and np.array_equal(z["ib"], ib)
        assert np.array_equal(z["v"], v[:, lo:hi]) and np.array_equal(z["ub"], ub[lo:hi])


def _check_avgblocks(res, k):
    """The CPU simulation of the averaging scheme on the ranks' block orders: per epoch every shard runs the oracle's epoch
    (kmf_train.pyx:243-270) on its ratings in the order the rank reports, from the common item state; the deltas are added in rank
    order and combined with scale 1/2 (sharding.combine)."""
    from mfrec_amd import sharding as SH
    from oracle import oracle as O
    idx, r = synth.make_ratings(NU, NI, NNZ, seed=41)
    u0, v0 = synth.init_normal(k, NI, 1), synth.init_normal(k, NU, 2)
    bounds = SH.user_blocks(idx, NU, 2)
    shards = [SH.shard(idx, r, bounds, g) for g in range(2)]
    u, ib, v, ub = u0.copy(), np.zeros(NI), v0.copy(), np.zeros(NU)
    for _ in range(E):
        du, dib = np.zeros_like(u), np.zeros_like(ib)
        for g, (sidx, sr, lo, n) in enumerate(shards):
            perm = res[g]["order"]
            assert sorted(perm.tolist()) == list(range(sr.shape[0]))
            ug, ibg = u.copy(), ib.copy()
            vg, ubg = np.ascontiguousarray(v[:, lo:lo + n]), np.ascontiguousarray(ub[lo:lo + n])
            O.train_linear_kernel(1, k, 0.1, HP[0], 0, 0, HP[1], HP[2], HP[3], 0.0, ug, vg, np.ascontiguousarray(sidx[perm]),
                                  np.ascontiguousarray(sr[perm]), ibg, ubg)
            v[:, lo:lo + n] = vg; ub[lo:lo + n] = ubg
            du += ug - u
            dib += ibg - ib
        u, ib = SH.combine(u, du, 0.5), SH.combine(ib, dib, 0.5)
    for z in res:
        lo, hi = int(z["lo"]), int(z["hi"])
        assert np.array_equal(z["u"], u) and np.array_equal(z["ib"], ib)
        assert np.array_equal(z["v"], v[:, lo:hi]) and np.array_equal(z["ub"], ub[lo:hi])


def test_averaging_mode_on_the_block_order_over_real_processes_equals_its_simulation(mock_lib, tmp_path):
    """ShardedKMFBlocks (mfx_dsgd_run_kmf_averaged): the averaging scheme with every shard's local epoch on ITS block order and one
    all-reduce per item block -- on the TRANSFER stream, under the next block's training (round 4) --, two real ranks, against the
    CPU simulation of the same scheme.  The stand-in delivers late: a missing event between the two streams reads stale rows."""
    res = _launch(mock_lib, tmp_path, 2, "avgblocks", 29761)
    _check_avgblocks(res, K)


def test_averaging_mode_with_static_blocks_and_a_rank_that_loses_a_co_residency_check(mock_lib, tmp_path):
    """The same with every block on the STATIC resident-side epoch (k = 32, forced), the all-reduce kernels of block j running beside
    block j + 1's persistent launch.  Rank 0's 5th static launch fails its registration (test hook: what a GPU shared with another
    process does): that block epoch and the plan's later ones modify nothing while the other blocks go on -- the ranks agree on the
    failure after the run, restore the snapshot taken when the call began, move every block to the level schedule and run the call
    again.  Same bits as the simulation; and without a failure (rank hooks off) the static blocks give them too."""
    for port, extra in ((29762, {"MFX_K1_RES": "static", "MFX_TEST_K": "32"}),
                        (29763, [{"MFX_K1_RES": "static", "MFX_TEST_K": "32", "MFX_GRAPH": "0", "MFX_TEST_REG_FAIL": "5"},
                                 {"MFX_K1_RES": "static", "MFX_TEST_K": "32", "MFX_GRAPH": "0"}])):
        res = _launch(mock_lib, tmp_path, 2, "avgblocks", port, extra)
        _check_avgblocks(res, 32)
        if isinstance(extra, list):
            assert int(res[0]["resident_blocks_after"]) == 0 and int(res[1]["resident_blocks_after"]) == 0   # BOTH ranks went to levels
        else:
            # (two processes share the card here: a REAL lost check can happen -- it did, once, in round 4 -- and is recovered the
            #  same way; all blocks or none)
            assert int(res[0]["resident_blocks_before"]) == 3 and int(res[0]["resident_blocks_after"]) in (0, 3)
            assert int(res[0]["resident_blocks_after"]) == int(res[1]["resident_blocks_after"])


def test_sharded_funk_pass_over_real_processes_equals_its_simulation(mock_lib, tmp_path):
    """The per-(feature, epoch) delta / all-reduce / combine of the sharded Funk trainer with two real ranks against
    its CPU simulation: every shard runs the oracle's epoch from the common item vector, the deltas are added in
    rank order and combined with scale 1/2 (gd_estimator.pyx:742-777 per shard; sharding.combine)."""
    from mfrec_amd import sharding as SH
    from oracle import oracle as O
    world, k, E = 2, 5, 2
    res = _launch(mock_lib, tmp_path, world, "funk", 29751)
    rs = np.random.RandomState(4)
    cells = rs.permutation(NU * NI)[:60000]
    fidx = np.stack([cells % NU, cells // NU], 1).astype(np.int32)
    fr = rs.randint(1, 6, len(cells)).astype(np.float64)
    bounds = SH.user_blocks(fidx, NU, world)
    shards = [SH.shard(fidx, fr, bounds, g) for g in range(world)]
    U = synth.init_const(k, NI)
    V = [synth.init_const(k, max(n, 1)) for _, _, _, n in shards]
    cache = [np.zeros(max(n, 1) * NI) for _, _, _, n in shards]
    hist = []
    for f in range(k):
        for _ in range(E):
            start = U[f].copy()
            dsum, se = None, 0.0
            for g, (sidx, sr, lo, n) in enumerate(shards):
                Ug = U.copy()
                rm = O.estimator_subloop(f, 1, 0.0, k, 0.1, 0.01, 0.05, Ug, V[g], sidx, sr, cache[g], max(n, 1), NI)
                se += rm * rm * len(sr)
                d = Ug[f] - start
                dsum = d if dsum is None else dsum + d
            U[f] = SH.combine(start, dsum, 1.0 / world)
            hist.append(np.sqrt(se / len(fr)))
        for g, (sidx, sr, lo, n) in enumerate(shards):
            O.predictor_subloop(f, 1, k, 0.1, U, V[g], sidx, sr, cache[g], max(n, 1), NI)
    for g, z in enumerate(res):
        n = shards[g][3]
        assert np.array_equal(z["u"], U) and np.array_equal(z["v"], V[g][:, :n])
        np.testing.assert_allclose(z["hist"][:len(hist)], hist, rtol=1e-11)


def test_row_sharded_als_over_real_processes_equals_one_process(mock_lib, tmp_path):
    import scipy.sparse as sp
    from mfrec_amd.lib.als_implicit import als_wrmf
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    idx, r, _, _ = _problem()
    res = _launch(mock_lib, tmp_path, 2, "als", 29731)
    M = sp.csr_matrix((np.ones(NNZ), (idx[:, 0], idx[:, 1])), shape=(NU, NI)); M.sum_duplicates()
    views = create_bool_sparse_row(M) + create_bool_sparse_col(M)
    rs = np.random.RandomState(3)
    u, v = rs.normal(0, 0.1, (24, NI)), rs.normal(0, 0.1, (24, NU))
    als_wrmf(2, 24, u, v, None, None, *views, NU, NI)
    for z in res:
        assert np.allclose(z["u"], u, rtol=1e-9, atol=1e-12) and np.allclose(z["v"], v, rtol=1e-9, atol=1e-12)
    assert np.array_equal(res[0]["u"], res[1]["u"]) and np.array_equal(res[0]["v"], res[1]["v"])


def test_bench_two_ranks_prints_one_line_with_both_modes(mock_lib, tmp_path):
    """bench.py exactly as the driver starts it for N = 2 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), both ranks on device 0, small problem."""
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29741", MFX_RCCL_LIB=mock_lib, MFX_TESTING="1", MFX_K1_RES="levels", MFX_BENCH_DEVICE="0", MFX_BENCH_DSGD_TIMEOUT="200")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                                       "--warmup", "1", "--users", "60000", "--items", "20000", "--nnz", "3000000"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=420) for p in procs]
    for rank, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d:\n%s" % (rank, e[-3000:])
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "users sharded over 2 GPUs" in d["config"]["workload"]
    assert d["also"]["dsgd_exact"]["value"] > 0 and "error" not in d["also"]["dsgd_exact"]


def test_bench_plain_invocation_launches_its_own_ranks(mock_lib):
    """`python bench.py --gpus 2 ...` with NO launcher (no WORLD_SIZE in the environment -- the shape of the driver's N = 1 command):
    bench.py starts its own two ranks before anything touches the GPU, rank 0 prints the one JSON line, the exit code is the worst
    child's.  Both ranks on device 0 through the stand-in; the static resident-side epoch is left ON (two grids share the GPU: a lost
    co-residency check is recovered inside the library, never an error)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MFX_RCCL_LIB=mock_lib, MFX_TESTING="1", MFX_BENCH_DEVICE="0", MFX_BENCH_DSGD_TIMEOUT="200")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--users", "60000", "--items", "20000", "--nnz", "3000000"], env=env, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "users sharded over 2 GPUs" in d["config"]["workload"]
    assert d["also"]["dsgd_exact"]["value"] > 0 and "error" not in d["also"]["dsgd_exact"]
    # the kernel named in the roofline is the one the epoch ran on
    assert ("k1_res" in d["roofline"]["kernel"]) == (d["config"]["epoch_kernel"].startswith("static"))
    # the line carries its own denominator: the SAME job on rank 0's one GPU, timed in the same run (a driver that divides the
    # N-GPU value by the N = 1 line's would compare cfg 3 with cfg 2)
    one = d["one_gpu_same_workload"]
    assert "error" not in one and one["value"] > 0 and one["unit"] == "rating-updates/s" and one["ms_per_step"] > 0
    assert abs(d["speedup_vs_one_gpu"] - d["value"] / one["value"]) < 1e-9 * d["speedup_vs_one_gpu"]
    assert abs(d["also"]["dsgd_exact"]["speedup_vs_one_gpu"] - d["also"]["dsgd_exact"]["value"] / one["value"]) < 1e-6
    # `value` is the averaging mode: its workload string must not claim the sequential loop's bits (also.dsgd_exact does)
    assert "bit-identical" not in d["config"]["workload"] and "AVERAGING" in d["config"]["workload"]
    assert "bit-identical" in d["also"]["dsgd_exact"]["workload"]
