"""CPU: the multi-GPU scheme's host logic and arithmetic.
  * partitioning properties;
  * world_size-2 gloo run (two real processes, torch.distributed on 127.0.0.1) of
    sharding.epoch_loop == the single-process G-rank simulation, bit for bit;
  * G = 1 == the sequential oracle."""
import os
import socket
import subprocess
import sys

import numpy as np

from mfrec_amd import sharding as SH
from mfrec_amd import synth
from oracle import oracle as O
from oracle import sharded_sim as SIM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NU, NI, NNZ, K, E = 300, 120, 9000, 8, 3
HP = (0.01, 0.1, 0.1, 0.007)


def _problem():
    idx, r = synth.make_ratings(NU, NI, NNZ, seed=77)
    return idx, r, synth.init_normal(K, NI, 1), synth.init_normal(K, NU, 2)


def test_user_blocks_balance_and_cover():
    idx, r, _, _ = _problem()
    for world in (1, 2, 3, 8):
        b = SH.user_blocks(idx, NU, world)
        assert b[0] == 0 and b[-1] == NU and np.all(np.diff(b) >= 0)
        sizes = []
        seen = 0
        for g in range(world):
            si, sr, lo, n = SH.shard(idx, r, b, g)
            assert si[:, 0].min(initial=0) >= 0 and si[:, 0].max(initial=-1) < max(n, 1)
            # order inside the shard is the original relative order
            sel = (idx[:, 0] >= lo) & (idx[:, 0] < lo + n)
            assert np.array_equal(sr, r[sel])
            sizes.append(len(sr)); seen += len(sr)
        assert seen == NNZ
        assert max(sizes) - min(sizes) <= 2 * (NNZ // NU + 60)      # balanced by nnz


def test_one_shard_is_the_oracle():
    idx, r, u0, v0 = _problem()
    u, v, ib, ub = SIM.simulate(0, 1, E, K, *HP, u0, v0, np.zeros(NI), np.zeros(NU), idx, r)
    u2, v2, ib2, ub2 = u0.copy(), v0.copy(), np.zeros(NI), np.zeros(NU)
    O.train_linear_kernel(E, K, 0.1, HP[0], 0, 0, HP[1], HP[2], HP[3], 0.0, u2, v2, idx, r, ib2, ub2)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
    assert np.array_equal(ib, ib2) and np.array_equal(ub, ub2)


def test_sharded_rmse_gap_is_reported_not_hidden():
    idx, r, u0, v0 = _problem()
    base = SIM.simulate(0, 1, E, K, *HP, u0, v0, np.zeros(NI), np.zeros(NU), idx, r)
    r1 = O.rmse(O.RMSE_KMF_LINEAR, base[0], base[1], idx, r, base[2], base[3])
    r0 = O.rmse(O.RMSE_KMF_LINEAR, u0, v0, idx, r, np.zeros(NI), np.zeros(NU))
    assert r1 < r0
    gaps = {}
    for world in (2, 8):
        for scale in (None, 1.0):            # parameter averaging / delta sum
            s = SIM.simulate(0, world, E, K, *HP, u0, v0, np.zeros(NI), np.zeros(NU), idx, r,
                             scale=scale)
            rg = O.rmse(O.RMSE_KMF_LINEAR, s[0], s[1], idx, r, s[2], s[3])
            gaps[(world, scale)] = rg - r1
            # a different trajectory (averaging dilutes the item updates by 1/G), but it
            # still trains: better than the untrained model, and the gap grows with G
            assert np.isfinite(rg) and rg < r0
    assert abs(gaps[(2, None)]) <= abs(gaps[(8, None)]) + 1e-12


WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from mfrec_amd import sharding as SH, synth
from oracle import oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
NU, NI, NNZ, K, E = %(dims)s
idx, r = synth.make_ratings(NU, NI, NNZ, seed=77)
u = synth.init_normal(K, NI, 1); v0 = synth.init_normal(K, NU, 2)
bounds = SH.user_blocks(idx, NU, world)
si, sr, lo, n = SH.shard(idx, r, bounds, rank)
v = np.ascontiguousarray(v0[:, lo:lo + n]); ib = np.zeros(NI); ub = np.zeros(n)
def local_epoch():
    O.train_linear_kernel(1, K, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, si, sr, ib, ub)
def allreduce_sum(a):
    # gather-then-sum in rank order: the same association as the simulation's loop
    t = torch.from_numpy(np.ascontiguousarray(a))
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    s = np.zeros_like(a)
    for p in parts:
        s += p.numpy()
    return s
SH.epoch_loop(E, u, ib, local_epoch, allreduce_sum, world, 1.0 / world)
np.savez(%(out)r %% rank, u=u, v=v, ib=ib, ub=ub, lo=lo, n=n)
dist.destroy_process_group()
"""


def test_two_process_gloo_equals_simulation(tmp_path):
    idx, r, u0, v0 = _problem()
    sim = SIM.simulate(0, 2, E, K, *HP, u0, v0, np.zeros(NI), np.zeros(NU), idx, r)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank%d.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "dims": (NU, NI, NNZ, K, E), "out": out})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    for rank in range(2):
        z = np.load(out % rank)
        assert np.array_equal(z["u"], sim[0]) and np.array_equal(z["ib"], sim[2])
        lo, n = int(z["lo"]), int(z["n"])
        assert np.array_equal(z["v"], sim[1][:, lo:lo + n])
        assert np.array_equal(z["ub"], sim[3][lo:lo + n])


RDZV_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
from mfrec_amd import rendezvous
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
uid = rendezvous.exchange_unique_id(rank, world, lambda: bytes(range(128)) if rank == 0 else b"", timeout_s=60)
open(%(out)r %% rank, "wb").write(uid)
"""


def test_file_rendezvous_delivers_the_unique_id(tmp_path):
    """bench.py's torch-free bootstrap: rank 0 publishes 128 bytes, the other ranks (same parent
    process, as under torch.distributed.run) receive exactly those bytes."""
    out = str(tmp_path / "uid%d.bin")
    script = tmp_path / "rdzv_worker.py"
    script.write_text(RDZV_WORKER % {"root": ROOT, "out": out})
    procs = []
    for rank in (2, 1, 0):                    # rank 0 last: the others must wait for the publish
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_PORT="47123")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=120) == 0
    for rank in range(3):
        assert open(out % rank, "rb").read() == bytes(range(128))
    # cleanup is keyed by (user, parent pid, port, run id, restart count): from this process the parent pid
    # differs, so remove the workers' directory by pattern
    import glob, shutil, tempfile
    from mfrec_amd import rendezvous
    hits = glob.glob(os.path.join(rendezvous._base(), "mfx_rdzv_u%d_p%d_47123_*" % (os.getuid(), os.getpid())))
    assert len(hits) == 1
    st = os.lstat(hits[0])
    assert (st.st_mode & 0o077) == 0 and st.st_uid == os.getuid()          # private to the user
    for h in hits:
        shutil.rmtree(h, ignore_errors=True)


def test_rendezvous_refuses_a_planted_directory(tmp_path, monkeypatch):
    """A directory at the rendezvous path that is not a private directory of this user (here: group/world
    accessible) is not trusted by the waiting ranks."""
    import pytest
    from mfrec_amd import rendezvous
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "47999")
    d = rendezvous._dir()
    os.mkdir(d, 0o755)
    with open(os.path.join(d, "unique_id.bin"), "wb") as f:
        f.write(b"x" * 128)
    with pytest.raises(TimeoutError):
        rendezvous.exchange_unique_id(1, 2, None, timeout_s=0.5)
