"""Out-of-band bootstrap for the RCCL communicator on ONE node, without torch.

`python -m torch.distributed.run` is only the launcher here: it exports RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT and forks one process per GPU.  Importing torch in
those processes would load PyTorch's *bundled* ROCm runtime next to the system one libmfx
links against (two HSA instances in one process: RCCL then fails with "no ROCm-capable
device"), so the 128-byte RCCL unique id is passed through a file that rank 0 publishes
atomically and the other ranks poll for.  All ranks share the launcher as parent process,
which makes (parent pid, MASTER_PORT) a per-run key.
"""
import os
import shutil
import tempfile
import time


def _dir():
    key = "mfx_rdzv_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
    return os.path.join(tempfile.gettempdir(), key)


def exchange_unique_id(rank, world, make_id, timeout_s=300.0):
    """rank 0 calls make_id() -> bytes and publishes it; every rank returns the same bytes."""
    if world <= 1:
        return make_id()
    d = _dir()
    path = os.path.join(d, "unique_id.bin")
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        uid = make_id()
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)                     # atomic publish
        return uid
    t0 = time.time()
    # a stale file of an earlier run with the same key is removed by rank 0 first; accept the
    # file only once it is younger than this process
    born = time.time() - 600.0
    try:
        born = os.stat("/proc/self").st_ctime
    except OSError:
        pass
    while True:
        try:
            st = os.stat(path)
            if st.st_size >= 128 and st.st_mtime >= born - 5.0:
                with open(path, "rb") as f:
                    return f.read()
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError("rank %d: no RCCL unique id at %s after %.0f s" % (rank, path, timeout_s))
        time.sleep(0.01)


def cleanup(rank):
    if rank == 0:
        shutil.rmtree(_dir(), ignore_errors=True)
