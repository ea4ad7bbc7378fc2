"""Out-of-band bootstrap for the RCCL communicator on ONE node, without torch.

`python -m torch.distributed.run` is only the launcher here: it exports RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT and forks one process per GPU.  Importing torch in
those processes would load PyTorch's *bundled* ROCm runtime next to the system one libmfx
links against (two HSA instances in one process: RCCL then fails with "no ROCm-capable
device"), so the 128-byte RCCL unique id is passed through a file that rank 0 publishes
atomically and the other ranks poll for.

The directory is private to the user (mode 0700, ownership and type checked with lstat: a
pre-planted symlink or a directory of another user is refused) and keyed by everything that
tells one launch attempt from another: the launcher's pid (all ranks share it as parent),
MASTER_PORT, torchelastic's run id and its restart count -- a restarted worker group never
picks up the previous attempt's id.
"""
import os
import shutil
import stat
import tempfile
import time


def _key():
    return "mfx_rdzv_u%d_p%d_%s_%s_r%s" % (
        os.getuid(), os.getppid(), os.environ.get("MASTER_PORT", "0"),
        "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum())[:32],
        os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))


def _base():
    # XDG_RUNTIME_DIR is per user and not world-writable where it exists
    b = os.environ.get("XDG_RUNTIME_DIR")
    return b if b and os.path.isdir(b) and os.access(b, os.W_OK) else tempfile.gettempdir()


def _dir():
    return os.path.join(_base(), _key())


def _check_private(d):
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("rendezvous directory %s is not a private directory of this user" % d)


def exchange_unique_id(rank, world, make_id, timeout_s=300.0):
    """rank 0 calls make_id() -> bytes and publishes it; every rank returns the same bytes."""
    if world <= 1:
        return make_id()
    d = _dir()
    path = os.path.join(d, "unique_id.bin")
    if rank == 0:
        if os.path.lexists(d):
            if os.path.islink(d):
                os.unlink(d)
            else:
                shutil.rmtree(d, ignore_errors=True)
        os.mkdir(d, 0o700)                        # fails if somebody re-created it meanwhile
        _check_private(d)
        uid = make_id()
        tmp = path + ".tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
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
            _check_private(d)
            st = os.lstat(path)
            if stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid() and st.st_size >= 128 \
                    and st.st_mtime >= born - 5.0:
                with open(path, "rb") as f:
                    return f.read()
        except (OSError, RuntimeError):
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError("rank %d: no RCCL unique id at %s after %.0f s" % (rank, path, timeout_s))
        time.sleep(0.01)


def cleanup(rank):
    if rank == 0:
        shutil.rmtree(_dir(), ignore_errors=True)
