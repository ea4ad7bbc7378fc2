"""Build mfrec_amd/libmfx.so in-tree with hipcc for gfx950 (python -m mfrec_amd.build).

Every source is compiled to its own object (in parallel, cached by mtime under build/obj) and the
objects are linked into the shared library."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmfx.so")
# -ffp-contract=off: the exactness contract forbids fusing a*b+c (csrc/mfx_sgd_kernels.h)
CFLAGS = ["-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
          "-Wall", "-Wno-unused-function", "-x", "hip"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def build(force=False, verbose=True):
    srcs = sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "mfx.h")]
    extra = os.environ.get("MFX_EXTRA_FLAGS", "").split()      # diagnostic builds only
    out = OUT if not extra else OUT.replace("libmfx.so", "libmfx_diag.so")
    if (not force and os.path.exists(out)
            and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in srcs + hdrs)):
        return out
    hipcc = os.environ.get("HIPCC", "hipcc")
    objdir = os.path.join(ROOT, "build", "obj_diag" if extra else "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time)):
            return obj
        cmd = [hipcc] + CFLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print("[mfrec_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1, 8)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-o", out, "-ldl", "-lpthread"]
    if verbose:
        print("[mfrec_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
