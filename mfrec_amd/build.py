"""Build mfrec_amd/libmfx.so in-tree with hipcc for gfx950 (python -m mfrec_amd.build)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmfx.so")
# -ffp-contract=off: the exactness contract forbids fusing a*b+c (csrc/mfx_sgd_kernels.h)
FLAGS = ["-O3", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
         "-Wall", "-Wno-unused-function", "-x", "hip"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def build(force=False, verbose=True):
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(os.path.dirname(HERE), "include", "mfx.h")]
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)):
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    extra = os.environ.get("MFX_EXTRA_FLAGS", "").split()      # diagnostic builds only
    cmd = [hipcc] + FLAGS + extra + srcs + ["-o", OUT if not extra else OUT.replace("libmfx.so", "libmfx_diag.so"), "-ldl", "-lpthread"]
    if verbose:
        print("[mfrec_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
