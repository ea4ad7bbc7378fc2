"""Build mfrec_amd/libmfx.so in-tree with hipcc for gfx950 (python -m mfrec_amd.build).

Every source is compiled to its own object (in parallel, cached by mtime under build/obj) with
-save-temps, the gfx950 assembly of exactly those objects is linted for the hazards the compiler cannot
see inside inline asm (mfrec_amd/isa_lint.py) -- a hazard FAILS the build -- and the objects are linked
into the shared library."""
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


def lint(srcs, isa_of, verbose=True):
    """ISA hazard lint of what is about to be linked; raises on any hazard."""
    from . import isa_lint
    total = {"dpp": 0, "dpp_hazards": 0, "wide_asm_stores": 0, "store_hazards": 0, "asm_loads": 0, "load_hazards": 0}
    for src in srcs:
        path = isa_of(src)
        if not os.path.exists(path):
            raise RuntimeError("no device assembly for %s (expected %s)" % (src, path))
        r = isa_lint.lint_file(path, verbose=True)
        for k in total:
            total[k] += r[k]
    if verbose:
        print("[mfrec_amd.build] ISA lint: %(dpp)d DPP instructions, %(dpp_hazards)d hazards; "
              "%(wide_asm_stores)d wide asm stores, %(store_hazards)d without s_nop >= 1; "
              "%(asm_loads)d asm loads, %(load_hazards)d named in flight" % total, flush=True)
    if total["dpp_hazards"] or total["store_hazards"] or total["load_hazards"]:
        raise RuntimeError("ISA hazard lint failed: %r" % (total,))
    return total


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

    def isa_of(src):
        stem = os.path.splitext(os.path.basename(src))[0]
        return os.path.join(objdir, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if (not force and os.path.exists(obj) and os.path.exists(isa_of(src))
                and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time)):
            return obj
        # -save-temps=obj: the device assembly of THIS object, with THESE flags, lands next to it
        cmd = [hipcc] + CFLAGS + extra + ["-save-temps=obj", "-c", src, "-o", obj]
        if verbose:
            print("[mfrec_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=objdir)
        # keep the object and the device assembly (the lint's input); the other temporaries (preprocessed
        # sources, bitcode, host assembly) are bulk that would travel with every snapshot
        stem = os.path.splitext(os.path.basename(src))[0]
        for f in os.listdir(objdir):
            if f.startswith(stem + "-") and not f.endswith("gfx950.s"):
                os.remove(os.path.join(objdir, f))
            elif f.startswith(os.path.basename(src) + "-") and f.endswith(".hipfb"):
                os.remove(os.path.join(objdir, f))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1, 8)) as ex:
        objs = list(ex.map(compile_one, srcs))
    lint(srcs, isa_of, verbose)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-o", out, "-ldl", "-lpthread"]
    if verbose:
        print("[mfrec_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
