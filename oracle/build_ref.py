#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds the *reference's own* Cython kernels OUTSIDE the repository.

The three hot-path modules of mlaprise/mfrec are Cython (.pyx -> C -> extension
module).  This recipe compiles them *from the sources where they lie* under
/root/reference, into $MFX_REF_OUT (default /tmp/mfx_ref) -- never into /root/repo:
the reference is Python/Cython and must not travel to the GPU box in any form (SURVEY 8(c)
usage rule), and everything under /root/repo does travel.

    cython -2 <ref>/mfrec/lib/X.pyx -o $MFX_REF_OUT/X.c
    gcc -O2 -fPIC -shared -ffp-contract=off $MFX_REF_OUT/X.c -o $MFX_REF_OUT/X.so
    rm $MFX_REF_OUT/X.c

The resulting modules are used, in the build container only, to
  * pin oracle/mfx_oracle.c (our plain-C restatement) bit-for-bit,
  * generate tests/golden/*.npz via oracle/make_golden.py, and
  * record the port-vs-reference speed ratio (tools/port_vs_reference.py ->
    profiles/r03_port_vs_reference.json).
On the GPU box neither /root/reference nor $MFX_REF_OUT exists: build() is a no-op (returns
False), load() returns None, and bench.py's cpu_baseline is the plain-C port.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("MFX_REF_OUT", "/tmp/mfx_ref")
if os.path.realpath(OUT).startswith(os.path.realpath(os.path.dirname(HERE)) + os.sep):
    raise RuntimeError("MFX_REF_OUT must lie outside the repository (the reference must not travel)")
REF = os.environ.get("MFX_REFERENCE_ROOT", "/root/reference")
MODULES = ("kmf_train", "gd_estimator", "als_implicit")


def build(force=False, verbose=True):
    src_dir = os.path.join(REF, "mfrec", "lib")
    if not os.path.isdir(src_dir):
        return False
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    inc_py = sysconfig.get_paths()["include"]
    inc_np = np.get_include()
    ext = ".so"
    for m in MODULES:
        pyx = os.path.join(src_dir, m + ".pyx")
        so = os.path.join(OUT, m + ext)
        if (not force and os.path.exists(so)
                and os.path.getmtime(so) >= os.path.getmtime(pyx)):
            continue
        c = os.path.join(OUT, m + ".c")
        cmd = [sys.executable, "-m", "cython", "-2", pyx, "-o", c]
        if verbose:
            print("[build_ref]", " ".join(cmd))
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-w",
               "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
               "-I", inc_py, "-I", inc_np, c, "-o", so]
        if verbose:
            print("[build_ref]", " ".join(cmd))
        subprocess.check_call(cmd)
        os.remove(c)
    return True


def load():
    """Import the built reference modules; returns dict name->module or None."""
    import importlib.util
    mods = {}
    for m in MODULES:
        so = os.path.join(OUT, m + ".so")
        if not os.path.exists(so):
            return None
        spec = importlib.util.spec_from_file_location(m, so)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods[m] = mod
    return mods


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("[build_ref] built" if ok else "[build_ref] reference not present; skipped")
