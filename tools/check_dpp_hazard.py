#!/usr/bin/env python3
"""Command-line face of mfrec_amd/isa_lint.py: (re)build libmfx.so's objects with the build's own flags
(-save-temps) and lint the gfx950 assembly that is linked.  The same lint runs inside every
`python -m mfrec_amd.build`; a hazard fails the build.  Usage: python tools/check_dpp_hazard.py"""
import glob
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from mfrec_amd import build as B          # noqa: E402
from mfrec_amd import isa_lint            # noqa: E402
from mfrec_amd.isa_lint import regs, scan, scan_asm_stores   # noqa: E402,F401  (re-exported for the tests)


def main():
    B.build(verbose=False)                # compiles what is stale, lints, links (raises on a hazard)
    objdir = os.path.join(root, "build", "obj")
    tot = {"dpp": 0, "dpp_hazards": 0, "wide_asm_stores": 0, "store_hazards": 0}
    for path in sorted(glob.glob(os.path.join(objdir, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        r = isa_lint.lint_file(path)
        print("%-60s %s" % (os.path.basename(path), r))
        for k in tot:
            tot[k] += r[k]
    print("checked %d wide asm stores, %d without a trailing s_nop >= 1" % (tot["wide_asm_stores"], tot["store_hazards"]))
    print("checked %d DPP instructions, %d hazards" % (tot["dpp"], tot["dpp_hazards"]))
    return 1 if (tot["dpp_hazards"] or tot["store_hazards"] or tot["wide_asm_stores"] == 0) else 0


if __name__ == "__main__":
    sys.exit(main())
