"""CPU: the hand-written f64 DPP instructions of the ALS kernels (inline asm, invisible to the
compiler's hazard recogniser) must never read a VGPR written by the two preceding instructions
without an s_nop.  Cross-compiles mfx_als.hip for gfx950 and scans the ISA (no GPU needed)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_dpp_hazard_lint():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazard.py")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 hazards" in out.stdout and "0 without a trailing s_nop" in out.stdout
    import re
    n_dpp = int(re.search(r"checked (\d+) DPP instructions", out.stdout).group(1))
    n_st = int(re.search(r"checked (\d+) wide asm stores", out.stdout).group(1))
    assert n_dpp > 1000 and n_st > 50          # the lint really saw the hand-written instructions


def test_lint_detects_a_planted_hazard():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dpp_hazard as L
    ok = ["v_mul_f64 v[2:3], v[4:5], v[6:7]", "v_add_f64 v[8:9], v[8:9], v[4:5]", "v_add_f64 v[10:11], v[8:9], v[4:5]",
          "v_fmac_f64_dpp v[0:1], -v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf"]
    assert L.scan(ok, verbose=False) == (1, 0)
    bad = ["v_mul_f64 v[2:3], v[4:5], v[6:7]",
           "v_fmac_f64_dpp v[0:1], -v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf"]
    assert L.scan(bad, verbose=False) == (1, 1)
    cmpx = ["v_cmpx_lt_i32_e32 15, v0", "v_mul_f64 v[8:9], v[4:5], v[6:7]",
            "v_fmac_f64_dpp v[0:1], -v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf"]
    assert L.scan(cmpx, verbose=False) == (1, 1)
    nop = ["v_mul_f64 v[2:3], v[4:5], v[6:7]", "s_nop 1",
           "v_fmac_f64_dpp v[0:1], -v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf"]
    assert L.scan(nop, verbose=False) == (1, 0)


def test_lint_detects_a_wide_asm_store_without_nop():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dpp_hazard as L
    bad = [";;#ASMSTART", "global_store_dwordx4 v[0:1], v[2:5], off sc0 sc1", ";;#ASMEND", "v_mul_f64 v[2:3], v[6:7], v[8:9]"]
    assert L.scan_asm_stores(bad) == (1, 1)
    ok = [";;#ASMSTART", "global_store_dwordx4 v[0:1], v[2:5], off sc0 sc1", "s_nop 1", ";;#ASMEND"]
    assert L.scan_asm_stores(ok) == (1, 0)


def test_lint_is_conservative_at_labels_and_rejects_s_nop_0():
    """(2) control flow -- a hand-written DPP instruction right behind a label or a branch cannot
    rely on what the linear scan saw before it; (3) `s_nop 0` is ONE wait state, a wide store needs two."""
    from mfrec_amd import isa_lint as L
    dpp = "v_fmac_f64_dpp v[0:1], -v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf"
    at_label = ["v_add_f64 v[8:9], v[8:9], v[4:5]", ".LBB0_3:", ";;#ASMSTART", dpp, ";;#ASMEND"]
    assert L.scan(at_label, verbose=False) == (1, 1)
    after_branch = ["s_cbranch_vccnz .LBB0_7", ";;#ASMSTART", dpp, ";;#ASMEND"]
    assert L.scan(after_branch, verbose=False) == (1, 1)
    padded = [".LBB0_3:", ";;#ASMSTART", "s_nop 1", dpp, ";;#ASMEND"]
    assert L.scan(padded, verbose=False) == (1, 0)
    compiler_dpp = [".LBB0_3:", "v_mov_b32_dpp v1, v2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"]
    assert L.scan(compiler_dpp, verbose=False) == (1, 0)          # the hazard recogniser's own business
    nop0 = [";;#ASMSTART", "global_store_dwordx4 v[0:1], v[2:5], off sc0 sc1", "s_nop 0", ";;#ASMEND"]
    assert L.scan_asm_stores(nop0) == (1, 1)
    drained = [";;#ASMSTART", "global_store_dwordx4 v[6:7], v[2:5] off sc1", "s_waitcnt vmcnt(0)", ";;#ASMEND"]
    assert L.scan_asm_stores(drained) == (1, 0)                   # rocprim's look-back stores: completed, hence safe


def test_the_build_lints_what_it_links():
    """the lint is part of mfrec_amd.build -- same flags, the objects that are linked."""
    import inspect
    from mfrec_amd import build as B
    src = inspect.getsource(B.build)
    assert "-save-temps=obj" in src and "lint(srcs, isa_of" in src
    assert "ISA hazard lint failed" in inspect.getsource(B.lint)
