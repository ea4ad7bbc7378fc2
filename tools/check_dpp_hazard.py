#!/usr/bin/env python3
"""Lint the generated gfx950 ISA of mfx_als.hip: a DPP instruction must not read (as its DPP
source, src0) a VGPR that one of the two preceding instructions wrote, unless an s_nop >= 1
separates them (VALU-write -> DPP-read needs 2 wait states and the compiler cannot see inside
inline asm).  Usage: python tools/check_dpp_hazard.py   (compiles to a temp dir with -save-temps)"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(lines, verbose=True):
    """-> (number of DPP instructions seen, number of hazards)"""
    hist_ops = []
    hist, bad, n = [], 0, 0          # hist: (VGPRs written, wait states if s_nop) of preceding instructions
    for line in lines:
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(";")[0].split(",")]
        hist_ops.append(t)
        if op.startswith("s_nop"):
            hist.append((set(), int(ops[0]) + 1))
            continue
        if "_dpp" in op and len(ops) >= 2:
            n += 1
            # VALU write of EXEC (v_cmpx*) needs 5 wait states before a DPP instruction
            recent, ws = list(hist_ops[-6:]), 0
            for prev in reversed(recent):
                if prev.startswith("s_nop"):
                    ws += int(prev.split()[1]) + 1 if len(prev.split()) > 1 else 1
                    continue
                if ws >= 5:
                    break
                if prev.startswith("v_cmpx"):
                    bad += 1
                    if verbose:
                        print("HAZARD (v_cmpx -> DPP):", t)
                    break
                ws += 1
            src0 = regs(ops[1].split()[0])
            wait = 0
            for w, nop in reversed(hist[-4:]):
                if nop:
                    wait += nop
                    continue
                if wait >= 2:
                    break
                if w & src0:
                    bad += 1
                    if verbose:
                        print("HAZARD:", t)
                    break
                wait += 1
        w = regs(ops[0].split()[0]) if op.startswith(("v_", "ds_read", "global_load", "buffer_load")) and ops else set()
        hist.append((w, 0))
    return n, bad


def scan_asm_stores(lines):
    """-> (asm stores of more than 8 bytes, those NOT followed by an s_nop inside their asm block).
    Such a store reads its data VGPRs after issue: the next instruction must not be a VALU write to them,
    and only the compiler's own stores get that wait state automatically."""
    n = bad = 0
    in_asm = False
    pending = False
    for line in lines:
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm, pending = True, False
        elif t.startswith(";;#ASMEND"):
            if pending:
                bad += 1
            in_asm = pending = False
        elif in_asm and t:
            if t.startswith(("global_store_dwordx4", "global_store_dwordx3", "buffer_store_dwordx4", "buffer_store_dwordx3")):
                if pending:
                    bad += 1
                n += 1
                pending = True
            elif t.startswith("s_nop"):
                pending = False
    return n, bad


def compile_isa(src):
    tmp = tempfile.mkdtemp()
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                           "-I" + root + "/include", "-I" + root + "/mfrec_amd/csrc", "-c",
                           root + "/mfrec_amd/csrc/" + src, "-save-temps", "-o", "x.o"], cwd=tmp)
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    return open(os.path.join(tmp, asm)).readlines()


def main():
    lines = compile_isa("mfx_sgd.hip")
    ns, bs = scan_asm_stores(lines)
    print("checked %d wide asm stores in mfx_sgd.hip, %d without a trailing s_nop" % (ns, bs))
    tmp = tempfile.mkdtemp()
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                           "-I" + root + "/include", "-I" + root + "/mfrec_amd/csrc", "-c",
                           root + "/mfrec_amd/csrc/mfx_als.hip", "-save-temps", "-o", "als.o"], cwd=tmp)
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    n, bad = scan(open(os.path.join(tmp, asm)))
    print("checked %d DPP instructions, %d hazards" % (n, bad))
    return 1 if (bad or bs or ns == 0) else 0


if __name__ == "__main__":
    sys.exit(main())
