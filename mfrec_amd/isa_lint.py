"""Hazard lint of the gfx950 ISA that mfrec_amd/build.py links into libmfx.so.

Hand-written instructions inside inline asm are invisible to the compiler's hazard recogniser.  Two kinds
are used by the kernels (mfx_als.hip: f64 DPP row broadcasts; mfx_sgd_kernels.h: 16-byte write-through
stores) and both have bitten before (an intermittent full-size mismatch, fixed by an `s_nop`):

  * VALU write -> DPP read of the same VGPR needs 2 wait states (and v_cmpx -> DPP 5);
  * a VMEM store of more than 8 bytes reads its data VGPRs after issue: the next instruction must not
    overwrite them, so an asm store must carry its own `s_nop 1` (2 wait states; `s_nop 0` is not enough).

  * a VMEM LOAD written in inline asm (rocprim's look-back scans; round 5's archived k_topn_mfma variant with hand-written
    wait counts) is invisible to the compiler's wait-count pass: nothing may NAME its destination VGPRs -- no copy of a
    live-range split, no spill, no re-use -- until a wait that covers it: `s_waitcnt vmcnt(0)` (anybody's), or an asm
    `s_waitcnt vmcnt(N)` once at least N younger asm loads have been issued (vmcnt counts in order).  scan_asm_loads().

The scan is linear, so at every label and after every branch the history is UNKNOWN: a hand-written DPP
instruction (one inside an ;;#ASMSTART block) that close to such a point must bring its own `s_nop`
(compiler-generated DPP moves are the hazard recogniser's business and are only checked against the
visible history).  build.py runs lint_file() on the `-save-temps` assembly of exactly the objects it links, with the build's own flags,
and fails the build on any hazard; tools/check_dpp_hazard.py is the command-line face.
"""
import re


def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


UNKNOWN = "unknown"          # history marker: control flow joins / leaves here


def scan(lines, verbose=True):
    """-> (number of DPP instructions seen, number of hazards)"""
    hist_ops = []
    hist, bad, n = [], 0, 0          # hist: (VGPRs written | UNKNOWN, wait states if s_nop) of preceding instructions
    in_asm = False
    for line in lines:
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.endswith(":") and not t.startswith((";", "//")):   # a label: predecessors unknown
            hist.append((UNKNOWN, 0))
            hist_ops.append(UNKNOWN)
            continue
        if not t or t.startswith((";", ".", "//")):
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
                if w == UNKNOWN:
                    if in_asm:                           # hand-written: nobody else will pad it
                        bad += 1
                        if verbose:
                            print("HAZARD (hand-written DPP right behind a label / branch, no s_nop):", t)
                    break
                if w & src0:
                    bad += 1
                    if verbose:
                        print("HAZARD:", t)
                    break
                wait += 1
        w = regs(ops[0].split()[0]) if op.startswith(("v_", "ds_read", "global_load", "buffer_load")) and ops else set()
        hist.append((w, 0))
        if op.startswith(("s_branch", "s_cbranch", "s_setpc", "s_swappc")):
            hist.append((UNKNOWN, 0))                    # the fall-through may be a branch target's twin
            hist_ops.append(UNKNOWN)
    return n, bad


WIDE_STORES = ("global_store_dwordx4", "global_store_dwordx3", "buffer_store_dwordx4", "buffer_store_dwordx3",
               "flat_store_dwordx4", "flat_store_dwordx3")


def scan_asm_stores(lines):
    """-> (asm stores of more than 8 bytes, those NOT followed by an `s_nop N` with N >= 1 -- or a full
    `s_waitcnt vmcnt(0)` -- inside their asm block).  gfx940+/gfx950 need 2 wait states between such a store and a VALU write of its data VGPRs;
    `s_nop 0` is one."""
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
            if t.startswith(WIDE_STORES):
                if pending:
                    bad += 1
                n += 1
                pending = True
            elif t.startswith("s_nop"):
                arg = t.split()[1] if len(t.split()) > 1 else "0"
                if arg.isdigit() and int(arg) >= 1:
                    pending = False
            elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                pending = False          # the store has completed (rocprim's look-back scan stores do this)
    return n, bad


ASM_LOADS = ("buffer_load_dword", "global_load_dword", "flat_load_dword")


def all_regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def scan_asm_loads(lines, verbose=True):
    """-> (asm VMEM loads, hazards).  Linear scan per kernel (state reset at every `.amdhsa_kernel` / function label is not needed: a
    kernel ends with everything waited for or the scan reports it).  In flight: the destination registers of every asm load, youngest
    last.  An asm `s_waitcnt vmcnt(N)` lands all but the N youngest; any `s_waitcnt` with vmcnt(0) lands all.  if / else twins that
    load the SAME registers follow each other in linear order: re-loading a register that is in flight replaces the older entry
    (the twin), it is not a hazard.  Every other instruction naming a register in flight is one.  A loop's back edge is not followed:
    what is in flight at the bottom of a loop body stays in flight through the code behind the loop, which is the stricter reading."""
    n = bad = 0
    in_asm = False
    flight = []                                   # [set of VGPRs] per load, oldest first
    for line in lines:
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            if t.startswith(".end_amdhsa_kernel") or t.startswith(".amdhsa_kernel"):
                if flight and t.startswith(".amdhsa_kernel"):
                    pass
            if t.endswith(":") and not t.startswith((";", "//")) and not t.startswith(".L"):   # a new function: nothing carries over
                if flight:
                    bad += 1
                    if verbose:
                        print("HAZARD (asm loads still in flight at the end of a function):", sorted(set().union(*flight))[:8])
                flight = []
            continue
        op = t.split()[0]
        body = t.split(";")[0]
        if op.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", body)
            if m:
                k = int(m.group(1))
                if k == 0:
                    flight = []
                elif in_asm:
                    flight = flight[-k:]
            continue
        if op == "s_endpgm":
            flight = []
            continue
        named = all_regs(body)
        if in_asm and op.startswith(ASM_LOADS):
            n += 1
            dst = all_regs(body.split(",")[0])
            flight = [f for f in flight if not (f & dst)]          # the twin of an if / else pair
            src = named - dst
            if any(f & src for f in flight):
                bad += 1
                if verbose:
                    print("HAZARD (address of an asm load in registers a load is still going to write):", t)
            flight.append(dst)
            continue
        if flight and any(f & named for f in flight):
            bad += 1
            if verbose:
                print("HAZARD (names registers an asm load has in flight):", t)
    return n, bad


def lint_file(path, verbose=True):
    """-> dict(dpp, dpp_hazards, wide_asm_stores, store_hazards, asm_loads, load_hazards) for one gfx950 assembly file."""
    with open(path) as f:
        lines = f.readlines()
    n, bad = scan(lines, verbose)
    ns, bs = scan_asm_stores(lines)
    nl, bl = scan_asm_loads(lines, verbose)
    return {"dpp": n, "dpp_hazards": bad, "wide_asm_stores": ns, "store_hazards": bs, "asm_loads": nl, "load_hazards": bl}
