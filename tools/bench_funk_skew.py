#!/usr/bin/env python3
"""Funk passes on skewed data, resident-side pass in its granule form (MFX_K2_BOX=0) against the mailbox form (2 / 4 / 8 mailboxes per poll;
a trailing "h", e.g. 8h, or "auto": with the hot-wave role where the plan chooses it -- without the "h" it is switched off, MFX_K2_HOT=0):
(a) the MovieLens-10M-shaped synthetic of bench_ml10m_shape.py, (b) Zipf-0.8 item degrees over cfg 2's shape, (c) cfg 2 uniform."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
k = 64
rs = np.random.RandomState(1)


def ml10m():
    nu, ni, nnz = 69878, 10677, 10000000
    pu, pi = 1.0 / np.arange(1, nu + 1) ** 0.4, 1.0 / np.arange(1, ni + 1) ** 0.47
    idx = np.stack([rs.choice(nu, nnz, p=pu / pu.sum()), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32)
    return "ml10m-shaped", nu, ni, np.ascontiguousarray(idx)


def zipf08():
    nu, ni, nnz = 100000, 50000, 10000000
    pi = 1.0 / np.arange(1, ni + 1) ** 0.8
    idx = np.stack([rs.randint(0, nu, nnz), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32)
    return "zipf-0.8 items, cfg2 shape", nu, ni, np.ascontiguousarray(idx)


def uniform():
    nu, ni, nnz = 100000, 50000, 10000000
    idx, _ = synth.make_ratings(nu, ni, nnz)
    return "cfg2 uniform shuffled", nu, ni, idx


for make in (uniform, ml10m, zipf08):
    name, nu, ni, idx = make()
    nnz = idx.shape[0]
    r = rs.randint(1, 6, nnz).astype(np.float64)
    outs = {}
    for box in sys.argv[1:] or ["0", "2", "8", "auto"]:
        os.environ.pop("MFX_K2_BOX", None); os.environ.pop("MFX_K2_HOT", None); os.environ.pop("MFX_K2_HOTU", None)
        if box.endswith("-nou"):                                # hot items only (no hot-user waves)
            os.environ["MFX_K2_HOTU"] = "0"
        elif box.startswith("auto-u"):                          # auto-u<degree>: hot users from that degree on
            os.environ["MFX_K2_HOTU"] = box[6:]
        if not box.startswith("auto"):
            os.environ["MFX_K2_BOX"] = box.rstrip("h")
            if not box.endswith("h"):
                os.environ["MFX_K2_HOT"] = "0"
        with SgdPlan(idx, r, nu, ni, k) as p:
            u, v = synth.init_const(k, ni), synth.init_const(k, nu)
            p.set_factors(u, v)
            p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
            p.set_factors(u, v)
            with Timer() as tf:
                hist = p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
            p.get_factors(u, v)
            outs[box] = (u.copy(), v.copy())
            print(json.dumps({"data": name, "MFX_K2_BOX": box, "kernel": p.kernel_name(3), "funk_ms_per_pass": tf.ms / max(len(hist), 1),
                              "passes": len(hist), "scalar_updates_per_s": nnz * len(hist) / (tf.ms * 1e-3)}), flush=True)
    keys = list(outs)
    print(json.dumps({"data": name, "all_forms_bit_equal": bool(all(np.array_equal(outs[keys[0]][0], outs[q][0]) and
                                                                   np.array_equal(outs[keys[0]][1], outs[q][1]) for q in keys[1:]))}), flush=True)
