#!/usr/bin/env python3
"""Full-k epoch rate at several ranks k on cfg 2's shape (shuffled order): how much do rows that are not a whole
number of 128-byte lines cost?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan
S.init(0)
nu, ni, nnz = 100000, 50000, 10000000
idx, r = synth.make_ratings(nu, ni, nnz)
ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 20, 32, 40, 48, 64, 96, 128]
out = {}
for k in ks:
    u0, v0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        S.sync(); t0 = time.time()
        p.run_kmf(0, 5, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        S.sync(); dt = (time.time() - t0) / 5
        out[k] = {"ms_per_epoch": dt * 1e3, "G_updates_per_s": nnz / dt / 1e9, "row_bytes": 8 * k,
                  "GB_s_rows": nnz * 4 * 8 * k / dt / 1e9, "kernel": p.kernel_name(0)}
    print(k, out[k], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/k_sweep.json", "w"), indent=1)
