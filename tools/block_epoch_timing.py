#!/usr/bin/env python3
"""One DSGD block of cfg 3 at G = 8 (125 K users x 25 K items, 1.56 M ratings, balanced order) and at G = 4 / G = 2:
epoch time of the level schedule against the static resident-side epoch, same plan data, one process."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan, Timer, LINEAR
from mfrec_amd.lib.datasets import balanced_order_gpu
S.init(0)
k = 64
for G in (8, 4, 2):
    nu, ni, nnz = 1000000 // G, 200000 // G, 100000000 // (G * G)
    idx, r = synth.make_ratings(nu, ni, nnz, seed=5)
    perm, _ = balanced_order_gpu(idx, nu, ni)
    idx, r = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    out = {"G": G, "block": [nu, ni, nnz]}
    for epoch in ("levels", "static"):
        try:
            with SgdPlan(idx, r, nu, ni, k, epoch=epoch) as p:
                p.set_factors(synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2), np.zeros(ni), np.zeros(nu))
                p.run_kmf(LINEAR, 2, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
                with Timer() as t:
                    p.run_kmf(LINEAR, 10, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
                out[epoch] = {"ms_per_epoch": t.ms / 10, "levels": p.info()["n_levels"], "kernel": p.kernel_name(LINEAR)}
        except Exception as e:
            out[epoch] = {"error": str(e)[:200]}
    print(json.dumps(out), flush=True)
