#!/usr/bin/env python3
"""Full-k kernel families side by side on cfg 2 (linear is bench.py's metric)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
nu, ni, nnz, k = 100000, 50000, 10000000, 64
idx, r = synth.make_ratings(nu, ni, nnz)
with SgdPlan(idx, r, nu, ni, k) as p:
    for mode, name in ((0, "linear"), (1, "logistic")):
        p.set_factors(synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8), np.zeros(ni), np.zeros(nu))
        p.run_kmf(mode, 2, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        with Timer() as tm:
            p.run_kmf(mode, 10, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        print(json.dumps({"mode": name, "ms_per_epoch": tm.ms / 10, "updates_per_s": nnz * 10 / (tm.ms * 1e-3)}))
    p.set_factors(synth.init_const(k, ni), synth.init_const(k, nu), np.zeros(ni), np.zeros(nu))
    with Timer() as tm:
        h = p.run_learned_bias(1, 1e9, 0.1, 0.001, 0.001, 0.001, 0.05, 0.01, 3.5, epoch_cap=1)
    print(json.dumps({"mode": "learned_bias (k passes, full dot each)", "passes": len(h), "ms_per_pass": tm.ms / len(h),
                      "updates_per_s": nnz * len(h) / (tm.ms * 1e-3)}))
