#!/usr/bin/env python3
"""Exact-mode throughput on power-law (Zipf) degrees: the structural limit (levels >= max degree)
and what the tail kernel recovers.  Not a BASELINE config."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
nu, ni, nnz, k = 100000, 50000, 10000000, int(os.environ.get("K", "64"))
rs = np.random.RandomState(1)
pi = 1.0 / np.arange(1, ni + 1) ** float(sys.argv[1] if len(sys.argv) > 1 else 0.8)
idx = np.stack([rs.randint(0, nu, nnz), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32)
r = rs.randint(1, 6, nnz).astype(np.float64)
for tail in ("1", "0"):
    os.environ["MFX_K1_TAIL"] = tail
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2), np.zeros(ni), np.zeros(nu))
        info = p.info()
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        with Timer() as tm:
            p.run_kmf(0, 3, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        if "funk" in sys.argv:
            p.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
            p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
            with Timer() as tf:
                hist = p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
            print(json.dumps({"tail_kernel": tail, "funk_ms_per_pass": tf.ms / max(len(hist), 1), "passes": len(hist)}))
    print(json.dumps({"tail_kernel": tail, "max_item_degree": int(np.bincount(idx[:, 1]).max()),
                      "levels": info["n_levels"], "launches_per_epoch": info["launches_per_epoch"],
                      "ms_per_epoch": tm.ms / 3, "updates_per_s": nnz * 3 / (tm.ms * 1e-3)}))
