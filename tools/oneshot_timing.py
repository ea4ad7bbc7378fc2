#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-shot drop-in (host arrays in, host arrays out): schedule
build + H2D + epochs + D2H, for DESIGN.md."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.lib import kmf_train as K
S.init(0)
nu, ni, nnz, k, E = 100000, 50000, 10000000, 64, 10
idx, r = synth.make_ratings(nu, ni, nnz)
u, v = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
ib, ub = np.zeros(ni), np.zeros(nu)
K.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)   # warm-up
t0 = time.perf_counter()
K.train_linear_kernel(E, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)
dt = time.perf_counter() - t0
print(json.dumps({"one_shot_seconds": dt, "epochs": E, "updates_per_s_incl_schedule_and_pcie": nnz * E / dt,
                  "MFX_SCHED": os.environ.get("MFX_SCHED", "auto")}))
from mfrec_amd.plan import SgdPlan
for sched in ("host", "device"):
    for rep in range(2):
        t0 = time.perf_counter()
        with SgdPlan(idx, r, nu, ni, k, schedule=sched) as p:
            dt = time.perf_counter() - t0
            info = p.info()
    print(json.dumps({"plan_create_seconds": dt, "schedule": sched, "builder": info["schedule_builder"],
                      "schedule_us": info["schedule_us"], "n_levels": info["n_levels"]}))
