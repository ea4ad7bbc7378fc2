#!/usr/bin/env python3
"""One-off scale validation: BASELINE cfg 3 whole (1 M x 200 K, 100 M ratings, k=64), ONE epoch on the GPU
(device-built schedule) against the oracle's sequential loop, bit for bit.  ~3 minutes, 60 GB of host RAM."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan
from oracle import oracle as O
S.init(0)
nu, ni, nnz, k = 1000000, 200000, 100000000, 64
t0 = time.time(); idx, r = synth.make_ratings(nu, ni, nnz); print("data %.1f s" % (time.time() - t0), flush=True)
u0, v0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
t0 = time.time()
with SgdPlan(idx, r, nu, ni, k) as p:
    info = p.info()
    p.set_factors(u, v, ib, ub)
    rm = p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007)
    p.get_factors(u, v, ib, ub)
print("gpu %.1f s, levels %d, builder %s" % (time.time() - t0, info["n_levels"], info["schedule_builder"]), flush=True)
u2, v2, ib2, ub2, rm2 = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu), np.zeros(1)
t0 = time.time()
O.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u2, v2, idx, r, ib2, ub2, 1, 1, 0, rm2)
print("oracle %.1f s" % (time.time() - t0), flush=True)
ok = bool(np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2) and np.array_equal(ub, ub2))
print(json.dumps({"config": "cfg3 whole, 1 epoch", "bit_identical": ok, "rmse_gpu": float(rm[0]), "rmse_oracle": float(rm2[0]),
                  "levels": info["n_levels"], "schedule_builder": info["schedule_builder"]}))
sys.exit(0 if ok else 1)
