#!/usr/bin/env python3
"""A MovieLens-10M-SHAPED synthetic (the reference's only published workload: gradient_descent.py:41-45, 69 878 users x
10 677 items, 10 M ratings; the real file is not available offline): power-law degrees on both sides (hottest item ~35 K
ratings, hottest user ~7 K), shuffled order.  Full-k SGD epochs (level schedule vs the plan's own choice) and Funk passes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
if os.environ.get("MFX_LIB"):
    S.LIB_PATH = os.environ["MFX_LIB"]
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
nu, ni, nnz, k = 69878, 10677, 10000000, int(os.environ.get("K", "64"))
rs = np.random.RandomState(1)
pu, pi = 1.0 / np.arange(1, nu + 1) ** 0.4, 1.0 / np.arange(1, ni + 1) ** 0.47
idx = np.ascontiguousarray(np.stack([rs.choice(nu, nnz, p=pu / pu.sum()), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32))
r = rs.randint(1, 6, nnz).astype(np.float64)
print(json.dumps({"max_item_degree": int(np.bincount(idx[:, 1]).max()), "max_user_degree": int(np.bincount(idx[:, 0]).max())}), flush=True)
u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
res = {}
for epoch in (("auto",) if os.environ.get("ONLY_AUTO") else ("levels", "auto")):
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    with SgdPlan(idx, r, nu, ni, k, epoch=epoch) as p:
        p.set_factors(u, v, ib, ub)
        info = p.info()
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        with Timer() as tm:
            p.run_kmf(0, 3, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        p.get_factors(u, v, ib, ub)
        name = p.kernel_name(0)
        if epoch == "auto":
            p.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
            p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
            with Timer() as tf:
                hist = p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
            print(json.dumps({"funk_kernel": p.kernel_name(3), "funk_ms_per_pass": tf.ms / max(len(hist), 1), "passes": len(hist),
                              "scalar_updates_per_s": nnz * len(hist) / (tf.ms * 1e-3)}), flush=True)
    res[epoch] = (u, v, ib, ub)
    print(json.dumps({"epoch": epoch, "kernel": name, "levels": info["n_levels"], "ms_per_epoch": tm.ms / 3,
                      "updates_per_s": nnz * 3 / (tm.ms * 1e-3)}), flush=True)
if "levels" in res:
    print(json.dumps({"auto_equals_levels_bit_for_bit": bool(all(np.array_equal(x, y) for x, y in zip(res["levels"], res["auto"])))}))
