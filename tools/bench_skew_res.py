#!/usr/bin/env python3
"""Power-law (Zipf) item degrees over cfg 2's shape: the level schedule (with its tail kernel) against the static
resident-side epoch (hot items live in one wave's registers).  usage: bench_skew_res.py [exponent]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
nu, ni, nnz, k = 100000, 50000, 10000000, 64
a = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
rs = np.random.RandomState(1)
pi = 1.0 / np.arange(1, ni + 1) ** a
idx = np.stack([rs.randint(0, nu, nnz), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32)
r = rs.randint(1, 6, nnz).astype(np.float64)
u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
res = {}
for epoch in ("levels", "static", "auto"):
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    with SgdPlan(idx, r, nu, ni, k, epoch=epoch) as p:
        p.set_factors(u, v, ib, ub)
        info = p.info()
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        with Timer() as tm:
            p.run_kmf(0, 2, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
        p.get_factors(u, v, ib, ub)
        name = p.kernel_name(0)
    res[epoch] = (u, v, ib, ub)
    print(json.dumps({"zipf": a, "epoch": epoch, "kernel": name, "max_item_degree": int(np.bincount(idx[:, 1]).max()),
                      "levels": info["n_levels"], "ms_per_epoch": tm.ms / 2, "updates_per_s": nnz * 2 / (tm.ms * 1e-3)}), flush=True)
same = all(np.array_equal(x, y) for x, y in zip(res["levels"], res["static"]))
print(json.dumps({"static_equals_levels_bit_for_bit": bool(same)}))
