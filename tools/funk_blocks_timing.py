#!/usr/bin/env python3
"""The Funk pass on a catalogue with more items than the resident-side pass has lanes (1 M items, BASELINE cfg 5's item side; 2 M users,
40 M ratings, k = 4 features x 2 epochs): the whole problem as ONE plan (the pass falls to the single-XCD pass / one launch per level)
against the item-block order (sharding.DsgdKMF.train_funk: blocks of <= 400 K items, every block on the persistent pass)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth, sharding as SH
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
nu, ni, nnz, k = 2000000, 1000000, 40000000, 4
idx, r = synth.make_ratings(nu, ni, nnz)
out = {"shape": "%d users x %d items, %d ratings, k = %d features x 2 epochs + commit passes" % (nu, ni, nnz, k)}
with SgdPlan(idx, r, nu, ni, k) as p:
    p.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
    p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
    p.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
    with Timer() as tm:
        hist = p.run_funk(2, 1e9, 0.1, 0.001, 0.05)
    out["one_plan"] = {"kernel": p.kernel_name(3), "ms_per_pass": tm.ms / len(hist), "levels": p.info()["n_levels"],
                       "scalar_updates_per_s": nnz * len(hist) / (tm.ms * 1e-3)}
print(json.dumps(out["one_plan"]), flush=True)
nb = -(-ni // SH.FUNK_ITEMS_PER_BLOCK)
t0 = time.time()
d = SH.DsgdKMF(idx, r, nu, ni, k, -1, 1, item_blocks_1gpu=nb)
build_s = time.time() - t0
d.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
d.train_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
d.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
S.sync()
t0 = time.time()
hist = d.train_funk(2, 1e9, 0.1, 0.001, 0.05)
S.sync()
dt = time.time() - t0
out["block_order"] = {"item_blocks": nb, "ms_per_pass": dt * 1e3 / len(hist), "scalar_updates_per_s": nnz * len(hist) / dt,
                      "plan_build_s": build_s}
d.close()
print(json.dumps(out["block_order"]), flush=True)
out["speedup"] = out["one_plan"]["ms_per_pass"] / out["block_order"]["ms_per_pass"]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_k2_block_order.json", "w"), indent=1)
print(json.dumps({"speedup": out["speedup"]}))
