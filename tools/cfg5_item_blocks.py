#!/usr/bin/env python3
"""BASELINE cfg 5 WHOLE (10 M users x 1 M items, 1 B ratings, k=64) on ONE MI355X on the ITEM-BLOCK order (sharding.DsgdKMF, world = 1):
1 M item rows do not fit the register file, 24 blocks of ~42 K do -- every block on the static resident-side epoch (its items resident,
the 10 M user rows travel).  Compare tools/cfg5_whole.py (level schedule on the given order: 0.56 s per epoch)."""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.sharding import DsgdKMF

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=10000000)
ap.add_argument("--items", type=int, default=1000000)
ap.add_argument("--nnz", type=int, default=1000000000)
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--blocks", type=int, default=0)
ap.add_argument("--user-blocks", type=int, default=3, help="user blocks (rows of a travelling side must stay below 2 GB)")
ap.add_argument("--balanced", action="store_true", help="every block in its own minimum-level order")
ap.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
a = ap.parse_args()
nu, ni, nnz, k = a.users, a.items, a.nnz, a.k
nb = a.blocks or -(-ni // 42000)
nb = -(-nb // a.user_blocks) * a.user_blocks
alive = [True]


def heartbeat():
    t0 = time.time()
    while alive[0]:
        time.sleep(45)
        if alive[0]:
            print("  ... %.0f s" % (time.time() - t0), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
S.init(0)
out = {"balanced": bool(a.balanced), "config": "cfg5 whole on one GPU, block order: %d users x %d items, %d ratings, k=%d, %d user x %d item blocks" % (nu, ni, nnz, k, a.user_blocks, nb)}
t0 = time.time()
idx, r = synth.make_ratings(nu, ni, nnz, workers=a.workers)
out["data_s"] = time.time() - t0
print("data %.1f s" % out["data_s"], flush=True)
t0 = time.time()
d = DsgdKMF(idx, r, nu, ni, k, -1, a.user_blocks, balanced=a.balanced, item_blocks_1gpu=nb)
out["create_s"] = time.time() - t0
info = d.info()
out.update({"resident_blocks": info["resident_blocks"], "levels_total": info["levels_total"], "largest_block": info["max_block"]})
print("create %.1f s: %s" % (out["create_s"], info), flush=True)
del idx, r
d.set_factors(synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8), np.zeros(ni), np.zeros(nu))
print("factors up", flush=True)
se0 = d.train(0, 1, 0.01, 0.1, 0.1, 0.007)
S.sync()
t0 = time.time()
se = d.train(0, a.epochs, 0.01, 0.1, 0.1, 0.007)
S.sync()
dt = time.time() - t0
out["epoch_s"] = dt / a.epochs
out["rating_updates_per_s"] = nnz * a.epochs / dt
out["rmse_online"] = [float(np.sqrt(se0[0] / nnz))] + [float(np.sqrt(x / nnz)) for x in se]
out["rmse_decreasing"] = bool(all(x > y for x, y in zip(out["rmse_online"][:-1], out["rmse_online"][1:])))
d.close()
alive[0] = False
print(json.dumps(out), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cfg5_item_blocks.json", "w"), indent=1)
