#!/usr/bin/env python3
"""Is the ONE-GPU block order (sharding.DsgdKMF(rank=-1, item_blocks_1gpu=n, balanced=True): the ratings grouped by item block, every
block in its own minimum-level order) as good a training order as the uniform shuffle the reference draws (base.py:1127-1129)?
Bit-identity to the oracle ON that order is pinned by the tests; MODEL QUALITY on it is what this measures: cfg 3 / 100 (10 K users x
2 K items, 1 M ratings of the synthetic planted model, k = 64), E epochs of the CPU oracle (train_linear_kernel) per order, final TRAIN
RMSE and HELD-OUT RMSE (200 K further ratings of the same generator) -- the block order against several uniform shuffles.
Writes profiles/r04_block_order_quality.json.  The order itself comes from the library (needs the GPU: the per-block minimum-level
order is built on the device); everything measured runs on the CPU oracle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth, sharding
from oracle import oracle as O

nu, ni, nnz, k, E = 10000, 2000, 1000000, 64, 10
n_test = 200000
LR, KU, KI, KB = 0.01, 0.1, 0.1, 0.007
idx_all, r_all = synth.make_ratings(nu, ni, nnz + n_test)
idx, r, tidx, tr = idx_all[:nnz], r_all[:nnz], np.ascontiguousarray(idx_all[nnz:]), np.ascontiguousarray(r_all[nnz:])
idx, r = np.ascontiguousarray(idx), np.ascontiguousarray(r)
u0, v0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)


def run(order):
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    i2, r2 = np.ascontiguousarray(idx[order]), np.ascontiguousarray(r[order])
    O.train_linear_kernel(E, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u, v, i2, r2, ib, ub)
    return float(O.rmse(O.RMSE_KMF_LINEAR, u, v, idx, r, ib, ub)), float(O.rmse(O.RMSE_KMF_LINEAR, u, v, tidx, tr, ib, ub))


S.init(0)
out = {"problem": "cfg 3 / 100: %d x %d, %d ratings (synthetic planted model, seed %d), k = %d, %d epochs of oracle.train_linear_kernel "
                  "(lr %g, K %g / %g / %g); held-out = %d further ratings of the same generator" % (nu, ni, nnz, synth.SEED_DATA, k, E, LR, KU, KI, KB, n_test),
       "orders": {}}
t0 = time.time()
for nb in (4, 8):
    for bal in (True, False):
        d = sharding.DsgdKMF(idx, r, nu, ni, k, -1, 1, balanced=bal, item_blocks_1gpu=nb)
        perm = d.order(); d.close()
        out["orders"]["block order, %d item blocks, %s inside a block" % (nb, "minimum-level order" if bal else "given order")] = run(perm)
out["orders"]["given order (the generator's: i.i.d.)"] = run(np.arange(nnz))
sh = []
for seed in range(6):
    sh.append(run(np.random.RandomState(100 + seed).permutation(nnz)))
    out["orders"]["uniform shuffle, seed %d" % (100 + seed)] = sh[-1]
sh = np.asarray(sh + [out["orders"]["given order (the generator's: i.i.d.)"]])
out["uniform_shuffles"] = {"n": int(sh.shape[0]), "train_rmse_mean": float(sh[:, 0].mean()), "train_rmse_min": float(sh[:, 0].min()),
                           "train_rmse_max": float(sh[:, 0].max()), "heldout_rmse_mean": float(sh[:, 1].mean()),
                           "heldout_rmse_min": float(sh[:, 1].min()), "heldout_rmse_max": float(sh[:, 1].max())}
b = out["orders"]["block order, 4 item blocks, minimum-level order inside a block"]
out["block_order_vs_shuffles"] = {"train_rmse_gap_to_mean": b[0] - out["uniform_shuffles"]["train_rmse_mean"],
                                  "heldout_rmse_gap_to_mean": b[1] - out["uniform_shuffles"]["heldout_rmse_mean"],
                                  "shuffle_spread_train": out["uniform_shuffles"]["train_rmse_max"] - out["uniform_shuffles"]["train_rmse_min"],
                                  "shuffle_spread_heldout": out["uniform_shuffles"]["heldout_rmse_max"] - out["uniform_shuffles"]["heldout_rmse_min"]}
out["cpu_seconds"] = time.time() - t0
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/block_order_quality.json", "w"), indent=1)
for kx, v in out["orders"].items():
    print("%-75s train %.6f  held-out %.6f" % (kx, v[0], v[1]))
print(json.dumps(out["block_order_vs_shuffles"]))
