#!/usr/bin/env python3
"""Single-GPU replay of the G-GPU DSGD schedule at BASELINE cfg 3 (fake world: all G x G blocks on this GPU,
rotation = device copy): how long ONE GPU's share of an epoch takes (= replay time / G), i.e. the compute side of
the exact multi-GPU mode without the xGMI hand-offs.  A projection input for DESIGN.md, not a multi-GPU measurement."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.sharding import DsgdKMF
S.init(0)
nu, ni, nnz, k = 1000000, 200000, 100000000, 64
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
order = sys.argv[2] if len(sys.argv) > 2 else "shuffled"
idx, r = synth.make_ratings(nu, ni, nnz)
t0 = time.time()
d = DsgdKMF(idx, r, nu, ni, k, -1, G, overlap=("noverlap" not in sys.argv), balanced=("balanced" in sys.argv))
t_create = time.time() - t0
d.set_factors(synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8), np.zeros(ni), np.zeros(nu))
d.train(0, 1, 0.01, 0.1, 0.1, 0.007)
S.sync()
E = 3
t0 = time.time()
se = d.train(0, E, 0.01, 0.1, 0.1, 0.007)
S.sync()
dt = (time.time() - t0) / E
info = d.info()
print(json.dumps({"G": G, "replay_s_per_epoch_all_G_ranks_on_one_gpu": dt, "per_gpu_compute_s_per_epoch": dt / G,
                  "projected_updates_per_s_without_transfers": nnz / (dt / G), "levels_total": info["levels_total"],
                  "levels_per_gpu_per_epoch": info["levels_total"] / G, "max_block": info["max_block"],
                  "min_block": info["min_block"], "create_s": t_create, "rmse_online_last": float(np.sqrt(se[-1] / nnz)),
                  "item_blocks": info["item_blocks"], "item_block_bytes": int(ni / info["item_blocks"] * (k + 1) * 8), "args": sys.argv[1:]}))
d.close()
