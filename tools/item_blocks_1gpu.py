#!/usr/bin/env python3
"""cfg 3 whole on ONE GPU on the item-block order (sharding.DsgdKMF, world = 1, any number of item blocks): block counts and the
per-block balanced order side by side.  usage: item_blocks_1gpu.py [nb ...]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.sharding import DsgdKMF
from mfrec_amd.plan import Timer
S.init(0)
nu, ni, nnz, k = 1000000, 200000, 100000000, 64
idx, r = synth.make_ratings(nu, ni, nnz)
u0, v0 = synth.init_normal(k, ni, synth.SEED_INIT), synth.init_normal(k, nu, synth.SEED_INIT + 1)
for nb in [int(x) for x in sys.argv[1:]] or [5, 6, 8]:
    for balanced in (False, True):
        t0 = time.perf_counter()
        d = DsgdKMF(idx, r, nu, ni, k, -1, 1, balanced=balanced, item_blocks_1gpu=nb)
        build = time.perf_counter() - t0
        d.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
        d.train(0, 1, 0.01, 0.1, 0.1, 0.007)
        with Timer() as tm:
            d.train(0, 3, 0.01, 0.1, 0.1, 0.007)
        info = d.info()
        d.close()
        print(json.dumps({"item_blocks": nb, "balanced": balanced, "ms_per_epoch": tm.ms / 3, "updates_per_s": nnz * 3 / (tm.ms * 1e-3),
                          "resident_blocks": info["resident_blocks"], "levels_total": info["levels_total"], "build_s": round(build, 1)}), flush=True)
