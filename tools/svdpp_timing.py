#!/usr/bin/env python3
"""How slow is the sequential SVD++-like loop on the GPU next to the reference's own C loop? (ML-100K shape)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.lib import gd_estimator as G
from oracle import oracle as O, build_ref
S.init(0)
out = []
for nu, ni, nnz in ((943, 1682, 100000), (6040, 3706, 1000000)):
    idx, r = synth.make_ratings(nu, ni, nnz, seed=1)
    pairs = np.array(sorted(set(map(tuple, idx.tolist()))), dtype=np.int32)
    counts = np.bincount(pairs[:, 0], minlength=nu); first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    h = np.ascontiguousarray(np.stack([np.where(counts > 0, first, -1), np.where(counts > 0, counts, -1)], axis=1).astype(np.int32))
    ib, ub = np.zeros(ni), np.zeros(nu)
    k, E = 2, 2
    def run(fn):
        u, v, y = synth.init_const(k, ni), synth.init_const(k, nu), np.zeros((k, ni))
        t0 = time.perf_counter()
        fn(E, E, 1e9, k, 0.1, 0.01, 0, 0, 0.05, 3.5, u, v, y, idx, r, pairs, h, ib, ub, nu, ni)
        return (time.perf_counter() - t0) / (k * E), u
    run(G.estimator_loop_with_implicit_feedback)
    tg, ug = run(G.estimator_loop_with_implicit_feedback)
    ref = build_ref.load()
    fn = ref["gd_estimator"].estimator_loop_with_implicit_feedback if ref else O.estimator_loop_with_implicit_feedback
    tc, uc = run(fn)
    out.append({"shape": [nu, ni, nnz], "lds_resident": bool((2 * ni + nu) * 8 <= 150 * 1024), "gpu_s_per_pass": tg,
                "gpu_us_per_rating": tg / nnz * 1e6, "cpu_s_per_pass": tc, "cpu_kind": "reference" if ref else "port",
                "bit_identical": bool(np.array_equal(ug, uc))})
print(json.dumps(out))
