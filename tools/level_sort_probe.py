#!/usr/bin/env python3
"""Does the ORDER OF THE RATINGS INSIDE A LEVEL matter for k1_level?  Ratings of one level are independent, so any
permutation inside a level is the same sequential algorithm (same factors, bit for bit).  The probe feeds the
library the same ratings in three input orders that all have the same levels -- as generated (ascending t inside a
level), sorted by user inside a level, sorted by item inside a level -- and times the full-k epoch on each."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan

LR, KU, KI, KB = 0.01, 0.1, 0.1, 0.007
S.init(0)
cfgs = {"cfg2": (100000, 50000, 10000000), "cfg3": (1000000, 200000, 100000000),
        "cfg5shard": (1250000, 1000000, 125000000), "cfg5third": (10000000, 1000000, 330000000)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2"]
k = 64
res = {}
for name in names:
    nu, ni, nnz = cfgs[name]
    idx, r = synth.make_ratings(nu, ni, nnz)
    lev = np.zeros(nnz, dtype=np.int32)
    n = C.c_int64(0)
    S.check(S.lib().mfx_schedule_levels_gpu(S.i32(idx, "idx", 2), nnz, nu, ni, lev.ctypes.data_as(S._ip), C.byref(n)))
    print(name, "levels", n.value, flush=True)
    u0, v0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
    out = {}
    ref = None
    for what in ("as-is", "by-user", "by-item", "by-item-block"):
        if what == "as-is":
            perm = None
        elif what == "by-user":
            perm = np.argsort((lev.astype(np.int64) << 32) | idx[:, 0], kind="stable")
        elif what == "by-item":
            perm = np.argsort((lev.astype(np.int64) << 32) | idx[:, 1], kind="stable")
        else:   # items in 8 contiguous blocks (one per XCD under round-robin dispatch would need more; locality only)
            perm = np.argsort((lev.astype(np.int64) << 32) | ((idx[:, 1].astype(np.int64) * 64 // ni) << 24) | (idx[:, 0] >> 8),
                              kind="stable")
        pi = idx if perm is None else np.ascontiguousarray(idx[perm])
        pr = r if perm is None else np.ascontiguousarray(r[perm])
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        with SgdPlan(pi, pr, nu, ni, k) as p:
            assert p.info()["n_levels"] == n.value
            p.set_factors(u, v, ib, ub)
            p.run_kmf(0, 1, LR, KU, KI, KB, want_rmse=False)
            p.get_factors(u, v, ib, ub)
            S.sync()
            t0 = time.time()
            p.run_kmf(0, 3, LR, KU, KI, KB, want_rmse=False)
            S.sync()
            dt = (time.time() - t0) / 3
        same = True
        if ref is None:
            ref = (u, v, ib, ub)
        else:
            same = all(np.array_equal(a, b) for a, b in zip(ref, (u, v, ib, ub)))
        out[what] = {"ms_per_epoch": dt * 1e3, "G_per_s": nnz / dt / 1e9, "same_bits_as_as_is": bool(same)}
        print(name, what, out[what], flush=True)
    res[name] = out
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/level_sort_probe.json", "w"), indent=1)
