#!/usr/bin/env python3
"""Diagnostic (libmfx_diag.so built with MFX_EXTRA_FLAGS=-DMFX_BURST_STATS): how the burst walk of the static resident-side epoch
covers a skewed problem -- MovieLens-10M shape by default, `zipf` = Zipf-0.8 items over cfg 2's shape."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
S.LIB_PATH = os.environ.get("MFX_LIB") or S.LIB_PATH.replace("libmfx.so", "libmfx_diag.so")
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
k = 64
rs = np.random.RandomState(1)
if len(sys.argv) > 1 and sys.argv[1] == "zipf":
    nu, ni, nnz = 100000, 50000, 10000000
    pi = 1.0 / np.arange(1, ni + 1) ** 0.8
    idx = np.ascontiguousarray(np.stack([rs.randint(0, nu, nnz), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32))
else:
    nu, ni, nnz = 69878, 10677, 10000000
    pu, pi = 1.0 / np.arange(1, nu + 1) ** 0.4, 1.0 / np.arange(1, ni + 1) ** 0.47
    idx = np.ascontiguousarray(np.stack([rs.choice(nu, nnz, p=pu / pu.sum()), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32))
r = rs.randint(1, 6, nnz).astype(np.float64)
u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
st = (C.c_ulonglong * 16)()
with SgdPlan(idx, r, nu, ni, k) as p:
    p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
    assert S.lib().mfx_debug_burst_stats(st, 1) == 0
    with Timer() as tm:
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
    assert S.lib().mfx_debug_burst_stats(st, 1) == 0
    name = p.kernel_name(0)
v = list(st)
print(name, "%.2f ms per epoch, max item degree %d" % (tm.ms, np.bincount(idx[:, 1]).max()))
print(json.dumps({"burst_calls": v[0], "calls_with_nothing_ready": v[1], "windows_walked": v[2], "ratings_in_bursts": v[3],
                  "windows_not_ready_in_full": v[4], "bursts_ended_by_a_window_with_nothing_ready": v[5], "windows_of_eight": v[6],
                  "ordinary_steps": v[7], "ratings_in_ordinary_steps": v[8], "idle_polls": v[9],
                  "ratings_per_window": v[3] / max(v[2], 1)}))
