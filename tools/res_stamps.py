#!/usr/bin/env python3
"""Diagnostic: where a step of the static resident-side epoch spends its time (k1_res / k1_res_look).
Needs mfrec_amd/libmfx_diag.so:  MFX_EXTRA_FLAGS=-DMFX_RES_STAMPS python -m mfrec_amd.build
usage: tools/res_stamps.py [order=shuffled|user-major|balanced] [ordered|butterfly]   (cfg 2, static binding)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
S.LIB_PATH = os.environ.get("MFX_LIB") or S.LIB_PATH.replace("libmfx.so", "libmfx_diag.so")
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
order = sys.argv[1] if len(sys.argv) > 1 else "shuffled"
dot = sys.argv[2] if len(sys.argv) > 2 else "ordered"          # "butterfly": the tolerance mode
nu, ni, nnz, k = 100000, 50000, 10000000, 64
idx, r = synth.make_ratings(nu, ni, nnz)
if order == "user-major":
    idx, r = synth.user_major(idx, r)
elif order == "balanced":
    from mfrec_amd.lib.datasets import balanced_order_gpu
    perm, _ = balanced_order_gpu(idx, nu, ni)
    idx, r = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
with SgdPlan(idx, r, nu, ni, k, epoch="static", dot=dot) as p:
    p.set_factors(synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8), np.zeros(ni), np.zeros(nu))
    p.run_kmf(0, 3, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
    with Timer() as tm:
        p.run_kmf(0, 1, 0.01, 0.1, 0.1, 0.007, want_rmse=False)
    name, info = p.kernel_name(0), p.info()
st = np.zeros(64 * 16, dtype=np.uint64)
S.lib().mfx_debug_res_stamps.argtypes = [C.c_void_p]
assert S.lib().mfx_debug_res_stamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.reshape(64, 16).astype(np.float64)
st = st[st[:, 10] > 0]
tot = st[:, :10].sum(axis=1)
ticks_per_us = np.median(tot) / (tm.ms * 1e3)
print("%s | %s order, %d levels | epoch %.3f ms | %d sampled waves, median %d steps + %d idle polls, %.1f ratings per step | %.1f ticks per us"
      % (name, order, info["n_levels"], tm.ms, st.shape[0], np.median(st[:, 10]), np.median(st[:, 11]), np.median(st[:, 12] / st[:, 10]), ticks_per_us))
names = ["0 poll wait + ballot", "1 idle polling", "2 row loads", "3 row updates, stores issued", "4 -", "5 store acknowledge",
         "6 publish, advance, next polls issued", "7 products -> LDS tile (tolerance mode: partials + butterfly)", "8 ordered sums (tolerance mode: none), errors, biases", "9"]
per_step = np.median(st[:, :10] / st[:, 10:11], axis=0) / ticks_per_us
share = np.median(st[:, :10] / tot[:, None], axis=0)
for i in range(9):
    print("   %-52s %6.3f us per step   %5.1f %% of the epoch" % (names[i], per_step[i], 100 * share[i]))
print("   step (all phases but idle polling): %.3f us" % (per_step.sum() - per_step[1]))
