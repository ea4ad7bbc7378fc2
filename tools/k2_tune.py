#!/usr/bin/env python3
"""Same-process A/B of the XCD-resident Funk pass's tuning bits (MFX_K2_TUNE; read by the library at every pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan, Timer
S.init(0)
nu, ni, nnz, k = 100000, 50000, 10000000, 64
idx, r = synth.make_ratings(nu, ni, nnz)
if "balanced" in sys.argv:
    from mfrec_amd.lib.datasets import balanced_order_gpu
    perm, _ = balanced_order_gpu(idx, nu, ni)
    idx, r = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
p = SgdPlan(idx, r, nu, ni, k)
p.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
p.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)
tunes = [int(x) for x in os.environ.get("TUNES", "0,2").split(",")]
for rep in range(3):
    out = []
    for t in tunes:
        os.environ["MFX_K2_TUNE"] = str(t)
        p.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
        with Timer() as tm:
            h = p.run_funk(2, 1e9, 0.1, 0.001, 0.05)
        out.append("tune %d: %.3f ms/pass" % (t, tm.ms / len(h)))
    print(" | ".join(out), flush=True)
print(p.kernel_name(3), p.info()["n_levels"])
