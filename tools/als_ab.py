#!/usr/bin/env python3
"""Same-process A/B of ALS row-kernel variants (boxes of the pool differ, so variants are never compared across calls):
MFX_ALS_OCC / MFX_ALS_TUNE / MFX_ALS_KERNEL are read by the library at every call.
usage: als_ab.py [k] [variant ...]   variant = OCC:TUNE[:KERNEL], e.g. 3:0 3:1 4:0 4:1"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from mfrec_amd import _shim as S, synth
if os.environ.get("MFX_LIB"):
    S.LIB_PATH = os.environ["MFX_LIB"]
from mfrec_amd.lib.als_implicit import als_wrmf
from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
S.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = sys.argv[2:] or ["3:0", "3:1", "3:3", "4:0", "4:1"]
shapes = {"short+long (100K x 20K, 10M: degrees 100 / 500)": (100000, 20000, 10000000),
          "short only (100K x 100K, 10M: degrees 100 / 100)": (100000, 100000, 10000000)}
rounds = int(os.environ.get("ROUNDS", "3"))
for name, (nu, ni, nnz) in shapes.items():
    idx, _ = synth.make_ratings(nu, ni, nnz)
    M = sp.csr_matrix((np.ones(nnz), (idx[:, 0], idx[:, 1])), shape=(nu, ni)); M.sum_duplicates()
    views = create_bool_sparse_row(M) + create_bool_sparse_col(M)
    rs = np.random.RandomState(0)
    u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    res, ref = {}, None
    for rd in range(rounds):
        for var in variants:
            parts = var.split(":")
            os.environ["MFX_ALS_OCC"], os.environ["MFX_ALS_TUNE"] = parts[0], parts[1]
            os.environ["MFX_ALS_KERNEL"] = parts[2] if len(parts) > 2 else ""
            u, v = u0.copy(), v0.copy()
            als_wrmf(2, k, u, v, None, None, *views, nu, ni)
            ms = S.lib().mfx_last_compute_ms() / 2
            res.setdefault(var, []).append(ms)
            if os.environ.get("NOCHECK") == "1":
                pass
            elif ref is None:
                ref = (u, v)
            else:
                err = max(np.abs(u - ref[0]).max(), np.abs(v - ref[1]).max())
                assert err < 1e-9, (var, err)
    print(name)
    for var in variants:
        print("   variant %-10s ms per epoch: %s  (min %.3f)" % (var, " ".join("%.3f" % x for x in res[var]), min(res[var])), flush=True)
