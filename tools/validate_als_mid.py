#!/usr/bin/env python3
"""One-off: ALS at k=128 on 20 K x 4 K with 1 M ratings (cfg 4 / 100), one epoch, against the oracle (1e-8)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from mfrec_amd import _shim as S, synth
from mfrec_amd.lib.als_implicit import als_wrmf
from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
from oracle import oracle as O
S.init(0)
nu, ni, nnz, k = 20000, 4000, 1000000, 128
idx, _ = synth.make_ratings(nu, ni, nnz)
M = sp.csr_matrix((np.ones(nnz), (idx[:, 0], idx[:, 1])), shape=(nu, ni)); M.sum_duplicates(); M.data[:] = 1.0
views = create_bool_sparse_row(M) + create_bool_sparse_col(M)
rs = np.random.RandomState(0)
u0, v0 = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
u, v, u2, v2 = u0.copy(), v0.copy(), u0.copy(), v0.copy()
t0 = time.time(); als_wrmf(1, k, u, v, None, None, *views, nu, ni, 1, 0.015); tg = time.time() - t0
t0 = time.time(); O.als_wrmf(1, k, u2, v2, None, None, *views, nu, ni, 1, 0.015); to = time.time() - t0
du = float(np.max(np.abs(u - u2) / (np.abs(u2) + 1e-3))); dv = float(np.max(np.abs(v - v2) / (np.abs(v2) + 1e-3)))
ok = bool(np.allclose(u, u2, rtol=1e-8, atol=1e-11) and np.allclose(v, v2, rtol=1e-8, atol=1e-11))
print(json.dumps({"config": "ALS k=128, 20K x 4K, 1M ratings, 1 epoch", "within_1e-8": ok, "max_rel_diff_u": du, "max_rel_diff_v": dv,
                  "gpu_seconds": tg, "oracle_seconds": to}))
sys.exit(0 if ok else 1)
