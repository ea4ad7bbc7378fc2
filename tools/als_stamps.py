#!/usr/bin/env python3
"""Diagnostic: phase shares of the ALS per-row kernel (needs mfrec_amd/libmfx_diag.so built
with MFX_EXTRA_FLAGS=-DMFX_ALS_STAMPS python -m mfrec_amd.build --force)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from mfrec_amd import _shim as S, synth
S.LIB_PATH = os.environ.get("MFX_LIB") or S.LIB_PATH.replace("libmfx.so", "libmfx_diag.so")
from mfrec_amd.lib.als_implicit import als_wrmf
from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
S.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
swap = len(sys.argv) > 2 and sys.argv[2] == "swap"      # last launch then covers the low-degree side
nu, ni, nnz = (20000, 100000, 10000000) if swap else (100000, 20000, 10000000)
if len(sys.argv) > 5:                                   # explicit sizes: k mode nu ni nnz
    nu, ni, nnz = (int(x) for x in sys.argv[3:6])
idx, _ = synth.make_ratings(nu, ni, nnz)
M = sp.csr_matrix((np.ones(nnz), (idx[:, 0], idx[:, 1])), shape=(nu, ni)); M.sum_duplicates()
urow, ucol = create_bool_sparse_row(M); irow, icol = create_bool_sparse_col(M)
rs = np.random.RandomState(0)
u, v = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni)
st = np.zeros(4096 * 16, dtype=np.uint64)
assert S.lib().mfx_debug_als_stamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.reshape(4096, 16).astype(np.float64)
st = st[st[:, 4] > 0]                                   # workgroups that ran
d = np.diff(st[:, :5], axis=1)
print("k=%d, last launch (rows of degree ~%d): median ticks  syrk+gather %.0f | write A %.0f | cholesky %.0f | back-subst %.0f | total %.0f"
      % ((k, nnz // ni) + tuple(np.median(d, axis=0)) + (np.median(st[:, 4] - st[:, 0]),)))
if st[:, 5:12].min() > 0:      # rows3 sub-phase stamps (wave 0, panel step 0)
    sub = np.median(np.diff(np.concatenate([st[:, 2:3], st[:, 5:12]], axis=1), axis=1), axis=0)
    print("  step 0: factor(0)+sync %.0f | tiles->Xs+sync %.0f | row solve %.0f | sync %.0f | border+diag mfma %.0f | factor(1) %.0f | rest+sync %.0f" % tuple(sub))
