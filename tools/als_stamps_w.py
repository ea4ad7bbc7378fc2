#!/usr/bin/env python3
"""Diagnostic (libmfx_diag.so with -DMFX_ALS_STAMPS [-DMFX_ALS_STAMP_BASE=N]): phase times of k_als_rowsw rows."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from mfrec_amd import _shim as S, synth
S.LIB_PATH = os.environ.get("MFX_LIB") or S.LIB_PATH.replace("libmfx.so", "libmfx_diag.so")
from mfrec_amd.lib.als_implicit import als_wrmf
from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
S.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
os.environ["MFX_ALS_KERNEL"] = "rowsw"
nu, ni, nnz = 20000, 100000, 10000000          # the LAST launch (item pass) covers 100 K rows of degree ~100
idx, _ = synth.make_ratings(nu, ni, nnz)
M = sp.csr_matrix((np.ones(nnz), (idx[:, 0], idx[:, 1])), shape=(nu, ni)); M.sum_duplicates()
urow, ucol = create_bool_sparse_row(M); irow, icol = create_bool_sparse_col(M)
rs = np.random.RandomState(0)
u, v = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni)
st = np.zeros(4096 * 16, dtype=np.uint64)
assert S.lib().mfx_debug_als_stamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.reshape(4096, 16)
st = st[st[:, 4] > 0].astype(np.float64)
print("k=%d rows stamped %d" % (k, len(st)))
seq = [0, 1, 2, 5, 6, 7, 8, 9, 10, 3, 4]
names = ["gather+syrk", "nops+assemble", "factor(0)", "panel0: tiles->Xs", "panel0: solve", "panel0: readback+operands+diag mfma+border", "panel0: factor(1)",
         "panel0: rest of trailing", "panels 1..%d" % (k // 16 - 1), "back-subst+store"]
d = np.diff(st[:, seq], axis=1)
for n, m in zip(names, np.median(d, axis=0)):
    print("  %-46s %8.0f" % (n, m))
print("  %-46s %8.0f" % ("row total", np.median(st[:, 4] - st[:, 0])))
