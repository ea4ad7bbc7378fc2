#!/usr/bin/env python3
"""Diagnostic (libmfx_diag.so built with -DMFX_ALS_STAMPS -DMFX_ALS_STAMP_BASE=N): do the rows that share a CU run their
matrix phases (gather + SYRK) at the same time?  For every stamped workgroup: how many OTHER stamped workgroups of the same CU
were inside their SYRK phase, on average, while it was inside its own."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from mfrec_amd import _shim as S, synth
S.LIB_PATH = os.environ.get("MFX_LIB") or S.LIB_PATH.replace("libmfx.so", "libmfx_diag.so")
from mfrec_amd.lib.als_implicit import als_wrmf
from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
S.init(0)
k = 128
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
ok = st[:, 4] > 0
st = st[ok]
t = st[:, :5].astype(np.float64)
cu = (st[:, 13].astype(np.int64) << 16) | (st[:, 12].astype(np.int64) & 0x7f00)
print("stamped workgroups: %d on %d CUs" % (len(st), len(set(cu.tolist()))))
d = np.diff(t, axis=1)
print("median ticks: syrk %.0f | assemble %.0f | cholesky %.0f | back-subst %.0f | row %.0f" % (tuple(np.median(d, axis=0)) + (np.median(t[:, 4] - t[:, 0]),)))
by = collections.defaultdict(list)
for i in range(len(st)):
    by[cu[i]].append(i)
ov_syrk, ov_any, conc = [], [], []
for key, ids in by.items():
    for i in ids:
        a0, a1 = t[i, 0], t[i, 1]
        o = 0.0; c = 0.0
        for j in ids:
            if j == i:
                continue
            o += max(0.0, min(a1, t[j, 1]) - max(a0, t[j, 0]))          # j's SYRK phase inside i's SYRK phase
            c += max(0.0, min(a1, t[j, 4]) - max(a0, t[j, 0]))          # j resident at all
        ov_syrk.append(o / (a1 - a0)); ov_any.append(c / (a1 - a0))
print("while a row is in its SYRK phase: %.2f other rows of its CU are in THEIR SYRK phase on average (of %.2f other rows resident); "
      "if phases were independent: %.2f" % (np.mean(ov_syrk), np.mean(ov_any), np.mean(ov_any) * np.median(d[:, 0]) / np.median(t[:, 4] - t[:, 0])))
# start-time offsets between consecutive workgroups of a CU
offs = []
for key, ids in by.items():
    s0 = np.sort(t[ids, 0])
    offs.extend(np.diff(s0).tolist())
offs = np.array(offs)
print("start-time gaps between consecutive rows of a CU (ticks): median %.0f, p10 %.0f, p90 %.0f; share below 5000 ticks: %.2f" % (
    np.median(offs), np.percentile(offs, 10), np.percentile(offs, 90), np.mean(offs < 5000)))

if st[:, 14].min() > 0 and st[:, 5].min() > 0:
    tt = st.astype(np.float64)
    seq = np.stack([tt[:, 0], tt[:, 14], tt[:, 5], tt[:, 6], tt[:, 7], tt[:, 8], tt[:, 1]], 1)
    dd = np.median(np.diff(seq, axis=1), axis=0)
    print("inside SYRK (median ticks): col ids %.0f | first gather + store %.0f | chunk 0 mfma + next gather + store %.0f | chunk 1 %.0f | chunk 2 %.0f | "
          "chunk 3 (4 rows) + drain %.0f" % tuple(dd))
