#!/usr/bin/env python3
"""One serving-form top-N query of 100 K users (cfg 2 shape) -- run under `rocprofv3 --kernel-trace --stats` for the kernels' own times;
prints how many users the fused path served and how many went through the sort path."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from mfrec_amd import _shim as S, synth
from mfrec_amd.recommendation import KMFRecommender, metrics
S.init(0)
nu, ni, nnz, k = 100000, int(os.environ.get("NI", "50000")), 10000000, 64
idx, r = synth.make_ratings(nu, ni, nnz)
R = KMFRecommender(nu, ni, {"nbr_features": k})
R.relationship_matrix = sp.csr_matrix((r, (idx[:, 0], idx[:, 1])), shape=(nu, ni)).tolil()
R.svd_u, R.svd_v = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
R.items_bias, R.users_bias, R.overall_bias = np.zeros(ni), np.zeros(nu), 3.0
R.neighborhood = ni
plan = metrics.TopNPlan(R, "predict_linear")
users = np.arange(int(os.environ.get("NQ", str(nu))), dtype=np.int32)       # NQ: users per query (default: all)
plan.query(users[:1000], 10)
s0 = (C.c_int64 * 2)(); S.check(S.lib().mfx_topn_stats(s0))
t0 = time.time()
for _ in range(3):
    plan.query(users, 10)
dt = (time.time() - t0) / 3
s1 = (C.c_int64 * 2)(); S.check(S.lib().mfx_topn_stats(s1))
nq = users.shape[0]
print("ni = %d: %.3f ns per (user, item) pair; " % (ni, dt * 1e9 / (nq * ni)) + "query of %d users: %.1f ms; fused-served %d, redone by the sort path %d (3 queries)" % (nq, dt * 1e3, s1[0] - s0[0], s1[1] - s0[1]))
plan.close()
