#!/usr/bin/env python3
"""The 100 K-user resident-model top-N query (cfg 2 shape) N times: every run must return the same rows bit for bit, and a sample of users must
equal the sort path's rows (MFX_TOPN_FUSED=0) -- the determinism / exactness check at the size the kernel is timed at (whole rounds + split tail,
bars, second fused pass).  python tools/stress_topn_big.py [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from mfrec_amd import _shim as S, synth
from mfrec_amd.recommendation import KMFRecommender, metrics
S.init(0)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nu, ni, nnz, k = 100000, 50000, 10000000, 64
idx, r = synth.make_ratings(nu, ni, nnz)
R = KMFRecommender(nu, ni, {"nbr_features": k})
R.relationship_matrix = sp.csr_matrix((r, (idx[:, 0], idx[:, 1])), shape=(nu, ni)).tolil()
R.svd_u, R.svd_v = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
R.items_bias, R.users_bias, R.overall_bias = synth.init_normal(1, ni, 9)[0] * 3.0, synth.init_normal(1, nu, 10)[0] * 3.0, 3.0
R.neighborhood = ni
bad = 0
for pred in ("predict_linear", "predict_logistic"):
    plan = metrics.TopNPlan(R, pred)
    users = np.arange(nu, dtype=np.int32)
    i0, s0 = plan.query(users, 10)
    for t in range(runs - 1):
        i1, s1 = plan.query(users, 10)
        if not (np.array_equal(i0, i1) and np.array_equal(s0, s1)):
            bad += 1; print("MISMATCH between runs", pred, t, int((i0 != i1).any(1).sum()), "rows")
    sample = np.concatenate([np.arange(0, 2000), np.arange(98304 - 500, nu)]).astype(np.int32)
    os.environ["MFX_TOPN_FUSED"] = "0"
    si, ss = plan.query(sample, 10)
    os.environ.pop("MFX_TOPN_FUSED")
    if not (np.array_equal(i0[sample], si) and np.array_equal(s0[sample], ss)):
        bad += 1; print("MISMATCH against the sort path", pred, int((i0[sample] != si).any(1).sum()), "rows")
    plan.close()
print("runs", runs, "per predictor; mismatches", bad)
sys.exit(1 if bad else 0)
