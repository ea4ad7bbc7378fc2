#!/usr/bin/env python3
"""Batched top-N (mfx_topn_batch) at cfg 2's shape: U users x 50 K candidate items, k = 64, N = 10, against the
single-user query loop it replaces."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from mfrec_amd import _shim as S, synth
from mfrec_amd.recommendation import KMFRecommender
S.init(0)
nu, ni, nnz, k = 100000, 50000, 10000000, 64
idx, r = synth.make_ratings(nu, ni, nnz)
R = KMFRecommender(nu, ni, {"nbr_features": k})
R.relationship_matrix = sp.csr_matrix((r, (idx[:, 0], idx[:, 1])), shape=(nu, ni)).tolil()
R.svd_u, R.svd_v = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
R.items_bias, R.users_bias, R.overall_bias = np.zeros(ni), np.zeros(nu), 3.0
R.neighborhood = ni
out = {}
for U in (1000, 10000, 100000):
    users = np.arange(U, dtype=np.int32)
    t0 = time.time()
    items, scores = R.find_recommended_items_batch(users, 10, "predict_linear")
    out[U] = {"seconds": time.time() - t0, "users_per_s": U / (time.time() - t0)}
    print(U, out[U], flush=True)
from mfrec_amd.recommendation import metrics
t0 = time.time()
plan = metrics.TopNPlan(R, "predict_linear")
out["plan_create_s"] = time.time() - t0
for U in (1, 64, 1000, 100000):
    users = np.arange(U, dtype=np.int32)
    plan.query(users, 10)
    reps = 20 if U <= 1000 else 2
    t0 = time.time()
    for _ in range(reps):
        pi, ps = plan.query(users, 10)
    dt = (time.time() - t0) / reps
    out["plan_query_%d_users_ms" % U] = dt * 1e3
    if U == 100000:
        assert np.array_equal(pi, items)
print({k: v for k, v in out.items() if str(k).startswith("plan")}, flush=True)
plan.close()
t0 = time.time()
for user in range(50):
    a, b = R.find_recommended_items(user_index=user, nbr_recommendations=10, predictor="predict_linear")
    assert a == [int(x) for x in items[user] if x >= 0]
out["single_user_loop_s_per_user"] = (time.time() - t0) / 50
print(out, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/topn_timing.json", "w"), indent=1)
