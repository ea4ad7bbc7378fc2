#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): Funk passes (K2) and ALS (K3)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

from mfrec_amd import _shim as S, synth  # noqa: E402
if os.environ.get("MFX_LIB"):
    S.LIB_PATH = os.environ["MFX_LIB"]
from mfrec_amd.plan import SgdPlan, Timer  # noqa: E402


def funk(a):
    nu, ni, nnz, k = a.users, a.items, a.nnz, a.k
    idx, r = synth.make_ratings(nu, ni, nnz)
    if a.balanced:
        from mfrec_amd.lib.datasets import balanced_order
        perm, _ = balanced_order(idx, nu, ni, seed=11)
        idx, r = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    plan = SgdPlan(idx, r, nu, ni, k)
    plan.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
    E = a.epochs
    plan.run_funk(1, 1e9, 0.1, 0.001, 0.05, epoch_cap=1)          # warm-up (k passes)
    plan.set_factors(synth.init_const(k, ni), synth.init_const(k, nu))
    t0 = time.perf_counter()
    hist = plan.run_funk(E, 1e9, 0.1, 0.001, 0.05)
    dt = time.perf_counter() - t0
    n = len(hist)
    print(json.dumps({"family": "funk (estimator_loop_without_bias)", "nu": nu, "ni": ni, "nnz": nnz,
                      "k": k, "feature_epochs": n, "seconds": dt,
                      "scalar_updates_per_s": nnz * n / dt, "row_equivalents_per_s": nnz * n / dt / k,
                      "ms_per_pass": dt * 1e3 / n, "levels": plan.info()["n_levels"],
                      "rmse_last": float(hist[-1])}))
    plan.close()


def als(a):
    from mfrec_amd.lib.als_implicit import als_wrmf
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    nu, ni, nnz, k = a.users, a.items, a.nnz, a.k
    idx, _ = synth.make_ratings(nu, ni, nnz)
    M = sp.csr_matrix((np.ones(nnz), (idx[:, 0], idx[:, 1])), shape=(nu, ni))
    M.sum_duplicates()
    urow, ucol = create_bool_sparse_row(M)
    irow, icol = create_bool_sparse_col(M)
    nz = int(M.nnz)
    rs = np.random.RandomState(0)
    u, v = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    als_wrmf(1, k, u, v, None, None, urow, ucol, irow, icol, nu, ni)      # warm-up incl. transfers
    t0 = time.perf_counter()
    als_wrmf(a.epochs, k, u, v, None, None, urow, ucol, irow, icol, nu, ni)
    dt = time.perf_counter() - t0
    dev_s = S.lib().mfx_last_compute_ms() * 1e-3
    # SURVEY 8(d) FLOP model per epoch
    F = 2 * (k * (k + 1) * nz) + 2 * k * k * (nu + ni) + (nu + ni) * (k ** 3 / 3 + 2 * k * k) + 4 * k * nz
    print(json.dumps({"family": "als_wrmf", "nu": nu, "ni": ni, "nnz_distinct": nz, "k": k,
                      "epochs": a.epochs, "seconds_incl_transfers": dt,
                      "device_s_per_epoch": dev_s / a.epochs, "rows_per_s": (nu + ni) * a.epochs / dev_s,
                      "model_tflop_per_epoch": F / 1e12, "model_tflops": F * a.epochs / dev_s / 1e12,
                      "mfma_f64_peak_frac": F * a.epochs / dev_s / 78.6e12,
                      "finite": bool(np.isfinite(u).all() and np.isfinite(v).all())}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("family", choices=["funk", "als"])
    ap.add_argument("--users", type=int, default=100000)
    ap.add_argument("--items", type=int, default=50000)
    ap.add_argument("--nnz", type=int, default=10000000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--balanced", action="store_true", help="balanced input order (fewest levels)")
    a = ap.parse_args()
    S.init(0)
    funk(a) if a.family == "funk" else als(a)
