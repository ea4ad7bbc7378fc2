#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): speed of the plain-C port (oracle/mfx_oracle.c, the CPU
baseline that travels to the GPU box) against the reference's own Cython kernels compiled OUTSIDE the
repository by oracle/build_ref.py.  Same inputs, same single thread, outputs compared on the way.

    python tools/port_vs_reference.py        ->  profiles/r03_port_vs_reference.json

bench.py prints the recorded ratios next to its cpu_baseline ("kind": "port"), so that the stand-in is shown
to be neither slower nor faster than the real Cython path by more than the stated factor."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from mfrec_amd import synth  # noqa: E402
from oracle import build_ref, oracle as O  # noqa: E402


def host():
    model = "unknown"
    for ln in open("/proc/cpuinfo"):
        if ln.startswith("model name"):
            model = ln.split(":", 1)[1].strip()
            break
    return {"cpu_model": model, "nproc": os.cpu_count()}


def best(fn, reps=3):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    if not build_ref.build(verbose=False):
        sys.exit("the reference is not present here")
    ref = build_ref.load()
    O.build()
    out = {"host": host(), "note": "single thread, best of 3; ratio > 1: the port is FASTER than the reference"}

    # K1: train_linear_kernel at cfg-2 factor sizes, 2 M ratings of the bench order
    nu, ni, k, n = 100000, 50000, 64, 2000000
    idx, r = synth.make_ratings(nu, ni, n)
    u0, v0 = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    res = {}

    def run_port():
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        t0 = time.perf_counter()
        O.train_linear_kernel(1, k, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub)
        res["port"] = (time.perf_counter() - t0, u, v)

    def run_ref():
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        t0 = time.perf_counter()
        ref["kmf_train"].train_linear_kernel(1, k, 0.1, 0.01, 0.0, 0.0, 0.1, 0.1, 0.007, 0.0, u, v, idx, r, ib, ub, 1, 1, 0)
        res["ref"] = (time.perf_counter() - t0, u, v)

    tp, tr = [], []
    for _ in range(3):
        run_port(); tp.append(res["port"][0])
        run_ref(); tr.append(res["ref"][0])
    assert np.array_equal(res["port"][1], res["ref"][1]) and np.array_equal(res["port"][2], res["ref"][2])
    out["k1"] = {"workload": "train_linear_kernel, %d x %d factors, k=%d, first %d ratings of the bench order" % (nu, ni, k, n),
                 "port_updates_per_s": n / min(tp), "reference_updates_per_s": n / min(tr),
                 "port_over_reference": min(tr) / min(tp), "outputs_bit_identical": True}

    # Funk: estimator_loop_without_bias, 8 features x 1 epoch
    kk = 8
    def funk(fn, extra):
        u, v = synth.init_const(kk, ni, 0.1), synth.init_const(kk, nu, 0.1)
        t0 = time.perf_counter()
        fn(1, 1, 1e9, kk, 0.1, 0.001, 0.05, u, v, idx, r, nu, ni, *extra)
        return time.perf_counter() - t0, u, v
    fp = [funk(O.estimator_loop_without_bias, ()) for _ in range(3)]
    fr = [funk(ref["gd_estimator"].estimator_loop_without_bias, (0,)) for _ in range(3)]
    assert np.array_equal(fp[0][1], fr[0][1]) and np.array_equal(fp[0][2], fr[0][2])
    tpf, trf = min(x[0] for x in fp), min(x[0] for x in fr)
    out["funk"] = {"workload": "estimator_loop_without_bias, 8 features x 1 epoch, same ratings",
                   "port_scalar_updates_per_s": n * kk / tpf, "reference_scalar_updates_per_s": n * kk / trf,
                   "port_over_reference": trf / tpf, "outputs_bit_identical": True}

    # ALS: cfg 4 / 400 at k = 128, 1 epoch (the reference calls np.linalg.inv per row)
    import scipy.sparse as sp
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    from threadpoolctl import threadpool_limits
    anu, ani, annz, ak = 2500, 500, 250000, 128
    aidx, _ = synth.make_ratings(anu, ani, annz)
    M = sp.csr_matrix((np.ones(annz, dtype=np.int8), (aidx[:, 0], aidx[:, 1])), shape=(anu, ani))
    M.sum_duplicates()
    views = create_bool_sparse_row(M) + create_bool_sparse_col(M)
    rs = np.random.RandomState(0)
    au0, av0 = rs.normal(0, 0.1, (ak, ani)), rs.normal(0, 0.1, (ak, anu))
    with threadpool_limits(limits=1):
        au, av = au0.copy(), av0.copy()
        t0 = time.perf_counter(); O.als_wrmf(1, ak, au, av, None, None, *views, anu, ani, 1, 0.015); ta = time.perf_counter() - t0
        bu, bv = au0.copy(), av0.copy()
        m, mi = np.zeros((ak, ak)), np.zeros((ak, ak))
        t0 = time.perf_counter(); ref["als_implicit"].als_wrmf(1, ak, bu, bv, m, mi, *views, anu, ani, 1, 0.015, 0); tb = time.perf_counter() - t0
    assert np.allclose(au, bu, rtol=1e-8, atol=1e-11) and np.allclose(av, bv, rtol=1e-8, atol=1e-11)
    out["als"] = {"workload": "als_wrmf, %d x %d, %d cells, k=%d, 1 epoch" % (anu, ani, int(M.nnz), ak),
                  "port_s": ta, "reference_s": tb, "port_over_reference": tb / ta}
    path = os.path.join(ROOT, "profiles", "r03_port_vs_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
