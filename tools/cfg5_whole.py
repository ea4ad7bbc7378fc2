#!/usr/bin/env python3
"""Scale run: BASELINE cfg 5 WHOLE (10 M users x 1 M items, 1 B ratings, k=64) on ONE MI355X -- 16 GB of records and
5.6 GB of factor rows fit the 288 GB of one GPU several times over, so the single-GPU exact path serves it.
Reports the device-side schedule build, epochs/s and the online RMSE of consecutive epochs; with --prefix N the
first N ratings (same 10 M x 1 M tables) are also trained for one epoch and compared BIT FOR BIT with the oracle's
sequential loop (the whole billion would keep the oracle busy for ~10 minutes; a heartbeat thread keeps the run
visibly alive meanwhile).  Host RAM: ~16 GB of ratings + ~6 GB per factor copy."""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mfrec_amd import _shim as S, synth
from mfrec_amd.plan import SgdPlan

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=10000000)
ap.add_argument("--items", type=int, default=1000000)
ap.add_argument("--nnz", type=int, default=1000000000)
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--prefix", type=int, default=0, help="oracle check on the first N ratings (0 = skip)")
ap.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
a = ap.parse_args()
nu, ni, nnz, k = a.users, a.items, a.nnz, a.k
LR, KU, KI, KB = 0.01, 0.1, 0.1, 0.007

alive = [True]


def heartbeat():
    t0 = time.time()
    while alive[0]:
        time.sleep(45)
        if alive[0]:
            print("  ... %.0f s" % (time.time() - t0), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
S.init(0)
out = {"config": "cfg5 whole on one GPU: %d users x %d items, %d ratings, k=%d" % (nu, ni, nnz, k)}
t0 = time.time()
idx, r = synth.make_ratings(nu, ni, nnz, workers=a.workers)
out["data_s"] = time.time() - t0
print("data %.1f s" % out["data_s"], flush=True)
u0, v0 = synth.init_normal(k, ni, 7), synth.init_normal(k, nu, 8)
print("init factors done", flush=True)

t0 = time.time()
with SgdPlan(idx, r, nu, ni, k) as p:
    info = p.info()
    out["plan_create_s"] = time.time() - t0
    out["levels"] = info["n_levels"]
    out["launches_per_epoch"] = info["launches_per_epoch"]
    out["schedule_builder"] = info["schedule_builder"]
    out["schedule_build_ms"] = info["schedule_us"] / 1e3
    out["device_bytes"] = info.get("device_bytes")
    out["kernel"] = p.kernel_name(0)
    print("plan %.1f s, %d levels, builder %s" % (out["plan_create_s"], info["n_levels"], info["schedule_builder"]), flush=True)
    p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    rm = p.run_kmf(0, 1, LR, KU, KI, KB)                       # warm-up epoch (graph instantiation)
    S.sync()
    t0 = time.time()
    rm2 = p.run_kmf(0, a.epochs, LR, KU, KI, KB)
    S.sync()
    dt = time.time() - t0
    out["epoch_s"] = dt / a.epochs
    out["rating_updates_per_s"] = nnz * a.epochs / dt
    out["rmse_online"] = [float(rm[0])] + [float(x) for x in rm2]
    out["rmse_decreasing"] = bool(all(x > y for x, y in zip(out["rmse_online"][:-1], out["rmse_online"][1:])))
print(json.dumps(out), flush=True)

if a.prefix:
    from oracle import oracle as O
    n = min(a.prefix, nnz)
    pidx, pr = np.ascontiguousarray(idx[:n]), np.ascontiguousarray(r[:n])
    del idx, r
    u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
    with SgdPlan(pidx, pr, nu, ni, k) as p:
        p.set_factors(u, v, ib, ub)
        rmg = p.run_kmf(0, 1, LR, KU, KI, KB)
        p.get_factors(u, v, ib, ub)
    print("prefix gpu done", flush=True)
    u2, v2, ib2, ub2, rmo = u0, v0, np.zeros(ni), np.zeros(nu), np.zeros(1)       # the oracle trains in place
    t0 = time.time()
    O.train_linear_kernel(1, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u2, v2, pidx, pr, ib2, ub2, 1, 1, 0, rmo)
    out["prefix"] = {"ratings": n, "oracle_s": time.time() - t0,
                     "bit_identical": bool(np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(ib, ib2)
                                           and np.array_equal(ub, ub2)),
                     "rmse_gpu": float(rmg[0]), "rmse_oracle": float(rmo[0])}
    print(json.dumps(out), flush=True)
alive[0] = False
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cfg5_whole.json", "w"), indent=1)
sys.exit(0 if (out["rmse_decreasing"] and (not a.prefix or out["prefix"]["bit_identical"])) else 1)
