#!/usr/bin/env python3
"""bench.py -- rating-updates/s of the full-k SGD hot path (train_linear_kernel semantics)
on BASELINE.json configs[1]: synthetic 100K users x 50K items, 10M ratings, k=64, f64.

A "step" is one epoch (one pass of the hot path over all ratings of this rank's shard).
Inputs (ratings in level-major order, schedule, factors) are resident in HBM when the
timed region starts.  At N > 1 (one process per GPU, torchrun) every rank holds a cfg-2
sized user shard (weak scaling) and the item factors are all-reduced over RCCL each epoch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL's cross-process buffer sharing needs this
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
LR, KU, KI, KB = 0.01, 0.1, 0.1, 0.007        # kmf.py:51-56 defaults


def algorithmic_bytes(k, nnz, nu, s=8, rec=16):
    """B_read = R + k*s + s + (k*s+s)/d_u, B_total adds the writes."""
    d_u = nnz / float(nu)
    b_read = rec + k * s + s + (k * s + s) / d_u
    b_total = b_read + k * s + s + (k * s + s) / d_u
    return b_read, b_total


def kernel_label(k, batch, storage):
    """Name of the k1_level instantiation mfx_sgd.hip's k1_pick selects (KP lanes per rating, features
    per lane, ratings per wave; 'vec16+wt' = 16-byte accesses per lane with write-through row stores)."""
    vec = int(os.environ.get("MFX_K1_VEC", "2") or 0)
    tag = "vec16+wt" if vec == 2 else "vec16"
    if storage == "f64" and vec and k % 2 == 0 and k > 4 and batch == (16 if k <= 32 else 8 if k <= 128 else 4 if k <= 256 else 2):
        return "k1_level<%d,%d,%d,linear,%s>" % (8 if k <= 16 else 16 if k <= 32 else 32 if k <= 64 else 64,
                                                 2 if k <= 128 else 4 if k <= 256 else 8, batch, tag)
    if storage == "f32" and vec and k % 4 == 0 and k > 8 and batch == (16 if k <= 32 else 8 if k <= 128 else 4 if k <= 256 else 2):
        return "k1_level<%d,%d,%d,linear,f32,%s>" % (8 if k <= 32 else 16 if k <= 64 else 32 if k <= 128 else 64,
                                                     4 if k <= 256 else 8, batch, tag)
    return "k1_level<%d,%d,%d,linear>" % ((8 if k <= 8 else 16 if k <= 16 else 32 if k <= 32 else 64),
                                          (1 if k <= 64 else 2 if k <= 128 else 4 if k <= 256 else 8), batch)


def cpu_baseline(idx, r, nu, ni, k, budget_s=12.0):
    """The reference CPU path timed on this host, 1 thread (the reference is single-threaded:
    GIL held, no OpenMP), on a bounded prefix of the same workload with the full-size factor
    arrays allocated.  kind = "reference": the reference's own Cython kernel
    (train_linear_kernel, kmf_train.pyx:195-277) built into oracle/_ref by oracle/build_ref.py,
    when that binary travelled with the snapshot; otherwise kind = "port": the plain-C
    restatement oracle/mfx_oracle.c (bit-identical output)."""
    from oracle import oracle as O
    from mfrec_amd import synth
    u, v = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    ib, ub = np.zeros(ni), np.zeros(nu)

    def port(n):
        O.train_linear_kernel(1, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u, v, idx[:n], r[:n], ib, ub)

    ref_fn = None
    try:
        from oracle import build_ref
        ref = build_ref.load()
        if ref is not None:
            def ref_fn(n):
                ref["kmf_train"].train_linear_kernel(1, k, 0.1, LR, 0.0, 0.0, KU, KI, KB, 0.0, u, v,
                                                     idx[:n], r[:n], ib, ub, 1, 1, 0)
    except Exception:
        ref_fn = None

    def timed(fn, budget):
        n = min(idx.shape[0], 200000)
        t0 = time.perf_counter(); fn(n); rate = n / (time.perf_counter() - t0)
        n2 = int(min(idx.shape[0], max(n, rate * budget)))
        t0 = time.perf_counter(); fn(n2); dt = time.perf_counter() - t0
        return n2 / dt, n2, dt

    def host_info():
        model = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        return {"cpu_model": model, "nproc": os.cpu_count()}

    p_rate, p_n, p_dt = timed(port, budget_s if ref_fn is None else 5.0)
    out = {"value": p_rate, "unit": "rating-updates/s", "cores": 1, "kind": "port", "host": host_info(),
           "sample": "first %d of %d ratings (%s), 1 epoch, full-size factors, "
                     "oracle/mfx_oracle.c single thread, %.1f s" % (p_n, idx.shape[0], "bench order", p_dt)}
    if ref_fn is not None:
        try:
            r_rate, r_n, r_dt = timed(ref_fn, budget_s)
            out = {"value": r_rate, "unit": "rating-updates/s", "cores": 1, "kind": "reference",
                   "host": host_info(),
                   "sample": "first %d of %d ratings (bench order), 1 epoch, full-size factors, the "
                             "reference's Cython train_linear_kernel (oracle/_ref), single thread, "
                             "%.1f s" % (r_n, idx.shape[0], r_dt),
                   "port_value": p_rate}
        except Exception as e:      # never let the reference leg break the bench line
            out["reference_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--users", type=int, default=100000, help="users per GPU")
    ap.add_argument("--items", type=int, default=50000)
    ap.add_argument("--nnz", type=int, default=10000000, help="ratings per GPU")
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--order", choices=["shuffled", "user-major", "balanced"], default="shuffled")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help=argparse.SUPPRESS)   # accepted, no-op
    ap.add_argument("--extra-balanced", action="store_true",
                    help="also time the same kernel on a balanced input order (reported under 'also')")
    ap.add_argument("--storage", choices=["f64", "f32"], default="f64",
                    help="f32 = NOT the reference's precision: rows stored as f32 (measured option only)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torchrun (one process per GPU)" % a.gpus)
        a.gpus = world

    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    from mfrec_amd import synth
    from mfrec_amd.plan import SgdPlan, Timer, LINEAR
    uid = None
    if world > 1:
        # torch.distributed.run is only the launcher: no torch import in the ranks (see
        # mfrec_amd/rendezvous.py); the RCCL unique id travels through a file.  Rank 0 builds (a
        # no-op when libmfx.so is current) before it publishes the id, the other ranks wait for the
        # id before they touch the library: eight ranks never compile into the same files at once.
        import ctypes as C
        from mfrec_amd import rendezvous
        if rank != 0:
            uid = rendezvous.exchange_unique_id(rank, world, None)
    B.build(verbose=False)
    # MFX_BENCH_DEVICE: rehearsal hook (several ranks on one GPU); the driver never sets it
    S.init(int(os.environ.get("MFX_BENCH_DEVICE", local)))

    if world > 1:
        def make_id():
            idbuf = C.create_string_buffer(S.UNIQUE_ID_BYTES)
            S.check(S.lib().mfx_comm_unique_id(idbuf))
            return idbuf.raw

        if rank == 0:
            uid = rendezvous.exchange_unique_id(rank, world, make_id)
        S.check(S.lib().mfx_comm_init(rank, world, uid))

    nu, ni, nnz, k = a.users, a.items, a.nnz, a.k
    # this rank's shard: its own users (local ids 0..nu-1), the shared item catalogue
    idx, r = synth.make_ratings(nu, ni, nnz, t0=rank * nnz, user_base=0)
    if a.order == "user-major":
        idx, r = synth.user_major(idx, r)
    elif a.order == "balanced":
        from mfrec_amd.lib.datasets import balanced_order_gpu
        perm, _ = balanced_order_gpu(idx, nu, ni)          # recolours the (already shuffled) synthetic order
        idx, r = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    u0 = synth.init_normal(k, ni, synth.SEED_INIT)
    v0 = synth.init_normal(k, nu, synth.SEED_INIT + 1 + rank)
    plan = SgdPlan(idx, r, nu, ni, k, storage=a.storage)
    plan.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
    info = plan.info()

    def run(n):
        if world > 1:
            plan.run_kmf_sharded(LINEAR, n, LR, KU, KI, KB, 1.0 / world)
        else:
            plan.run_kmf(LINEAR, n, LR, KU, KI, KB, want_rmse=False)

    def barrier():
        S.sync()                                      # hipStreamSynchronize on the library stream
        if world > 1:
            S.check(S.lib().mfx_comm_barrier())       # RCCL all-reduce + sync = cross-rank barrier

    run(max(a.warmup, 0))
    barrier()
    t0 = time.perf_counter()
    with Timer() as tm:
        run(a.steps)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:                                     # MAX over ranks
        import ctypes as C
        w = C.c_double(wall)
        S.check(S.lib().mfx_comm_allreduce_max(C.byref(w)))
        wall = w.value

    # Dominant kernel = k1_level (one launch per dependency level).  Its average launch
    # duration is taken from the HIP events that bracket the timed region on the library's
    # stream: device time of the K epochs / (K * launches per epoch).  The epoch graph holds
    # nothing else but two tiny reduction kernels, so this is an upper bound of the kernel's
    # own duration (inter-kernel gaps included) and agrees with rocprofv3's average.
    launches = info["launches_per_epoch"]
    b_read, b_total = algorithmic_bytes(k, nnz, nu)
    avg_launch_s = tm.ms * 1e-3 / (a.steps * launches)
    achieved = (b_read * nnz / launches) / avg_launch_s
    # cross-check: one instrumented epoch with an event pair around every launch
    kern_ms, _ = plan.profile_kmf(LINEAR, LR, KU, KI, KB)

    # HBM traffic per launch: from the committed rocprofv3 PMC summary of this same command
    # (profiles/rNN_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)
    traffic = traffic_src = None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        if cands and a.order == "shuffled" and a.storage == "f64" and (nu, ni, nnz, k) == (100000, 50000, 10000000, 64):
            tj = json.load(open(cands[-1]))
            traffic = (tj["read_bytes_per_rating"] + tj["write_bytes_per_rating"]) * nnz / launches
            traffic_src = os.path.basename(cands[-1])
    except Exception:
        traffic = None

    if rank == 0:
        value = world * nnz * a.steps / wall
        line = {
            "metric": "rating-updates/s per epoch at k=%d" % k, "value": value,
            "unit": "rating-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": wall * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if a.storage == "f64" else "f64 arithmetic on f32-stored rows (NOT the reference precision)",
            "data": "synthetic",
            "config": {"workload": "%s: synthetic %dK users x %dK items, %dM ratings per GPU, "
                                   "k=%d, full-k SGD (train_linear_kernel), %s order, "
                                   "exact level schedule" % (
                                       "cfg2" if (nu, ni, nnz, k) == (100000, 50000, 10000000, 64) else "custom",
                                       nu // 1000, ni // 1000, nnz // 1000000, k, a.order),
                       "users_per_gpu": nu, "items": ni, "ratings_per_gpu": nnz, "k": k,
                       "levels": info["n_levels"], "launches_per_epoch": info["launches_per_epoch"],
                       "sharding": "users x%d, item-delta all-reduce (RCCL)" % world if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC, %s)" % traffic_src,
                         "algorithmic_bytes_per_launch": b_read * nnz / launches,
                         "kernel": kernel_label(k, info["ratings_per_wave"], a.storage),
                         "bytes_per_update_read": b_read, "bytes_per_update_total": b_total,
                         "avg_launch_us": avg_launch_s * 1e6, "launches_per_epoch": launches,
                         "event_pair_avg_launch_us": kern_ms * 1e3 / launches,
                         "frac_total_bytes": (b_total * nnz / launches) / avg_launch_s / HBM_PEAK},
            "device_ms_per_step": tm.ms / a.steps,
            "schedule_build_ms": info["schedule_us"] / 1e3,
        }
        if world == 1 and a.order == "shuffled" and a.storage == "f64" and a.extra_balanced:
            # secondary, clearly labelled data point (not `value`): the same exact kernel on a
            # balanced input order (fewest dependency levels)
            try:
                from mfrec_amd.lib.datasets import balanced_order_gpu
                perm, rounds = balanced_order_gpu(idx, nu, ni)
                with SgdPlan(np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm]), nu, ni, k) as p2:
                    p2.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
                    p2.run_kmf(LINEAR, 1, LR, KU, KI, KB, want_rmse=False)
                    with Timer() as t2:
                        p2.run_kmf(LINEAR, a.steps, LR, KU, KI, KB, want_rmse=False)
                line["also"] = {"balanced_input_order": {"value": nnz * a.steps / (t2.ms * 1e-3),
                                                         "unit": "rating-updates/s", "levels": rounds,
                                                         "ms_per_step": t2.ms / a.steps}}
            except Exception as e:                  # never let the extra break the bench line
                line["also"] = {"balanced_input_order_error": repr(e)}
        if not a.no_cpu and world == 1:          # rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(idx, r, nu, ni, k)
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        S.check(S.lib().mfx_comm_barrier())
        S.check(S.lib().mfx_comm_destroy())
        from mfrec_amd import rendezvous
        rendezvous.cleanup(rank)


if __name__ == "__main__":
    main()
