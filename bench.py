#!/usr/bin/env python3
"""bench.py -- rating-updates/s of the full-k SGD hot path (train_linear_kernel semantics).

N = 1 (default):  BASELINE.json configs[1] -- synthetic 100K users x 50K items, 10M ratings, k=64, f64,
    shuffled order, exact (bit-identical to the sequential loop); the library picks the epoch kernel (cfg 2:
    the static resident-side epoch, ONE persistent launch per epoch).  `value` is that.  The same JSON line
    carries an `also` block with the other kernels of the path, each with its own roofline and CPU baseline
    (SURVEY 8(d)): K1 on the level schedule, on the user-major and the balanced order, K2 (Funk per-feature
    pass) on cfg 2, K3 (ALS) on cfg 4 at full size, K1 on cfg 3 whole in shuffled order (the 1-GPU point of the
    strong-scaling curve below) and in user-major order (user rows resident: also.k1_user_resident_cfg3).
N > 1 (one process per GPU under torch.distributed.run): BASELINE.json configs[2] -- 1M users x 200K
    items, 100M ratings, k=64 -- STRONG scaling: the same 100M ratings are sharded by user over the N
    GPUs, item factors replicated, RCCL all-reduce of the item deltas every epoch (north_star's scheme:
    `value`); the exact block-rotation mode (DSGD, ncclSend/Recv ring) is timed next to it under `also`.

A "step" is one epoch (one pass of the hot path over all ratings of the job).  Inputs (ratings in
level-major order, schedule, factors) are resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL's cross-process buffer sharing needs this
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
F64_MFMA_PEAK = 78.6e12    # FLOP/s, SURVEY 8(d) (datasheet f64 matrix peak)
LR, KU, KI, KB = 0.01, 0.1, 0.1, 0.007        # kmf.py:51-56 defaults
FUNK_LR, FUNK_K, FUNK_INIT = 0.001, 0.05, 0.1  # gradient_descent.py:80-84 defaults
CFG2 = (100000, 50000, 10000000, 64)
CFG3 = (1000000, 200000, 100000000, 64)
# one producer-store -> consumer-poll hop through the coherent path, other CU (tools/hop_latency.hip); overridden by the
# newest profiles/rNN_hop_latency.json
HOP_US_DEFAULT = 0.6


def algorithmic_bytes(k, nnz, nu, s=8, rec=16):
    """B_read = R + k*s + s + (k*s+s)/d_u, B_total adds the writes."""
    d_u = nnz / float(nu)
    b_read = rec + k * s + s + (k * s + s) / d_u
    b_total = b_read + k * s + s + (k * s + s) / d_u
    return b_read, b_total


def host_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        share = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        share = os.cpu_count()
    return {"cpu_model": model, "nproc": os.cpu_count(), "cpu_share": share}


def port_vs_reference():
    """Speed of oracle/mfx_oracle.c relative to the reference's own compiled Cython kernels, recorded ONCE in the
    build container by tools/port_vs_reference.py (the reference never travels to the GPU box)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_port_vs_reference.json")) as f:
            d = json.load(f)
        return {"k1_port_over_reference": d["k1"]["port_over_reference"],
                "funk_port_over_reference": d["funk"]["port_over_reference"],
                "als_port_over_reference": d["als"]["port_over_reference"],
                "measured_on": d.get("host", {}).get("cpu_model"), "source": "profiles/r03_port_vs_reference.json"}
    except Exception:
        return None


def timed_prefix(fn, n_total, budget_s, n_probe=200000):
    """fn(n) processes the first n ratings; size n so that the run takes about budget_s."""
    n = min(n_total, n_probe)
    t0 = time.perf_counter(); fn(n); rate = n / (time.perf_counter() - t0)
    n2 = int(min(n_total, max(n, rate * budget_s)))
    t0 = time.perf_counter(); fn(n2); dt = time.perf_counter() - t0
    return n2 / dt, n2, dt


# ------------------------------------------------------------------------------------------------
# CPU baselines (test infrastructure used as the measured-beside baseline; never in the product path)
# ------------------------------------------------------------------------------------------------
def cpu_baseline_k1(idx, r, nu, ni, k, budget_s=10.0, replicas=True):
    """train_linear_kernel (kmf_train.pyx:195-277), 1 thread (the reference is single-threaded: GIL held,
    no OpenMP), on a bounded prefix of the same workload with the full-size factor arrays allocated.
    kind "port" = oracle/mfx_oracle.c, whose output is bit-identical to the reference's Cython kernel
    (oracle/make_golden.py); `port_vs_reference` = the speed ratio of the two measured in the build container."""
    from oracle import oracle as O
    from mfrec_amd import synth
    u, v = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    ib, ub = np.zeros(ni), np.zeros(nu)

    def port(n):
        O.train_linear_kernel(1, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u, v, idx[:n], r[:n], ib, ub)

    p_rate, p_n, p_dt = timed_prefix(port, idx.shape[0], budget_s)
    out = {"value": p_rate, "unit": "rating-updates/s", "cores": 1, "kind": "port", "host": host_info(),
           "sample": "first %d of %d ratings (bench order), 1 epoch, full-size factors, "
                     "oracle/mfx_oracle.c single thread, %.1f s" % (p_n, idx.shape[0], p_dt),
           "port_vs_reference": port_vs_reference()}
    pv = out["port_vs_reference"]
    if pv:      # what the reference's own Cython kernel would run at on this host, by the recorded ratio
        out["reference_estimate"] = p_rate / pv["k1_port_over_reference"]
    if replicas:
        try:
            out["replicas"] = cpu_replica_aggregate(nu, ni, k)
        except Exception as e:
            out["replicas"] = {"error": repr(e)}
    return out


def physical_cores():
    """Physical cores this job may use: distinct (package, core) pairs among the allowed CPUs."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    seen = set()
    for c in allowed:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            seen.add(("?", str(c)))
    return max(1, len(seen)), len(allowed)


def cpu_quota():
    """CPUs this job may actually burn: the cgroup bandwidth limit (v2 cpu.max, v1 cfs quota / period), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def cpu_replica_aggregate(nu, ni, k, n_ratings=1500000):
    """SURVEY 8(d) "N-process replica aggregate": N independent single-thread replicas of the same kernel, N =
    the PHYSICAL cores this job may use (the affinity mask's distinct cores, capped by the cgroup's CPU bandwidth
    quota and at 128 processes; a replica holds ~120 MB at cfg 2), each on its own copy of the factors and the first n_ratings ratings, started together; aggregate =
    N * n_ratings / slowest replica.  A generous upper bound for "the reference on all host cores" (the
    reference itself has no parallel path)."""
    cores, threads = physical_cores()
    quota = cpu_quota()
    # one replica per physical core this job can actually run on: the cgroup's CPU bandwidth caps it (measured on this
    # pool: 128 replicas under a 16-CPU quota aggregate to 6 M/s, 16 replicas to 16 M/s)
    n = max(1, min(128, cores, int(quota) if quota else cores))
    start = time.time() + 12.0 + 0.05 * n
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-replica", str(n_ratings),
                               "--users", str(nu), "--items", str(ni), "--k", str(k), "--start-at", repr(start)],
                              stdout=subprocess.PIPE, env=dict(os.environ, OMP_NUM_THREADS="1"))
             for _ in range(n)]
    res = []
    for p in procs:
        o, _ = p.communicate(timeout=240)
        res.append(json.loads(o.decode().strip().splitlines()[-1]))
    slowest = max(x["seconds"] for x in res)
    late = max(x["late_s"] for x in res)
    return {"value": n * n_ratings / slowest, "unit": "rating-updates/s", "processes": n,
            "physical_cores": cores, "hw_threads": threads, "cgroup_cpu_quota": quota,
            "kind": "port", "ratings_per_replica": n_ratings, "slowest_replica_s": slowest,
            "start_skew_s": late,
            "note": "N independent single-thread replicas run side by side; not a parallel algorithm"}


def cpu_replica_main(a):
    """Child of cpu_replica_aggregate: one replica."""
    from mfrec_amd import synth
    from oracle import oracle as O
    nu, ni, k, n = a.users, a.items, a.k, a.cpu_replica
    idx, r = synth.make_ratings(nu, ni, n, workers=1)
    u, v = synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2)
    ib, ub = np.zeros(ni), np.zeros(nu)
    late = max(0.0, time.time() - a.start_at)
    while time.time() < a.start_at:
        time.sleep(0.001)
    t0 = time.perf_counter()
    O.train_linear_kernel(1, k, 0.1, LR, 0, 0, KU, KI, KB, 0.0, u, v, idx, r, ib, ub)
    print(json.dumps({"seconds": time.perf_counter() - t0, "late_s": late, "kind": "port"}))


def cpu_baseline_funk(idx, r, nu, ni, budget_s=8.0):
    """estimator_loop_without_bias (gd_estimator.pyx:691-779): k = 8 features x 1 epoch each (+ the
    cache-commit passes) on a bounded prefix, 1 thread."""
    from oracle import oracle as O
    from mfrec_amd import synth
    kk = 8

    def run(n):
        u, v = synth.init_const(kk, ni, FUNK_INIT), synth.init_const(kk, nu, FUNK_INIT)
        O.estimator_loop_without_bias(1, 1, 1e9, kk, FUNK_INIT, FUNK_LR, FUNK_K, u, v, idx[:n], r[:n], nu, ni)

    rate, n, dt = timed_prefix(run, idx.shape[0], budget_s / 2.0, n_probe=500000)
    return {"value": rate * kk, "unit": "scalar-updates/s", "cores": 1, "kind": "port",
            "sample": "first %d of %d ratings, 8 features x 1 epoch (+ commit passes), oracle/mfx_oracle.c, "
                      "single thread, %.1f s" % (n, idx.shape[0], dt)}


def als_flops(k, nz, nu, ni):
    """SURVEY 8(d) FLOP model of one ALS epoch."""
    return 2 * (k * (k + 1) * nz) + 2 * k * k * (nu + ni) + (nu + ni) * (k ** 3 / 3 + 2 * k * k) + 4 * k * nz


def cpu_baseline_als(k):
    """als_wrmf (als_implicit.pyx:211-352) on the cfg-4 problem divided by 200 in every dimension (5K x 1K,
    500K ratings, same k; SURVEY 8(d) proposes / 100, which needs a minute and a half of CPU), 1 epoch,
    1 thread; scaled to cfg 4 by the FLOP model of SURVEY 8(d)."""
    import scipy.sparse as sp
    from oracle import oracle as O
    from mfrec_amd import synth
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    nu, ni, nnz = 5000, 1000, 500000
    idx, _ = synth.make_ratings(nu, ni, nnz)
    M = sp.csr_matrix((np.ones(nnz, dtype=np.int8), (idx[:, 0], idx[:, 1])), shape=(nu, ni))
    M.sum_duplicates()
    nz = int(M.nnz)
    views = create_bool_sparse_row(M) + create_bool_sparse_col(M)
    rs = np.random.RandomState(0)
    u, v = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    # ONE thread, as the reference runs: its per-row np.linalg.inv would otherwise fan a 128 x 128 LAPACK
    # call out over every core of the host (measured: 13x slower on a 256-thread box)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    t0 = time.perf_counter()
    try:
        O.als_wrmf(1, k, u, v, None, None, *views, nu, ni, 1, 0.015)
    finally:
        if limit is not None:
            limit.restore_original_limits()
    dt = time.perf_counter() - t0
    F = als_flops(k, nz, nu, ni)
    return {"value": (nu + ni) / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "model_gflops": F / dt / 1e9, "seconds": dt,
            "sample": "cfg 4 / 200: %d x %d, %d distinct ratings, k=%d, 1 epoch, oracle/mfx_oracle.c, single thread"
                      % (nu, ni, nz, k)}


# ------------------------------------------------------------------------------------------------
# GPU measurements
# ------------------------------------------------------------------------------------------------
def epoch_launches(info):
    """Launches of the dominant kernel per epoch: one per dependency level (k1_level / k1_tail), or ONE persistent
    launch (k1_res: MFX_EPOCH_TICKET / _STATIC; the two 16-byte-per-entity granule kernels and the two reduction
    kernels around it are inside the timed region and count against it)."""
    return 1 if info.get("epoch_kernel", "levels") != "levels" else info["launches_per_epoch"]


def k1_measure(plan, nnz, nu, k, steps, warmup, run, label_plan=None):
    """Timed region of a full-k SGD plan: HIP events on the library stream + roofline of k1_level."""
    from mfrec_amd.plan import Timer, LINEAR
    info = plan.info()
    run(max(warmup, 0))
    with Timer() as tm:
        run(steps)
    launches = epoch_launches(info)
    b_read, b_total = algorithmic_bytes(k, nnz, nu)
    avg_launch_s = tm.ms * 1e-3 / (steps * launches)
    achieved = (b_read * nnz / launches) / avg_launch_s
    return {"device_ms_per_step": tm.ms / steps, "value": nnz * steps / (tm.ms * 1e-3),
            "levels": info["n_levels"], "launches_per_epoch": launches, "epoch_kernel": info.get("epoch_kernel", "levels"),
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": None,
                         "algorithmic_bytes_per_launch": b_read * nnz / launches,
                         "kernel": (label_plan or plan).kernel_name(LINEAR),
                         "bytes_per_update_read": b_read, "bytes_per_update_total": b_total,
                         "avg_launch_us": avg_launch_s * 1e6, "launches_per_epoch": launches,
                         "frac_total_bytes": (b_total * nnz / launches) / avg_launch_s / HBM_PEAK,
                         "us_per_chain_level": (tm.ms * 1e3 / steps) / max(info["n_levels"], 1)},
            "schedule_build_ms": info["schedule_us"] / 1e3}


def also_k1_order(idx, r, nu, ni, k, u0, v0, steps, warmup, order, epoch="auto"):
    from mfrec_amd import synth
    from mfrec_amd.plan import SgdPlan, LINEAR
    if order == "user-major":
        i2, r2 = synth.user_major(idx, r)
    elif order == "balanced":
        from mfrec_amd.lib.datasets import balanced_order_gpu
        perm, _ = balanced_order_gpu(idx, nu, ni)
        i2, r2 = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    else:
        i2, r2 = idx, r
    with SgdPlan(i2, r2, nu, ni, k, epoch=epoch) as p:
        p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
        m = k1_measure(p, idx.shape[0], nu, k, steps, warmup,
                       lambda n: p.run_kmf(LINEAR, n, LR, KU, KI, KB, want_rmse=False))
    m["unit"] = "rating-updates/s"
    m["workload"] = "%dK x %dK, %dM ratings, %s order, exact, full-k SGD, epoch kernel: %s" % (
        nu // 1000, ni // 1000, idx.shape[0] // 1000000, order, m["epoch_kernel"])
    return m


def also_k1_skew(nu, ni, nnz, k, u0, v0, a):
    """Power-law item degrees (Zipf exponent a) over cfg 2's shape -- the shape of real rating sets (the reference's only
    published workload is MovieLens-10M, gradient_descent.py:41-45).  Level count >= the hottest item's degree; the library
    finds the chain on the resident side and walks it in bursts."""
    from mfrec_amd.plan import SgdPlan, LINEAR
    rs = np.random.RandomState(1)
    pi = 1.0 / np.arange(1, ni + 1) ** a
    idx = np.ascontiguousarray(np.stack([rs.randint(0, nu, nnz), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32))
    r = rs.randint(1, 6, nnz).astype(np.float64)
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(u0, v0, np.zeros(ni), np.zeros(nu))
        m = k1_measure(p, nnz, nu, k, 3, 1, lambda n: p.run_kmf(LINEAR, n, LR, KU, KI, KB, want_rmse=False))
    m["unit"] = "rating-updates/s"
    m["max_item_degree"] = int(np.bincount(idx[:, 1]).max())
    m["workload"] = "cfg2 shape with Zipf-%.1f item degrees (hottest item: %d ratings = the minimum level count), exact, " \
                    "epoch kernel: %s; the level schedule with its tail kernel runs this at 15-16 M updates/s" % (
                        a, m["max_item_degree"], m["epoch_kernel"])
    return m


def ml10m_shape(nnz=10000000, seed=1):
    """The SHAPE of the reference's only published workload (gradient_descent.py:41-45: MovieLens-10M, 69 878 users x 10 677
    items, 10 M ratings): a synthetic with power-law degrees on both sides (the real file is not available offline), shuffled."""
    nu, ni = 69878, 10677
    rs = np.random.RandomState(seed)
    pu, pi = 1.0 / np.arange(1, nu + 1) ** 0.4, 1.0 / np.arange(1, ni + 1) ** 0.47
    idx = np.ascontiguousarray(np.stack([rs.choice(nu, nnz, p=pu / pu.sum()), rs.choice(ni, nnz, p=pi / pi.sum())], 1).astype(np.int32))
    r = rs.randint(1, 6, nnz).astype(np.float64)
    return nu, ni, idx, r


# the dependent f64 arithmetic of ONE rating in exact mode (64 ordered adds + the error / bias chain, profiles/r04_k1_res_stamps.txt):
# what a rating of the hottest entity costs at the very least, its rows being forwarded in registers
DEP_US_PER_RATING = 0.65


def also_k1_ml10m(k):
    """K1 (full-k SGD, exact) on the MovieLens-10M SHAPE: the epoch is at least as long as the hottest item's chain --
    max degree x the dependent arithmetic of one rating -- which is the floor printed beside the byte roofline."""
    from mfrec_amd import synth
    from mfrec_amd.plan import SgdPlan, LINEAR
    nu, ni, idx, r = ml10m_shape()
    nnz = idx.shape[0]
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(synth.init_normal(k, ni, 1), synth.init_normal(k, nu, 2), np.zeros(ni), np.zeros(nu))
        m = k1_measure(p, nnz, nu, k, 3, 1, lambda n: p.run_kmf(LINEAR, n, LR, KU, KI, KB, want_rmse=False))
    m["unit"] = "rating-updates/s"
    m["max_item_degree"] = int(np.bincount(idx[:, 1]).max())
    m["max_user_degree"] = int(np.bincount(idx[:, 0]).max())
    floor_ms = m["max_item_degree"] * DEP_US_PER_RATING * 1e-3
    m["roofline"]["chain_floor_ms"] = floor_ms
    m["roofline"]["chain_floor"] = "hottest item's degree x %.2f us of dependent f64 arithmetic per rating (exact mode)" % DEP_US_PER_RATING
    m["roofline"]["frac_of_chain_floor"] = floor_ms / m["device_ms_per_step"]
    # the OTHER chain of this shape: the heaviest user's row travels, every one of its ratings is a hand-off between two waves
    # (round 5: this, not the hot item, was the epoch -- 7 369 hand-offs x 8.7 us -- until only hot items burst)
    m["roofline"]["heaviest_user_hand_offs"] = m["max_user_degree"]
    m["roofline"]["us_per_hand_off_if_that_chain_is_the_epoch"] = m["device_ms_per_step"] * 1e3 / max(m["max_user_degree"], 1)
    m["north_star_floor_updates_per_s"] = 200e6
    m["workload"] = "MovieLens-10M SHAPE (synthetic, power-law degrees on both sides): 69 878 x 10 677, 10 M ratings, shuffled, k=%d, " \
                    "full-k SGD (train_linear_kernel), exact; epoch kernel: %s" % (k, m["epoch_kernel"])
    return m


def train_rmse(u, v, ib, ub, idx, r, nu, ni, k):
    """sqrt(mean((r - predict(item, user))^2)) over the training ratings, f64, batched predict kernel
    (metrics.py:69-73 with kmf.py:88-94's predict)."""
    import ctypes as C
    from mfrec_amd import _shim as S
    out = C.c_double(0.0)
    S.check(S.lib().mfx_predict_rmse(1, k, S.f64(u, "u", 2), S.f64(v, "v", 2), S.i32(idx, "idx", 2), S.f64(r, "r", 1),
                                     S.f64(ib, "ib", 1), S.f64(ub, "ub", 1), 0.0, idx.shape[0], nu, ni, None, C.byref(out)))
    return out.value


def also_k1_tolerance(idx, r, nu, ni, k, u0, v0, steps, warmup, order, epochs=10, variant="butterfly"):
    """The two tolerance modes of the full-k epoch, each timed, with its final train RMSE after `epochs` epochs against the exact
    mode's (= the oracle's, bit for bit) from the same initial factors on the same order.
    variant "butterfly": north_star's own kernel wording (wave-shuffle reduction of the k-dim dot, FMA), f64 rows;
    variant "f32": f32 row storage, f64 arithmetic, ordered dot (half the travelling bytes)."""
    from mfrec_amd.plan import SgdPlan, LINEAR
    if order == "balanced":
        from mfrec_amd.lib.datasets import balanced_order_gpu
        perm, _ = balanced_order_gpu(idx, nu, ni)
        idx, r = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    res = {}
    kw = {"dot": "butterfly"} if variant == "butterfly" else {"storage": "f32"}
    for name, kwargs in (("exact", {}), (variant, kw)):
        u, v, ib, ub = u0.copy(), v0.copy(), np.zeros(ni), np.zeros(nu)
        with SgdPlan(idx, r, nu, ni, k, **kwargs) as p:
            p.set_factors(u, v, ib, ub)
            p.run_kmf(LINEAR, epochs, LR, KU, KI, KB, want_rmse=False)
            p.get_factors(u, v, ib, ub)
            res[name] = train_rmse(u, v, ib, ub, idx, r, nu, ni, k)
            if name == variant:
                m = k1_measure(p, idx.shape[0], nu, k, steps, warmup,
                               lambda n: p.run_kmf(LINEAR, n, LR, KU, KI, KB, want_rmse=False))
    m["unit"] = "rating-updates/s"
    if variant == "butterfly":
        m["dtype"] = "f64 (butterfly dot + FMA: a different summation order than the reference's)"
        m["workload"] = "cfg2 shape, %s order, full-k SGD in TOLERANCE mode: butterfly k-dot (DPP + cross-row swaps, fixed " \
                        "association) + FMA -- deterministic f64, not bit-identical to the reference" % order
    else:
        m["dtype"] = "f64 arithmetic on f32-stored rows (NOT the reference precision)"
        m["workload"] = "cfg2 shape, %s order, full-k SGD on f32-STORED rows (widened on load, rounded once per update), f64 " \
                        "arithmetic, ordered dot -- half the travelling bytes; not the reference's precision" % order
        # the byte model at s = 4: 16-B record + k*4 + 8 (f64 bias) + amortised user row
        b_read = 16 + k * 4 + 8 + (k * 4 + 8) / (idx.shape[0] / float(nu))
        ro = m["roofline"]
        ro["bytes_per_update_read"] = b_read
        ro["achieved"] = m["value"] * b_read / 1e9
        ro["frac"] = m["value"] * b_read / HBM_PEAK
        ro["algorithmic_bytes_per_launch"] = b_read * idx.shape[0] / ro["launches_per_epoch"]
        # ... and the write side at s = 4: item row + f64 item bias every rating, user row + bias once per run
        b_total = b_read + k * 4 + 8 + (k * 4 + 8) / (idx.shape[0] / float(nu))
        ro["bytes_per_update_total"] = b_total
        ro["frac_total_bytes"] = m["value"] * b_total / HBM_PEAK
    m["final_train_rmse_exact_mode"] = res["exact"]
    m["final_train_rmse"] = res[variant]
    m["rmse_delta_vs_exact_after_%d_epochs" % epochs] = abs(res[variant] - res["exact"])
    m["rmse_budget"] = 1e-5
    return m


def also_funk(idx, r, nu, ni, k, order, epochs_per_feature=2):
    """K2: estimator_loop_without_bias semantics, k features x epochs_per_feature epochs (the stopping rule
    is disabled with min_improvement = 1e9) + the cache-commit passes."""
    from mfrec_amd import synth
    from mfrec_amd.plan import SgdPlan, Timer
    if order == "balanced":
        from mfrec_amd.lib.datasets import balanced_order_gpu
        perm, _ = balanced_order_gpu(idx, nu, ni)
        idx, r = np.ascontiguousarray(idx[perm]), np.ascontiguousarray(r[perm])
    nnz = idx.shape[0]
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(synth.init_const(k, ni, FUNK_INIT), synth.init_const(k, nu, FUNK_INIT))
        p.run_funk(1, 1e9, FUNK_INIT, FUNK_LR, FUNK_K, epoch_cap=1)        # warm-up: k passes, graph capture
        p.set_factors(synth.init_const(k, ni, FUNK_INIT), synth.init_const(k, nu, FUNK_INIT))
        with Timer() as tm:
            hist = p.run_funk(epochs_per_feature, 1e9, FUNK_INIT, FUNK_LR, FUNK_K)
        info = p.info()
        kname = p.kernel_name(3)
    n = len(hist)
    rate = nnz * n / (tm.ms * 1e-3)
    # k2_res / k2_res_box / k2_xcd_pass: ONE persistent launch per (feature, epoch) pass (the mailbox fill / seed and the
    # squared-error reduction beside it are inside the timed region and count against it); k2_level: one launch per level
    persistent = kname.startswith("k2_xcd_pass") or kname.startswith("k2_res")
    launches = 1 if persistent else info["funk_launches_per_epoch"]
    ms_pass = tm.ms / n
    how = ("one persistent launch per pass, lane = item, V[f,user] travels through per-rating mailboxes" if (kname.startswith("k2_res") and "mailbox" in kname)
           else "one persistent launch per pass, lane = item, V[f,user] travels inside 16-byte granules" if kname.startswith("k2_res")
           else "one persistent single-XCD launch per pass" if kname.startswith("k2_xcd_pass") else "one launch per dependency level")
    # algorithmic bytes: 24 B per scalar update (16-B record + 8-B cache word)
    roof = {"bound": "hbm", "achieved": rate * 24 / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": rate * 24 / HBM_PEAK, "bytes_per_scalar_update": 24, "kernel": kname,
            "avg_launch_us": ms_pass * 1e3 / max(launches, 1), "launches_per_pass": launches, "traffic": None}
    if persistent:
        # the pass is a chain of `levels` dependent hand-offs inside one launch: what one hop of the chain costs, and the floor
        # the minimum hop of this part sets under it
        roof["us_per_chain_level"] = ms_pass * 1e3 / max(info["n_levels"], 1)
        roof.update(chain_floor(info["n_levels"], ms_pass))
    else:
        roof["us_per_level"] = ms_pass * 1e3 / max(info["n_levels"], 1)
    tr = k2_traffic(kname) if order == "shuffled" and (nu, ni, nnz, k) == CFG2 else None
    if tr:
        roof["traffic"] = tr["bytes_per_pass"]
        roof["traffic_unit"] = "HBM-side bytes per launch = per pass (PMC, %s): %.1f B read + %.1f B written per scalar update" % (
            tr["source"], tr["read_bytes_per_update"], tr["write_bytes_per_update"])
    return {"value": rate, "unit": "scalar-updates/s", "row_equivalents_per_s": rate / k,
            "feature_epochs": n, "ms_per_pass": ms_pass, "levels": info["n_levels"],
            "launches_per_pass": launches, "rmse_online_last": float(hist[-1]),
            "workload": "cfg2 shape, %s order, k=%d features x %d epochs each + cache-commit passes "
                        "(estimator_loop_without_bias), exact (bit-identical to the sequential loop); %s" % (order, k, epochs_per_feature, how),
            "roofline": roof}


def k2_traffic(kname):
    """HBM-side bytes per pass of the Funk kernel from the committed rocprofv3 PMC summary (profiles/rNN_k2_traffic.json: FETCH_SIZE
    / WRITE_SIZE in separate passes, tools/prof_k2.sh); only when it was taken on the kernel that ran here."""
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k2_traffic.json")))
        if not cands:
            return None
        tj = json.load(open(cands[-1]))
        if tj.get("kernel") != kname:
            return None
        tj["source"] = os.path.basename(cands[-1])
        return tj
    except Exception:
        return None


def also_funk_ml10m(with_cpu):
    """K2 on the SHAPE of the reference's only published workload (gradient_descent.py:41-45: MovieLens-10M, 69 878 users x
    10 677 items, 10 M ratings, 40 features, 49 minutes on a Core i5 2.3 GHz with the class defaults = 275 epochs per
    feature): a synthetic with power-law degrees on both sides (the real file is not available offline), shuffled order,
    k = 40 features x 2 epochs each timed.  A pass is as long as the longest chain of the data: the hottest items (~39 K ratings)
    and the hottest users (~7 K ratings) own a whole wave each (DESIGN.md section 4, K2 hot-wave roles)."""
    from mfrec_amd import synth
    from mfrec_amd.plan import SgdPlan, Timer
    nnz, k = 10000000, 40
    nu, ni, idx, r = ml10m_shape(nnz)
    with SgdPlan(idx, r, nu, ni, k) as p:
        p.set_factors(synth.init_const(k, ni, FUNK_INIT), synth.init_const(k, nu, FUNK_INIT))
        p.run_funk(1, 1e9, FUNK_INIT, FUNK_LR, FUNK_K, epoch_cap=1)
        p.set_factors(synth.init_const(k, ni, FUNK_INIT), synth.init_const(k, nu, FUNK_INIT))
        with Timer() as tm:
            hist = p.run_funk(2, 1e9, FUNK_INIT, FUNK_LR, FUNK_K)
        info = p.info()
        kname = p.kernel_name(3)
    n = len(hist)
    rate = nnz * n / (tm.ms * 1e-3)
    ms_pass = tm.ms / n
    passes_ref = 40 * 275
    out = {"value": rate, "unit": "scalar-updates/s", "feature_epochs": n, "ms_per_pass": ms_pass, "levels": info["n_levels"],
           "max_item_degree": int(np.bincount(idx[:, 1]).max()), "max_user_degree": int(np.bincount(idx[:, 0]).max()),
           "rmse_online_last": float(hist[-1]),
           "workload": "MovieLens-10M SHAPE (synthetic, power-law degrees on both sides): 69 878 x 10 677, 10 M ratings, shuffled, "
                       "k=40 features x 2 epochs each + cache-commit passes (estimator_loop_without_bias), exact",
           "reference_published": {"seconds": 2940.0, "what": "GDRecommender.train, MovieLens-10M, 40 features (class default 275 epochs "
                                   "per feature = 11 000 passes), Core i5 2.3 GHz", "source": "gradient_descent.py:41-45",
                                   "scalar_updates_per_s": nnz * passes_ref / 2940.0},
           "projected_seconds_for_the_published_run": passes_ref * ms_pass * 1e-3,
           "roofline": {"bound": "hbm", "achieved": rate * 24 / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": rate * 24 / HBM_PEAK,
                        "bytes_per_scalar_update": 24, "kernel": kname, "avg_launch_us": ms_pass * 1e3,
                        "us_per_level": ms_pass * 1e3 / max(info["n_levels"], 1), "traffic": None,
                        "note": "a latency chain, not a stream: the pass is as long as the hottest item's chain (0.115 us per rating on its own wave)"}}
    if with_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline_funk(idx, r, nu, ni, budget_s=6.0)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def also_als(idx3, nu, ni, k):
    """K3: als_wrmf on BASELINE cfg 4 at full size (1M x 200K, 100M ratings -> distinct cells, k=128)."""
    import scipy.sparse as sp
    from mfrec_amd import _shim as S
    from mfrec_amd.lib.als_implicit import als_wrmf
    from mfrec_amd.lib.datasets import create_bool_sparse_col, create_bool_sparse_row
    nnz = idx3.shape[0]
    M = sp.csr_matrix((np.ones(nnz, dtype=np.int8), (idx3[:, 0], idx3[:, 1])), shape=(nu, ni))
    M.sum_duplicates()
    nz = int(M.nnz)
    views = create_bool_sparse_row(M) + create_bool_sparse_col(M)
    del M
    rs = np.random.RandomState(0)
    u, v = rs.normal(0, 0.1, (k, ni)), rs.normal(0, 0.1, (k, nu))
    epochs = 2
    als_wrmf(1, k, u, v, None, None, *views, nu, ni)            # warm-up incl. transfers
    als_wrmf(epochs, k, u, v, None, None, *views, nu, ni)
    dev_s = S.lib().mfx_last_compute_ms() * 1e-3 / epochs
    F = als_flops(k, nz, nu, ni)
    return {"value": (nu + ni) / dev_s, "unit": "rows/s", "seconds_per_epoch": dev_s, "epochs_timed": epochs,
            "model_tflop_per_epoch": F / 1e12, "finite": bool(np.isfinite(u).all() and np.isfinite(v).all()),
            "workload": "cfg4: %dM x %dK, %d distinct ratings, k=%d, ALS/WRMF (als_wrmf), device time of the "
                        "epochs (HIP events), factors and CSR views resident" % (nu // 1000000, ni // 1000, nz, k),
            "roofline": {"bound": "mfma", "achieved": F / dev_s / 1e12, "peak": F64_MFMA_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": F / dev_s / F64_MFMA_PEAK, "kernel": "k_als_rows3",
                         "traffic": None, "flop_model": "SURVEY 8(d)"}}


def also_topn(idx, r, nu, ni, k, with_cpu):
    """Evaluation side: top-10 of every item for 100 K users in one call (mfx_topn_batch, fused
    kernel) on cfg 2's shape, and -- as the CPU baseline -- the reference's own procedure for ONE user: a Python
    loop calling predict() per item (mf.py:161-173), timed on a bounded number of users."""
    import ctypes as C
    import scipy.sparse as sp
    from mfrec_amd import _shim as S
    from mfrec_amd import synth
    from mfrec_amd.recommendation import KMFRecommender
    R = KMFRecommender(4, 6, {"nbr_features": k})
    R._nbr_users, R._nbr_items = nu, ni
    R.relationship_matrix = sp.csr_matrix((r, (idx[:, 0], idx[:, 1])), shape=(nu, ni)).tolil()
    R.svd_u, R.svd_v = synth.init_normal(k, ni, synth.SEED_INIT), synth.init_normal(k, nu, synth.SEED_INIT + 1)
    R.items_bias, R.users_bias, R.overall_bias = np.zeros(ni), np.zeros(nu), 3.0
    R.neighborhood = ni
    users = np.arange(nu, dtype=np.int32)
    N = 10
    R.find_recommended_items_batch(users[:1000], N, "predict_linear")            # warm-up
    st0 = (C.c_int64 * 2)(); S.check(S.lib().mfx_topn_stats(st0))
    t = time.perf_counter()
    items, scores = R.find_recommended_items_batch(users, N, "predict_linear")
    wall = time.perf_counter() - t
    st1 = (C.c_int64 * 2)(); S.check(S.lib().mfx_topn_stats(st1))
    flops = 2.0 * k * nu * ni                                                     # one multiply + one add per term
    out = {"value": nu / wall, "unit": "users/s", "seconds": wall, "users": nu, "candidates": ni, "N": N,
           "users_redone_by_sort_path": int(st1[1] - st0[1]),
           "workload": "cfg2 shape: top-%d of %dK candidate items for each of %dK users, k=%d, predict_linear, rated items "
                       "excluded (mf.py:144-193 rules), end to end from host arrays (CSR marshalling of 10M ratings, "
                       "H2D, fused kernel, D2H)" % (N, ni // 1000, nu // 1000, k),
           "roofline": None}
    # the serving form: model resident (mfx_topn_plan_*), users queried by id
    from mfrec_amd.recommendation import metrics as _metrics
    t = time.perf_counter()
    plan = _metrics.TopNPlan(R, "predict_linear")
    serving = {"plan_create_s": time.perf_counter() - t}
    for U in (1, 1000, nu):
        q = np.arange(U, dtype=np.int32)
        plan.query(q, N)
        reps = 20 if U <= 1000 else 3
        t = time.perf_counter()
        for _ in range(reps):
            pi, _ps = plan.query(q, N)
        serving["query_%d_users_ms" % U] = (time.perf_counter() - t) / reps * 1e3
        assert np.array_equal(pi, items[:U])
    plan.close()
    out["serving_plan"] = serving
    # the contraction scores = V^T U runs on the matrix cores (k_topn_mfma: approximate scores SELECT, k_topn_exactify restores the
    # reference's bits for the N selected or flags the user): priced against the f64 MFMA peak on the resident-model query of all
    # users (gather of the users' columns + kernel + exactify + D2H of the lists; the kernel alone: profiles/r05_topn_kernel_stats.csv)
    q_s = serving["query_%d_users_ms" % nu] * 1e-3
    out["roofline"] = {"bound": "mfma", "achieved": flops / q_s / 1e12, "peak": F64_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                       "frac": flops / q_s / F64_MFMA_PEAK, "kernel": "k_topn_mfma + k_topn_exactify", "traffic": None,
                       "flops": "2 k users items (one multiply + one add per term)",
                       "note": "resident-model query of all %d users: %.1f ms; `value` is the one-shot call from host arrays" % (nu, q_s * 1e3)}
    if with_cpu:
        t = time.perf_counter()
        done = 0
        while time.perf_counter() - t < 6.0 and done < 50:
            user = done
            rated = set(R.relationship_matrix.rows[user]) | {user}
            sc = np.zeros(ni)
            for i in range(ni):                                                   # the reference's loop, predict() per item
                sc[i] = 0.0 if i in rated else R.predict_linear(i, user)
            nzi = sc.nonzero()[0]
            top = nzi[np.argsort(-sc[nzi], kind="stable")][:N]
            assert list(top) == [int(x) for x in items[user] if x >= 0]
            done += 1
        cw = time.perf_counter() - t
        out["cpu_baseline"] = {"value": done / cw, "unit": "users/s", "cores": 1, "kind": "port",
                               "sample": "%d users, the reference's per-item predict() loop (mf.py:161-173) in NumPy, "
                                         "%.1f s; each list checked against the GPU's" % (done, cw)}
    return out


def averaging_gap(world, epochs, k):
    """Final train-RMSE gap of the user-sharded averaging scheme at `world` ranks to the sequential loop, both run by the
    CPU oracle on cfg 3 / 100 (10 K x 2 K, 1 M ratings) from the same start."""
    from mfrec_amd import synth
    from oracle import oracle as O
    from oracle import sharded_sim as SIM
    nu, ni, nnz = CFG3[0] // 100, CFG3[1] // 100, CFG3[2] // 100
    idx, r = synth.make_ratings(nu, ni, nnz)
    u0, v0 = synth.init_normal(k, ni, synth.SEED_INIT), synth.init_normal(k, nu, synth.SEED_INIT + 1)
    res = {}
    for w in (1, world):
        u, v, ib, ub = SIM.simulate(0, w, epochs, k, LR, KU, KI, KB, u0, v0, np.zeros(ni), np.zeros(nu), idx, r)
        res[w] = float(O.rmse(O.RMSE_KMF_LINEAR, u, v, idx, r, ib, ub))
    return {"value": res[world] - res[1], "rmse_sequential": res[1], "rmse_averaging": res[world], "epochs": epochs,
            "measured_on": "CPU simulation (oracle/sharded_sim.py) of the same scheme, cfg 3 / 100: %d x %d, %d ratings, "
                           "%d shards" % (nu, ni, nnz, world), "budget": 1e-5}


def fail_line(rank, world, what, err):
    """One greppable line per failing rank (RCCL's own text is inside the library's error message; with
    NCCL_DEBUG=WARN, set below for N > 1, RCCL also prints its warnings to stderr)."""
    sys.stderr.write("[bench rank %d/%d] %s FAILED: %s\n" % (rank, world, what, err))
    sys.stderr.flush()


def self_launch(n):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the launcher.
    It has not touched the GPU, loaded libmfx or imported the shim; it starts N fresh child processes of this same script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would set them (one rank per GPU, device = LOCAL_RANK),
    lets rank 0 print the one JSON line on the inherited stdout (the other ranks' stdout goes to stderr), and returns the worst
    child exit code.  No exec.  A rank that dies takes the job with it: the survivors get a grace period (their own watchdogs
    print where they were), then they are ended by pid."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env0 = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                TORCHELASTIC_RUN_ID="selflaunch%d" % os.getpid(), TORCHELASTIC_RESTART_COUNT="0")
    kids = []

    def die_with_parent():
        # (child side, before exec'ing python: if the launcher is killed -- a driver's timeout --, the ranks get SIGTERM instead of
        #  living on as orphans that hold the GPUs)
        try:
            import ctypes
            import signal
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)        # PR_SET_PDEATHSIG
        except Exception:
            pass

    for rk in range(n):
        env = dict(env0, RANK=str(rk), LOCAL_RANK=str(rk))
        kids.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                     stdout=None if rk == 0 else sys.stderr, preexec_fn=die_with_parent))

    def forward(signum, frame):                      # SIGTERM / SIGINT to the launcher ends the ranks too
        for p in kids:
            if p.poll() is None:
                p.terminate()
    import signal
    signal.signal(signal.SIGTERM, forward)
    signal.signal(signal.SIGINT, forward)
    grace = float(os.environ.get("MFX_BENCH_LAUNCH_GRACE", "60"))
    failed_at = None
    while True:
        codes = [p.poll() for p in kids]
        if all(c is not None for c in codes):
            break
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.time()
            sys.stderr.write("[bench launcher] rank(s) %s ended with an error; the others get %.0f s\n"
                             % ([i for i, c in enumerate(codes) if c not in (None, 0)], grace))
            sys.stderr.flush()
        if failed_at is not None and time.time() - failed_at > grace:
            for p in kids:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10.0
            while time.time() < t_end and any(p.poll() is None for p in kids):
                time.sleep(0.1)
            for p in kids:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    codes = [p.wait() for p in kids]
    worst = 0
    for c in codes:
        if c != 0:
            worst = c if c > 0 else 128 - c          # a signal's negative code -> the shell's 128 + n
            break
    if worst:
        sys.stderr.write("[bench launcher] child exit codes: %r\n" % (codes,))
    return worst


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["auto", "cfg2", "cfg3"], default="auto",
                    help="auto = cfg2 at N=1 (the config the metric is quoted on), cfg3 user-sharded at N>1")
    ap.add_argument("--users", type=int, default=None)
    ap.add_argument("--items", type=int, default=None)
    ap.add_argument("--nnz", type=int, default=None, help="ratings of the whole job")
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--order", choices=["shuffled", "user-major", "balanced"], default="shuffled")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1: strong = the job's ratings are divided among the GPUs (default); weak = every "
                         "GPU gets a job-sized user shard of its own")
    ap.add_argument("--epoch", choices=["auto", "levels", "ticket", "static"], default="auto",
                    help="kernel of a full-k epoch (include/mfx.h MFX_EPOCH_*): auto = the library's cost model")
    ap.add_argument("--avg-epoch", choices=["blocks", "levels"], default="blocks",
                    help="N > 1: the local epoch of a user shard -- `blocks`: its ratings on the item-block order, every block on the "
                         "static resident-side epoch, one all-reduce per item block; `levels`: the given order on the level schedule")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` block (secondary kernels)")
    ap.add_argument("--no-extra", action="store_true", help=argparse.SUPPRESS)   # accepted, no-op
    ap.add_argument("--also-only", default="", help="comma-separated subset of the `also` legs (profiling runs)")
    ap.add_argument("--storage", choices=["f64", "f32", "bf16"], default="f64",
                    help="f32 = NOT the reference's precision: rows stored as f32 (measured option only)")
    ap.add_argument("--cpu-replica", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--start-at", type=float, default=0.0, help=argparse.SUPPRESS)
    return ap.parse_args()


class Job(object):
    """What the stages of one bench run hand to each other (one process = one rank)."""

    def mark(self, what):      # MFX_BENCH_TRACE=1: one stderr line per stage (which stage a traced run died in)
        self.where[0] = what
        if os.environ.get("MFX_BENCH_TRACE") == "1":
            sys.stderr.write("[bench] stage: %s\n" % what); sys.stderr.flush()

    def barrier(self):
        self.S.sync()                                      # hipStreamSynchronize on the library stream
        if self.multi:
            self.S.check(self.S.lib().mfx_comm_barrier())  # RCCL all-reduce + sync = cross-rank barrier

    def max_over_ranks(self, x):
        if not self.multi:
            return x
        import ctypes as C
        w = C.c_double(x)
        self.S.check(self.S.lib().mfx_comm_allreduce_max(C.byref(w)))
        return w.value


def init_ranks(a):
    """Rank / world from the launcher's environment, the run's watchdog, the workload (cfg 2 at N = 1, cfg 3 at N > 1)."""
    J = Job()
    J.a = a
    J.rank = int(os.environ.get("RANK", "0"))
    J.world = int(os.environ.get("WORLD_SIZE", "1"))
    J.local = int(os.environ.get("LOCAL_RANK", "0"))
    if J.world != a.gpus:
        a.gpus = J.world
    if J.world > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")
    # MFX_BENCH_FORCE_MULTI=1 (rehearsal on a one-GPU box; the driver never sets it): run the N > 1 code path
    # -- cfg 3, sharded plan, real RCCL all-reduce on a forced 1-rank communicator, the DSGD leg -- with world 1
    J.multi = J.world > 1 or os.environ.get("MFX_BENCH_FORCE_MULTI") == "1"
    if J.multi and J.world == 1:
        os.environ["MFX_FORCE_COMM"] = "1"
    J.where = ["start"]
    if J.world > 1:
        # a rank that never comes back from a collective (a peer died, a link is down) would sit there until the
        # driver's own limit: say which rank and where, then end the process with an error
        import threading

        def stuck():
            fail_line(J.rank, J.world, "watchdog", "no progress after %s s, last stage: %s"
                      % (os.environ.get("MFX_BENCH_TIMEOUT", "1500"), J.where[0]))
            os._exit(3)

        main_dog = threading.Timer(float(os.environ.get("MFX_BENCH_TIMEOUT", "1500")), stuck)
        main_dog.daemon = True
        main_dog.start()
    cfg = CFG2 if (a.workload == "cfg2" or (a.workload == "auto" and not J.multi)) else CFG3
    J.nu, J.ni, J.nnz = a.users or cfg[0], a.items or cfg[1], a.nnz or cfg[2]
    J.k = a.k
    J.cfg_name = {CFG2: "cfg2", CFG3: "cfg3"}.get((J.nu, J.ni, J.nnz, J.k), "custom")
    return J


def init_device_and_comm(J):
    """Build (a no-op when libmfx.so is current), HIP device = LOCAL_RANK, and at N > 1 the RCCL communicator."""
    from mfrec_amd import _shim as S
    from mfrec_amd import build as B
    J.S = S
    rank, world, local = J.rank, J.world, J.local
    uid = None
    if J.multi:
        # torch.distributed.run is only the launcher: no torch import in the ranks (see
        # mfrec_amd/rendezvous.py); the RCCL unique id travels through a file.  Rank 0 builds (a
        # no-op when libmfx.so is current) before it publishes the id, the other ranks wait for the
        # id before they touch the library: eight ranks never compile into the same files at once.
        import ctypes as C
        from mfrec_amd import rendezvous
        if rank != 0:
            uid = rendezvous.exchange_unique_id(rank, world, None)
    B.build(verbose=False)
    J.where[0] = "device / RCCL communicator init"
    try:
        # MFX_BENCH_DEVICE: rehearsal hook (several ranks on one GPU); the driver never sets it
        # one process per GPU: device = LOCAL_RANK; a launcher that narrows every rank's view to its own GPU
        # (HIP_VISIBLE_DEVICES per process) leaves a single visible device 0
        ndev = max(S.device_count(), 1)
        if not (ndev == 1 or local < ndev):
            raise SystemExit("LOCAL_RANK %d but only %d visible devices: two ranks would share a GPU" % (local, ndev))
        S.init(int(os.environ.get("MFX_BENCH_DEVICE", local if local < ndev else 0)))
        if J.multi:
            def make_id():
                idbuf = C.create_string_buffer(S.UNIQUE_ID_BYTES)
                S.check(S.lib().mfx_comm_unique_id(idbuf))
                return idbuf.raw

            if rank == 0:
                uid = rendezvous.exchange_unique_id(rank, world, make_id)
            S.check(S.lib().mfx_comm_init(rank, world, uid))
    except Exception as e:
        fail_line(rank, world, "device / RCCL communicator init", e)
        raise


def build_plan(J):
    """Synthetic ratings of the whole job (a pure function of the counter: every rank generates them and keeps its own
    users -- strong scaling: the same ratings at every N), this rank's shard, initial factors, the plan (schedule build and
    upload: everything resident in HBM before the timed region)."""
    from mfrec_amd import synth
    from mfrec_amd.plan import SgdPlan, LINEAR
    a, rank, world, multi = J.a, J.rank, J.world, J.multi
    nu, ni, nnz, k = J.nu, J.ni, J.nnz, J.k
    J.mark("synthetic data + plan (schedule build, upload)")
    if world > 1 and a.scaling == "weak":
        idx, r = synth.make_ratings(nu, ni, nnz, t0=rank * nnz, user_base=0)
        my_idx, my_r, nu_local, job_nnz = idx, r, nu, world * nnz
        bounds = None
    else:
        idx, r = synth.make_ratings(nu, ni, nnz)
        job_nnz = nnz
        if multi:
            from mfrec_amd import sharding as SH
            bounds = SH.user_blocks(idx, nu, world)
            my_idx, my_r, _, nu_local = SH.shard(idx, r, bounds, rank)
        else:
            bounds = None
            my_idx, my_r, nu_local = idx, r, nu
    if a.order == "user-major":
        my_idx, my_r = synth.user_major(my_idx, my_r)
    elif a.order == "balanced":
        from mfrec_amd.lib.datasets import balanced_order_gpu
        perm, _ = balanced_order_gpu(my_idx, nu_local, ni)     # recolours the (already shuffled) synthetic order
        my_idx, my_r = np.ascontiguousarray(my_idx[perm]), np.ascontiguousarray(my_r[perm])
    u0 = synth.init_normal(k, ni, synth.SEED_INIT)
    v0_full = synth.init_normal(k, nu, synth.SEED_INIT + 1) if (not multi or a.scaling == "strong") else None
    if multi and a.scaling == "strong":
        lo = int(bounds[rank])
        v0 = np.ascontiguousarray(v0_full[:, lo:lo + nu_local])
    elif world > 1:
        v0 = synth.init_normal(k, nu, synth.SEED_INIT + 1 + rank)
    else:
        v0 = v0_full
    J.idx, J.r, J.job_nnz, J.bounds = idx, r, job_nnz, bounds
    J.my_idx, J.my_r, J.nu_local, J.my_nnz = my_idx, my_r, nu_local, my_r.shape[0]
    J.u0, J.v0, J.v0_full = u0, v0, v0_full
    # (N > 1: the user-sharded averaging mode.  A shard holds ALL items -- too many for the static resident-side epoch at cfg 3 --, so
    #  its local epoch runs on the item-block order (sharding.ShardedKMFBlocks: every block on the static epoch, one all-reduce of the
    #  item deltas per block), or, --avg-epoch levels, the given order on the level schedule with one all-reduce per epoch)
    J.use_blocks = multi and a.avg_epoch == "blocks" and a.storage == "f64" and a.order == "shuffled" and ni > 42000
    J.sb = None
    if J.use_blocks:
        from mfrec_amd.sharding import ShardedKMFBlocks

        class BlocksPlan(object):                       # what the code below asks of a plan
            def __init__(self):
                t0 = time.perf_counter()
                self.sb = ShardedKMFBlocks(my_idx, my_r, max(nu_local, 1), ni, k, rank, world, already_local=True)
                self.build_us = (time.perf_counter() - t0) * 1e6

            def set_factors(self, u, v, ib, ub):
                self.sb.d.set_factors(u, v, ib, ub)

            def info(self):
                i = self.sb.info()
                res = i["resident_blocks"] == i["item_blocks"]
                return {"n_levels": i["levels_total"], "launches_per_epoch": i["item_blocks"] if res else i["launches_per_epoch"],
                        "schedule_us": self.build_us, "epoch_kernel": "static, one launch per item block (%d blocks)" % i["item_blocks"]
                        if res else "levels per item block (%d blocks)" % i["item_blocks"], "item_blocks": i["item_blocks"]}

            def kernel_name(self, mode):
                i = self.sb.info()
                if i["resident_blocks"] == i["item_blocks"]:
                    return "k1_res (static), one persistent launch per item block; RCCL all-reduce of the block's item deltas on the " \
                           "transfer stream, under the next block's launch"
                return "k1_level per item block (%d of %d blocks on the static epoch); RCCL all-reduce per block on the transfer " \
                       "stream" % (i["resident_blocks"], i["item_blocks"])

            def close(self):
                self.sb.close()

        J.plan = BlocksPlan()
        J.sb = J.plan.sb
    else:
        J.plan = SgdPlan(my_idx, my_r, max(nu_local, 1), ni, k, storage=a.storage, epoch=a.epoch if not multi else "levels")
    J.plan.set_factors(u0, v0, np.zeros(ni), np.zeros(max(nu_local, 1)))
    J.last_rmse = {}
    J.LINEAR = LINEAR


def run_epochs(J, n):
    """n epochs of the headline mode on this rank's plan (the hot path; nothing else)."""
    if n <= 0:
        return
    a, plan, world = J.a, J.plan, J.world
    if J.use_blocks:
        se = J.sb.train(J.LINEAR, n, LR, KU, KI, KB, 1.0 / world)
        J.last_rmse["se_local"] = float(se[-1])
    elif J.multi:
        rm, se = plan.run_kmf_sharded(J.LINEAR, n, LR, KU, KI, KB, 1.0 / world)
        J.last_rmse["se_local"] = float(se[-1])
    else:
        plan.run_kmf(J.LINEAR, n, LR, KU, KI, KB, want_rmse=False)


def timed_region(J):
    """W untimed warm-up epochs, then EXACTLY K epochs between a barrier + stream synchronisation on both sides; the MAX
    over the ranks of the wall time; HIP events on the library's stream around the same K epochs."""
    a, S = J.a, J.S
    J.mark("warmup / timed region")
    try:
        run_epochs(J, max(a.warmup, 0))
        J.barrier()
        S.check(S.lib().mfx_timer_start())
        t0 = time.perf_counter()
        run_epochs(J, a.steps)
        J.barrier()
        wall = time.perf_counter() - t0
        import ctypes as C
        ms = C.c_double(0.0)
        S.check(S.lib().mfx_timer_stop(C.byref(ms)))
        J.dev_ms = ms.value
        J.wall = J.max_over_ranks(wall)
    except Exception as e:
        fail_line(J.rank, J.world, "timed region", e)
        raise


def headline_line(J):
    """The JSON line of rank 0 (without `also` and `cpu_baseline`): value, config, roofline of the dominant kernel."""
    a, plan, multi, world, use_blocks = J.a, J.plan, J.multi, J.world, J.use_blocks
    nu, ni, nnz, k, my_nnz, job_nnz = J.nu, J.ni, J.nnz, J.k, J.my_nnz, J.job_nnz
    dev_ms, wall = J.dev_ms, J.wall
    # Dominant kernel = k1_res (ONE persistent launch per epoch) or k1_level (one launch per dependency level).  Its
    # average launch duration is taken from the HIP events that bracket the timed region on the library's stream:
    # device time of the K epochs / (K * launches per epoch).  The epoch holds nothing else but the tiny granule /
    # reduction kernels (and, at N > 1, the delta / all-reduce / combine step), so this is an upper bound of the
    # kernel's own duration and agrees with rocprofv3's average (profiles/).
    # the kernel that RAN: a static epoch that lost its co-residency check sends the plan -- all blocks of a block plan, on every
    # rank -- to the level schedule for good (and the library repeats the call), so the labels are read after the timed region
    info = plan.info()
    kernel_name = plan.kernel_name(J.LINEAR)
    J.mark("per-launch event pairs (profile_kmf)")
    resident = use_blocks or ((not multi) and info.get("epoch_kernel", "levels") != "levels" and a.storage == "f64")
    launches = info["launches_per_epoch"] if use_blocks else (epoch_launches(info) if resident else info["launches_per_epoch"])
    b_read, b_total = algorithmic_bytes(k, nnz, nu)
    avg_launch_s = dev_ms * 1e-3 / (a.steps * launches)
    achieved = (b_read * my_nnz / launches) / avg_launch_s
    # cross-check on the level schedule: an event pair around every launch (the resident-side epoch IS one launch)
    kern_ms = plan.profile_kmf(J.LINEAR, LR, KU, KI, KB)[0] if not resident else dev_ms / a.steps

    # HBM traffic per launch: from the committed rocprofv3 PMC summary of this same command
    # (profiles/rNN_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)
    traffic = traffic_src = None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        if cands and world == 1 and a.order == "shuffled" and a.storage == "f64" and (nu, ni, nnz, k) == CFG2:
            tj = json.load(open(cands[-1]))
            # the summary is stamped with the kernel it was taken on: a stale one (another kernel) is dropped
            if tj.get("kernel") == kernel_name:
                traffic = (tj["read_bytes_per_rating"] + tj["write_bytes_per_rating"]) * nnz / launches
                traffic_src = os.path.basename(cands[-1])
    except Exception:
        traffic = None

    if J.rank != 0:
        return None
    value = job_nnz * a.steps / wall
    epoch_kernel = info.get("epoch_kernel", "levels") if (not multi or use_blocks) else "levels"
    if not multi:
        how = "exact (bit-identical to the sequential loop)"
        shard = ""
    else:
        # north_star's scheme: a DIFFERENT trajectory than the sequential loop (rmse_gap_vs_sequential below says by how much);
        # the mode that keeps the sequential loop's bits is also.dsgd_exact
        how = "parameter AVERAGING (not the sequential loop's trajectory: see rmse_gap_vs_sequential)"
        shard = ", users sharded over %d GPUs, item factors replicated, RCCL all-reduce of the item deltas every epoch " \
                "(scale 1/%d)" % (world, world)
    line = {
        "metric": "rating-updates/s per epoch at k=%d" % k, "value": value,
        "unit": "rating-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall * 1e3 / a.steps, "higher_is_better": True,
        "scaling": "n/a" if not multi else ("weak" if a.scaling == "weak" else "strong"),
        "vs_baseline": None,
        "dtype": "f64" if a.storage == "f64" else "f64 arithmetic on %s-stored rows (NOT the reference precision)" % a.storage,
        "data": "synthetic",
        "config": {"workload": "%s: synthetic %dK users x %dK items, %dM ratings%s, k=%d, full-k SGD "
                               "(train_linear_kernel), %s order, %s, epoch kernel: %s%s" % (
                                   J.cfg_name, nu // 1000, ni // 1000, nnz // 1000000,
                                   " per GPU" if (world > 1 and a.scaling == "weak") else "", k, a.order, how, epoch_kernel, shard),
                   "users": nu, "items": ni, "ratings": job_nnz, "ratings_this_gpu": my_nnz, "k": k,
                   "levels": info["n_levels"], "launches_per_epoch": launches,
                   "epoch_kernel": epoch_kernel,
                   "sharding": ("users x%d, item-delta all-reduce (RCCL)%s" % (world, " per item block; a shard's local order: "
                                "its item-block order" if use_blocks else "")) if multi else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "algorithmic_bytes_per_launch": b_read * my_nnz / launches,
                     "kernel": kernel_name,
                     "bytes_per_update_read": b_read, "bytes_per_update_total": b_total,
                     "avg_launch_us": avg_launch_s * 1e6, "launches_per_epoch": launches,
                     "event_pair_avg_launch_us": kern_ms * 1e3 / launches,
                     "frac_total_bytes": (b_total * my_nnz / launches) / avg_launch_s / HBM_PEAK},
        "device_ms_per_step": dev_ms / a.steps,
        "schedule_build_ms": info["schedule_us"] / 1e3,
    }
    if traffic is not None:
        line["roofline"]["traffic_unit"] = "bytes per launch (PMC, %s)" % traffic_src
    if not multi and resident:
        # the epoch is a chain of dependent hand-offs, not a stream: the floor that chain sets
        line["roofline"].update(chain_floor(info["n_levels"], avg_launch_s * 1e3))
        # what one level of the dependency DAG costs inside the launch (~3.7 us = one dependent step when the epoch is
        # a pure chain, more when a level holds more ratings than the waves turn over per step)
        line["roofline"]["us_per_chain_level"] = avg_launch_s * 1e6 / max(info["n_levels"], 1)
    if multi and "se_local" in J.last_rmse:
        line["rmse_online_last_rank0_shard"] = float(np.sqrt(J.last_rmse["se_local"] / max(my_nnz, 1)))
    if multi:
        # `value` is the AVERAGING mode (north_star's user shards + item-delta all-reduce): a different trajectory than
        # the sequential loop.  Its final-RMSE gap is measured here on a bounded CPU simulation of the same algorithm
        # (test infrastructure: oracle/sharded_sim.py) at cfg 3 / 100; the mode that keeps the reference's RMSE is DSGD.
        try:
            line["rmse_gap_vs_sequential"] = averaging_gap(world, min(a.steps, 5), k)
        except Exception as e:
            line["rmse_gap_vs_sequential"] = {"error": repr(e)}
        line["parity_mode"] = "also.dsgd_exact: block rotation, bit-identical to the sequential loop on its block order " \
                              "(RMSE gap 0); `value` is parameter averaging and misses north_star's 1e-5 by the gap above"
    return line


def chain_floor(levels, launch_ms):
    """The epoch / pass as a chain: `levels` dependent hand-offs, each at least one producer-store -> consumer-poll hop through
    the coherent path.  The hop is the MINIMUM measured on this part in isolation (tools/hop_latency.hip, committed under
    profiles/): nothing of the kernel's own work is in it, so launch time / floor says how much of the launch is chain."""
    hop_us, src = HOP_US_DEFAULT, "default"
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hop_latency.json")))
        if cands:
            hj = json.load(open(cands[-1]))
            hop_us, src = float(hj["min_hop_us"]), os.path.basename(cands[-1])
    except Exception:
        pass
    floor_ms = levels * hop_us * 1e-3
    return {"chain_levels": levels, "min_hop_us": hop_us, "min_hop_source": src, "chain_floor_ms": floor_ms,
            "frac_of_chain_floor": floor_ms / launch_ms if launch_ms > 0 else None}


def one_gpu_same_workload(J):
    """N > 1, rank 0 only, AFTER the timed region (no collective inside: the other ranks wait at their next barrier):
    the SAME job -- every rating of it, the same k -- on this rank's ONE GPU, so that the line carries its own
    denominator: the one-GPU block order (sharding.DsgdKMF, world = 1: item blocks on the static resident-side epoch,
    exact on that order), the local epoch every user shard of the sharded run uses."""
    from mfrec_amd import synth
    from mfrec_amd.plan import Timer
    from mfrec_amd.sharding import DsgdKMF, one_gpu_block_counts
    a, nu, ni, nnz, k = J.a, J.nu, J.ni, J.nnz, J.k
    steps = max(1, min(a.steps, 3))
    nb = one_gpu_block_counts(nu, ni, k)[1]
    t0 = time.perf_counter()
    dk = DsgdKMF(J.idx, J.r, nu, ni, k, -1, 1, balanced=True, item_blocks_1gpu=nb)
    build_s = time.perf_counter() - t0
    try:
        dk.set_factors(J.u0, J.v0_full if J.v0_full is not None else synth.init_normal(k, nu, synth.SEED_INIT + 1),
                       np.zeros(ni), np.zeros(nu))
        dk.train(J.LINEAR, 1, LR, KU, KI, KB)
        J.S.sync()
        t0 = time.perf_counter()
        with Timer() as tm:
            dk.train(J.LINEAR, steps, LR, KU, KI, KB)
        J.S.sync()
        wall = time.perf_counter() - t0
        info = dk.info()
    finally:
        dk.close()
    return {"value": nnz * steps / wall, "unit": "rating-updates/s", "ms_per_step": wall * 1e3 / steps,
            "device_ms_per_step": tm.ms / steps, "steps": steps, "warmup": 1, "item_blocks": nb,
            "blocks_on_the_resident_epoch": info.get("resident_blocks"), "plan_build_s": round(build_s, 1),
            "workload": "the same %d ratings, k=%d, on rank 0's ONE GPU: one-GPU block order (%d item blocks, each in its "
                        "minimum-level order, static resident-side epoch), exact on that order; wall time between stream "
                        "synchronisations" % (nnz, k, nb)}


def dsgd_leg(J, line, also, emit):
    """N > 1: the exact block-rotation mode on the same ratings, timed the same way.  The main line exists
    already: a watchdog prints it (without this leg) and ends the process if the leg -- whose send/recv ring no
    one-GPU box can rehearse -- does not come back, so the averaging-mode `value` survives a hang here."""
    import threading
    a, rank, world = J.a, J.rank, J.world
    limit_s = float(os.environ.get("MFX_BENCH_DSGD_TIMEOUT", "240"))

    def give_up():
        also["dsgd_exact"] = {"error": "no result after %.0f s (watchdog): leg abandoned" % limit_s}
        fail_line(rank, world, "dsgd leg", "watchdog after %.0f s" % limit_s)
        emit()
        os._exit(4)          # a hang in a GPU collective is a failure of the run, whatever was printed

    J.where[0] = "dsgd leg (block rotation: ncclSend/Recv ring)"
    dog = threading.Timer(limit_s, give_up)
    dog.daemon = True
    dog.start()
    try:
        from mfrec_amd import sharding as SH
        J.plan.close()
        J.plan = None
        nu, ni, k = J.nu, J.ni, J.k
        # item blocks per GPU: as many as it takes for a block's items to fit the static resident-side epoch (cfg 3: 3 at
        # N = 2, 2 at N = 4, 1 at N = 8; a block rests m - 1 steps in transit)
        m_blocks = max(1, -(-(ni // world) // 42000))
        d = SH.DsgdKMF(J.idx, J.r, nu, ni, k, rank, world, user_bounds=J.bounds, balanced=True,
                       item_blocks_1gpu=(world * m_blocks if m_blocks > 1 else None))
        d.set_factors(J.u0, J.v0, np.zeros(ni), np.zeros(max(J.nu_local, 1)))
        d.train(J.LINEAR, max(a.warmup, 1), LR, KU, KI, KB)
        J.barrier()
        t0 = time.perf_counter()
        se = d.train(J.LINEAR, a.steps, LR, KU, KI, KB)
        J.barrier()
        w2 = J.max_over_ranks(time.perf_counter() - t0)
        di = d.info()
        also["dsgd_exact"] = {"value": J.job_nnz * a.steps / w2, "unit": "rating-updates/s",
                              "ms_per_step": w2 * 1e3 / a.steps, "rmse_online_last": float(np.sqrt(se[-1] / J.job_nnz)),
                              "launches_per_epoch_per_gpu": di["launches_per_epoch"],
                              "workload": "%s: the same ratings, %d x %d user/item blocks, every block in its own "
                                          "minimum-level (balanced) order, block rotation over a ncclSend/Recv ring; "
                                          "bit-identical to the sequential loop on that order "
                                          "(tests/test_gpu_fakeworld.py)" % (J.cfg_name, world, world * m_blocks),
                              "blocks_on_the_resident_epoch": di.get("resident_blocks")}
        one = line.get("one_gpu_same_workload") if line else None
        if one and "value" in one:
            also["dsgd_exact"]["speedup_vs_one_gpu"] = also["dsgd_exact"]["value"] / one["value"]
        d.close()
    except Exception as e:
        fail_line(rank, world, "dsgd leg", e)
        also["dsgd_exact"] = {"error": repr(e)}
    dog.cancel()


def also_legs(J, also):
    """N = 1: the other kernels of the path (secondary, clearly labelled; never `value`)."""
    from mfrec_amd import synth
    from mfrec_amd.plan import SgdPlan
    a, idx, r, nu, ni, nnz, k, u0 = J.a, J.idx, J.r, J.nu, J.ni, J.nnz, J.k, J.u0
    LINEAR = J.LINEAR
    v0n = J.v0
    only = set(x for x in a.also_only.split(",") if x)

    def leg(name, fn):
        if only and name not in only:
            return
        if os.environ.get("MFX_BENCH_TRACE") == "1":
            sys.stderr.write("[bench] leg %s\n" % name); sys.stderr.flush()
        t = time.perf_counter()
        try:
            also[name] = fn()
            also[name]["leg_wall_s"] = round(time.perf_counter() - t, 1)
        except Exception as e:                      # never let a secondary leg break the bench line
            also[name] = {"error": repr(e)}

    J.plan.close()
    J.plan = None
    leg("k1_levels", lambda: also_k1_order(idx, r, nu, ni, k, u0, v0n, a.steps, a.warmup, "shuffled", epoch="levels"))
    leg("k1_user_major", lambda: also_k1_order(idx, r, nu, ni, k, u0, v0n, a.steps, a.warmup, "user-major"))
    leg("k1_balanced", lambda: also_k1_order(idx, r, nu, ni, k, u0, v0n, a.steps, a.warmup, "balanced"))
    leg("k1_tolerance_mode", lambda: also_k1_tolerance(idx, r, nu, ni, k, u0, v0n, a.steps, a.warmup, "shuffled"))
    leg("k1_tolerance_mode_balanced", lambda: also_k1_tolerance(idx, r, nu, ni, k, u0, v0n, a.steps, a.warmup, "balanced"))
    leg("k1_f32_rows", lambda: also_k1_tolerance(idx, r, nu, ni, k, u0, v0n, a.steps, a.warmup, "shuffled", variant="f32"))
    leg("k1_f32_rows_balanced", lambda: also_k1_tolerance(idx, r, nu, ni, k, u0, v0n, a.steps, a.warmup, "balanced", variant="f32"))
    leg("k1_zipf08", lambda: also_k1_skew(nu, ni, nnz, k, u0, v0n, 0.8))
    leg("k1_ml10m_shape", lambda: also_k1_ml10m(k))
    leg("k2_funk", lambda: also_funk(idx, r, nu, ni, k, "shuffled"))
    leg("k2_funk_balanced", lambda: also_funk(idx, r, nu, ni, k, "balanced"))
    if not a.no_cpu and "k2_funk" in also and "error" not in also["k2_funk"]:
        try:
            also["k2_funk"]["cpu_baseline"] = cpu_baseline_funk(idx, r, nu, ni)
        except Exception as e:
            also["k2_funk"]["cpu_baseline"] = {"error": repr(e)}

    leg("k2_funk_ml10m_shape", lambda: also_funk_ml10m(not a.no_cpu))
    leg("topn_batch", lambda: also_topn(idx, r, nu, ni, k, not a.no_cpu))

    def cfg3_and_als():
        nu3, ni3, nnz3, k3 = CFG3
        idx3, r3 = synth.make_ratings(nu3, ni3, nnz3)
        out = {}
        with SgdPlan(idx3, r3, nu3, ni3, k3) as p3:
            p3.set_factors(synth.init_normal(k3, ni3, synth.SEED_INIT), synth.init_normal(k3, nu3, synth.SEED_INIT + 1),
                           np.zeros(ni3), np.zeros(nu3))
            m = k1_measure(p3, nnz3, nu3, k3, a.steps, a.warmup,
                           lambda n: p3.run_kmf(LINEAR, n, LR, KU, KI, KB, want_rmse=False))
        m["unit"] = "rating-updates/s"
        m["workload"] = "cfg3 whole on ONE GPU (1M x 200K, 100M ratings, k=64), given (shuffled) order"
        out["k1_cfg3_whole_1gpu"] = m
        # the same 100 M ratings in get_ratings(randomize_order=False)'s order (base.py:1115-1126): user rows
        # resident, one persistent launch per epoch (MFX_EPOCH_TICKET by the cost model)
        try:
            i3u, r3u = synth.user_major(idx3, r3)
            with SgdPlan(i3u, r3u, nu3, ni3, k3) as p3:
                p3.set_factors(synth.init_normal(k3, ni3, synth.SEED_INIT), synth.init_normal(k3, nu3, synth.SEED_INIT + 1),
                               np.zeros(ni3), np.zeros(nu3))
                m2 = k1_measure(p3, nnz3, nu3, k3, a.steps, a.warmup,
                                lambda n: p3.run_kmf(LINEAR, n, LR, KU, KI, KB, want_rmse=False))
            m2["unit"] = "rating-updates/s"
            m2["workload"] = "cfg3 whole on ONE GPU, user-major order (get_ratings(randomize_order=False)): user rows " \
                             "resident in registers, item rows travel behind chain counters; epoch kernel: %s" % m2["epoch_kernel"]
            out["k1_user_resident_cfg3"] = m2
            del i3u, r3u
        except Exception as e:
            out["k1_user_resident_cfg3"] = {"error": repr(e)}
        # the same 100 M ratings on the item-block order (sharding.DsgdKMF, world = 1): 200 K items do not fit the register file
        # at once, four blocks of 50 K do -- every block on the static resident-side epoch; exact on that order.  Twice: the
        # given order inside the blocks, and every block in its own minimum-level (balanced) order (`value`)
        try:
            from mfrec_amd.sharding import DsgdKMF
            from mfrec_amd.plan import Timer
            from mfrec_amd.sharding import one_gpu_block_counts
            nb = one_gpu_block_counts(nu3, ni3, k3)[1]
            res_b = {}
            for bal in (False, True):
                t0 = time.perf_counter()
                dk = DsgdKMF(idx3, r3, nu3, ni3, k3, -1, 1, balanced=bal, item_blocks_1gpu=nb)
                build_s = time.perf_counter() - t0
                dk.set_factors(synth.init_normal(k3, ni3, synth.SEED_INIT), synth.init_normal(k3, nu3, synth.SEED_INIT + 1),
                               np.zeros(ni3), np.zeros(nu3))
                dk.train(LINEAR, max(a.warmup, 1), LR, KU, KI, KB)
                with Timer() as tmb:
                    se_b = dk.train(LINEAR, a.steps, LR, KU, KI, KB)
                info_b = dk.info()
                dk.close()
                res_b[bal] = (nnz3 * a.steps / (tmb.ms * 1e-3), tmb.ms / a.steps, info_b, build_s, float(np.sqrt(se_b[-1] / nnz3)))
            rate_b, ms_b, info_b, build_s, rm_b = res_b[True]
            br3 = algorithmic_bytes(k3, nnz3, nu3)[0]
            out["k1_cfg3_item_blocks_1gpu"] = {
                "value": rate_b, "unit": "rating-updates/s", "device_ms_per_step": ms_b, "item_blocks": nb,
                "blocks_on_the_resident_epoch": info_b["resident_blocks"], "levels_total": info_b["levels_total"],
                "plan_build_s": round(build_s, 1), "rmse_online_last": rm_b,
                "given_order_inside_the_blocks": {"value": res_b[False][0], "device_ms_per_step": res_b[False][1],
                                                  "levels_total": res_b[False][2]["levels_total"], "plan_build_s": round(res_b[False][3], 1)},
                "roofline": {"bound": "hbm", "achieved": rate_b * br3 / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": rate_b * br3 / HBM_PEAK, "traffic": None,
                             "kernel": "k1_res (static), one persistent launch per item block and epoch",
                             "note": "measured ceiling of the access pattern: 5.5 G ratings/s (tools/rmw_ceiling.hip)"},
                "workload": "cfg3 whole on ONE GPU, ITEM-BLOCK order: the given (shuffled) ratings grouped by %d item blocks (stable), "
                            "every block in its own minimum-level order; exact on that order (bit-identical to the sequential loop: "
                            "tests/test_gpu_fakeworld.py, tests/test_gpu_scale.py).  The N = 1 point of the curve `--gpus N` continues "
                            "(the N > 1 line times this same job again as `one_gpu_same_workload`): at N > 1 every user shard runs this "
                            "very local epoch (item blocks on the static resident-side epoch)" % nb}
        except Exception as e:
            out["k1_cfg3_item_blocks_1gpu"] = {"error": repr(e)}
        del r3
        t = time.perf_counter()
        try:
            out["k3_als_cfg4"] = also_als(idx3, nu3, ni3, 128)
            out["k3_als_cfg4"]["leg_wall_s"] = round(time.perf_counter() - t, 1)
        except Exception as e:
            out["k3_als_cfg4"] = {"error": repr(e)}
        return out

    if not only or (only & {"k1_cfg3_whole_1gpu", "k1_user_resident_cfg3", "k1_cfg3_item_blocks_1gpu", "k3_als_cfg4"}):
        if os.environ.get("MFX_BENCH_TRACE") == "1":
            sys.stderr.write("[bench] leg cfg3 + als\n"); sys.stderr.flush()
        try:
            also.update(cfg3_and_als())
        except Exception as e:
            also["k1_cfg3_whole_1gpu"] = {"error": repr(e)}
    if not a.no_cpu and "k3_als_cfg4" in also and "error" not in also["k3_als_cfg4"]:
        try:
            cb = cpu_baseline_als(128)
            # scaled to cfg 4 by the FLOP model: seconds per epoch the CPU path would need
            cb["cfg4_epoch_s_extrapolated"] = also["k3_als_cfg4"]["model_tflop_per_epoch"] * 1e3 / cb["model_gflops"]
            also["k3_als_cfg4"]["cpu_baseline"] = cb
        except Exception as e:
            also["k3_als_cfg4"]["cpu_baseline"] = {"error": repr(e)}


def main():
    a = parse_args()
    if a.cpu_replica:
        a.users = a.users or CFG2[0]; a.items = a.items or CFG2[1]
        return cpu_replica_main(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process is only the launcher (nothing above has touched the GPU or loaded
        # the library) -- N fresh child ranks, rank 0's JSON line relayed, worst child exit code returned
        return self_launch(a.gpus)

    J = init_ranks(a)
    init_device_and_comm(J)
    build_plan(J)
    timed_region(J)
    line = headline_line(J)
    rank, multi = J.rank, J.multi

    also = {}
    printed = []

    def emit():
        if rank == 0 and not printed:
            printed.append(1)
            if also:
                line["also"] = also
            print(json.dumps(line), flush=True)

    if multi and a.scaling == "strong" and rank == 0 and os.environ.get("MFX_BENCH_NO_ONE_GPU") != "1":
        # the line's own denominator: the same job on ONE GPU (rank 0's), timed here -- after `value` is safe, before the DSGD leg
        J.mark("one-GPU run of the same workload (rank 0 only; the other ranks wait at their next barrier)")
        try:
            one = one_gpu_same_workload(J)
            line["one_gpu_same_workload"] = one
            line["speedup_vs_one_gpu"] = line["value"] / one["value"]
        except Exception as e:
            fail_line(rank, J.world, "one-GPU run of the same workload", e)
            line["one_gpu_same_workload"] = {"error": repr(e)}
            line["speedup_vs_one_gpu"] = None

    if multi and a.scaling == "strong" and not a.no_also:
        dsgd_leg(J, line, also, emit)

    if rank == 0 and not multi and not a.no_also and a.storage == "f64" and (J.nu, J.ni, J.nnz, J.k) == CFG2 \
            and a.order == "shuffled":
        also_legs(J, also)

    if rank == 0:
        if not a.no_cpu and not multi:          # rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline_k1(J.idx, J.r, J.nu, J.ni, J.k)
    emit()
    if J.plan is not None:
        J.plan.close()
    if multi:
        J.S.check(J.S.lib().mfx_comm_barrier())
        J.S.check(J.S.lib().mfx_comm_destroy())
        from mfrec_amd import rendezvous
        rendezvous.cleanup(rank)


if __name__ == "__main__":
    sys.exit(main() or 0)
