#!/usr/bin/env python3
"""Copy the summaries tools/prof_r02.sh left under gpurun_out/prof_r02 into profiles/ (the tracked evidence) and derive
r02_traffic.json (HBM-side bytes per launch / per rating of k1_level: FETCH_SIZE x2 on gfx950 + WRITE_SIZE)."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "prof_r02"), os.path.join(ROOT, "profiles")
for src, dst in (("kernel_stats_k1.csv", "r02_bench_kernel_stats.csv"), ("kernel_stats_k2.csv", "r02_k2_funk_kernel_stats.csv"),
                 ("kernel_stats_k1v.csv", "r02_k1_tolerance_kernel_stats.csv"), ("kernel_stats_als.csv", "r02_als_k128_kernel_stats.csv"),
                 ("bench_k1.json", "r02_bench_line_under_rocprof.json"), ("bench_als.json", "r02_als_k128_line.json")):
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))
s = json.load(open(os.path.join(O, "summary.json")))
keep = {k: [e for e in s[k] if ("k1_level" in e["kernel"] or "als_rows" in e["kernel"])] for k in ("fetch", "write", "alspmc")}
json.dump(keep, open(os.path.join(P, "r02_pmc_summary.json"), "w"), indent=1)
line = json.loads([l for l in open(os.path.join(O, "bench_fetch.json")) if l.startswith("{")][0])
f = [e for e in keep["fetch"] if "k1_level" in e["kernel"]][0]
w = [e for e in keep["write"] if "k1_level" in e["kernel"]][0]
ratings = line["config"]["ratings"] / line["config"]["launches_per_epoch"]
rd, wr = f["avg"] * 1024 * 2, w["avg"] * 1024
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                     "--warmup 1 --no-cpu --no-also; tools/prof_r02.sh + tools/collect_prof_r02.py",
           "kernel": f["kernel"], "launches_profiled": f["dispatches"],
           "FETCH_SIZE_avg_per_launch_KB": f["avg"], "WRITE_SIZE_avg_per_launch_KB": w["avg"],
           "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
           "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "ratings_per_launch": ratings,
           "read_bytes_per_rating": rd / ratings, "write_bytes_per_rating": wr / ratings},
          open(os.path.join(P, "r02_traffic.json"), "w"), indent=1)
a = {e["counter"]: e["avg"] for e in keep["alspmc"]}
print("k1 traffic per rating: read %.0f B write %.0f B" % (rd / ratings, wr / ratings))
print("ALS MFMA busy: %.1f %%" % (100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024)))
