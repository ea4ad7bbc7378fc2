#!/usr/bin/env python3
"""Copy the round-5 summaries that tools/prof_r05.sh left under
gpurun_out/ into profiles/ (the tracked evidence) and derive the traffic files bench.py attaches to its roofline objects:
r05_traffic.json (k1_res: FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes + WRITE_SIZE, separate --pmc passes).  (K2 and
K3 kernels are unchanged since round 4: profiles/r04_k2_*, r04_als_* stand.)"""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def line_of(path):
    return json.loads([l for l in open(path) if l.startswith("{")][0])


def cp(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst)); print("  ", dst)
    else:
        print("   MISSING", src)


O = os.path.join(G, "prof_r05")
if os.path.isdir(O):
    print("prof_r05:")
    cp(os.path.join(O, "kernel_stats_all.csv"), "r05_bench_kernel_stats.csv")
    cp(os.path.join(O, "kernel_stats_head.csv"), "r05_bench_headline_kernel_stats.csv")
    cp(os.path.join(O, "bench_all.json"), "r05_bench_line_under_rocprof.json")
    cp(os.path.join(O, "bench_head.json"), "r05_bench_headline_line_under_rocprof.json")
    cp(os.path.join(O, "bench_line.json"), "r05_bench_line.json")
    cp(os.path.join(O, "multi_1rank.json"), "r05_bench_multi_path_rehearsal_1rank.json")
    cp(os.path.join(O, "multi_4ranks.json"), "r05_bench_selflaunch_4ranks_1gpu.json")
    cp(os.path.join(O, "multi_4ranks.err"), "r05_bench_selflaunch_4ranks_1gpu.stderr.txt")
    s = json.load(open(os.path.join(O, "summary.json")))
    keep = {k: [e for e in s[k] if "k1_res" in e["kernel"]] for k in ("fetch", "write") if isinstance(s.get(k), list)}
    json.dump(keep, open(os.path.join(P, "r05_pmc_summary.json"), "w"), indent=1)
    if keep.get("fetch") and keep.get("write"):
        line = line_of(os.path.join(O, "bench_fetch.json"))
        f, w = keep["fetch"][0], keep["write"][0]
        ratings = line["config"]["ratings"] / line["config"]["launches_per_epoch"]
        rd, wr = f["avg"] * 1024 * 2, w["avg"] * 1024
        d = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                       "--no-cpu --no-also; tools/prof_r05.sh + tools/collect_prof_r05.py",
             "kernel": line["roofline"]["kernel"], "pmc_kernel": f["kernel"], "launches_profiled": f["dispatches"],
             "FETCH_SIZE_avg_per_launch_KB": f["avg"], "WRITE_SIZE_avg_per_launch_KB": w["avg"],
             "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact. The x2 also "
                           "doubles the 64-byte requests of the 16-byte granule polls and record loads, which are tallied at their size: the read "
                           "figure is an upper bound",
             "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "ratings_per_launch": ratings,
             "read_bytes_per_rating": rd / ratings, "write_bytes_per_rating": wr / ratings,
             "read_bytes_per_rating_uncorrected": rd / 2 / ratings,
             "read_bytes_per_rating_rows_corrected_only": rd / 2 / ratings + 256.0}
        json.dump(d, open(os.path.join(P, "r05_traffic.json"), "w"), indent=1)
        print("   r05_traffic.json: read %.0f B (rows-only correction %.0f) + written %.0f B per rating" % (
            d["read_bytes_per_rating"], d["read_bytes_per_rating_rows_corrected_only"], d["write_bytes_per_rating"]))

