#!/usr/bin/env python3
"""Copy the summaries tools/prof_r03.sh left under gpurun_out/prof_r03 into profiles/ (the tracked evidence) and derive
r03_traffic.json / r03_traffic_cfg3_user_resident.json: HBM-side bytes per launch / per rating of k1_res from FETCH_SIZE
(x2 on gfx950, as MI355X_MICROARCH.md prescribes) and WRITE_SIZE, each collected in its own --pmc pass."""
import json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "prof_r03"), os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(O, "kernel_stats_all.csv"), os.path.join(P, "r03_bench_kernel_stats.csv"))
shutil.copy(os.path.join(O, "bench_all.json"), os.path.join(P, "r03_bench_line_under_rocprof.json"))
s = json.load(open(os.path.join(O, "summary.json")))
keep = {k: [e for e in s[k] if "k1_res" in e["kernel"]] for k in ("fetch", "write", "fetch3", "write3")}
json.dump(keep, open(os.path.join(P, "r03_pmc_summary.json"), "w"), indent=1)


def traffic(fk, wk, line_file, out_name, cmd):
    line = json.loads([l for l in open(os.path.join(O, line_file)) if l.startswith("{")][0])
    f, w = keep[fk][0], keep[wk][0]
    ratings = line["config"]["ratings"] / line["config"]["launches_per_epoch"]
    rd, wr = f["avg"] * 1024 * 2, w["avg"] * 1024
    d = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + cmd +
                   "; tools/prof_r03.sh + tools/collect_prof_r03.py",
         "kernel": line["roofline"]["kernel"], "pmc_kernel": f["kernel"], "launches_profiled": f["dispatches"],
         "FETCH_SIZE_avg_per_launch_KB": f["avg"], "WRITE_SIZE_avg_per_launch_KB": w["avg"],
         "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact. "
                       "The x2 also doubles the 64-byte requests of the 16-byte granule polls and record loads, which are "
                       "tallied at their size: the read figure is an upper bound",
         "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "ratings_per_launch": ratings,
         "read_bytes_per_rating": rd / ratings, "write_bytes_per_rating": wr / ratings,
         "read_bytes_per_rating_uncorrected": rd / 2 / ratings,
         # the x2 applied ONLY to what is known to be fetched as 128-byte requests -- the travelling row of every rating (k = 64:
         # 512 B, tallied as 256) -- and the rest (16-byte granule polls, records, run tables: 64-byte requests) at its tallied size
         "read_bytes_per_rating_rows_corrected_only": rd / 2 / ratings + 256.0}
    json.dump(d, open(os.path.join(P, out_name), "w"), indent=1)
    print("%s: read %.0f B (uncorrected %.0f) + written %.0f B per rating" % (out_name, d["read_bytes_per_rating"],
                                                                             d["read_bytes_per_rating_uncorrected"], d["write_bytes_per_rating"]))


traffic("fetch", "write", "bench_fetch.json", "r03_traffic.json", "python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also")
traffic("fetch3", "write3", "bench_fetch3.json", "r03_traffic_cfg3_user_resident.json",
        "python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also --users 1000000 --items 200000 --nnz 100000000 --order user-major")
