#!/usr/bin/env python3
"""Copy the round-4 summaries that tools/prof_r04.sh, tools/prof_k2.sh, tools/prof_als_r04.sh and tools/res_stamps.py left under
gpurun_out/ into profiles/ (the tracked evidence) and derive the traffic files bench.py attaches to its roofline objects:
r04_traffic.json (k1_res: FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes + WRITE_SIZE, separate --pmc passes) and
r04_k2_traffic.json (the Funk pass)."""
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


O = os.path.join(G, "prof_r04")
if os.path.isdir(O):
    print("prof_r04:")
    cp(os.path.join(O, "kernel_stats_all.csv"), "r04_bench_kernel_stats.csv")
    cp(os.path.join(O, "kernel_stats_head.csv"), "r04_bench_headline_kernel_stats.csv")
    cp(os.path.join(O, "bench_all.json"), "r04_bench_line_under_rocprof.json")
    cp(os.path.join(O, "bench_head.json"), "r04_bench_headline_line_under_rocprof.json")
    cp(os.path.join(O, "bench_line.json"), "r04_bench_line.json")
    cp(os.path.join(O, "multi_1rank.json"), "r04_bench_multi_path_rehearsal_1rank.json")
    cp(os.path.join(O, "multi_4ranks.json"), "r04_bench_selflaunch_4ranks_1gpu.json")
    cp(os.path.join(O, "multi_4ranks.err"), "r04_bench_selflaunch_4ranks_1gpu.stderr.txt")
    s = json.load(open(os.path.join(O, "summary.json")))
    keep = {k: [e for e in s[k] if "k1_res" in e["kernel"]] for k in ("fetch", "write") if isinstance(s.get(k), list)}
    json.dump(keep, open(os.path.join(P, "r04_pmc_summary.json"), "w"), indent=1)
    if keep.get("fetch") and keep.get("write"):
        line = line_of(os.path.join(O, "bench_fetch.json"))
        f, w = keep["fetch"][0], keep["write"][0]
        ratings = line["config"]["ratings"] / line["config"]["launches_per_epoch"]
        rd, wr = f["avg"] * 1024 * 2, w["avg"] * 1024
        d = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                       "--no-cpu --no-also; tools/prof_r04.sh + tools/collect_prof_r04.py",
             "kernel": line["roofline"]["kernel"], "pmc_kernel": f["kernel"], "launches_profiled": f["dispatches"],
             "FETCH_SIZE_avg_per_launch_KB": f["avg"], "WRITE_SIZE_avg_per_launch_KB": w["avg"],
             "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact. The x2 also "
                           "doubles the 64-byte requests of the 16-byte granule polls and record loads, which are tallied at their size: the read "
                           "figure is an upper bound",
             "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "ratings_per_launch": ratings,
             "read_bytes_per_rating": rd / ratings, "write_bytes_per_rating": wr / ratings,
             "read_bytes_per_rating_uncorrected": rd / 2 / ratings,
             "read_bytes_per_rating_rows_corrected_only": rd / 2 / ratings + 256.0}
        json.dump(d, open(os.path.join(P, "r04_traffic.json"), "w"), indent=1)
        print("   r04_traffic.json: read %.0f B (rows-only correction %.0f) + written %.0f B per rating" % (
            d["read_bytes_per_rating"], d["read_bytes_per_rating_rows_corrected_only"], d["write_bytes_per_rating"]))

K2 = os.path.join(G, "k2_pmc")
if os.path.isdir(K2):
    print("k2_pmc:")
    cp(os.path.join(K2, "kernel_stats.csv"), "r04_k2_kernel_stats.csv")
    cp(os.path.join(K2, "summary.json"), "r04_k2_pmc.json")
    cp(os.path.join(K2, "bench_stats.json"), "r04_k2_bench_line_under_rocprof.json")
    s = json.load(open(os.path.join(K2, "summary.json")))
    main = [k for k in s if k.startswith("k2_res")]
    if main:
        c = s[main[0]]
        line = line_of(os.path.join(K2, "bench_p1.json"))["also"]["k2_funk"]
        nnz = 10000000
        rd, wr = c["FETCH_SIZE"]["avg"] * 1024, c["WRITE_SIZE"]["avg"] * 1024
        d = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu "
                       "--also-only k2_funk; tools/prof_k2.sh + tools/collect_prof_r04.py",
             "kernel": line["roofline"]["kernel"], "pmc_kernel": main[0], "launches_profiled": c["FETCH_SIZE"]["dispatches"],
             "note": "the pass's requests are 64-byte ones (16-byte mailbox polls, 16-byte records, 8-byte cache words and stores): FETCH_SIZE is "
                     "taken at its tallied size, no x2 (that correction is for 128-byte requests)",
             "read_bytes_per_pass": rd, "write_bytes_per_pass": wr, "bytes_per_pass": rd + wr,
             "read_bytes_per_update": rd / nnz, "write_bytes_per_update": wr / nnz}
        json.dump(d, open(os.path.join(P, "r04_k2_traffic.json"), "w"), indent=1)
        print("   r04_k2_traffic.json: %.1f B read + %.1f B written per scalar update" % (d["read_bytes_per_update"], d["write_bytes_per_update"]))

A = os.path.join(G, "als_pmc_r04")
if os.path.isdir(A):
    print("als_pmc_r04:")
    cp(os.path.join(A, "kernel_stats.csv"), "r04_als_cfg4_kernel_stats.csv")
    cp(os.path.join(A, "summary.json"), "r04_als_cfg4_pmc_per_dispatch.json")
    cp(os.path.join(A, "dispatch_times.txt"), "r04_als_cfg4_dispatch_times.txt")

st = sorted(glob.glob(os.path.join(G, "res_stamps_*.txt")))
if st:
    with open(os.path.join(P, "r04_k1_res_stamps.txt"), "w") as o:
        o.write("# tools/res_stamps.py (libmfx_diag.so, -DMFX_RES_STAMPS): shader-clock stamps at the phase boundaries of every step of every 64th wave,\n"
                "# cfg 2, static binding; the runs of round 4 in the order they were taken (each file = one gpurun call; the kernels changed in between:\n"
                "# 1 = after the s_waitcnt fix, with the look-ahead variant (since archived); 2 = + readlane broadcasts and the pipelined ordered sum;\n"
                "# 3 = finer stamps inside the compute phase; 4 = + uniform set skips and s_setprio (dropped); 5 = exact vs butterfly-dot, set-major;\n"
                "# 6 = butterfly-dot stage-major over all sets (dropped); 7 = the final kernel on the balanced (stream-bound: 6.3 us steps, 8.8 ratings each) and the\n"
                "# user-major order (chain-bound: 1 266 steps of 3.5 us); 8 = the tolerance mode with the LDS tree sum (kept)).  With the stamps an epoch takes ~2.9-3.0 ms instead of 2.7.\n")
        for f in st:
            o.write("\n## %s\n" % os.path.basename(f)); o.write(open(f).read())
    print("   r04_k1_res_stamps.txt (%d runs)" % len(st))
for f, dst in (("ab_look_waitfix.txt", "r04_k1_look_ab_waitfix.txt"), ("ab_look_readlane.txt", "r04_k1_look_ab_readlane.txt"), ("ab_look_setskip.txt", "r04_k1_look_ab_setskip.txt")):
    cp(os.path.join(G, f), dst)
