#!/bin/bash
# Round-2 rocprofv3 evidence.  One profiled run per kernel family (the full default bench under the tracer
# is too much trace): K1 (the bench `value`), K2 (XCD-resident Funk pass), K3 (ALS cfg 4 / 10), then
# FETCH_SIZE / WRITE_SIZE of k1_level in separate PMC passes.  Progress lines keep the call alive.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
export MFX_BENCH_TRACE=1
echo "[prof] k1 trace"; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-also > $O/bench_k1.json 2> $O/k1.log
echo "[prof] k2 trace"; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --also-only k2_funk,k2_funk_balanced > $O/bench_k2.json 2> $O/k2.log
# (the K1 order / tolerance variants are the same k1_level kernel; a traced run of all three legs in one
# process made rocprofv3 abort with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT on this pool -- each leg alone traces
# fine (tools/prof_probe*.py) -- so only the tolerance variant is traced, by itself)
echo "[prof] k1 tolerance trace"; timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1v -- python3 tools/prof_probe2.py full > $O/bench_k1v.json 2> $O/k1v.log
echo "[prof] als trace"; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/als -- python3 tools/bench_families.py als --users 100000 --items 20000 --nnz 10000000 --k 128 --epochs 1 > $O/bench_als.json 2> $O/als.log
echo "[prof] als pmc"; timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/alspmc -- python3 tools/bench_families.py als --users 100000 --items 20000 --nnz 10000000 --k 128 --epochs 1 > $O/bench_alspmc.json 2> $O/alspmc.log
echo "[prof] fetch"; timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_fetch.json 2> $O/fetch.log
echo "[prof] write"; timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_write.json 2> $O/write.log
echo "[prof] summarise"
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
for name in ("k1", "k2", "k1v", "als"):
    st = glob.glob(O + "/%s/**/*kernel_stats.csv" % name, recursive=True)
    if st:
        rows = [r for r in csv.DictReader(open(st[0]))]
        out["kernel_stats_" + name] = rows
        with open(O + "/kernel_stats_%s.csv" % name, "w") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
        print("==", name)
        for r in rows[:6]:
            print("  ", r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
    else:
        out["kernel_stats_" + name] = "missing"; print("==", name, "missing")
for name in ("fetch", "write", "alspmc"):
    f = glob.glob(O + "/%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        out[name] = "no counter file"; continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[name] = [{"kernel": k[0], "counter": k[1], "dispatches": v[0], "sum": v[1], "avg": v[1] / v[0]} for k, v in agg.items()]
    for e in out[name]:
        if "k1_level" in e["kernel"] or "als_rows" in e["kernel"]:
            print(name, e["kernel"][:60], e["counter"], "dispatches", e["dispatches"], "avg", e["avg"])
json.dump(out, open(O + "/summary.json", "w"), indent=1)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
du -sh $O; (tail -q -n 3 $O/*.log | grep -i -E "malformed|abort" | head) || true
