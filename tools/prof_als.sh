#!/bin/bash
# rocprofv3 for the ALS kernels: kernel stats + MFMA busy counters (separate PMC pass)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_als_${1:-r1}
rm -rf $O; mkdir -p $O
ARGS="als --users 100000 --items 20000 --nnz 10000000 --k 128 --epochs 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_families.py $ARGS > $O/bench.json 2> $O/trace.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -- python3 tools/bench_families.py $ARGS > $O/bench_pmc.json 2> $O/pmc.log
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
st = glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)
if st:
    out["kernel_stats"] = [r for r in csv.DictReader(open(st[0]))]
f = glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True)
if f:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"], r["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out["pmc"] = [{"kernel": k[0], "counter": k[1], "dispatches": v[0], "sum": v[1]} for k, v in agg.items()]
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for r in out.get("kernel_stats", [])[:8]:
    print(r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
for e in out.get("pmc", []):
    if "als" in e["kernel"] or "gram" in e["kernel"]:
        print(e["kernel"][:60], e["counter"], e["dispatches"], e["sum"])
PY
cat $O/bench.json | tail -1
find $O -name "*kernel_trace.csv" -delete
