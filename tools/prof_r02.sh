#!/bin/bash
# Round-2 rocprofv3 evidence: kernel-trace stats of the default bench (incl. the `also` legs: K2 XCD pass,
# ALS cfg 4, K1 variants), then FETCH_SIZE / WRITE_SIZE of k1_level in separate PMC passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $O/bench_trace.json 2> $O/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_fetch.json 2> $O/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_write.json 2> $O/write.log
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
st = glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)
if st:
    rows = [r for r in csv.DictReader(open(st[0]))]
    out["kernel_stats"] = rows
    with open(O + "/kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for name in ("fetch", "write"):
    f = glob.glob(O + "/%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        out[name] = "no counter file"; continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[name] = [{"kernel": k[0], "counter": k[1], "dispatches": v[0], "sum": v[1], "avg": v[1] / v[0]} for k, v in agg.items()]
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for r in out.get("kernel_stats", [])[:14]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
for name in ("fetch", "write"):
    if isinstance(out[name], list):
        for e in out[name]:
            if "k1_level" in e["kernel"]:
                print(name, e["counter"], "dispatches", e["dispatches"], "avg/launch", e["avg"])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
du -sh $O
