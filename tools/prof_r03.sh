#!/bin/bash
# Round-3 rocprofv3 evidence.  (1) the DEFAULT bench traced as ONE process (the run that aborted in round 2);
# (2) FETCH_SIZE / WRITE_SIZE of the headline kernel in separate PMC passes; (3) the cfg-3 user-resident leg's traffic.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
export MFX_BENCH_TRACE=1
# (MFX_GRAPH=0: the same kernels launched eagerly.  Under rocprofv3 --kernel-trace a RE-launch of an executable hipGraph that is
#  submitted when the process has pushed ~16 384 AQL packets through the stream's queue crashes inside the tracer's queue
#  interception -- library-free reproducer: tools/graph_trace_repro.hip, evidence: profiles/r03_rocprofv3_graph_abort/)
echo "[prof] default bench, one process, kernel trace"
MFX_GRAPH=0 timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/all -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $O/bench_all.json 2> $O/all.log
echo "[prof] rc=$? ; $(grep -c 'stage\|leg' $O/all.log) stage lines; last: $(grep 'stage\|leg' $O/all.log | tail -1)"
echo "[prof] headline leg alone, kernel trace"
MFX_GRAPH=0 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/head -- python3 bench.py --no-cpu --no-also > $O/bench_head.json 2> $O/head.log
echo "[prof] fetch"; timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_fetch.json 2> $O/fetch.log
echo "[prof] write"; timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_write.json 2> $O/write.log
echo "[prof] cfg3 user-resident fetch"; timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also --users 1000000 --items 200000 --nnz 100000000 --order user-major > $O/bench_fetch3.json 2> $O/fetch3.log
echo "[prof] cfg3 user-resident write"; timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also --users 1000000 --items 200000 --nnz 100000000 --order user-major > $O/bench_write3.json 2> $O/write3.log
echo "[prof] summarise"
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
st = glob.glob(O + "/all/**/*kernel_stats.csv", recursive=True)
if st:
    rows = [r for r in csv.DictReader(open(st[0]))]
    with open(O + "/kernel_stats_all.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    for r in rows[:14]:
        print("  ", r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
else:
    print("== kernel stats of the one-process run: missing")
for name in ("fetch", "write", "fetch3", "write3"):
    f = glob.glob(O + "/%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        out[name] = "no counter file"; continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[name] = [{"kernel": k[0], "counter": k[1], "dispatches": v[0], "sum": v[1], "avg": v[1] / v[0]} for k, v in agg.items()]
    for e in out[name]:
        if "k1_res" in e["kernel"] or "k1_level" in e["kernel"]:
            print(name, e["kernel"][:70], e["counter"], "dispatches", e["dispatches"], "avg", e["avg"])
json.dump(out, open(O + "/summary.json", "w"), indent=1)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
du -sh $O; (grep -i -E "malformed|abort|INVALID" $O/*.log | head) || true
