#!/bin/bash
# Round-5 rocprofv3 evidence + rehearsals.  (1) the DEFAULT bench traced as ONE process; (2) the headline leg alone; (3) FETCH_SIZE /
# WRITE_SIZE of the headline kernel in separate PMC passes; (4) the N > 1 code path rehearsed on this one-GPU box: one rank with a
# forced 1-rank RCCL communicator (real librccl), and a PLAIN `python bench.py --gpus 4` (self-launch) over the RCCL stand-in.
# (tools/prof_k2.sh and tools/prof_als_r05.sh are the per-kernel passes of K2 and K3.)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r05
rm -rf $O; mkdir -p $O
export MFX_BENCH_TRACE=1
# (MFX_GRAPH=0 under the tracer: a re-launched hipGraph crossing ~16 384 AQL packets crashes rocprofv3's queue interception --
#  tools/graph_trace_repro.hip, profiles/r03_rocprofv3_graph_abort/)
echo "[prof] default bench, one process, kernel trace"
MFX_GRAPH=0 timeout -k 5 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/all -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $O/bench_all.json 2> $O/all.log
echo "[prof] rc=$? ; last: $(grep 'stage\|leg' $O/all.log | tail -1)"
echo "[prof] headline leg alone, kernel trace"
MFX_GRAPH=0 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/head -- python3 bench.py --no-cpu --no-also > $O/bench_head.json 2> $O/head.log
echo "[prof] fetch"; timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_fetch.json 2> $O/fetch.log
echo "[prof] write"; timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also > $O/bench_write.json 2> $O/write.log
unset MFX_BENCH_TRACE
echo "[prof] the plain bench line (no tracer)"
timeout -k 5 900 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "rc=$?"
echo "[prof] N > 1 path, one rank, forced 1-rank RCCL communicator (real librccl)"
MFX_BENCH_FORCE_MULTI=1 timeout -k 5 600 python3 bench.py --steps 5 --warmup 2 > $O/multi_1rank.json 2> $O/multi_1rank.err; echo "rc=$?"
echo "[prof] plain python bench.py --gpus 4: self-launched ranks over the RCCL stand-in, all on this GPU (static epoch left ON)"
MFX_TESTING=1 MFX_RCCL_LIB=$PWD/build/libmock_rccl.so MFX_BENCH_DEVICE=0 timeout -k 5 900 python3 bench.py --gpus 4 --steps 3 --warmup 1 > $O/multi_4ranks.json 2> $O/multi_4ranks.err; echo "rc=$?"
echo "[prof] summarise"
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
for tag in ("all", "head"):
    st = glob.glob(O + "/%s/**/*kernel_stats.csv" % tag, recursive=True)
    if st:
        rows = [r for r in csv.DictReader(open(st[0]))]
        with open(O + "/kernel_stats_%s.csv" % tag, "w") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
        print("==", tag)
        for r in rows[:10]:
            print("  ", r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
    else:
        print("== kernel stats of", tag, ": missing")
for name in ("fetch", "write"):
    f = glob.glob(O + "/%s/**/*counter_collection.csv" % name, recursive=True)
    if not f:
        out[name] = "no counter file"; continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[name] = [{"kernel": k[0], "counter": k[1], "dispatches": v[0], "sum": v[1], "avg": v[1] / v[0]} for k, v in agg.items()]
    for e in out[name]:
        if "k1_res" in e["kernel"]:
            print(name, e["kernel"][:70], e["counter"], "dispatches", e["dispatches"], "avg", e["avg"])
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for f in ("bench_line", "multi_1rank", "multi_4ranks"):
    try:
        d = json.loads([l for l in open(O + "/%s.json" % f) if l.startswith("{")][0])
        print(f, "value %.3f G/s" % (d["value"] / 1e9), "n_gpus", d["n_gpus"], d["config"]["epoch_kernel"], "|", d["roofline"]["kernel"][:90],
              "| dsgd", (d.get("also", {}).get("dsgd_exact", {}) or {}).get("value"))
    except Exception as e:
        print(f, "NO LINE", repr(e))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +5M -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; (grep -i -E "malformed|abort|INVALID" $O/*.log | head) || true
