#!/bin/bash
# Round 4: evidence for the Funk per-feature pass (K2) ALONE -- the cfg 2 shuffled leg of bench.py (`--also-only k2_funk --no-cpu`).
# Pass 0: --kernel-trace --stats (per-kernel durations); then separate --pmc passes (never combined with other trace domains):
# FETCH_SIZE, WRITE_SIZE, and three SQ sets.  Per-kernel averages of every counter -> gpurun_out/k2_pmc/summary.json,
# the kernel-stats csv -> gpurun_out/k2_pmc/kernel_stats.csv.  tools/collect_prof_r04.py copies the summaries into profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/k2_pmc; rm -rf $O; mkdir -p $O
export MFX_GRAPH=0
ARGS="--steps 2 --warmup 1 --no-cpu --also-only k2_funk"
echo "[k2] pass 0: kernel trace + stats"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $ARGS > $O/bench_stats.json 2> $O/stats.log || echo "stats pass failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1)); echo "[k2] pmc pass $i: $set"
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 bench.py $ARGS > $O/bench_p$i.json 2> $O/p$i.log || echo "pmc pass $i failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json, os, shutil
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "::k2_" in kn:
            short = kn[kn.index("::k2_") + 2:].split("(")[0]          # e.g. k2_res_box<2>
            a = agg[short][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: {"dispatches": v[0], "avg": v[1] / v[0]} for c, v in cs.items()} for k, cs in agg.items()}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for k in sorted(out):
    print(k)
    for c in sorted(out[k]):
        print("   %-28s dispatches %6d  avg per dispatch %.6g" % (c, out[k][c]["dispatches"], out[k][c]["avg"]))
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, O + "/kernel_stats.csv")
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
