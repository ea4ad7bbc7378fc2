#!/bin/bash
# Round 4: cfg 4 at FULL size (1 M x 200 K, 100 M ratings, k = 128), one ALS epoch, the two half-passes of k_als_rows3 SEPARATELY
# (per dispatch, in dispatch order: the user half-pass first -- rows of degree ~100 gathering ITEM rows, a 205 MB table --, then the
# item half-pass -- rows of degree ~500 gathering USER rows, a 1 GB table).  Pass 0: kernel trace + stats; then separate --pmc passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/als_pmc_r04; rm -rf $O; mkdir -p $O
CMD="python3 tools/bench_families.py als --users 1000000 --items 200000 --nnz 100000000 --k ${K:-128} --epochs 1"
echo "[als] pass 0: kernel trace + stats"
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.json 2> $O/stats.log || echo "stats pass failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
  i=$((i+1)); echo "[als] pmc pass $i: $set"
  timeout -k 5 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- $CMD > $O/p$i.json 2> $O/p$i.log || echo "pmc pass $i failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json, shutil
O = sys.argv[1]
out = collections.defaultdict(dict)     # kernel -> counter -> [values in dispatch order]
for f in sorted(glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "::k_als_" in r["Kernel_Name"] or "::k_gram" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        kn = r["Kernel_Name"]; short = kn[kn.index("::k_") + 2:].split("(")[0]
        out[short].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for k in sorted(out):
    print(k)
    for c in sorted(out[k]):
        print("   %-30s %s" % (c, "  ".join("%.5g" % v for v in out[k][c])))
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, O + "/kernel_stats.csv")
# per-dispatch durations of the row kernels from the trace of pass 0
for f in glob.glob(O + "/stats/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "::k_als_" in r["Kernel_Name"] or "::k_gram" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(O + "/dispatch_times.txt", "w") as o:
        for r in rows:
            kn = r["Kernel_Name"]; short = kn[kn.index("::k_") + 2:].split("(")[0]
            line = "%-28s %.3f ms  grid %s" % (short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "?"))
            print(line); o.write(line + "\n")
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
