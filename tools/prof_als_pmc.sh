#!/bin/bash
# Where does k_als_rows3 spend its cycles?  Several separate --pmc passes over the ALS leg (cfg 4 / 10, k=128).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/als_pmc; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); echo "[pmc] pass $i: $set"
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 tools/bench_families.py als --users 100000 --items 20000 --nnz 10000000 --k ${K:-128} --epochs 1 > $O/p$i.json 2> $O/p$i.log || echo "pass $i failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "als_rows" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(O + "/summary.txt", "w") as out:
    for k in sorted(agg):
        line = "%-32s dispatches %d  avg per dispatch %.4g" % (k, agg[k][0], agg[k][1] / agg[k][0])
        print(line); out.write(line + "\n")
PY
find $O -name "*counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace.csv" -delete
