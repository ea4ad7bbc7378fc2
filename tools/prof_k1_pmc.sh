#!/bin/bash
# Where does the resident-side epoch (k1_res) spend its cycles?  Separate --pmc passes over the headline run (cfg 2, static) and
# over cfg 3 whole in user-major order (ticket); averages per dispatch -> gpurun_out/k1_pmc/summary_*.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/k1_pmc; rm -rf $O; mkdir -p $O
run() {   # name, bench args...
  local name=$1; shift
  local i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    i=$((i+1)); echo "[pmc] $name pass $i: $set"
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${name}_p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-also "$@" > $O/${name}_p$i.json 2> $O/${name}_p$i.log || echo "pass $i failed"
  done
  python3 - "$O" "$name" <<'PY'
import csv, glob, sys, collections
O, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(O + "/%s_p*/**/*counter_collection.csv" % name, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k1_res" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(O + "/summary_%s.txt" % name, "w") as out:
    for k in sorted(agg):
        line = "%-28s dispatches %d  avg per dispatch %.5g" % (k, agg[k][0], agg[k][1] / agg[k][0])
        print(line); out.write(line + "\n")
PY
}
run cfg2_static
run cfg3_ticket --users 1000000 --items 200000 --nnz 100000000 --order user-major
find $O -name "*counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace.csv" -delete
du -sh $O
