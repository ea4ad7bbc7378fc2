#!/bin/bash
# Round 5: what bounds k_topn_mfma at 100 K users x 50 K items?  (1) the candidate blocks split over blockIdx.y (3 125 workgroups on 512
# slots = 6.1 rounds -> 7: the split shortens the last round); (2) HBM-side bytes (every workgroup streams the whole item matrix for its 32
# users: 80 GB if nothing hit in L2); (3) matrix-pipe / vector / wait shares.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_topn4; rm -rf $O; mkdir -p $O
for sp in 1 2 3 4 8; do
  echo "splits $sp: $(MFX_TOPN_SPLITS=$sp timeout -k 5 300 python3 tools/topn_kernel_probe.py 2>&1 | tail -1)" | tee -a $O/splits.txt
done
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc$i -- python3 tools/topn_kernel_probe.py > $O/pmc$i.txt 2> $O/pmc$i.log || echo "pmc pass $i ($c) failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_topn_mfma" not in r["Kernel_Name"]: continue
        if int(r["Grid_Size"]) < 256 * 3000: continue            # the 100 K-user launches only
        out[r["Counter_Name"]].setdefault("v", []).append(float(r["Counter_Value"]))
res = {c: {"launches": len(v["v"]), "avg": sum(v["v"]) / len(v["v"])} for c, v in out.items()}
json.dump(res, open(O + "/pmc_summary.json", "w"), indent=1)
for c in sorted(res): print("  %-30s %d launches, avg %.6g" % (c, res[c]["launches"], res[c]["avg"]))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
