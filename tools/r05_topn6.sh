#!/bin/bash
# Round 5: the product top-N kernel (compiler-managed loads): stress of the small golden queries, the top-N tests, the timing.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_topn6; rm -rf $O; mkdir -p $O
timeout -k 5 300 python3 tools/stress_topn.py ${REPS:-300} 2>&1 | tail -4 | tee $O/stress.txt
if grep -q "Memory access fault\|MISMATCH" $O/stress.txt; then echo "stress failed: stopping"; exit 1; fi
timeout -k 5 600 python3 -m pytest tests/test_gpu_recommenders.py tests/test_gpu_topn_golden.py tests/test_gpu_predict.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
for ni in 50000 200000; do echo "$(NI=$ni timeout -k 5 300 python3 tools/topn_kernel_probe.py 2>&1 | tail -1)" | tee -a $O/query.txt; done
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/topn_kernel_probe.py > $O/t.txt 2>&1
python3 - "$O/t" <<'PY' | tee -a $O/kernel.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f))
      if "k_topn_mfma" in r["Kernel_Name"] and int(r.get("Grid_Size_X") or r.get("Grid_Size")) >= 256 * 3000 ]
print("k_topn_mfma %s ms (100 K-user launches)" % " ".join("%.2f" % x for x in d))
PY
cp $(find $O/t -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/t
