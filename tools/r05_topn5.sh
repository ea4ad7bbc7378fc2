#!/bin/bash
# Round 5: the top-N kernel with the per-user bar: parity tests, then the kernel's own time (kernel trace) and the query time.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_topn5; rm -rf $O; mkdir -p $O
timeout -k 5 900 python3 -m pytest tests/test_gpu_recommenders.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
for ni in 50000 200000; do
echo "$(NI=$ni timeout -k 5 300 python3 tools/topn_kernel_probe.py 2>&1 | tail -1)" | tee -a $O/query.txt
done
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
