#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_topn; rm -rf $O; mkdir -p $O
for kn in mfma tiled; do
  MFX_TOPN_KERNEL=$kn timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$kn -- python3 tools/topn_kernel_probe.py > $O/$kn.txt 2>&1 || echo "failed $kn"
  tail -1 $O/$kn.txt
  f=$(find $O/$kn -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200; cp "$f" $O/${kn}_kernel_stats.csv
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
