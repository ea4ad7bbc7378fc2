#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 5 900 python3 -m pytest tests/test_gpu_topn_golden.py tests/test_gpu_recommenders.py -x -q 2>&1 | tail -3
for kn in mfma tiled; do
  echo -n "$kn: "; MFX_TOPN_KERNEL=$kn timeout -k 5 600 python3 tools/topn_kernel_probe.py 2>&1 | tail -1
done
