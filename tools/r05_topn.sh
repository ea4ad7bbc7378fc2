#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_topn; mkdir -p $O
timeout -k 5 900 python3 -m pytest tests/test_gpu_topn_golden.py tests/test_gpu_recommenders.py -x -q 2>&1 | tail -8
echo "== mfma"; timeout -k 5 600 python3 tools/topn_timing.py 2>&1 | tail -3 | cut -c1-400
echo "== tiled"; MFX_TOPN_KERNEL=tiled timeout -k 5 600 python3 tools/topn_timing.py 2>&1 | tail -3 | cut -c1-400
