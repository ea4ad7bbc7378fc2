#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_burst; mkdir -p $O
timeout -k 5 300 python3 tools/bench_ml10m_shape.py > $O/ml10m.txt 2>&1 || echo "ml10m failed"
cat $O/ml10m.txt
timeout -k 5 600 python3 -m pytest tests/test_gpu_resident.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5
