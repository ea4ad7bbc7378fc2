#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_burst_final; rm -rf $O; mkdir -p $O
timeout -k 5 900 python3 -m pytest tests/test_gpu_resident.py tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
echo "ml: $(timeout -k 5 300 python3 tools/bench_ml10m_shape.py 2>&1 | grep -E 'bit_for_bit|"epoch"' | tr '\n' ' ')" | tee -a $O/legs.txt
timeout -k 5 200 python3 tools/bench_skew_knob.py ml 2>&1 | tail -1 | tee -a $O/legs.txt
timeout -k 5 200 python3 tools/bench_skew_knob.py zipf 2>&1 | tail -1 | tee -a $O/legs.txt
timeout -k 5 600 python3 tools/stress_res.py 1 2>&1 | tail -6 | tee -a $O/stress.txt
