#!/bin/bash
# Round 5: MFX_K1_BURST_RUN = ratings a slot must have ahead for a burst to be taken, on the two power-law legs of the bench.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_burst_knobs; mkdir -p $O
for run in ${RUNS:-3 500 750 1000 1500}; do
  echo "burst_run $run: $(MFX_K1_BURST_RUN=$run timeout -k 5 200 python3 tools/bench_skew_knob.py ml 2>&1 | tail -1)" | tee -a $O/knobs2.txt
done
for run in ${ZRUNS:-3 200 1000 4000 16000}; do
  echo "burst_run $run: $(MFX_K1_BURST_RUN=$run timeout -k 5 200 python3 tools/bench_skew_knob.py zipf 2>&1 | tail -1)" | tee -a $O/knobs2.txt
done
