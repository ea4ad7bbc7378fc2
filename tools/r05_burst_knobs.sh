#!/bin/bash
# Round 5: MFX_K1_BURST_HEAD: 1 = a burst's first window is the slot's head rating out of the caller's registers (no round trip before the
# rows); 0 = round 4's prologue (eight records, then their granules).  Both power-law legs, alternating; bit-exactness first.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_burst_knobs; mkdir -p $O
echo "check head=1: $(MFX_K1_BURST_HEAD=1 timeout -k 5 300 python3 tools/bench_ml10m_shape.py 2>&1 | grep 'bit_for_bit' | tail -1)" | tee -a $O/knobs3.txt
for h in 0 1 0 1; do
  echo "head $h: $(MFX_K1_BURST_HEAD=$h timeout -k 5 200 python3 tools/bench_skew_knob.py ml 2>&1 | tail -1)" | tee -a $O/knobs3.txt
done
for h in 0 1; do
  echo "head $h: $(MFX_K1_BURST_HEAD=$h timeout -k 5 200 python3 tools/bench_skew_knob.py zipf 2>&1 | tail -1)" | tee -a $O/knobs3.txt
done
for run in 300 500 937 1500 2500; do
  echo "head 1 run $run: $(MFX_K1_BURST_HEAD=1 MFX_K1_BURST_RUN=$run timeout -k 5 200 python3 tools/bench_skew_knob.py ml 2>&1 | tail -1)" | tee -a $O/knobs3.txt
done
