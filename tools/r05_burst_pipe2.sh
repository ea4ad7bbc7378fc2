#!/bin/bash
# Round 5: the archived pipelined burst walk (profiles/experiments/r05_k1_burst_pipelined.hip.txt, built into libmfx_diag.so with
# -DMFX_K1_BURST_PIPE) once more, now that only hot slots burst: both power-law legs, alternating with the product on one box.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_burst_pipe2; rm -rf $O; mkdir -p $O
echo "check pipe: $(MFX_LIB=$PWD/mfrec_amd/libmfx_diag.so timeout -k 5 300 python3 tools/bench_ml10m_shape.py 2>&1 | grep 'bit_for_bit' | tail -1)" | tee -a $O/ab.txt
for rep in 1 2; do for v in product pipe; do
  if [ $v = pipe ]; then export MFX_LIB=$PWD/mfrec_amd/libmfx_diag.so; else unset MFX_LIB; fi
  echo "$v $(timeout -k 5 200 python3 tools/bench_skew_knob.py ml 2>&1 | tail -1)" | tee -a $O/ab.txt
  echo "$v $(timeout -k 5 200 python3 tools/bench_skew_knob.py zipf 2>&1 | tail -1)" | tee -a $O/ab.txt
done; done
export MFX_LIB=$PWD/mfrec_amd/libmfx_diag.so
for run in 300 2000 4000; do echo "pipe run $run $(MFX_K1_BURST_RUN=$run timeout -k 5 200 python3 tools/bench_skew_knob.py ml 2>&1 | tail -1)" | tee -a $O/ab.txt; done
for run in 1000 8000; do echo "pipe run $run $(MFX_K1_BURST_RUN=$run timeout -k 5 200 python3 tools/bench_skew_knob.py zipf 2>&1 | tail -1)" | tee -a $O/ab.txt; done
