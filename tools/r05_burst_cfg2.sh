#!/bin/bash
# Round 5: the burst variant FORCED on the uniform headline workload (cfg 2 shuffled / balanced), by the run threshold: does a short item
# edge (consecutive ratings of a slot back to back) buy anything where both kinds of edge matter equally?
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_burst_cfg2; rm -rf $O; mkdir -p $O
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '%.3f G/s  %.3f ms' % (d['value']/1e9, d['ms_per_step']), d['roofline']['kernel'][:64])"; }
timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 | one "product shuffled" | tee -a $O/ab.txt
for run in 3 10 30 100 200 1000; do
  MFX_K1_BURST=1 MFX_K1_BURST_RUN=$run timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 | one "burst run $run shuffled" | tee -a $O/ab.txt
done
timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 --order balanced | one "product balanced" | tee -a $O/ab.txt
for run in 3 30 200; do
  MFX_K1_BURST=1 MFX_K1_BURST_RUN=$run timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 --order balanced | one "burst run $run balanced" | tee -a $O/ab.txt
done
