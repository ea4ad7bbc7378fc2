#!/bin/bash
# Round 5: k1_level's ordered sum with the next eight products' LDS reads issued before the current eight are added (the resident epoch's
# form), against the loop that waits for LDS in every iteration (libmfx_base.so = the tree before), alternating on one box.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_level_sum; rm -rf $O; mkdir -p $O
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '%.3f G/s  %.3f ms' % (d['value']/1e9, d['ms_per_step']), d['roofline']['kernel'][:50])"; }
for rep in 1 2 3; do for v in base new; do
  if [ $v = base ]; then export MFX_LIB=$PWD/mfrec_amd/libmfx_base.so; else unset MFX_LIB; fi
  timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 10 --warmup 3 --epoch levels | one "$v levels shuffled"
done; done 2>&1 | tee $O/ab.txt
unset MFX_LIB
timeout -k 5 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
