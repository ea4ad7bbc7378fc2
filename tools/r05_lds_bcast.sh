#!/bin/bash
# Round 5: the step's cross-lane broadcasts (travelling ids, gradients) through LDS (-DMFX_K1_LDS_BCAST=1, libmfx_diag.so) against the
# v_readlane + select chains of the product, alternating on one box; bit-exactness of the variant first.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_lds_bcast; rm -rf $O; mkdir -p $O
MFX_LIB=$PWD/mfrec_amd/libmfx_diag.so timeout -k 5 600 python3 -m pytest tests/test_gpu_resident.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '%.3f G/s  %.3f ms' % (d['value']/1e9, d['ms_per_step']), d['roofline']['kernel'][:50])"; }
for rep in 1 2 3; do for v in readlane lds; do
  if [ $v = lds ]; then export MFX_LIB=$PWD/mfrec_amd/libmfx_diag.so; else unset MFX_LIB; fi
  timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 | one "$v shuffled"
  timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 --order balanced | one "$v balanced"
done; done 2>&1 | tee $O/ab.txt
