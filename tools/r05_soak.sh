#!/bin/bash
# Round 5 soak of the final tree: the resident kernels' stress (many epochs against the level schedule), the top-N stress (the golden
# fixture's fused queries, 20 000 calls), the two fuzz files under six more seeds, the multi-process recovery / self-launch tests twice.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_soak; rm -rf $O; mkdir -p $O
{ echo "== tools/stress_res.py 2"; timeout -k 5 900 python3 tools/stress_res.py 2 2>&1 | tail -12
  echo "== tools/stress_topn.py 1000"; timeout -k 5 600 python3 tools/stress_topn.py 1000 2>&1 | tail -3
  for seed in 11 12 13 14 15 16; do echo "== fuzz seed $seed"; MFX_FUZZ_SEED=$seed timeout -k 5 600 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_funk_hot_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
  for i in 1 2; do echo "== multiproc $i"; timeout -k 5 600 python3 -m pytest tests/test_gpu_multiproc.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
} 2>&1 | tee $O/soak.log
