#!/bin/bash
# k2_res_box with the mailbox poll issued one round ahead (MFX_K2_POLL_PIPE=1) against the product form, same box, alternating
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_k2; mkdir -p $O
for rep in 1 2 3; do for b in 0 1; do
  echo -n "poll_pipe $b: "; MFX_K2_POLL_PIPE=$b timeout -k 5 300 python3 tools/bench_families.py funk --k 16 --epochs 2 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms per pass, %.2f G/s' % (d['ms_per_pass'], d['scalar_updates_per_s']/1e9))"
  echo -n "   balanced $b: "; MFX_K2_POLL_PIPE=$b timeout -k 5 300 python3 tools/bench_families.py funk --k 16 --epochs 2 --balanced 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms per pass, %.2f G/s' % (d['ms_per_pass'], d['scalar_updates_per_s']/1e9))"
done; done | tee $O/poll_pipe.txt
MFX_K2_POLL_PIPE=1 timeout -k 5 600 python3 -m pytest tests/test_gpu_funk_res.py tests/test_gpu_funk_hot_fuzz.py -x -q 2>&1 | tail -3
