#!/bin/bash
# register-diet experiment: 14 slots x 4 096 waves at four waves per SIMD (MFX_K1_NR=7) against 18 x 3 072
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_nr7; mkdir -p $O
for rep in 1 2; do for nr in 9 7; do
  echo "== MFX_K1_NR=$nr"
  MFX_K1_NR=$nr timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel'], d['roofline'].get('frac_of_chain_floor'))"
  MFX_K1_NR=$nr timeout -k 5 300 python3 bench.py --no-also --no-cpu --steps 20 --warmup 5 --order balanced | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('balanced', d['value']/1e9, d['ms_per_step'], d['roofline']['kernel'])"
done; done 2>&1 | tee $O/ab.txt
