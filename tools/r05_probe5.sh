#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_p5; rm -rf $O; mkdir -p $O
ROUNDS=2 timeout -k 5 600 python3 tools/als_ab.py ${K:-128} 3:0 3:0:n2 > $O/ab.txt 2>&1 || echo "ab failed"
cat $O/ab.txt | tail -20
