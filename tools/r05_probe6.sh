#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_p6; rm -rf $O; mkdir -p $O
for k in 96 128; do timeout -k 5 300 python3 tools/als_stamps_w.py $k > $O/stamps_w_$k.txt 2>&1 || echo "stamps failed"; cat $O/stamps_w_$k.txt; done
