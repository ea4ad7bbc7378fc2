#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_p3; rm -rf $O; mkdir -p $O


timeout -k 5 300 python3 tools/als_convoy.py > $O/convoy.txt 2>&1 || echo "convoy failed"
cat $O/convoy.txt
