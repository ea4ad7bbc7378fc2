#!/bin/bash
# knock-out diagnostics of k_als_rows3 (diag build -DMFX_ALS_KO): what each phase costs in throughput under load
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_p4; rm -rf $O; mkdir -p $O
MFX_LIB=$GRAFT_REPO_ROOT/mfrec_amd/libmfx_diag.so NOCHECK=1 ROUNDS=2 timeout -k 5 800 python3 tools/als_ab.py 128 3:0 3:4 3:8 3:16 3:32 3:64 3:128 3:256 3:512 3:24 3:28 3:540 3:96 3:416 3:1020 > $O/ko.txt 2>&1 || echo "ko failed"
cat $O/ko.txt
