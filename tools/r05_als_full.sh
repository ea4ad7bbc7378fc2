#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in libmfx_r04als.so libmfx.so libmfx_r04als.so libmfx.so; do
  echo -n "$lib: "; MFX_LIB=$GRAFT_REPO_ROOT/mfrec_amd/$lib timeout -k 5 500 python3 tools/bench_families.py als --users 1000000 --items 200000 --nnz 100000000 --k 128 --epochs 2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['device_s_per_epoch'], d['mfma_f64_peak_frac'])"
done
