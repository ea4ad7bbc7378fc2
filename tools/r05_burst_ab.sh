#!/bin/bash
# same-box A/B: the round-4 burst walk (libmfx_r04burst.so, built from the stashed tree) against the tree's
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_burst; mkdir -p $O
for rep in 1 2; do
for lib in libmfx_r04burst.so libmfx.so; do
  echo "== $lib"; ONLY_AUTO=1 MFX_LIB=$GRAFT_REPO_ROOT/mfrec_amd/$lib timeout -k 5 300 python3 tools/bench_ml10m_shape.py 2>&1 | grep -E '"epoch"' 
done; done | tee $O/ab.txt
