#!/bin/bash
# Round 5, second GPU session: issue throughput of the DPP f64 forms; same-process A/B of the ALS row kernel with
# issue priorities by hardware wave slot (MFX_ALS_TUNE=1) and at four workgroups per CU (MFX_ALS_OCC=4, spills today).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_p2; rm -rf $O; mkdir -p $O
timeout -k 5 120 build/tools/simd_probe > $O/simd_probe.txt 2>&1 || echo "simd_probe failed"
grep -E "placement|throughput" $O/simd_probe.txt
timeout -k 5 60 build/tools/hop_latency $O/hop_latency.json > $O/hop_latency.txt 2>&1 || echo "hop failed"
tail -2 $O/hop_latency.txt
timeout -k 5 600 python3 tools/als_ab.py 128 3:0 3:1 3:3 > $O/als_ab.txt 2>&1 || echo "ab failed"
cat $O/als_ab.txt
