#!/bin/bash
# k = 48 / 64: k_als_rows2 against k_als_rows3 at occupancy hints 3..6
set -o pipefail
O=gpurun_out/sweep_als_small; mkdir -p $O; : > $O/log.txt
for k in 48 64; do
  for v in "MFX_ALS_ROWS2=1" "MFX_ALS_ROWS2=0 MFX_ALS_OCC=3" "MFX_ALS_ROWS2=0 MFX_ALS_OCC=4" "MFX_ALS_ROWS2=0 MFX_ALS_OCC=5" "MFX_ALS_ROWS2=0 MFX_ALS_OCC=6"; do
    echo "== k=$k $v" >> $O/log.txt
    env $v timeout -k 10 120 python3 tools/bench_families.py als --users 100000 --items 20000 --nnz 10000000 --k $k --epochs 2 >> $O/log.txt 2>&1 || exit 1
  done
done
grep -o "== .*\|device_s_per_epoch\": [0-9.]*" $O/log.txt | paste - -
