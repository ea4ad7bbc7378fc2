#!/bin/bash
# ALS row-kernel variants: k_als_rows3 (register tiles; occupancy 3 / 2) against k_als_rows2
set -o pipefail
O=gpurun_out/sweep_als; mkdir -p $O; : > $O/log.txt
for k in ${KS:-80 96 112 128}; do
  for v in "MFX_ALS_ROWS2=1" "MFX_ALS_OCC=2" "MFX_ALS_OCC=3"; do
    echo "== k=$k $v" >> $O/log.txt
    env $v timeout -k 10 120 python3 tools/bench_families.py als --users 100000 --items 20000 --nnz 10000000 --k $k --epochs 2 >> $O/log.txt 2>&1 || exit 1
  done
done
grep -o "== .*\|device_s_per_epoch\": [0-9.]*" $O/log.txt | paste - -
