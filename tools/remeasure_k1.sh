#!/bin/bash
# the K1 lines quoted in DESIGN.md section 4 "Measured": default, orders, shard shapes, storage
set -o pipefail
O=gpurun_out/remeasure; mkdir -p $O
b() { name=$1; shift; timeout -k 10 600 python bench.py --no-cpu "$@" > $O/$name.json || exit 1; python3 -c "import sys,json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],3), 'ms', d['config']['levels'], 'levels', round(d['roofline']['avg_launch_us'],2), 'us/launch', d['roofline']['kernel'])"; }
b default
b user_major --order user-major
b balanced --order balanced
b f32 --storage f32
b cfg3_shard --users 125000 --items 200000 --nnz 12500000
b cfg5_shard --users 1250000 --items 1000000 --nnz 125000000 --steps 5 --warmup 1
b cfg3_full --users 1000000 --items 200000 --nnz 100000000 --steps 5 --warmup 1
b k128 --k 128
