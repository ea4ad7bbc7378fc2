#!/bin/bash
# K2 persistent-pass experiments (tools/k2_xcd.hip), all variants in one GPU call
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/k2_xcd; mkdir -p $O; : > $O/log2.txt
run() { echo "== $*" >> $O/log2.txt; timeout -k 5 60 tools/_k2_xcd_w8_p1 10000000 "$@" >> $O/log2.txt 2>&1 || echo "rc=$?" >> $O/log2.txt; }
for g in 64 128 256 512 1024; do run 4 $g 0; done
run 4 128 1; run 4 256 1
for g in 512 1024 2048; do run 5 $g 0; run 5 $g 1; done
grep -E "^==|K2 pass|rc=" $O/log2.txt
