#!/bin/bash
# Round 5: knock-outs of k_topn_mfma (diagnostic library built with MFX_EXTRA_FLAGS=-DMFX_TOPN_KO): 1 = no selection, 2 = no matrix
# instructions (hashed scores), 4 = no B loads, 8 = no rated-item masks.  Results of the knocked-out runs are garbage (the sort path
# then redoes the users: the QUERY time means nothing); the kernel's own duration from the kernel trace is the point.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_topn_ko; rm -rf $O; mkdir -p $O
export MFX_LIB=$PWD/mfrec_amd/libmfx_diag.so
for ko in ${KOS:-0 1 2 4 8 6 7 9 15}; do
  MFX_TOPN_KO=$ko timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$ko -- python3 tools/topn_kernel_probe.py > $O/t$ko.txt 2>&1 || echo "ko $ko failed"
  python3 - "$O/t$ko" "$ko" <<'PY' | tee -a $O/ko.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f))
      if "k_topn_mfma" in r["Kernel_Name"] and int(r.get("Grid_Size_X") or r.get("Grid_Size")) >= 256 * 3000 ]
print("ko %2s: k_topn_mfma %s ms (100 K-user launches)" % (sys.argv[2], " ".join("%.2f" % x for x in d)))
PY
  rm -rf $O/t$ko
done
