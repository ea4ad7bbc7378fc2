#!/bin/bash
# Round 5, first GPU session: where does a short ALS row's time go?  SIMD placement of the waves, the price of a
# dependent vector instruction beside f64 MFMA streams, the phase stamps of k_als_rows3, instruction-cache counters.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_p1; rm -rf $O; mkdir -p $O
echo "[p1] simd_probe"; timeout -k 5 120 build/tools/simd_probe > $O/simd_probe.txt 2>&1 || echo "simd_probe failed"
echo "[p1] baseline cfg4/10"; timeout -k 5 300 python3 tools/bench_families.py als --users 100000 --items 20000 --nnz 10000000 --k 128 --epochs 3 > $O/als_base.json 2> $O/als_base.log || echo "baseline failed"
echo "[p1] stamps"; timeout -k 5 300 python3 tools/als_stamps.py 128 swap > $O/als_stamps_short.txt 2>&1 || echo "stamps failed"
timeout -k 5 300 python3 tools/als_stamps.py 128 > $O/als_stamps_long.txt 2>&1 || echo "stamps failed"
echo "[p1] counters"; rocprofv3 -L > $O/counters_all.txt 2>&1 || true
grep -o -i -E "\b(SQC_[A-Z_0-9]*ICACHE[A-Z_0-9]*|SQ_[A-Z_0-9]*IFETCH[A-Z_0-9]*|SQC_[A-Z_0-9]*INST[A-Z_0-9]*)\b" $O/counters_all.txt | sort -u > $O/counters_icache.txt
cat $O/counters_icache.txt
CMD="python3 tools/bench_families.py als --users 100000 --items 20000 --nnz 10000000 --k 128 --epochs 1"
i=0
for c in $(cat $O/counters_icache.txt | head -12); do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc$i -- $CMD > $O/pmc$i.json 2> $O/pmc$i.log || echo "pmc $c failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "::k_als_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        kn = r["Kernel_Name"]; short = kn[kn.index("::k_") + 2:].split("(")[0]
        out[short].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
json.dump(out, open(O + "/icache_summary.json", "w"), indent=1)
for k in sorted(out):
    print(k)
    for c in sorted(out[k]):
        print("   %-34s %s" % (c, "  ".join("%.5g" % v for v in out[k][c])))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
