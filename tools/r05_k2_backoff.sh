#!/bin/bash
# k2_res_box poll back-off sweep (MFX_K2_BACKOFF = cap of the per-lane exponential back-off): ms per pass, then HBM-side traffic
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_k2; rm -rf $O; mkdir -p $O
for rep in 1 2; do for b in 0 1 2 3 4; do
  echo -n "backoff $b: "; MFX_K2_BACKOFF=$b timeout -k 5 300 python3 tools/bench_families.py funk --k 16 --epochs 2 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms per pass, %.2f G/s' % (d['ms_per_pass'], d['scalar_updates_per_s']/1e9))"
done; done | tee $O/sweep.txt
for b in 0 2 3; do for c in FETCH_SIZE WRITE_SIZE; do
  MFX_K2_BACKOFF=$b timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${b}_$c -- python3 tools/bench_families.py funk --k 16 --epochs 2 > /dev/null 2> $O/pmc_${b}_$c.log || echo "pmc failed"
done; done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for b in (0, 2, 3):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(O + "/pmc_%d_%s/**/*counter_collection.csv" % (b, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k2_res_box" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        tot[c] = (sum(vals) / max(len(vals), 1), len(vals))
    # FETCH_SIZE / WRITE_SIZE are in KB per dispatch; a pass = 10 M scalar updates
    print("backoff %d: FETCH_SIZE %.0f KB (x%d dispatches), WRITE_SIZE %.0f KB per pass -> %.1f B fetched (uncorrected), %.1f B written per scalar update"
          % (b, tot["FETCH_SIZE"][0], tot["FETCH_SIZE"][1], tot["WRITE_SIZE"][0], tot["FETCH_SIZE"][0] * 1024 / 1e7, tot["WRITE_SIZE"][0] * 1024 / 1e7))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
