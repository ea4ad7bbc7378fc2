#!/bin/bash
# A/B of the static resident-side epoch on cfg 2: MFX_K1_LOOK=1 (k1_res_look, two ratings per slot and step) vs 0 (k1_res), three orders.
# usage: tools/ab_look.sh [tag]   -> gpurun_out/ab_look_<tag>.txt
tag=${1:-x}
out=gpurun_out/ab_look_$tag.txt; : > $out
for L in 1 0; do for O in shuffled user-major balanced; do
  MFX_K1_LOOK=$L timeout -k 10 200 python bench.py --no-cpu --no-also --steps 20 --warmup 5 --order $O --epoch static > gpurun_out/b_look${L}_$O.json 2>gpurun_out/b_look${L}_$O.err
  python - "$L" "$O" >> $out <<'PY'
import json, sys
L, O = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open("gpurun_out/b_look%s_%s.json" % (L, O)) if l.startswith("{")][0])
    print("look=%s %-10s %.3f G/s %.3f ms  levels %d  %s" % (L, O, d["value"] / 1e9, d["ms_per_step"], d["config"]["levels"], d["roofline"]["kernel"]))
except Exception as e:
    print("look=%s %s FAILED %r" % (L, O, e))
PY
done; done
cat $out
