#!/bin/bash
# tuning sweep of the full-k kernel (run on the GPU box): ratings per wave
for b in 16 8 4; do
  echo "== batch $b"
  MFX_K1_BATCH=$b python bench.py --steps 10 --warmup 2 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.3fG  ms/step %.3f  avg_launch_us %.2f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_us']))"
done
