#!/bin/bash
# usage (GPU box, repo root): bash tools/batch_sweep.sh > gpurun_out/r3_batch_sweep.txt
# ms per step / molecules per second / edge kernel A's fraction of the fp32 MFMA peak against the per-GPU batch, configs #2 and #3
echo "# python bench.py --headline-only --no-cpu-baseline --batch B --steps 100 (20 for B >= 1024) --warmup 5 [--guided], one MI355X"
echo "# B  config  ms/step  molecules/s  edge_a2 fraction of the fp32 MFMA peak"
for B in 8 32 64 128 256 512 1024 2048; do
  steps=100; [ $B -ge 1024 ] && steps=20
  for cfg in simple guided; do
    flag=""; [ $cfg = guided ] && flag="--guided"
    python bench.py --headline-only --no-cpu-baseline --batch $B --steps $steps --warmup 5 $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print($B, '$cfg', round(d['ms_per_step'],3), round(d['value'],2), round(d['roofline']['frac'],3))"
  done
done
