#!/bin/bash
# usage (GPU box, repo root): bash tools/batch_sweep_r4.sh > gpurun_out/r5_batch_sweep.txt
# ms per step / molecules per second against the per-GPU batch, configs #2 and #3, exact fp32 path and split float16 path side by side
echo "# python bench.py --headline-only --no-cpu-baseline --batch B --steps 100 (20 for B >= 1024) --warmup 5 [--guided]; MOLDIFF_MATRIX_PATH=split_f16 for the split columns; one MI355X"
echo "# B  config  exact: ms/step molecules/s edge_a frac(fp32 peak) | split: ms/step molecules/s edge_a x fp32 peak | speed-up"
for B in 64 256 1024 2048; do
  steps=100; [ $B -ge 1024 ] && steps=20
  for cfg in simple guided; do
    flag=""; [ $cfg = guided ] && flag="--guided"
    A=$(python bench.py --headline-only --no-cpu-baseline --batch $B --steps $steps --warmup 5 $flag 2>/dev/null | tail -1)
    S=$(MOLDIFF_MATRIX_PATH=split_f16 python bench.py --headline-only --no-cpu-baseline --batch $B --steps $steps --warmup 5 $flag 2>/dev/null | tail -1)
    python - "$A" "$S" <<PY
import json,sys
a,s=json.loads(sys.argv[1]),json.loads(sys.argv[2])
print($B, '$cfg', round(a['ms_per_step'],3), round(a['value'],2), round(a['roofline']['frac'],3), '|', round(s['ms_per_step'],3), round(s['value'],2), round(s['roofline']['frac_of_fp32_mfma_peak'],3), '|', round(a['ms_per_step']/s['ms_per_step'],2))
PY
  done
done
