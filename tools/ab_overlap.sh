#!/bin/bash
# A/B of the guided step (config #3): guidance chain in line vs on a side stream (MDX_BENCH_OVERLAP=1), with stream priorities.
# usage (GPU box): bash tools/ab_overlap.sh  -> gpurun_out/ab_overlap.txt
OUT=gpurun_out/ab_overlap.txt; : > $OUT
run() { echo "== $1" >> $OUT; env $1 python bench.py --guided --headline-only --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $OUT; }
run "X=0"
run "MDX_BENCH_OVERLAP=1"
run "MDX_BENCH_OVERLAP=1 MDX_SIDE_PRIORITY=-1"
run "X=1"
cat $OUT
