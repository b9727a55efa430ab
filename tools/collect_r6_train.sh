#!/bin/bash
# usage (GPU box, via gpurun, from the repo root): bash tools/collect_r6_train.sh  -> gpurun_out/r6_train_* (the training half of tools/collect_r6.sh
# + tools/r6_trace_last_step.sh, for a HEAD whose sampling kernels did not change since the last full collection)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/prof_tr
  sfx=""; [ $v = 0 ] && sfx="_unfused"
  MDX_TRAIN_FUSED=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 10 > $OUT/r6_train_bench_fp16_under_rocprof$sfx.json 2> /dev/null
  find /tmp/prof_tr -name "*kernel_stats.csv" -exec cp {} $OUT/r6_train_fp16_kernel_stats$sfx.csv \;
done
for rep in 1 2; do
  python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 > $OUT/r6_train_bench_fp16_$rep.json 2> /dev/null
  MDX_TRAIN_FAST=0 python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 > $OUT/r6_train_bench_fp16_pybodies_$rep.json 2> /dev/null
done
python $ROOT/bench.py --train --precision f32 --no-cpu-baseline --steps 8 --warmup 3 > $OUT/r6_train_bench_f32.json 2> /dev/null
python $ROOT/tools/profile_train_host.py > $OUT/r6_train_host_profile_fast.txt 2>&1
bash $ROOT/tools/r6_trace_last_step.sh > /dev/null 2>&1
cd $ROOT
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_train_bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['ms_per_step'], 2), round(d['host_issue_ms_per_step'], 2), round(d['host_only_ms_per_step'], 2), d.get('loss_fixed_probe_before_after'))
    except Exception as e:
        print(f, e)
P
wc -l gpurun_out/r6_train_trace_last_step.txt
