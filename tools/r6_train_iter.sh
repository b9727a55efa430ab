#!/bin/bash
# usage (GPU box, via gpurun, from the repo root): bash tools/r6_train_iter.sh [tag]  -> gpurun_out/r6t_<tag>_*
# One iteration of the round-6 training-step work: the training tests, the fp16 step with the queued weight gradients at several
# rows-per-block settings (and with the queue off), then kernel statistics + an ordered kernel trace of the default setting.
set -u
TAG=${1:-a}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fused_train.py tests/test_gpu_train_ops.py tests/test_loss.py -x -q -m gpu > $OUT/r6t_${TAG}_tests.txt 2>&1
tail -3 $OUT/r6t_${TAG}_tests.txt
cd /tmp && export TMPDIR=/tmp
# the box's host cores are shared: every setting is measured REPS times, the summary prints each and the minimum counts
for rep in $(seq 1 ${REPS:-3}); do
for rows in ${SWEEP:-0 2048}; do
  if [ $rows = 0 ]; then
    MDX_WGRAD_GROUPED=0 python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 > $OUT/r6t_${TAG}_train_off_$rep.json 2> /dev/null
  else
    MDX_WGRAD_ROWS=$rows python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 > $OUT/r6t_${TAG}_train_rows${rows}_$rep.json 2> /dev/null
  fi
done
MDX_TRAIN_FAST=0 python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 > $OUT/r6t_${TAG}_train_pybodies_$rep.json 2> /dev/null
done
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 10 > $OUT/r6t_${TAG}_train_under_rocprof.json 2> /dev/null
find /tmp/prof_tr -name "*kernel_stats.csv" -exec cp {} $OUT/r6t_${TAG}_train_kernel_stats.csv \;
# ordered trace of the LAST step only (name, start, duration), compact
python - <<'P' > $OUT/r6t_${TAG}_train_trace_last_step.txt 2>&1
import csv, glob, re
f = glob.glob('/tmp/prof_tr/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one step = between two adamw_amp_kernel launches
idx = [i for i, r in enumerate(rows) if 'adamw_amp_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::|at::native::|void ', '', r['Kernel_Name'])[:70]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = e
P
cd $ROOT
for f in $OUT/r6t_${TAG}_train_*.json; do python - "$f" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'ms', round(d['ms_per_step'], 2), 'host', round(d['host_issue_ms_per_step'], 2), 'loss', d['loss_first_last'], 'GB', round(d['peak_hbm_gb'], 1))
except Exception as e:
    print(sys.argv[1], 'unparsed', e)
P
done
