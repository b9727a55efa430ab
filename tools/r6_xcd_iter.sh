#!/bin/bash
# usage (GPU box, via gpurun, from the repo root): bash tools/r6_xcd_iter.sh  -> gpurun_out/r6x_*
# A/B of the XCD-aware block dealing of the grouped weight-gradient kernel (csrc/mdx_train.hip wgrad_grouped_kernel, MDX_WGRAD_XCD):
# the training tests, then the fp16 step with the dealing off / on at two rows-per-block settings, three repetitions each (shared host
# cores: the minimum counts), then kernel statistics of both settings.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_fused_train.py tests/test_gpu_trainer.py -x -q -m gpu > $OUT/r6x_tests.txt 2>&1
tail -3 $OUT/r6x_tests.txt
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for xcd in 0 1; do
    for rows in ${SWEEP:-2048 4096}; do
      MDX_WGRAD_XCD=$xcd MDX_WGRAD_ROWS=$rows python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 2> /dev/null | tail -1 > $OUT/r6x_train_xcd${xcd}_rows${rows}_$rep.json
    done
  done
done
for xcd in 0 1; do
  rm -rf /tmp/prof_tr
  MDX_WGRAD_XCD=$xcd rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 10 > /dev/null 2>&1
  find /tmp/prof_tr -name "*kernel_stats.csv" -exec cp {} $OUT/r6x_train_kernel_stats_xcd$xcd.csv \;
done
cd $ROOT
python - <<'P'
import json, glob, csv
for f in sorted(glob.glob('gpurun_out/r6x_train_xcd*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'ms', round(d['ms_per_step'], 2), 'host', round(d['host_issue_ms_per_step'], 2), 'loss', d['loss_first_last'])
    except Exception as e:
        print(f, 'unparsed', e)
for x in (0, 1):
    try:
        rows = list(csv.DictReader(open(f'gpurun_out/r6x_train_kernel_stats_xcd{x}.csv')))
        w = [(r['Name'][:60], int(r['Calls']), float(r['AverageNs']) / 1e3) for r in rows if 'wgrad_grouped' in r['Name']]
        print('xcd', x, 'total ms/step', round(sum(float(r['TotalDurationNs']) for r in rows) / 16e6, 3), w)
    except Exception as e:
        print('stats', x, e)
P
