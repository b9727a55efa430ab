#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh r3   -> gpurun_out/<tag>_* (copy what is to be judged into profiles/)
# One pass per evidence kind, never a PMC pass together with a trace domain:
#   <tag>_bench.json                      python bench.py (the driver's command)
#   <tag>_kernel_stats.csv                rocprofv3 --kernel-trace --stats over bench.py --steps 95 --headline-only      (config #2)
#   <tag>_bench_under_rocprof.json        the JSON line of that same run (its hipEvent durations must agree with the csv)
#   <tag>_guided_kernel_stats.csv         the same with --guided                                                          (config #3)
#   <tag>_pmc_summary.json / _guided_     tools/pmc_summary.py: three separate --pmc passes each
#   <tag>_train_*                         bench.py --train in fp32 and fp16 + kernel stats of the fp16 run
set -u
TAG=${1:-r3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
for cfg in simple guided; do
  flag=""; name=${TAG}_kernel_stats; jn=${TAG}_bench_under_rocprof
  if [ $cfg = guided ]; then flag="--guided"; name=${TAG}_guided_kernel_stats; jn=${TAG}_guided_bench_under_rocprof; fi
  rm -rf /tmp/prof_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python $ROOT/bench.py --steps 95 --warmup 5 --headline-only --no-cpu-baseline $flag > $OUT/$jn.json 2> /dev/null
  find /tmp/prof_$cfg -name "*kernel_stats.csv" -exec cp {} $OUT/$name.csv \;
done
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json > $OUT/${TAG}_pmc.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_guided_pmc_summary.json --guided >> $OUT/${TAG}_pmc.log 2>&1
python $ROOT/bench.py --train > $OUT/${TAG}_train_bench.json 2> /dev/null
python $ROOT/bench.py --train --precision fp16 > $OUT/${TAG}_train_bench_fp16.json 2> /dev/null
python $ROOT/bench.py --train --model bondpred --precision fp16 --no-cpu-baseline > $OUT/${TAG}_train_bench_bondpred_fp16.json 2> /dev/null
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 10 > /dev/null 2>&1
find /tmp/prof_tr -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_train_fp16_kernel_stats.csv \;
ls -la $OUT | grep ${TAG}_
