#!/bin/bash
# usage (GPU box, via gpurun, from the repo root): bash tools/collect_r6.sh   -> gpurun_out/r6_*  (copy what is to be judged into profiles/)
# Everything profiles/r6_* holds, taken at one HEAD on one box.  One pass per evidence kind; PMC passes never share a run with a trace.
#   r6_bench.json / r6_bench_full.json         python bench.py (default: one whole 1000-step run) -- the stdout line (<= 6 KB) and the full tree
#   r6_bench_driver.json                       python bench.py --gpus 1 --steps 20 --warmup 5 -- the driver's command
#   r6_kernel_stats.csv / r6_guided_*          rocprofv3 --kernel-trace --stats, exact fp32 path (configs #2 / #3), + the JSON line of the same run
#   r6_pmc_summary.json / r6_guided_pmc_summary.json    tools/pmc_summary.py: three separate --pmc passes each
#   r6_train_fp16_kernel_stats.csv             kernel statistics of 10 fp16 training steps (fused EdgeBlock kernels on) + the unfused twin
#   r6_train_bench_fp16_{1,2}.json, ..._pybodies_*  the training line outside the profiler (C++ / Python operator bodies)
#   r6_train_host_profile(_fast).txt           tools/profile_train_host.py: host time per phase + cProfile of the step
#   (profiles/r6_kernel_resources.txt is produced on the build host: tools/resusage.sh over every csrc/*.hip)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/r6_bench.json 2> $OUT/r6_bench.err
cp $ROOT/bench_full.json $OUT/r6_bench_full.json
python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r6_bench_driver.json 2> /dev/null
for cfg in simple guided; do
  flag=""; name=r6_kernel_stats; jn=r6_bench_under_rocprof
  if [ $cfg = guided ]; then flag="--guided"; name=r6_guided_kernel_stats; jn=r6_guided_bench_under_rocprof; fi
  rm -rf /tmp/prof_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python $ROOT/bench.py --steps 95 --warmup 5 --headline-only --no-cpu-baseline $flag > $OUT/$jn.json 2> /dev/null
  find /tmp/prof_$cfg -name "*kernel_stats.csv" -exec cp {} $OUT/$name.csv \;
done
python $ROOT/tools/pmc_summary.py $OUT/r6_pmc_summary.json > $OUT/r6_pmc.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/r6_guided_pmc_summary.json --guided >> $OUT/r6_pmc.log 2>&1
for v in 1 0; do
  rm -rf /tmp/prof_tr
  sfx=""; [ $v = 0 ] && sfx="_unfused"
  MDX_TRAIN_FUSED=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 10 > $OUT/r6_train_bench_fp16_under_rocprof$sfx.json 2> /dev/null
  find /tmp/prof_tr -name "*kernel_stats.csv" -exec cp {} $OUT/r6_train_fp16_kernel_stats$sfx.csv \;
done
# the training step outside the profiler: C++ operator bodies (default) and the Python bodies, two runs each (the box's host cores are shared)
for rep in 1 2; do
  python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 > $OUT/r6_train_bench_fp16_$rep.json 2> /dev/null
  MDX_TRAIN_FAST=0 python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 > $OUT/r6_train_bench_fp16_pybodies_$rep.json 2> /dev/null
done
python $ROOT/tools/profile_train_host.py > $OUT/r6_train_host_profile_fast.txt 2>&1
MDX_TRAIN_FAST=0 python $ROOT/tools/profile_train_host.py > $OUT/r6_train_host_profile.txt 2>&1
cd $ROOT
ls -la $OUT | grep r6_
