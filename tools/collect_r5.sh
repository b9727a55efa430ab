#!/bin/bash
# usage (GPU box, via gpurun, from the repo root): bash tools/collect_r5.sh   -> gpurun_out/r5_*  (copy what is to be judged into profiles/)
# Everything profiles/r5_* holds, taken at one HEAD on one box.  One pass per evidence kind; PMC passes never share a run with a trace.
#   r5_bench.json                              python bench.py -- the driver's command (exact headline + configs.{simple,guided,simple_split,guided_split,guided_mixed,train})
#   r5_kernel_stats.csv / r5_guided_*          rocprofv3 --kernel-trace --stats, exact fp32 path (configs #2 / #3), + the JSON line of the same run
#   r5_split_kernel_stats.csv / r5_split_guided_*   the same with MOLDIFF_MATRIX_PATH=split_f16
#   r5_pmc_summary.json / r5_split_pmc_summary.json / r5_guided_* / r5_split_guided_*   tools/pmc_summary.py: three separate --pmc passes each
#   r5_train_fp16_kernel_stats.csv             kernel statistics of 10 fp16 training steps
#   r5_split_accuracy.txt, r5_parity_both_paths.txt (tail statistic, stress weights, eight objectives: both matrix paths vs float64), r5_trace_edge_bwd2.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/r5_bench.json 2> $OUT/r5_bench.err
for path in exact_f32 split_f16; do
  for cfg in simple guided; do
    pre=r5_; [ $path = split_f16 ] && pre=r5_split_
    flag=""; name=${pre}kernel_stats; jn=${pre}bench_under_rocprof
    if [ $cfg = guided ]; then flag="--guided"; name=${pre}guided_kernel_stats; jn=${pre}guided_bench_under_rocprof; fi
    rm -rf /tmp/prof_$cfg
    MOLDIFF_MATRIX_PATH=$path rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python $ROOT/bench.py --steps 95 --warmup 5 --headline-only --no-cpu-baseline $flag > $OUT/$jn.json 2> /dev/null
    find /tmp/prof_$cfg -name "*kernel_stats.csv" -exec cp {} $OUT/$name.csv \;
  done
done
python $ROOT/tools/pmc_summary.py $OUT/r5_pmc_summary.json > $OUT/r5_pmc.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/r5_guided_pmc_summary.json --guided >> $OUT/r5_pmc.log 2>&1
MOLDIFF_MATRIX_PATH=split_f16 python $ROOT/tools/pmc_summary.py $OUT/r5_split_pmc_summary.json >> $OUT/r5_pmc.log 2>&1
MOLDIFF_MATRIX_PATH=split_f16 python $ROOT/tools/pmc_summary.py $OUT/r5_split_guided_pmc_summary.json --guided >> $OUT/r5_pmc.log 2>&1
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 10 > $OUT/r5_train_bench_fp16_under_rocprof.json 2> /dev/null
find /tmp/prof_tr -name "*kernel_stats.csv" -exec cp {} $OUT/r5_train_fp16_kernel_stats.csv \;
cd $ROOT
python tools/split_accuracy.py > $OUT/r5_split_accuracy.txt 2>/dev/null
python -m pytest tests/test_gpu_round5.py tests/test_gpu_sampling.py tests/test_gpu_fullsize.py -q -m gpu -s -k "tail_statistic or stress or eight or mixed" 2>&1 | grep -v "^\s*$" > $OUT/r5_parity_both_paths.txt
python tools/trace_edge2.py w > $OUT/r5_trace_edge_bwd2.txt 2>&1
ls -la $OUT | grep r5_
