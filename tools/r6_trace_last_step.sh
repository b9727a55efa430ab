#!/bin/bash
# usage (GPU box, via gpurun, from the repo root): bash tools/r6_trace_last_step.sh -> gpurun_out/r6_train_trace_last_step.txt, r6_train_bench_fp16_probe.json
# The ordered kernel trace (start, duration, gap to the previous kernel's end, name) of the LAST fp16 training step under rocprofv3, and
# the training line outside the profiler (with the fixed-probe loss pair).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 20 --warmup 6 2> $OUT/r6_train_bench_fp16_probe.err | tail -1 > $OUT/r6_train_bench_fp16_probe.json
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 10 > /dev/null 2>&1
python - <<'P' > $OUT/r6_train_trace_last_step.txt 2>&1
import csv, glob, re
f = glob.glob('/tmp/prof_tr/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_amp_kernel' in r['Kernel_Name']]   # one step = between two adamw_amp_kernel launches
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::|at::native::|void ', '', r['Kernel_Name'])[:70]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = e
P
wc -l $OUT/r6_train_trace_last_step.txt; cat $OUT/r6_train_bench_fp16_probe.json | head -c 3000; tail -3 $OUT/r6_train_bench_fp16_probe.err
