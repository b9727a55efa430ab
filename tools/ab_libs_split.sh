#!/bin/bash
# A/B of library variants on the SPLIT float16 path, configs #2 and #3: tools/ab_libs_split.sh name1 name2 ...  -> gpurun_out/ab_libs_split.txt
OUT=gpurun_out/ab_libs_split.txt; : > $OUT
for n in "$@"; do
  lib=moldiff_amd/libmoldiff_hip_$n.so; [ $n = main ] && lib=moldiff_amd/libmoldiff_hip.so
  for flag in "" "--guided"; do
    echo "== $n $flag" >> $OUT
    MOLDIFF_MATRIX_PATH=split_f16 python tools/bench_with_lib.py $lib $flag --headline-only --no-cpu-baseline --steps 60 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> $OUT
  done
done
cat $OUT
