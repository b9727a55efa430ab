#!/bin/bash
# A/B of library variants on the guided step: tools/ab_libs.sh name1 name2 ...  (libmoldiff_hip_<name>.so; "main" = the product build)
# -> gpurun_out/ab_libs.txt
OUT=gpurun_out/ab_libs.txt; : > $OUT
for n in "$@"; do
  lib=moldiff_amd/libmoldiff_hip_$n.so; [ $n = main ] && lib=moldiff_amd/libmoldiff_hip.so
  echo "== $n" >> $OUT
  python tools/bench_with_lib.py $lib --guided --headline-only --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> $OUT
done
cat $OUT
