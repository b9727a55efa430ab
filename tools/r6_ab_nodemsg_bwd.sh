#!/bin/bash
# usage (GPU box, via gpurun, from the repo root): bash tools/r6_ab_nodemsg_bwd.sh  -> prints nodemsg_bwd / nodemsg_fwd us per launch per variant
# Ablation / variant libraries built on the build host into tools/_ab/lib_<tag>.so are swapped in for moldiff_amd/libmoldiff_hip.so of the
# box's scratch copy, one after the other (`base` = the shipped library); rocprofv3 --kernel-trace --stats of 6 fp16 training steps each.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cp $ROOT/moldiff_amd/libmoldiff_hip.so /tmp/lib_base.so
cd /tmp && export TMPDIR=/tmp
for tag in base $(ls $ROOT/tools/_ab | sed -n 's/^lib_\(.*\)\.so$/\1/p'); do
  if [ $tag = base ]; then cp /tmp/lib_base.so $ROOT/moldiff_amd/libmoldiff_hip.so; else cp $ROOT/tools/_ab/lib_$tag.so $ROOT/moldiff_amd/libmoldiff_hip.so; fi
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o p -- python $ROOT/bench.py --train --precision fp16 --no-cpu-baseline --steps 6 --warmup 2 > /tmp/line_$tag.json 2>/dev/null
  python - $tag <<'P'
import sys, glob, csv, json, os
tag = sys.argv[1]
f = glob.glob(f'/tmp/p_{tag}/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
pick = {r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34]: round(float(r['AverageNs']) / 1e3, 1) for r in rows if any(k in r['Name'] for k in os.environ.get('AB_KERNELS', 'nodemsg,bondffn,posffn').split(','))}
try:
    ms = round(json.loads(open(f'/tmp/line_{tag}.json').read().strip().splitlines()[-1])['ms_per_step'], 2)
except Exception as e:
    ms = repr(e)
print(tag, 'step(under rocprof)', ms, pick)
P
done
cp /tmp/lib_base.so $ROOT/moldiff_amd/libmoldiff_hip.so
