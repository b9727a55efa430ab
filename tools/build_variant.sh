#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA_FLAGS" [files...]  -> moldiff_amd/libmoldiff_hip_NAME.so
# Rebuilds only the listed .hip files (default: mdx_edge2.hip) with EXTRA_FLAGS into /tmp/mdxv_NAME and links them with
# the main build's other objects.  Development A/B tool (tools/bench_with_lib.py runs bench.py against the result).
set -e
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-mdx_edge2.hip}
cd "$(dirname "$0")/../moldiff_amd/csrc"
make -j8 >/dev/null
mkdir -p /tmp/mdxv_$NAME
OBJS=""
for o in $(sed -n "s/^OBJS *= *//p" Makefile); do
  src=${o%.o}.hip
  if echo " $FILES " | grep -q " $src "; then
    /opt/rocm/bin/hipcc $FLAGS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -c $src -o /tmp/mdxv_$NAME/$o
    OBJS="$OBJS /tmp/mdxv_$NAME/$o"
  else
    OBJS="$OBJS $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmoldiff_hip_$NAME.so $OBJS
echo built moldiff_amd/libmoldiff_hip_$NAME.so
