#!/bin/bash
# usage (GPU box, via gpurun): bash tools/final_collect.sh   -> gpurun_out/r3_* (copy into profiles/)
# Everything profiles/r3_* holds, taken at one HEAD on one box: tools/collect_profiles.sh (bench line, rocprofv3 kernel statistics,
# PMC passes, training benches), the batch sweep, the training GEMM micro-benchmark and host profile, the phase traces of the three
# row-owner edge kernels (needs moldiff_amd/libmoldiff_hip_trace2.so: tools/build_variant.sh trace2 -DMDX_TRACE2 mdx_edge2.hip
# mdx_edge2b.hip mdx_bwd2.hip) and the work-queue A/B.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/collect_profiles.sh r3 > gpurun_out/collect.log 2>&1
bash tools/batch_sweep.sh > gpurun_out/r3_batch_sweep.txt 2>&1
python tools/ubench_train_gemms.py > gpurun_out/r3_ubench_train_gemms.txt 2>&1
python tools/profile_train_cpu.py fp16 > gpurun_out/r3_train_host_profile.txt 2>&1
MDX_NO_TAIL_SPLIT=1 python tools/trace_edge2.py a > gpurun_out/r3_trace_edge_a2.txt 2>&1
python tools/trace_edge2.py b > gpurun_out/r3_trace_edge_b2.txt 2>&1
python tools/trace_edge2.py w > gpurun_out/r3_trace_edge_bwd2.txt 2>&1
for v in "MDX_STATIC_SPLIT=1" "MDX_STATIC_SPLIT=0"; do echo $v; env $v python bench.py --headline-only --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"; done > gpurun_out/r3_workqueue_ab.txt 2>&1
tail -3 gpurun_out/r3_trace_edge_a2.txt gpurun_out/r3_workqueue_ab.txt gpurun_out/r3_train_host_profile.txt | head -40
