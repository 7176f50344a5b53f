#!/bin/bash
# Where an LM iteration of ba_lm_kernel goes (its own wall_clock64 stamps, workgroup 0 + min/mean/max over the workgroups):
# one solve alone on the GPU, and the same under the default bench load.  Writes gpurun_out/lm_iteration_split.txt.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/lm_iteration_split.txt
echo "# $(sha256sum stereo_vo_amd/libsvo_hip.so | cut -c1-16) libsvo_hip.so; commit ${1:-unknown}" > $OUT
echo "== one stream alone on the GPU, device LM (SVO_BA_DEVICE_LM=1), bench workload (5-keyframe window, ~5.9 k observations, 94 chunks / 47 workgroups)" >> $OUT
SVO_TIMING=1 SVO_BA_TRACE=1 SVO_BA_DEVICE_LM=1 python bench.py --steps 10 --warmup 2 --streams 1 --groups 0 --no-cpu-baseline --no-other-workloads --no-streaming 2>&1 | grep "svo ba" | grep -v "^\[lm\]" >> $OUT
echo "== lane 0 of group 0 under the default load (48 lanes in 2 groups)" >> $OUT
SVO_TIMING=1 SVO_BA_TRACE=1 python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>&1 | grep "svo ba" | head -12 >> $OUT
cat $OUT
