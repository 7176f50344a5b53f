#!/bin/bash
# Builds and runs the stand-alone experiments behind DESIGN.md sections 6 and 9 on the GPU box; writes gpurun_out/exp/*.txt
# (copied to profiles/r02_exp_*.txt when they are to be committed).
set -euo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
for e in l2_invalidate launch_rate concurrency pingpong placement; do
  hipcc --offload-arch=gfx950 -O2 -w -o /tmp/exp_$e tools/exp/$e.hip -lpthread
  timeout -k 5 150 /tmp/exp_$e > $OUT/$e.txt 2>&1
  echo "== $e"; tail -3 $OUT/$e.txt
done
g++ -O2 -o /tmp/exp_chol tools/exp/chol_time.cpp -Lstereo_vo_amd -lsvo_hip -Wl,-rpath,$PWD/stereo_vo_amd
/tmp/exp_chol > $OUT/chol_time.txt 2>&1; grep -m1 "model name" /proc/cpuinfo >> $OUT/chol_time.txt
