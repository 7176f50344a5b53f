#!/bin/bash
# round 5, tenth sweep: repeats of the configurations of sweep h that were fast twice
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_j.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env SVO_BA_OVERFLOW=1 "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
for i in 1 2 3 4; do run "128/4, 32 queues ($i)" GPU_MAX_HW_QUEUES=32 -- --streams 128 --groups 4; done
for i in 1 2 3 4; do run "128/4, lines 1/1/2, 16 queues ($i)" SVO_GROUP_CHAIN_LINES=1 SVO_GROUP_BA_LINES=2 -- --streams 128 --groups 4; done
for i in 1 2; do run "128/4, 32 queues, own adjuster streams ($i)" GPU_MAX_HW_QUEUES=32 SVO_GROUP_OWN_BA_STREAMS=1 -- --streams 128 --groups 4; done
for i in 1 2; do run "128/4, 32 queues, wide only ($i)" GPU_MAX_HW_QUEUES=32 SVO_BA_OVERFLOW=0 -- --streams 128 --groups 4; done
for i in 1 2; do run "160/5, 40 queues ($i)" GPU_MAX_HW_QUEUES=40 -- --streams 160 --groups 5; done
for i in 1 2; do run "96/3, 24 queues ($i)" GPU_MAX_HW_QUEUES=24 -- --streams 96 --groups 3; done
for i in 1 2; do run "48/2, 16 queues ($i)" X=1 -- ; done
for i in 1 2; do run "128/4, lines 1/1/2, 16 queues, BA low priority ($i)" SVO_GROUP_CHAIN_LINES=1 SVO_GROUP_BA_LINES=2 SVO_GROUP_BA_PRIORITY=low -- --streams 128 --groups 4; done
