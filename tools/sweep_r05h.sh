#!/bin/bash
# round 5, eighth sweep: 128 lanes / 4 groups with the admission overflow: stream -> hardware-queue binding made deterministic
# (lines touched at creation, adjusters on the group's lines): which line / queue counts are fast, and are they reproducible?
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_h.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env SVO_BA_OVERFLOW=1 "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline --streams 128 --groups 4 "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
for i in 1 2; do run "touch order BA-chain-LK, 16 queues ($i)" X=1 -- ; done
for i in 1 2; do run "touch order LK-chain-BA, 16 queues ($i)" SVO_GROUP_TOUCH=2 -- ; done
for i in 1 2; do run "no touch, adjusters on the lines ($i)" SVO_GROUP_TOUCH=0 -- ; done
for i in 1 2; do run "no touch, own adjuster streams = round 5 so far ($i)" SVO_GROUP_TOUCH=0 SVO_GROUP_OWN_BA_STREAMS=1 -- ; done
for i in 1 2; do run "28 queues ($i)" GPU_MAX_HW_QUEUES=28 -- ; done
for i in 1 2; do run "32 queues ($i)" GPU_MAX_HW_QUEUES=32 -- ; done
for i in 1 2; do run "lines 1/1/2, 16 queues ($i)" SVO_GROUP_CHAIN_LINES=1 SVO_GROUP_BA_LINES=2 -- ; done
for i in 1 2; do run "lines 1/2/2, 20 queues ($i)" SVO_GROUP_BA_LINES=2 GPU_MAX_HW_QUEUES=20 -- ; done
for i in 1 2; do run "lines 1/2/3, 24 queues ($i)" SVO_GROUP_BA_LINES=3 GPU_MAX_HW_QUEUES=24 -- ; done
