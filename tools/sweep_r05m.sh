#!/bin/bash
# round 5: the new default line (128 lanes / 4 groups, lines 1/1/2, 16 hardware queues, overflow by lane count), repeated
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_m.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
for i in 1 2 3 4 5 6; do run "default ($i)" X=1 -- ; done
for i in 1 2 3; do run "default, solve lines low priority ($i)" SVO_GROUP_BA_PRIORITY=low -- ; done
for i in 1 2; do run "default, 40 steps ($i)" X=1 -- --steps 40; done
for i in 1 2; do run "96 / 3 ($i)" X=1 -- --streams 96 --groups 3; done
for i in 1 2; do run "default, overflow off ($i)" SVO_BA_OVERFLOW=0 -- ; done
