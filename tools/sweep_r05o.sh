#!/bin/bash
# round 5: a group whose solves are all refused (budget held by other groups) launches the compact form instead of a host-driven solve
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_o.txt
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
for i in 1 2 3 4 5 6; do run "stalled group -> compact ($i)" X=1 -- ; done
for i in 1 2 3 4; do run "stalled group -> host-driven solve = before ($i)" SVO_GROUP_STALLED_COMPACT=0 -- ; done
for i in 1 2 3 4; do run "stalled -> compact, budget 125 % ($i)" SVO_BA_BUDGET_PERCENT=125 -- ; done
for i in 1 2; do run "stalled -> compact, 96 / 4 ($i)" X=1 -- --streams 96 --groups 4; done
