#!/bin/bash
# round 5, third sweep: admission overflow into the compact form at higher lane counts (run on the GPU box)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_c.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 240 python bench.py --no-cpu-baseline --no-other-workloads --no-single --no-streaming "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
SVO_TIMING=1 timeout -k 10 200 python bench.py --workload ba50k --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5_ba5.log 2>&1
grep "device-side step control" gpurun_out/r5_ba5.log | tail -1 | tee -a $OUT
grep -o '"value": [0-9.]*' gpurun_out/r5_ba5.log | head -1 | tee -a $OUT
run "overflow 96 / 3" SVO_BA_OVERFLOW=1 -- --streams 96 --groups 3
run "overflow 96 / 3 (again)" SVO_BA_OVERFLOW=1 -- --streams 96 --groups 3
run "wide 96 / 3" X=1 -- --streams 96 --groups 3
run "overflow 128 / 4" SVO_BA_OVERFLOW=1 -- --streams 128 --groups 4
run "overflow 96 / 4" SVO_BA_OVERFLOW=1 -- --streams 96 --groups 4
run "overflow 64 / 2, budget 75 %" SVO_BA_OVERFLOW=1 SVO_BA_BUDGET_PERCENT=75 -- --streams 64 --groups 2
run "overflow 96 / 3, budget 75 %" SVO_BA_OVERFLOW=1 SVO_BA_BUDGET_PERCENT=75 -- --streams 96 --groups 3
run "overflow 96 / 3, budget 50 %" SVO_BA_OVERFLOW=1 SVO_BA_BUDGET_PERCENT=50 -- --streams 96 --groups 3
run "overflow 128 / 4, budget 50 %" SVO_BA_OVERFLOW=1 SVO_BA_BUDGET_PERCENT=50 -- --streams 128 --groups 4
