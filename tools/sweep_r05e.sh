#!/bin/bash
# round 5, fifth sweep: lanes x groups, wide form only vs overflow, repeated (the 128-lane / 4-group figure varies from run to run)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_e.txt
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
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
for i in 1 2 3 4; do run "wide 128 / 4 ($i)" X=1 -- --streams 128 --groups 4; done
for i in 1 2; do run "wide 96 / 3 ($i)" X=1 -- --streams 96 --groups 3; done
for i in 1 2; do run "wide 96 / 4 ($i)" X=1 -- --streams 96 --groups 4; done
for i in 1 2; do run "wide 112 / 4 ($i)" X=1 -- --streams 112 --groups 4; done
for i in 1 2; do run "wide 160 / 5 ($i)" X=1 -- --streams 160 --groups 5; done
run "wide 192 / 6" X=1 -- --streams 192 --groups 6
run "wide 256 / 8" X=1 -- --streams 256 --groups 8
for i in 1 2; do run "overflow 128 / 4 ($i)" SVO_BA_OVERFLOW=1 -- --streams 128 --groups 4; done
run "wide 128 / 4, 40 steps" X=1 -- --streams 128 --groups 4 --steps 40
run "wide 48 / 2 (round 4's default)" X=1 --
