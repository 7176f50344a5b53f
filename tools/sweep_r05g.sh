#!/bin/bash
# round 5, seventh sweep: more lanes PER GROUP (SVO_MAX_LANES 32 -> 64): fewer host threads and HIP streams for the same lanes
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_g.txt
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
for i in 1 2; do run "128 / 2 = 64 per group ($i)" X=1 -- --streams 128 --groups 2; done
run "128 / 2, stagger 30" X=1 -- --streams 128 --groups 2 --stagger-ms 30
for i in 1 2; do run "96 / 2 = 48 per group ($i)" X=1 -- --streams 96 --groups 2; done
for i in 1 2; do run "128 / 3 ($i)" X=1 -- --streams 128 --groups 3; done
for i in 1 2; do run "192 / 3 = 64 per group ($i)" X=1 -- --streams 192 --groups 3; done
run "192 / 4 = 48 per group" X=1 -- --streams 192 --groups 4
run "256 / 4 = 64 per group" X=1 -- --streams 256 --groups 4
run "64 / 1" X=1 -- --streams 64 --groups 1
run "128 / 2, 8 BA lines" SVO_GROUP_BA_LINES=8 -- --streams 128 --groups 2
run "128 / 2, 3 chain lines" SVO_GROUP_CHAIN_LINES=3 -- --streams 128 --groups 2
run "128 / 2, 2 LK lines" SVO_GROUP_LK_LINES=2 -- --streams 128 --groups 2
for i in 1 2; do run "128 / 4 = 32 per group ($i)" X=1 -- --streams 128 --groups 4; done
