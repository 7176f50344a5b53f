#!/bin/bash
# round 5: around 96 lanes / 3 groups + 1 compact line
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_u.txt
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
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "96/3 + 2 compact lines" SVO_GROUP_COMPACT_LINES=2 -- --streams 96 --groups 3
run "96/3 + 1 compact, 2 chain lines" SVO_GROUP_COMPACT_LINES=1 SVO_GROUP_CHAIN_LINES=2 -- --streams 96 --groups 3
run "96/3 + 1 compact, gather 250" SVO_GROUP_COMPACT_LINES=1 SVO_GROUP_GATHER_US=250 -- --streams 96 --groups 3
run "96/3 + 1 compact, budget 75 %" SVO_GROUP_COMPACT_LINES=1 SVO_BA_BUDGET_PERCENT=75 -- --streams 96 --groups 3
run "96/3 + 1 compact, budget 125 %" SVO_GROUP_COMPACT_LINES=1 SVO_BA_BUDGET_PERCENT=125 -- --streams 96 --groups 3
run "84/3 + 1 compact" SVO_GROUP_COMPACT_LINES=1 -- --streams 84 --groups 3
run "108/3 + 1 compact" SVO_GROUP_COMPACT_LINES=1 -- --streams 108 --groups 3
run "120/3 + 1 compact" SVO_GROUP_COMPACT_LINES=1 -- --streams 120 --groups 3
run "128/4 + 1 compact (lines 1/1/2+1)" SVO_GROUP_COMPACT_LINES=1 -- --streams 128 --groups 4
run "96/3 + 1 compact, 1 wide line" SVO_GROUP_COMPACT_LINES=1 SVO_GROUP_BA_LINES=1 -- --streams 96 --groups 3
run "96/3 + 1 compact, 3 wide lines" SVO_GROUP_COMPACT_LINES=1 SVO_GROUP_BA_LINES=3 -- --streams 96 --groups 3
run "96/3 + 1 compact (again)" SVO_GROUP_COMPACT_LINES=1 -- --streams 96 --groups 3
