#!/bin/bash
# round 5: compact lines — solves the admission budget refuses leave at once in the one-workgroup form on lines of their own
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_s.txt
: > $OUT
( SVO_GROUP_COMPACT_LINES=1 SVO_BA_BUDGET_PERCENT=40 timeout -k 10 400 python -m pytest tests/test_group.py -m gpu -x -q ) > gpurun_out/r5_sweep_s_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r5_sweep_s_tests.log | tee -a $OUT
if [ $rc -ne 0 ]; then echo "tests failed: no bench" | tee -a $OUT; exit 1; fi
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  t=$(grep -o '"launches_per_step_of_group_0": {[^}]*}' gpurun_out/r5_sweep_tmp.log | head -1)
  echo "$label rc=$rc $v $p $h $t" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "48/2" X=1 -- --streams 48 --groups 2
run "48/2 + 1 compact line" SVO_GROUP_COMPACT_LINES=1 -- --streams 48 --groups 2
run "64/2 + 1 compact line" SVO_GROUP_COMPACT_LINES=1 -- --streams 64 --groups 2
run "64/2 + 2 compact lines, 3 wide" SVO_GROUP_COMPACT_LINES=2 SVO_GROUP_BA_LINES=3 -- --streams 64 --groups 2
run "96/3 lines 1/1/2 + 1 compact" SVO_GROUP_COMPACT_LINES=1 -- --streams 96 --groups 3
run "128/4 lines 1/1/1 + 1 compact" SVO_GROUP_COMPACT_LINES=1 SVO_GROUP_BA_LINES=1 -- --streams 128 --groups 4
run "128/4 lines 1/1/2 + 1 compact (20 streams)" SVO_GROUP_COMPACT_LINES=1 -- --streams 128 --groups 4
run "128/4 lines 1/1/2 + 2 compact (24 streams, 24 queues)" SVO_GROUP_COMPACT_LINES=2 GPU_MAX_HW_QUEUES=24 -- --streams 128 --groups 4
run "96/2 (48 per group) + 2 compact" SVO_GROUP_COMPACT_LINES=2 -- --streams 96 --groups 2
run "48/2 gather 250 (again)" SVO_GROUP_GATHER_US=250 -- --streams 48 --groups 2
run "48/2 gather 250 + 1 compact" SVO_GROUP_GATHER_US=250 SVO_GROUP_COMPACT_LINES=1 -- --streams 48 --groups 2
