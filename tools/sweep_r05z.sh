#!/bin/bash
# round 5 diagnostic: do the wide solves slow the tracker down by sharing its CUs (instruction cache, LDS)?  SVO_BA_CU_SHARE=n puts a
# NOTE: the group lines honoured SVO_BA_CU_SHARE in the working tree of this experiment only; the result is in profiles/r05_exp_lanes_groups_honest.txt
# group's solve lines on n CUs of every 32 and the tracking line on the others (the admission budget shrinks with n: the frame rate
# is expected to FALL — what is looked at is the tracking launch's duration next to confined solves)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_z.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  t=$(grep -o '"launches_per_step_of_group_0": {[^}]*}' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l $t" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "default (solves on every CU)" X=1 --
run "solve lines on 16 of every 32 CUs, tracking line on the other 16" SVO_BA_CU_SHARE=16 --
run "solve lines on 8 of every 32 CUs, tracking line on the other 24" SVO_BA_CU_SHARE=8 --
run "solve lines on 12 of 32" SVO_BA_CU_SHARE=12 --
run "48/2, default" X=1 -- --streams 48 --groups 2
run "48/2, solve lines on 16 of 32" SVO_BA_CU_SHARE=16 -- --streams 48 --groups 2
run "default, every solve host-driven (no wide solve kernel at all)" SVO_GROUP_HOST_SOLVES=1 --
