#!/bin/bash
# round 5: split prepare — corner detection on the first chain line next to the pyramids on the tracking line at the start of a batch
# NOTE: the knob this sweep toggles (SVO_GROUP_SPLIT_PREPARE) lived in the working tree of the experiment only; the result is in profiles/r05_exp_lanes_groups_honest.txt
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_x.txt
: > $OUT
( timeout -k 10 700 python -m pytest tests/test_group.py tests/test_pipeline.py -m gpu -x -q ) > gpurun_out/r5_sweep_x_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r5_sweep_x_tests.log | tee -a $OUT
if [ $rc -ne 0 ]; then echo "tests failed: no bench" | tee -a $OUT; exit 1; fi
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  sv=$(grep -o '"streaming": {"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  sb=$(grep -o '"bit_identical_to_resident": [a-z]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  b=$(grep -o '"host_thread_busy_us_of_call_us": \[[0-9., ]*\]' gpurun_out/r5_sweep_tmp.log | head -1)
  echo "$label rc=$rc $v $p $h $b $sv $sb" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "default, split prepare (1)" X=1 --
run "default, prepare on one stream (1)" SVO_GROUP_SPLIT_PREPARE=0 --
run "default, split prepare (2)" X=1 --
run "default, prepare on one stream (2)" SVO_GROUP_SPLIT_PREPARE=0 --
run "48/2, split prepare" X=1 -- --streams 48 --groups 2
run "48/2, prepare on one stream" SVO_GROUP_SPLIT_PREPARE=0 -- --streams 48 --groups 2
run "32/1, split prepare" X=1 -- --streams 32 --groups 1
run "32/1, prepare on one stream" SVO_GROUP_SPLIT_PREPARE=0 -- --streams 32 --groups 1
