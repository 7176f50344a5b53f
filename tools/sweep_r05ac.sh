#!/bin/bash
# round 5: the tracking launch is tail-dominated (average tracker wavefront 78 us, launch 650 us): a second tracking line per group that may
# NOTE: the f32 iteration is not in the tree: tools/exp/patches/lk_f32_iteration.patch applies to csrc/lk.hip; results in profiles/r05_exp_lanes_groups_honest.txt
# depart next to a launch that is already in its tail (SVO_GROUP_LK_LINES=2, SVO_GROUP_LK_OVERLAP_US)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_ac.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  t=$(grep -o '"track": \[[0-9., ]*\]' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $t $l" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "default (96/3, lines 1/1/2+1)" X=1 --
run "96/3, lines 2/1/1+1, second tracking launch next to one older than 150 us" SVO_GROUP_LK_LINES=2 SVO_GROUP_LK_OVERLAP_US=150 SVO_GROUP_BA_LINES=1 --
run "96/3, lines 2/1/1+1, older than 300 us" SVO_GROUP_LK_LINES=2 SVO_GROUP_LK_OVERLAP_US=300 SVO_GROUP_BA_LINES=1 --
run "96/3, lines 2/1/1+1, static split of the lanes over two tracking lines" SVO_GROUP_LK_LINES=2 SVO_GROUP_BA_LINES=1 --
run "96/3, lines 1/1/1+1 (the same solve lines, one tracking line)" SVO_GROUP_BA_LINES=1 --
run "48/2, lines 2/2/4, older than 150 us" SVO_GROUP_LK_LINES=2 SVO_GROUP_LK_OVERLAP_US=150 -- --streams 48 --groups 2
run "48/2, lines 2/2/4, older than 300 us" SVO_GROUP_LK_LINES=2 SVO_GROUP_LK_OVERLAP_US=300 -- --streams 48 --groups 2
