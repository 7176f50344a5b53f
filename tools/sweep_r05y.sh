#!/bin/bash
# round 5: tracker wavefronts per SIMD (SVO_GROUP_LK_WAVES = 4 / 5 / 6: 126 / 96 / 80 VGPRs) — is the plateau occupancy?
# NOTE: the knob this sweep toggles (SVO_GROUP_LK_WAVES) lived in the working tree of the experiment only; the result is in profiles/r05_exp_lanes_groups_honest.txt
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_y.txt
: > $OUT
for w in 5 6; do
( SVO_GROUP_LK_WAVES=$w timeout -k 10 400 python -m pytest tests/test_group.py -m gpu -x -q -k "oracles or headline or separate" ) > gpurun_out/r5_sweep_y_tests.log 2>&1
rc=$?
echo "SVO_GROUP_LK_WAVES=$w: $(tail -1 gpurun_out/r5_sweep_y_tests.log)" | tee -a $OUT
if [ $rc -ne 0 ]; then tail -20 gpurun_out/r5_sweep_y_tests.log | tee -a $OUT; echo "tests failed: no bench" | tee -a $OUT; exit 1; fi
done
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
run "default, 4 tracker wavefronts per SIMD" X=1 --
run "default, 5" SVO_GROUP_LK_WAVES=5 --
run "default, 6" SVO_GROUP_LK_WAVES=6 --
run "default, 4 (2)" X=1 --
run "default, 5 (2)" SVO_GROUP_LK_WAVES=5 --
run "default, 6 (2)" SVO_GROUP_LK_WAVES=6 --
run "48/2, 4" X=1 -- --streams 48 --groups 2
run "48/2, 6" SVO_GROUP_LK_WAVES=6 -- --streams 48 --groups 2
run "32/1, 4" X=1 -- --streams 32 --groups 1
run "32/1, 6" SVO_GROUP_LK_WAVES=6 -- --streams 32 --groups 1
