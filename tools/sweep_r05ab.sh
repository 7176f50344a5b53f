#!/bin/bash
# round 5: the LK iteration's bilinear target values by full-rate f32 FMAs on an f32 copy of the staged region (exact: every value is a
# NOTE: the f32 iteration is not in the tree: tools/exp/patches/lk_f32_iteration.patch applies to csrc/lk.hip; results in profiles/r05_exp_lanes_groups_honest.txt
# multiple of 2^-9 below 2^14) instead of byte permutes + 16-bit dot products: parity first, then the bench
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_ab.txt
: > $OUT
( timeout -k 10 700 python -m pytest tests/test_frontend.py tests/test_group.py tests/test_pipeline.py -m gpu -x -q ) > gpurun_out/r5_sweep_ab_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r5_sweep_ab_tests.log | tee -a $OUT
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r5_sweep_ab_tests.log | cut -c1-300 | tee -a $OUT; echo "tests failed: no bench" | tee -a $OUT; exit 1; fi
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  sg=$(grep -o '"single_stream": {"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $sg $p $h $l" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "default (1)" X=1 --
run "default (2)" X=1 --
run "default (3)" X=1 --
run "48/2" X=1 -- --streams 48 --groups 2
run "32/1" X=1 -- --streams 32 --groups 1 --no-single
timeout -k 10 280 python bench.py --workload hd10k --no-cpu-baseline > gpurun_out/r5_sweep_tmp.log 2>&1; echo "hd10k rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1) $(grep -o '"all_frames_identical_to_the_oracle": [a-z]*' gpurun_out/r5_sweep_tmp.log | head -1)" | tee -a $OUT
