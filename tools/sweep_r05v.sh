#!/bin/bash
# round 5: tracker iteration with its two sums finished in LDS (119 VALU instructions per iteration instead of 132 / 146): parity
# NOTE: the LDS form of the sums lived in the working tree of the experiment only (described in csrc/lk.hip at the iteration sums); result in profiles/r05_exp_lanes_groups_honest.txt
# tests, then the new default (96 lanes / 3 groups / 1 compact line) and round 4's shape
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_v.txt
: > $OUT
( timeout -k 10 600 python -m pytest tests/test_frontend.py tests/test_group.py tests/test_pipeline.py -m gpu -x -q ) > gpurun_out/r5_sweep_v_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r5_sweep_v_tests.log | tee -a $OUT
if [ $rc -ne 0 ]; then echo "tests failed: no bench" | tee -a $OUT; exit 1; fi
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
run "default = 96/3 + 1 compact line (1)" X=1 --
run "default (2)" X=1 --
run "default (3)" X=1 --
run "48/2 (1)" X=1 -- --streams 48 --groups 2
run "48/2 (2)" X=1 -- --streams 48 --groups 2
run "108/3" X=1 -- --streams 108 --groups 3
run "32/1" X=1 -- --streams 32 --groups 1
