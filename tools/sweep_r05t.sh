#!/bin/bash
# round 5: tracker with the template subtraction folded into the bilinear value's rounding (132 instead of 146 VALU instructions per
# LK iteration): parity tests, then the default bench three times and the HD workload
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_t.txt
: > $OUT
( timeout -k 10 600 python -m pytest tests/test_frontend.py tests/test_group.py tests/test_pipeline.py -m gpu -x -q ) > gpurun_out/r5_sweep_t_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r5_sweep_t_tests.log | tee -a $OUT
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
run "48/2 (1)" X=1 --
run "48/2 (2)" X=1 --
run "48/2 (3)" X=1 --
run "96/3 + 1 compact line" SVO_GROUP_COMPACT_LINES=1 -- --streams 96 --groups 3
run "96/3 + 1 compact line (2)" SVO_GROUP_COMPACT_LINES=1 -- --streams 96 --groups 3
run "72/3 + 1 compact line" SVO_GROUP_COMPACT_LINES=1 -- --streams 72 --groups 3
run "32/1 (one group)" X=1 -- --streams 32 --groups 1
timeout -k 10 280 python bench.py --workload hd10k --no-cpu-baseline > gpurun_out/r5_sweep_tmp.log 2>&1; echo "hd10k rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)" | tee -a $OUT
