#!/bin/bash
# round 5: keyframe chain with the stereo + triangulation launch queued right behind the PnP launch (no host turn) against one host
# turn per launch; group tests first (bit-parity of both forms), then the default bench interleaved
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_p.txt
: > $OUT
( timeout -k 10 500 python -m pytest tests/test_group.py tests/test_pnp.py -m gpu -x -q ) > gpurun_out/r5_sweep_p_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r5_sweep_p_tests.log | tee -a $OUT
if [ $rc -ne 0 ]; then echo "tests failed: no bench" | tee -a $OUT; exit 1; fi
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  t=$(grep -o '"track": \[[0-9., ]*\]' gpurun_out/r5_sweep_tmp.log | head -1)
  echo "$label rc=$rc $v $p $h $t" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
for i in 1 2 3; do
  run "fused chain ($i)" X=1 -- --steps 20 --warmup 5
  run "one host turn per launch ($i)" SVO_GROUP_CHAIN_FUSED=0 -- --steps 20 --warmup 5
  run "unfused, reset fill on the null stream, not waited for ($i)" SVO_GROUP_CHAIN_FUSED=0 SVO_GROUP_RESET_FILL=1 -- --steps 20 --warmup 5
  run "unfused, no reset fill ($i)" SVO_GROUP_CHAIN_FUSED=0 SVO_GROUP_RESET_FILL=2 -- --steps 20 --warmup 5
done
run "fused, single group of 32" X=1 -- --streams 32 --groups 1
run "unfused, single group of 32" SVO_GROUP_CHAIN_FUSED=0 -- --streams 32 --groups 1
