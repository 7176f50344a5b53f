#!/bin/bash
# round 5: do the tracker's wavefronts (4 x 126 VGPRs fill a SIMD's register file) keep the short kernels of the chain and the solves out?
# the tracking line on n of every 32 CUs only (SVO_GROUP_LK_CU_KEEP), and the chain lines at high stream priority (SVO_GROUP_CHAIN_PRIORITY)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_aa.txt
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
run "default" X=1 --
run "tracking line on 28 of every 32 CUs" SVO_GROUP_LK_CU_KEEP=28 --
run "tracking line on 24 of every 32 CUs" SVO_GROUP_LK_CU_KEEP=24 --
run "tracking line on 20 of every 32 CUs" SVO_GROUP_LK_CU_KEEP=20 --
run "chain lines at high stream priority" SVO_GROUP_CHAIN_PRIORITY=1 --
run "chain lines high priority, tracking on 24 of 32" SVO_GROUP_CHAIN_PRIORITY=1 SVO_GROUP_LK_CU_KEEP=24 --
