#!/bin/bash
# round 5, second sweep: fused pyramid A/B, admission overflow into the compact form, bulk-path step control (run on the GPU box)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_b.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 240 python bench.py --no-cpu-baseline --no-other-workloads --no-single --no-streaming "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
run "default (wide, fused pyramid) 48 / 2" X=1 -- 
run "default, per-level pyramid launches" SVO_PYR_PER_LEVEL=1 --
run "default again" X=1 --
run "overflow into compact, 48 / 2" SVO_BA_OVERFLOW=1 --
run "overflow into compact, 64 / 2" SVO_BA_OVERFLOW=1 -- --streams 64 --groups 2
run "wide 64 / 2" X=1 -- --streams 64 --groups 2
run "overflow into compact, 96 / 3" SVO_BA_OVERFLOW=1 -- --streams 96 --groups 3
run "wide 72 / 3" X=1 -- --streams 72 --groups 3
run "overflow 72 / 3" SVO_BA_OVERFLOW=1 -- --streams 72 --groups 3
SVO_TIMING=1 timeout -k 10 200 python bench.py --workload ba50k --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5_ba4.log 2>&1
grep "device-side step control" gpurun_out/r5_ba4.log | tail -1 | tee -a $OUT
grep -o '"value": [0-9.]*' gpurun_out/r5_ba4.log | head -1 | tee -a $OUT
