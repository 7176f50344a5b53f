#!/bin/bash
# round 5, fourth sweep: is "admission overflow into the compact form + more lanes" real?  Repeats, more lanes, parity vs the CPU oracle.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_d.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  c=$(grep -o '"parity_vs_cpu": {[^}]*}' gpurun_out/r5_sweep_tmp.log | head -1 | cut -c1-260)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l $c" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
run "overflow 128 / 4 with cpu parity" SVO_BA_OVERFLOW=1 -- --streams 128 --groups 4
cp gpurun_out/r5_sweep_tmp.log gpurun_out/r5_b_overflow128.log
run "overflow 128 / 4 (2)" SVO_BA_OVERFLOW=1 -- --streams 128 --groups 4 --no-cpu-baseline
run "overflow 128 / 4 (3)" SVO_BA_OVERFLOW=1 -- --streams 128 --groups 4 --no-cpu-baseline
run "wide 128 / 4" X=1 -- --streams 128 --groups 4 --no-cpu-baseline
run "overflow 112 / 4" SVO_BA_OVERFLOW=1 -- --streams 112 --groups 4 --no-cpu-baseline
run "overflow 160 / 5" SVO_BA_OVERFLOW=1 -- --streams 160 --groups 5 --no-cpu-baseline
run "overflow 192 / 6" SVO_BA_OVERFLOW=1 -- --streams 192 --groups 6 --no-cpu-baseline
run "overflow 128 / 4, 24 hw queues" SVO_BA_OVERFLOW=1 GPU_MAX_HW_QUEUES=24 -- --streams 128 --groups 4 --no-cpu-baseline
run "overflow 128 / 4, 5 compact waves" SVO_BA_OVERFLOW=1 SVO_BA_COMPACT_WAVES=5 -- --streams 128 --groups 4 --no-cpu-baseline
run "overflow 128 / 4, budget 125 %" SVO_BA_OVERFLOW=1 SVO_BA_BUDGET_PERCENT=125 -- --streams 128 --groups 4 --no-cpu-baseline
run "overflow 64 / 2 (32-lane groups)" SVO_BA_OVERFLOW=1 -- --streams 64 --groups 2 --no-cpu-baseline
run "overflow 32 / 1" SVO_BA_OVERFLOW=1 -- --streams 32 --groups 1 --no-cpu-baseline
