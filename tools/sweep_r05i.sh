#!/bin/bash
# round 5, ninth sweep: solve lines in their own priority class (= their own pool of hardware queues)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_i.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env SVO_BA_OVERFLOW=1 "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
for i in 1 2; do run "128/4, BA lines low priority, 16 queues per class ($i)" SVO_GROUP_BA_PRIORITY=low -- --streams 128 --groups 4; done
for i in 1 2; do run "128/4, BA lines high priority ($i)" SVO_GROUP_BA_PRIORITY=high -- --streams 128 --groups 4; done
for i in 1 2; do run "128/4, BA low, 12 queues per class ($i)" SVO_GROUP_BA_PRIORITY=low GPU_MAX_HW_QUEUES=12 -- --streams 128 --groups 4; done
for i in 1 2; do run "128/4, BA low, 2 BA lines, 12 queues ($i)" SVO_GROUP_BA_PRIORITY=low SVO_GROUP_BA_LINES=2 GPU_MAX_HW_QUEUES=12 -- --streams 128 --groups 4; done
for i in 1 2; do run "128/4, BA low, wide only (no overflow) ($i)" SVO_GROUP_BA_PRIORITY=low SVO_BA_OVERFLOW=0 -- --streams 128 --groups 4; done
for i in 1 2; do run "96/3, BA low ($i)" SVO_GROUP_BA_PRIORITY=low -- --streams 96 --groups 3; done
for i in 1 2; do run "48/2, BA low ($i)" SVO_GROUP_BA_PRIORITY=low -- ; done
for i in 1 2; do run "160/5, BA low ($i)" SVO_GROUP_BA_PRIORITY=low -- --streams 160 --groups 5; done
