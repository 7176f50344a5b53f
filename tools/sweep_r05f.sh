#!/bin/bash
# round 5, sixth sweep: 128 lanes / 4 groups with the admission overflow — what makes the figure vary (22-36 k frames/s)?
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_f.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env SVO_BA_OVERFLOW=1 "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline --streams 128 --groups 4 "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $l" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
}
for i in 1 2; do run "no stagger ($i)" X=1 -- ; done
for i in 1 2 3; do run "stagger 15 ms ($i)" X=1 -- --stagger-ms 15; done
for i in 1 2; do run "stagger 30 ms ($i)" X=1 -- --stagger-ms 30; done
for i in 1 2; do run "stagger 15, 2 BA lines ($i)" SVO_GROUP_BA_LINES=2 -- --stagger-ms 15; done
for i in 1 2; do run "stagger 15, 6 BA lines ($i)" SVO_GROUP_BA_LINES=6 -- --stagger-ms 15; done
run "stagger 15, 1 chain line" SVO_GROUP_CHAIN_LINES=1 -- --stagger-ms 15
run "stagger 15, 3 chain lines" SVO_GROUP_CHAIN_LINES=3 -- --stagger-ms 15
run "stagger 15, 12 hw queues" GPU_MAX_HW_QUEUES=12 -- --stagger-ms 15
run "stagger 15, 20 hw queues" GPU_MAX_HW_QUEUES=20 -- --stagger-ms 15
run "stagger 15, budget 150 %" SVO_BA_BUDGET_PERCENT=150 -- --stagger-ms 15
run "stagger 15, 40 steps" X=1 -- --stagger-ms 15 --steps 40
run "stagger 15, batch 32" X=1 -- --stagger-ms 15 --batch 32 --steps 10
