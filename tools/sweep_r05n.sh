#!/bin/bash
# round 5: the default line without the overflow (wide solves only), repeated; budget, lanes, lines
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_n.txt
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
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
for i in 1 2 3 4 5 6; do run "default = 128/4, lines 1/1/2, 16 queues, wide solves only ($i)" X=1 -- ; done
for i in 1 2; do run "default, 40 steps ($i)" X=1 -- --steps 40; done
for i in 1 2; do run "default, budget 150 % ($i)" SVO_BA_BUDGET_PERCENT=150 -- ; done
for i in 1 2; do run "160 / 5, lines 1/1/1 ($i)" SVO_GROUP_CHAIN_LINES=1 SVO_GROUP_BA_LINES=1 -- --streams 160 --groups 5; done
for i in 1 2; do run "128 / 4, lines 1/2/1 ($i)" SVO_GROUP_CHAIN_LINES=2 SVO_GROUP_BA_LINES=1 -- ; done
for i in 1 2; do run "96 / 3, lines 1/1/2 ($i)" X=1 -- --streams 96 --groups 3; done
for i in 1 2; do run "96 / 4 = 24 per group ($i)" X=1 -- --streams 96 --groups 4; done
for i in 1 2; do run "64 / 4 = 16 per group ($i)" X=1 -- --streams 64 --groups 4; done
