#!/bin/bash
# round 5: lanes x solve form x BA lines (run on the GPU box; writes gpurun_out/r5_sweep_a.txt)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_a.txt
: > $OUT
run() {  # label, env..., -- bench args
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
run "compact 48 lanes / 2 groups, 8 BA lines" SVO_GROUP_BA_LINES=8 -- --streams 48 --groups 2 --steps 10
run "compact 64 / 2, 8 lines" SVO_GROUP_BA_LINES=8 -- --streams 64 --groups 2 --steps 10
run "compact 96 / 3, 8 lines" SVO_GROUP_BA_LINES=8 -- --streams 96 --groups 3 --steps 10
run "compact 128 / 4, 8 lines" SVO_GROUP_BA_LINES=8 -- --streams 128 --groups 4 --steps 10
run "compact 128 / 4, 4 lines" SVO_GROUP_BA_LINES=4 -- --streams 128 --groups 4 --steps 10
run "compact 96 / 3, 8 lines, 3 waves" SVO_GROUP_BA_LINES=8 SVO_BA_COMPACT_WAVES=3 -- --streams 96 --groups 3 --steps 10
run "compact 128 / 4, 8 lines, 24 hw queues" SVO_GROUP_BA_LINES=8 GPU_MAX_HW_QUEUES=24 -- --streams 128 --groups 4 --steps 10
run "wide 64 / 2" SVO_BA_FORM=wide -- --streams 64 --groups 2 --steps 10
run "wide 96 / 3" SVO_BA_FORM=wide -- --streams 96 --groups 3 --steps 10
