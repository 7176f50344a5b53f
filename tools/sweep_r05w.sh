#!/bin/bash
# round 5: what is the chip full OF?  (a) the pause between two polls of a waiting solve workgroup (SVO_LM_POLL_SLEEPS x 128 cycles),
# NOTE: SVO_LM_POLL_SLEEPS lived in the working tree of the experiment only; the result is in profiles/r05_exp_lanes_groups_honest.txt
# (b) how busy a group's driving host thread is ("host_thread_busy_us_of_call_us" in launches_per_step_of_group_0)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_w.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  b=$(grep -o '"host_thread_busy_us_of_call_us": \[[0-9., ]*\]' gpurun_out/r5_sweep_tmp.log | head -1)
  l=$(grep -o '"avg_launch_us": [0-9.]*' gpurun_out/r5_sweep_tmp.log | tr '\n' ' ')
  echo "$label rc=$rc $v $p $h $b $l" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "default, poll pause 1 x 128 cycles (as shipped)" X=1 --
run "default, poll pause 4" SVO_LM_POLL_SLEEPS=4 --
run "default, poll pause 16" SVO_LM_POLL_SLEEPS=16 --
run "default, poll pause 0" SVO_LM_POLL_SLEEPS=0 --
run "48/2, poll pause 1" X=1 -- --streams 48 --groups 2
run "48/2, poll pause 8" SVO_LM_POLL_SLEEPS=8 -- --streams 48 --groups 2
run "32/1, poll pause 1" X=1 -- --streams 32 --groups 1
run "32/1, poll pause 8" SVO_LM_POLL_SLEEPS=8 -- --streams 32 --groups 1
