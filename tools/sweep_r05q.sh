#!/bin/bash
# round 5, after bench.py learned to fail when a group's thread dies (earlier sweeps of this round counted steps that a dead thread
# never ran: see profiles/README.md): lanes x groups x lines, gather policy, fused chain — every line with the work actually done
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_q.txt
: > $OUT
run() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  p=$(grep -o '"lane_steps_that_differ_from_step_0": [0-9]*' gpurun_out/r5_sweep_tmp.log | head -1)
  h=$(grep -o '"host_cores_busy": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  t=$(grep -o '"launches_per_step_of_group_0": {[^}]*}' gpurun_out/r5_sweep_tmp.log | head -1)
  echo "$label rc=$rc $v $p $h $t" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r5_sweep_tmp.log | cut -c1-400 | tee -a $OUT; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping the sweep" | tee -a $OUT; exit 1; fi
}
run "48/2 (round 4's shape: lines 1/2/4)" X=1 -- --streams 48 --groups 2
run "48/2 unfused chain" SVO_GROUP_CHAIN_FUSED=0 -- --streams 48 --groups 2
run "48/2 gather 100 us" SVO_GROUP_GATHER_US=100 -- --streams 48 --groups 2
run "48/2 gather 250 us" SVO_GROUP_GATHER_US=250 -- --streams 48 --groups 2
run "64/2" X=1 -- --streams 64 --groups 2
run "64/2 gather 250" SVO_GROUP_GATHER_US=250 -- --streams 64 --groups 2
run "96/3 (lines 1/1/2)" X=1 -- --streams 96 --groups 3
run "128/4 (lines 1/1/2)" X=1 -- --streams 128 --groups 4
run "128/4 gather 100" SVO_GROUP_GATHER_US=100 -- --streams 128 --groups 4
run "128/4 gather 250" SVO_GROUP_GATHER_US=250 -- --streams 128 --groups 4
run "128/4 gather 500" SVO_GROUP_GATHER_US=500 -- --streams 128 --groups 4
run "64/4 (lines 1/1/2)" X=1 -- --streams 64 --groups 4
run "96/4 (lines 1/1/2)" X=1 -- --streams 96 --groups 4
run "48/2 overflow -> compact" SVO_BA_OVERFLOW=1 -- --streams 48 --groups 2
run "64/2 overflow -> compact" SVO_BA_OVERFLOW=1 -- --streams 64 --groups 2
run "48/2 again" X=1 -- --streams 48 --groups 2
