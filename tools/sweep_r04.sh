#!/bin/bash
# Developer aid (round 4): the default bench line against group shape / bus lines / batch size.  Run on the GPU box.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # label, env..., -- args
  local label="$1"; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local v
  v=$(env "${envs[@]}" python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, %.2f host cores, differ %d' % (d['value'], d['config']['host_cores_busy'], d['parity_self']['lane_steps_that_differ_from_step_0']))")
  echo "$label: $v"
}
run "default (48 lanes, 2 groups, batch 16)" X=1 --
run "lk lines 2" SVO_GROUP_LK_LINES=2 --
run "chain lines 4" SVO_GROUP_CHAIN_LINES=4 --
run "lk 2 + chain 4" SVO_GROUP_LK_LINES=2 SVO_GROUP_CHAIN_LINES=4 --
run "3 groups of 16" X=1 -- --groups 3
run "4 groups of 12, 24 queues" GPU_MAX_HW_QUEUES=24 -- --groups 4
run "batch 32" X=1 -- --batch 32
run "batch 32, 3 groups" X=1 -- --batch 32 --groups 3
run "64 lanes, 2 groups" X=1 -- --streams 64
run "64 lanes, 4 groups, 24 queues" GPU_MAX_HW_QUEUES=24 -- --streams 64 --groups 4
run "ba lines 6" SVO_GROUP_BA_LINES=6 --
