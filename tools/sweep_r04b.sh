#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  local label="$1"; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local v
  v=$(env "${envs[@]}" python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, %.2f host cores, differ %d, ba launches/lanes %s' % (d['value'], d['config']['host_cores_busy'], d['parity_self']['lane_steps_that_differ_from_step_0'], d['config']['launches_per_step_of_group_0']['bundle_adjust']))")
  echo "$label: $v"
}
run "default" X=1 --
run "default again" X=1 --
run "ba lines 2" SVO_GROUP_BA_LINES=2 --
run "ba lines 3" SVO_GROUP_BA_LINES=3 --
run "ba lines 1" SVO_GROUP_BA_LINES=1 --
