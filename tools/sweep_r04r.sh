#!/bin/bash
# a chain line's stereo + triangulation launch BEFORE its PnP launch (the shorter one first) against after it
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --steps 10 --warmup 3 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, differ %d' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0']))"; }
echo "warm-up run: $(one)"
for r in 1 2 3; do
  echo "stereo first: $(SVO_GROUP_TRI_FIRST=1 one)"
  echo "pnp first (shipped): $(SVO_GROUP_TRI_FIRST=0 one)"
done
