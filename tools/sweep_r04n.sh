#!/bin/bash
# stereo + triangulation launches of a chain line on their own stream (beside the PnP launch that departs with them) against behind it
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --steps 10 --warmup 3 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, host cores %.1f, differ %d' % (d['value'], d['config']['host_cores_busy'], d['parity_self']['lane_steps_that_differ_from_step_0']))"; }
echo "warm-up run: $(one)"
for r in 1 2; do
  echo "own stream: $(SVO_GROUP_TRI_STREAM=1 one)"
  echo "same stream: $(SVO_GROUP_TRI_STREAM=0 one)"
done
echo "own stream, 20 queues: $(GPU_MAX_HW_QUEUES=20 SVO_GROUP_TRI_STREAM=1 one)"
