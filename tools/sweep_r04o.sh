#!/bin/bash
# keyframe-chain lines per group (SVO_GROUP_CHAIN_LINES; shipped: 2), final round-4 code
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --steps 10 --warmup 3 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, host cores %.1f, differ %d' % (d['value'], d['config']['host_cores_busy'], d['parity_self']['lane_steps_that_differ_from_step_0']))"; }
echo "warm-up run: $(one)"
for r in 1 2; do
  echo "chain lines 1: $(SVO_GROUP_CHAIN_LINES=1 one)"
  echo "chain lines 2: $(SVO_GROUP_CHAIN_LINES=2 one)"
done
echo "chain lines 3: $(SVO_GROUP_CHAIN_LINES=3 one)"
