#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { env "$@" python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, differ %d, ba %s' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0'], d['config']['launches_per_step_of_group_0']['bundle_adjust']))"; }
echo "4 ba lines: $(one X=1)"
echo "5 ba lines: $(one SVO_GROUP_BA_LINES=5)"
echo "6 ba lines: $(one SVO_GROUP_BA_LINES=6)"
echo "3 ba lines: $(one SVO_GROUP_BA_LINES=3)"
echo "4 ba lines, 3 chain lines: $(one SVO_GROUP_CHAIN_LINES=3)"
echo "4 ba lines, 12 queues: $(one GPU_MAX_HW_QUEUES=12)"
echo "4 ba lines, 20 queues: $(one GPU_MAX_HW_QUEUES=20)"
