#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { env "$@" python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, differ %d, ba %s' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0'], d['config']['launches_per_step_of_group_0']['bundle_adjust']))"; }
echo "launch at once, 4 lines, express off : $(one SVO_GROUP_EXPRESS=0)"
echo "wait for assembly (old), express off : $(one SVO_GROUP_EXPRESS=0 SVO_GROUP_BA_WAIT_ASSEMBLY=1)"
echo "launch at once, 6 lines, 24 queues   : $(one SVO_GROUP_EXPRESS=0 SVO_GROUP_BA_LINES=6 GPU_MAX_HW_QUEUES=24)"
echo "launch at once, 8 lines, 24 queues   : $(one SVO_GROUP_EXPRESS=0 SVO_GROUP_BA_LINES=8 GPU_MAX_HW_QUEUES=24)"
echo "launch at once, 4 lines, express on  : $(one X=1)"
echo "launch at once, 6 lines, express on, 24q: $(one SVO_GROUP_BA_LINES=6 GPU_MAX_HW_QUEUES=24)"
