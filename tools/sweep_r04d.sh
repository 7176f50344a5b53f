#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { env "$@" python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, differ %d, track %s' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0'], d['config']['launches_per_step_of_group_0']['track']))"; }
echo "express on : $(one X=1)"
echo "express off: $(one SVO_GROUP_EXPRESS=0)"
echo "express on : $(one X=1)"
echo "express off: $(one SVO_GROUP_EXPRESS=0)"
