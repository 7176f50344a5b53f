#!/bin/bash
# tracker tail: issue priority of wavefronts that pass N iterations
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { env "$@" python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, differ %d, lk %.0f us x %d' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0'], d['roofline']['tracker_kernel']['avg_launch_us'], d['roofline']['tracker_kernel']['launches']))"; }
echo "off: $(one X=1)"
echo "after 8: $(one SVO_GROUP_LK_PRIO_AFTER=8)"
echo "after 16: $(one SVO_GROUP_LK_PRIO_AFTER=16)"
echo "after 32: $(one SVO_GROUP_LK_PRIO_AFTER=32)"
echo "off: $(one X=1)"
echo "after 4: $(one SVO_GROUP_LK_PRIO_AFTER=4)"
