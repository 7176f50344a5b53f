#!/bin/bash
# lanes and groups against the final round-4 code
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, host cores %.1f, differ %d' % (d['value'], d['config']['host_cores_busy'], d['parity_self']['lane_steps_that_differ_from_step_0']))"; }
for s in 44 48 52 56; do echo "$s lanes, 2 groups: $(one --streams $s)"; done
