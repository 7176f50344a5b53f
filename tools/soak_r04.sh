#!/bin/bash
# soak of the final round-4 code: 500 steps of the default line, 200 steps at 64 lanes (admission under pressure); every lane of every step against step 0
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['parity_self']; print('%.0f frames/s, %d lanes x %d steps checked, %d differ' % (d['value'], p['lanes_checked'], p['steps_checked'], p['lane_steps_that_differ_from_step_0']))"; }
echo "48 lanes, 500 steps: $(one --steps 500 --warmup 3)"
echo "64 lanes, 200 steps: $(one --steps 200 --warmup 3 --streams 64)"
