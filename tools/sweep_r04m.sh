#!/bin/bash
# lanes against the admission budget of the device-resident solves (SVO_BA_BUDGET_PERCENT), final round-4 code
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --steps 8 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, host cores %.1f, differ %d' % (d['value'], d['config']['host_cores_busy'], d['parity_self']['lane_steps_that_differ_from_step_0']))"; }
for s in ${SWEEP_LANES:-48 56 64}; do
  for b in ${SWEEP_BUDGETS:-100 150}; do echo "$s lanes, budget $b %: $(SVO_BA_BUDGET_PERCENT=$b one --streams $s)"; done
done
