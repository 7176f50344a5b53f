#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export SVO_EXP_POSE_CAP=64
one() { env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, differ %d, ba %s' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0'], d['config']['launches_per_step_of_group_0']['bundle_adjust']))"; }
echo "budget 100: $(one X=1)"
echo "budget 75: $(one SVO_EXP_BUDGET_PCT=75)"
echo "budget 50: $(one SVO_EXP_BUDGET_PCT=50)"
echo "budget 125: $(one SVO_EXP_BUDGET_PCT=125)"
echo "budget 100: $(one X=1)"
echo "budget 150: $(one SVO_EXP_BUDGET_PCT=150)"
