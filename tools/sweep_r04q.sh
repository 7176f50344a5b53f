#!/bin/bash
# the stereo + triangulation launch's XCD-aware corner -> workgroup map (SVO_GROUP_TRI_XCD) against blockIdx = (corner, lane); the tracker's map on in both
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --steps 10 --warmup 3 --no-single --no-cpu-baseline --no-other-workloads --no-streaming "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['tracker_kernel']; print('%.0f frames/s, differ %d, lk %.0f us x %d' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0'], t['avg_launch_us'], t['launches']))"; }
echo "warm-up run: $(one)"
for r in 1 2 3; do
  echo "stereo xcd map: $(SVO_GROUP_TRI_XCD=1 one)"
  echo "stereo corner x lane grid: $(SVO_GROUP_TRI_XCD=0 one)"
done
