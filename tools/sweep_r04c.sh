#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, differ %d, ba %s' % (d['value'], d['parity_self']['lane_steps_that_differ_from_step_0'], d['config']['launches_per_step_of_group_0']['bundle_adjust']))"; }
cp stereo_vo_amd/libsvo_hip.so /tmp/default.so
echo "default (2 waves/SIMD budget, LK LDS trimmed): $(one)"
echo "default again: $(one)"
for v in 3 4; do cp build/variants/libsvo_w$v.so stereo_vo_amd/libsvo_hip.so; echo "solve kernel at $v waves/SIMD: $(one)"; done
cp /tmp/default.so stereo_vo_amd/libsvo_hip.so
