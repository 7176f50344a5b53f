#!/bin/bash
# hardware queues against the final round-4 code (two launches per keyframe outside the solve)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { env "$@" python bench.py --steps 10 --warmup 2 --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s, host cores %.1f' % (d['value'], d['config']['host_cores_busy']))"; }
for q in 16 12 8 4 16; do echo "GPU_MAX_HW_QUEUES=$q: $(one GPU_MAX_HW_QUEUES=$q)"; done
