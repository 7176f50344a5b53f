#!/bin/bash
# single stream: width of the host-driven deterministic kernels
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { env "$@" python bench.py --steps 20 --warmup 3 --streams 1 --groups 0 --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s' % d['value'])"; }
echo "128 threads: $(one X=1)"
echo "256 threads: $(one SVO_BA_DET_THREADS=256)"
echo "128 threads again: $(one X=1)"
echo "256 threads again: $(one SVO_BA_DET_THREADS=256)"
SVO_TIMING=1 SVO_BA_DET_THREADS=256 python bench.py --steps 5 --warmup 2 --streams 1 --groups 0 --no-cpu-baseline --no-other-workloads --no-streaming 2>&1 | grep "svo ba" | tail -3
