#!/bin/bash
# single stream: width of the host-driven deterministic kernels, and the device LM alone
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
one() { env "$@" python bench.py --steps 20 --warmup 3 --streams 1 --groups 0 --no-cpu-baseline --no-other-workloads --no-streaming 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s' % d['value'])"; }
echo "128 threads: $(one X=1)"
echo "64 threads: $(one SVO_BA_DET_THREADS=64)"
echo "128 threads again: $(one X=1)"
echo "64 threads again: $(one SVO_BA_DET_THREADS=64)"
echo "device LM: $(one SVO_BA_DEVICE_LM=1)"
SVO_TIMING=1 SVO_BA_DET_THREADS=64 python bench.py --steps 5 --warmup 2 --streams 1 --groups 0 --no-cpu-baseline --no-other-workloads --no-streaming 2>&1 | grep -i "ba\b\|ba:\|linear\|iter" | tail -12
