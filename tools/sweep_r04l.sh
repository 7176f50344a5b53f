#!/bin/bash
# where the lanes' time goes at 48 and at 64 lanes (SVO_GROUP_TRACE), final round-4 code
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/trace
for s in 48 64; do
  SVO_GROUP_TRACE=1 python bench.py --steps 4 --warmup 2 --streams $s --no-single --no-cpu-baseline --no-other-workloads --no-streaming 2> gpurun_out/trace/t$s.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s lanes: %.0f frames/s' % d['value'])"
  python tools/trace_gaps.py gpurun_out/trace/t$s.txt > gpurun_out/trace/gaps$s.txt
  head -30 gpurun_out/trace/gaps$s.txt
  grep -v "^\[svo group\]" gpurun_out/trace/t$s.txt | tail -5
  gzip -f gpurun_out/trace/t$s.txt
done
