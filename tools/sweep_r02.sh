#!/bin/bash
# developer aid: aggregate throughput against the stream count (host threads = 2 per stream, CPU quota 16 cores per GPU)
set -uo pipefail
cd "$(dirname "$0")/.."
for s in 1 4 8 10 12 16; do
  echo -n "streams=$s: "
  timeout -k 10 200 python bench.py --no-cpu-baseline --streams $s --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
