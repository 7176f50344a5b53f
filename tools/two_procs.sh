#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# experiment: P bench processes on the same GPU at once, S streams each (is the ceiling the GPU or the per-process runtime?)
P=${1:-2}; S=${2:-4}
pids=""
for i in $(seq 1 $P); do
  timeout -k 10 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --streams $S > gpurun_out/proc_$i.log 2>&1 &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
for i in $(seq 1 $P); do tail -1 gpurun_out/proc_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['single_stream'] if 'single_stream' in d else '')"; done
