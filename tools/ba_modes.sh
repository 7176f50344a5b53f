#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# developer aid: per-rank BA iteration time of an N-rank run, emulated on one GPU, for each accumulation mode
for n in 1 2 4 8; do for m in deterministic atomics mfma; do
  if [ $n = 1 ] && [ $m = deterministic ]; then continue; fi
  echo -n "shard 1/$n $m: "
  SVO_BA_SHARD_OF=$n SVO_BA_ACC=$m timeout -k 10 200 python bench.py --workload ba50k --steps 20 --warmup 2 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_us": [0-9.]*' | tr '\n' ' '; echo
done; done
