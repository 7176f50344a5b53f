#!/bin/bash
# Developer aid: kernel durations of the group kernels against the number of lanes of ONE pipeline group (rocprofv3 kernel trace).
set -euo pipefail
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/lanes_trace"; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for L in 1 4 12; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/l$L" -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-single --streams $L --groups 1 > "$OUT/bench_l$L.log" 2>&1
  python3 tools/trace_by_grid.py "$OUT/l$L/p_kernel_trace.csv" > "$OUT/by_grid_l$L.txt" || true
  python3 tools/gpu_busy.py "$OUT/l$L/p_kernel_trace.csv" 20 > "$OUT/busy_l$L.txt" || true
  rm -f "$OUT/l$L/p_kernel_trace.csv"
  grep '^{"metric"' "$OUT/bench_l$L.log" | cut -c1-200
done
