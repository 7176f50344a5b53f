#!/bin/bash
# rocprofv3 kernel-trace evidence for the default bench command and the single-stream shape (run on the GPU box).
#   $1 = tag for the output directory (default r02)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
export TMPDIR=/tmp
TAG="${1:-r02}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s8" -o p -- python3 bench.py --no-cpu-baseline > "$OUT/bench_s8.log" 2>&1
grep '^{"metric"' "$OUT/bench_s8.log" | tail -1 > "$OUT/bench_s8.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s1" -o p -- python3 bench.py --no-cpu-baseline --streams 1 > "$OUT/bench_s1.log" 2>&1
grep '^{"metric"' "$OUT/bench_s1.log" | tail -1 > "$OUT/bench_s1.json"
python3 tools/gpu_busy.py "$OUT/s8/p_kernel_trace.csv" 0.1 > "$OUT/gpu_busy_s8.txt" || true
rm -f "$OUT"/s8/p_kernel_trace.csv "$OUT"/s1/p_kernel_trace.csv  # large; the stats carry what is committed
cut -d, -f1-5 "$OUT/s8/p_kernel_stats.csv" | cut -c1-40,80- | head -24
echo ---- single stream
cut -d, -f1-5 "$OUT/s1/p_kernel_stats.csv" | cut -c1-40,80- | head -24
cut -c1-300 "$OUT/bench_s8.json"; echo; cut -c1-300 "$OUT/bench_s1.json"
