#!/bin/bash
# rocprofv3 evidence of a round (run on the GPU box, writes gpurun_out/prof_<tag>/; tools/prof_summary.py then copies
# what is worth committing into profiles/).  Counter passes are their own runs with --kernel-trace only.
#   $1 = tag (default r04)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
export TMPDIR=/tmp
TAG="${1:-r05}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
# every file of this round comes from ONE tree: its identity goes into the output directory (the GPU box has no .git: the
# caller passes the commit, tools/prof_round.sh <tag> <commit>), together with the library's own build stamp
echo "commit ${2:-unknown}; libsvo_hip.so sha256 $(sha256sum stereo_vo_amd/libsvo_hip.so | cut -c1-16); $(date -u +%Y-%m-%dT%H:%MZ)" > "$OUT/STAMP.txt"
echo "[1/5] kernel trace, default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s8" -o p -- python3 bench.py --no-cpu-baseline --no-other-workloads --no-single --no-streaming > "$OUT/bench_s8.log" 2>&1
grep '^{"metric"' "$OUT/bench_s8.log" | tail -1 > "$OUT/bench_s8.json"
python3 tools/gpu_busy.py "$OUT/s8/p_kernel_trace.csv" 100 > "$OUT/gpu_busy_s8.txt" || true
python3 tools/trace_by_grid.py "$OUT/s8/p_kernel_trace.csv" > "$OUT/by_grid_s8.txt" || true
rm -f "$OUT/s8/p_kernel_trace.csv"
echo "[2/5] kernel trace, single stream"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s1" -o p -- python3 bench.py --no-cpu-baseline --no-other-workloads --streams 1 --groups 0 > "$OUT/bench_s1.log" 2>&1
grep '^{"metric"' "$OUT/bench_s1.log" | tail -1 > "$OUT/bench_s1.json"
python3 tools/trace_by_grid.py "$OUT/s1/p_kernel_trace.csv" > "$OUT/by_grid_s1.txt" || true
rm -f "$OUT/s1/p_kernel_trace.csv"
echo "[3/5] SQ counters, one pipeline group of 32 streams"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d "$OUT/sq" -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-single --no-streaming --streams 32 --groups 1 > "$OUT/bench_sq.log" 2>&1
rm -f "$OUT/sq/p_kernel_trace.csv"
echo "[4/5] FETCH_SIZE / WRITE_SIZE, one pipeline group of 32 streams"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-single --no-streaming --streams 32 --groups 1 > "$OUT/bench_$C.log" 2>&1
  rm -f "$OUT/$C/p_kernel_trace.csv"
done
echo "[5/5] kernel trace, ba50k"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ba" -o p -- python3 bench.py --workload ba50k --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_ba.log" 2>&1
grep '^{"metric"' "$OUT/bench_ba.log" | tail -1 > "$OUT/bench_ba.json"
rm -f "$OUT/ba/p_kernel_trace.csv"
echo "[6/6] f64 MFMA counters, ba50k"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$OUT/ba_pmc" -o p -- python3 bench.py --workload ba50k --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_ba_pmc.log" 2>&1 || true
rm -f "$OUT/ba_pmc/p_kernel_trace.csv"
python3 tools/prof_summary.py "$OUT" "$OUT/summary" "$TAG"
