set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_s8x
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s8 -o p -- python3 bench.py --no-cpu-baseline > $OUT/bench_s8.log 2>&1
python3 tools/trace_by_grid.py $OUT/s8/p_kernel_trace.csv > $OUT/by_grid_s8.txt || true
python3 tools/gpu_busy.py $OUT/s8/p_kernel_trace.csv 100 > $OUT/gpu_busy_s8.txt || true
rm -f $OUT/s8/p_kernel_trace.csv
grep '^{"metric"' $OUT/bench_s8.log | tail -1 | cut -c1-200
