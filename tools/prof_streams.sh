#!/bin/bash
# kernel trace of the default bench shape at S streams: is the GPU busy all the time?
export TMPDIR=/tmp
S=${1:-8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_s$S
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --streams $S > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
python3 tools/gpu_busy.py $OUT/p_kernel_trace.csv 0.5
head -8 $OUT/p_kernel_stats.csv | cut -c1-150
