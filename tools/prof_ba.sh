#!/bin/bash
# rocprofv3 kernel stats of the BA-50k workload (developer aid; summaries worth keeping are copied to profiles/)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ba
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ba -- python3 bench.py --workload ba50k --steps 30 --warmup 5 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
