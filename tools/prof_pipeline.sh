#!/bin/bash
# rocprofv3 evidence for the default bench command (run on the GPU box):
#   1. kernel-trace stats of `python3 bench.py`               -> gpurun_out/prof_pipe/stats
#   2. PMC pass FETCH_SIZE, 3. PMC pass WRITE_SIZE (own runs) -> per-kernel HBM KB per launch (traffic.json)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pipe
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 bench.py > $OUT/bench_stats.log 2>&1 || exit 1
grep '^{"metric"' $OUT/bench_stats.log | tail -1 > $OUT/bench_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/bench_$C.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, json, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_pipe"
res = collections.defaultdict(dict)
for c, key in (("FETCH_SIZE", "fetch_kb_per_launch"), ("WRITE_SIZE", "write_kb_per_launch")):
    f = glob.glob(out + "/" + c + "/*counter_collection.csv")[0]
    tot = collections.Counter(); calls = collections.Counter()
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != c: continue
        k = row["Kernel_Name"].split("(")[0]
        tot[k] += float(row["Counter_Value"]); calls[k] += 1
    for k in tot: res[k][key] = tot[k] / calls[k]
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print({k: v for k, v in res.items() if k in ("lk_fb_kernel", "corner_response_kernel")})
PY
head -12 $OUT/stats/p_kernel_stats.csv | cut -c1-60,140-240
