#!/bin/bash
# SQ counters of the pipeline kernels (single stream, own PMC run): instruction mix and wait fractions of lk_fb_kernel
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lk_pmc
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $OUT -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_lk_pmc"
f = glob.glob(out + "/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen: seen.add(key); calls[k] += 1
with open(out + "/summary.txt", "w") as w:
    for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"])[:8]:
        a = agg[k]; c = calls[k]
        wc = max(a["SQ_WAVE_CYCLES"], 1.0)
        line = ("%s calls=%d per launch: waves=%.0f valu=%.0f lds=%.0f salu=%.0f wave_cycles(quad)=%.0f | of wave cycles: parked (waitcnt/barrier) %.1f%%, issue-stalled %.1f%%, issuing %.1f%%"
                % (k, c, a["SQ_WAVES"] / c, a["SQ_INSTS_VALU"] / c, a["SQ_INSTS_LDS"] / c, a["SQ_INSTS_SALU"] / c, a["SQ_WAVE_CYCLES"] / c,
                   100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc))
        print(line); w.write(line + "\n")
PY
