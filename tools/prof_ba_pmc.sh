#!/bin/bash
# MFMA counters of the BA-50k workload (own run: --pmc with --kernel-trace only)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ba_pmc
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT -o ba -- python3 bench.py --workload ba50k --steps 30 --warmup 5 > $OUT/bench.log 2>&1
ls $OUT
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_ba_pmc"
f = glob.glob(out + "/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen: seen.add(key); calls[k] += 1
with open(out + "/summary.txt", "w") as w:
    for k in agg:
        line = k + " calls=%d " % calls[k] + " ".join("%s=%.0f" % (c, v / calls[k]) for c, v in sorted(agg[k].items())) + "  (per launch)"
        print(line); w.write(line + "\n")
PY
