set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bac
rm -rf $OUT; mkdir -p $OUT
for T in 1 8; do
  GPU_MAX_HW_QUEUES=16 SVO_BA_FUSED_REDUCE=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$T -o p -- python3 tools/ba_concurrent.py $T > $OUT/t$T.log 2>&1
  tail -1 $OUT/t$T.log | cut -c1-70
  python3 tools/trace_by_grid.py $OUT/t$T/p_kernel_trace.csv | head -5
  python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/t$T/p_kernel_trace.csv")))
qs=collections.Counter(r.get("Queue_Id","?") for r in rows)
print("queues used:", dict(qs))
PY
  rm -f $OUT/t$T/p_kernel_trace.csv
done
