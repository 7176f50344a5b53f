#!/bin/bash
# round 5: where a lane's wall clock goes (SVO_GROUP_TRACE -> tools/lane_time.py) at 48/2 and 128/4, and whether kernels of one
# stream may overlap (tools/exp/anyorder.hip)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_r.txt
: > $OUT
export GPU_MAX_HW_QUEUES=16
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/exp_anyorder tools/exp/anyorder.hip && timeout -k 5 60 /tmp/exp_anyorder 2>&1 | tee -a $OUT
for cfg in "48 2" "128 4" "32 1"; do
  set -- $cfg
  SVO_GROUP_TRACE=1 timeout -k 10 280 python bench.py --no-other-workloads --no-single --no-streaming --no-cpu-baseline --streams $1 --groups $2 --steps 6 --warmup 2 > gpurun_out/r5_trace_$1_$2.json 2> gpurun_out/r5_trace_$1_$2.log
  rc=$?
  echo "== $1 lanes / $2 groups rc=$rc $(grep -o '"value": [0-9.]*' gpurun_out/r5_trace_$1_$2.json | head -1)" | tee -a $OUT
  python tools/lane_time.py gpurun_out/r5_trace_$1_$2.log | tee -a $OUT
  rm -f gpurun_out/r5_trace_$1_$2.log
  if [ $rc -ne 0 ]; then exit 1; fi
done
