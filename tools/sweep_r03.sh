#!/bin/bash
# bench.py sweeps of the pipeline-group knobs (streams, groups, bus lines, hardware queues); one line per run
run() { # streams groups lk chain ba queues [lk overlap us]
  SVO_GROUP_LK_OVERLAP_US=${7:-0} SVO_GROUP_LK_LINES=$3 SVO_GROUP_CHAIN_LINES=$4 SVO_GROUP_BA_LINES=$5 GPU_MAX_HW_QUEUES=$6 timeout -k 10 100 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-other-workloads --no-single --streams $1 --groups $2 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sw.json")); print("streams $1 groups $2 lk $3 chain $4 ba $5 queues $6 overlap ${7:-0}:", round(d["value"]), "frames/s, host cores", d["config"].get("host_cores_busy"), {k:v for k,v in d["config"]["launches_per_step_of_group_0"].items() if k in ("track","bundle_adjust","pnp_hypotheses")})
except Exception as e: print("streams $1 groups $2 lk $3 chain $4 ba $5 queues $6: ERR", open("gpurun_out/sw.err").read()[-300:])
PY
}
for cfg in "$@"; do run $cfg; done
