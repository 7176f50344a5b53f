run() { # streams groups lk chain ba queues
  SVO_GROUP_LK_LINES=$3 SVO_GROUP_CHAIN_LINES=$4 SVO_GROUP_BA_LINES=$5 GPU_MAX_HW_QUEUES=$6 timeout -k 10 60 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --streams $1 --groups $2 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sw.json")); print("s$1 g$2 lk$3 ch$4 ba$5 q$6:", round(d["value"]), "cores", d["config"].get("host_cores_busy"), {k:v for k,v in d["config"]["launches_per_step"].items() if k in ("track","bundle_adjust","pnp_hypotheses")})
except Exception as e: print("s$1 g$2 lk$3 ch$4 ba$5 q$6: ERR", open("gpurun_out/sw.err").read()[-300:])
PY
}
run 8 1 1 1 2 4
run 8 1 1 1 4 8
run 8 1 2 2 4 8
run 8 1 2 4 8 16
run 8 1 1 2 8 16
run 8 2 1 1 2 8
run 8 2 1 2 4 16
run 16 1 2 4 8 16
run 16 2 2 2 4 16
