#!/bin/bash
# Developer experiment: the pipeline groups' tracker launches restricted to K of every 32 CUs (SVO_GROUP_LK_CU_KEEP)
for K in "$@"; do
  SVO_GROUP_LK_CU_KEEP=$K timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --no-single 2>/dev/null > gpurun_out/keep.json
  python - "$K" <<'PY'
import json, sys
d = json.load(open("gpurun_out/keep.json"))
print("LK on", sys.argv[1], "of 32 CUs:", round(d["value"]), "frames/s, host cores", d["config"]["host_cores_busy"], ", LK launch", round(d["roofline"]["dominant_kernel"]["avg_launch_us"]), "us")
PY
done
