#!/bin/bash
# round 5: the full default line (single stream, streaming, other workloads in the same process) against hardware-queue counts
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_l.txt
: > $OUT
full() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 420 python bench.py --no-cpu-baseline "$@" > gpurun_out/r5_full_tmp.log 2>&1
  rc=$?
  python3 - "$label" $rc >> $OUT <<'PY'
import json, sys
label, rc = sys.argv[1], sys.argv[2]
b = None
for ln in open("gpurun_out/r5_full_tmp.log"):
    if ln.startswith('{"metric"'):
        b = json.loads(ln)
if b is None:
    print(label, "rc", rc, "no line")
else:
    ow = b.get("other_workloads", {})
    print("%-44s rc %s  value %6.0f  single %5.0f  streaming %6.0f | kitti_stream %4.0f  ba50k %5.0f / %5.0f  hd10k %4.0f" % (
        label, rc, b["value"], (b.get("single_stream") or {}).get("value", 0), (b.get("streaming") or {}).get("value", 0),
        ow.get("kitti_stream_4541_frames", {}).get("value", 0), ow.get("ba50k_sparse", {}).get("value", 0), ow.get("ba50k_dense", {}).get("value", 0),
        ow.get("hd_1280x720_10k", {}).get("value", 0)))
PY
  tail -1 $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping" | tee -a $OUT; exit 1; fi
}
full "128/4, 32 queues (default)" X=1 --
full "128/4, 16 queues" GPU_MAX_HW_QUEUES=16 --
full "128/4, 24 queues, lines 1/2/3" GPU_MAX_HW_QUEUES=24 SVO_GROUP_BA_LINES=3 --
full "128/4, 16 queues, lines 1/1/2" GPU_MAX_HW_QUEUES=16 SVO_GROUP_CHAIN_LINES=1 SVO_GROUP_BA_LINES=2 --
full "48/2, 16 queues (round 4's shape)" GPU_MAX_HW_QUEUES=16 -- --streams 48 --groups 2
