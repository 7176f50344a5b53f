#!/bin/bash
# round 5: what slowed the single-stream workloads between round 4 and the first full round-5 line?
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_sweep_k.txt
: > $OUT
one() {
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --no-cpu-baseline --no-other-workloads "$@" > gpurun_out/r5_sweep_tmp.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/r5_sweep_tmp.log | head -1)
  echo "$label rc=$rc $v" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping" | tee -a $OUT; exit 1; fi
}
one "single stream, 16 queues" GPU_MAX_HW_QUEUES=16 -- --streams 1 --groups 0 --steps 100
one "single stream, 32 queues" GPU_MAX_HW_QUEUES=32 -- --streams 1 --groups 0 --steps 100
one "single stream, 16 queues, per-process admission" GPU_MAX_HW_QUEUES=16 SVO_BA_XPROC=0 -- --streams 1 --groups 0 --steps 100
one "single stream, 16 queues, per-level pyramid" GPU_MAX_HW_QUEUES=16 SVO_PYR_PER_LEVEL=1 -- --streams 1 --groups 0 --steps 100
one "single stream, 16 queues, host-driven solves" GPU_MAX_HW_QUEUES=16 SVO_BA_DEVICE_LM=0 -- --streams 1 --groups 0 --steps 100
one "kitti_stream 1200 frames, 16 queues" GPU_MAX_HW_QUEUES=16 -- --workload kitti_stream --frames 1200
one "kitti_stream 1200 frames, 16 queues, host-driven solves" GPU_MAX_HW_QUEUES=16 SVO_BA_DEVICE_LM=0 -- --workload kitti_stream --frames 1200
one "kitti_stream 1200 frames, 32 queues, host-driven solves" GPU_MAX_HW_QUEUES=32 SVO_BA_DEVICE_LM=0 -- --workload kitti_stream --frames 1200
one "hd10k, 16 queues" GPU_MAX_HW_QUEUES=16 -- --workload hd10k
one "hd10k, 32 queues" GPU_MAX_HW_QUEUES=32 -- --workload hd10k
one "hd10k, 16 queues, per-level pyramid" GPU_MAX_HW_QUEUES=16 SVO_PYR_PER_LEVEL=1 -- --workload hd10k
one "ba50k, 16 queues (host-driven step control)" GPU_MAX_HW_QUEUES=16 -- --workload ba50k --steps 30 --warmup 5
one "ba50k, 32 queues" GPU_MAX_HW_QUEUES=32 -- --workload ba50k --steps 30 --warmup 5
