#!/bin/bash
# Rehearsal of the N>1 code paths on a ONE-GPU box, THROUGH bench.py's own spawner (`bench.py --gpus 2`: HIP-free GPU count in
# the parent, torch.distributed.run started from a process that never mapped libamdhip64).  The count is stubbed
# (SVO_BENCH_GPU_COUNT=2: the box has one GPU) and both ranks are mapped onto GPU 0 (SVO_BENCH_FORCE_DEVICE=0).  RCCL refuses
# two ranks on one device, so the process group is gloo and the bundle adjustment's collective goes through the callback
# (same library code path up to the all-reduce call itself); with two GPUs the default (RCCL from the library) is what runs.
set -euo pipefail
cd "$(dirname "$0")/.."
export SVO_BENCH_GPU_COUNT=2 SVO_BENCH_FORCE_DEVICE=0 SVO_BENCH_BACKEND=gloo
timeout -k 10 500 python bench.py --gpus 2 --steps 4 --warmup 1 --streams 4 --no-other-workloads "$@"
timeout -k 10 500 python bench.py --gpus 2 --workload ba50k --steps 6 --warmup 2 "$@"
