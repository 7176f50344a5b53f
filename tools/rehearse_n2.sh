#!/bin/bash
# Rehearsal of the N>1 code paths on a ONE-GPU box: 2 ranks share GPU 0.  RCCL refuses two ranks on one device, so
# the process group is gloo and the bundle adjustment's collective goes through the callback (same library code path
# up to the all-reduce call itself); with two GPUs the default (RCCL from the library) is what runs.
set -euo pipefail
cd "$(dirname "$0")/.."
export SVO_BENCH_FORCE_DEVICE=0 SVO_BENCH_BACKEND=gloo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 --streams 4 "$@"
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload ba50k --steps 6 --warmup 2 "$@"
