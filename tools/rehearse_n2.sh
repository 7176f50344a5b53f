#!/bin/bash
# rehearsal of the N>1 bench path on a one-GPU box: 2 ranks share GPU 0, gloo instead of RCCL
export SVO_BENCH_FORCE_DEVICE=0 SVO_BENCH_BACKEND=gloo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 --streams 4 "$@"
