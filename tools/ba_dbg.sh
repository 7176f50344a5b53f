#!/bin/bash
# developer aid: time the bulk linearisation kernel with parts disabled (SVO_BA_DBG bit 0: no MFMA phase, bit 1: no tile flush)
for d in ${DBGS:-0 1 2 3}; do
  echo "== SVO_BA_DBG=$d"
  SVO_BA_DBG=$d timeout -k 10 200 python bench.py --workload ba50k --steps 10 --warmup 2 2>&1 | grep -o '"avg_launch_us": [0-9.]*'
done
