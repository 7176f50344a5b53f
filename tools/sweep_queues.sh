#!/bin/bash
# sweep the number of HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) at S streams per GPU
for Q in 4 8 16 32; do for S in 8 12; do
  echo -n "Q=$Q S=$S "; GPU_MAX_HW_QUEUES=$Q timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --streams $S 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done; done
