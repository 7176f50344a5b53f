#!/bin/bash
# sweep streams-per-GPU (run on the GPU box); two repetitions each to see the run-to-run spread
for S in 1 2 4 6 8 12; do for rep in 1 2; do
  echo -n "S=$S "; timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --streams $S 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d.get('parity_vs_cpu'))"
done; done
