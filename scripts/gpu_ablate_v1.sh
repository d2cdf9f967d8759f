#!/bin/bash
mkdir -p gpurun_out
for d in 0 1 2 4 6 7; do
  echo "== NEGF_GJ_DEBUG=$d" >> gpurun_out/ablate.log
  NEGF_GJ_DEBUG=$d timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ms_per_step', round(j['ms_per_step'],3), 'inverse_ms', round(j['roofline']['avg_launch_ms'],3))
    elif 'Error' in l or 'error' in l: print(l.strip())
" >> gpurun_out/ablate.log 2>&1 || exit 1
done
cat gpurun_out/ablate.log
