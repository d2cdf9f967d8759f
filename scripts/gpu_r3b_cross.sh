#!/bin/bash
# round 3 (second session): where the single-workgroup inverse and the windowed inverse cross today
mkdir -p gpurun_out
for lm in 257 129; do
echo "== NEGF_GJ_LARGE_MIN=$lm" 
NEGF_GJ_LARGE_MIN=$lm timeout -k 10 200 python scripts/time_midsize.py 160 200 224 256 2>&1 | grep "^n="
done > gpurun_out/r3b_cross.log 2>&1
cat gpurun_out/r3b_cross.log
SIZES="500" bash scripts/gpu_kstats.sh > gpurun_out/r3b_kstats500.log 2>&1; cat gpurun_out/r3b_kstats500.log
