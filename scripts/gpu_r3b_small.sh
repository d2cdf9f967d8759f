#!/bin/bash
# round 3 (second session): the windowed path below n = 257 with the lean window kernel (4 + 2 waves, two per CU)
mkdir -p gpurun_out
NEGF_GJ_LARGE_MIN=65 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "G_of_E_per_energy or windowed_inverse or singular_and_nan or GrInt_GrLessInt_const or full_size_properties_C2" > gpurun_out/r3b_small_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r3b_small_tests.log; if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" gpurun_out/r3b_small_tests.log | head -20; exit $rc; fi
for lm in 257 129; do
echo "== NEGF_GJ_LARGE_MIN=$lm"
NEGF_GJ_LARGE_MIN=$lm timeout -k 10 200 python scripts/time_midsize.py 130 160 200 224 256 2>&1 | grep "^n="
done > gpurun_out/r3b_small_time.log 2>&1
cat gpurun_out/r3b_small_time.log
