#!/bin/bash
# round 3 (second session): window pairs (gj_colupdate2*_kernel) -- parity subset, then A/B timing
# NEGF_GJ_PAIR: 0 windows one by one, 1 pairs with two 4-wave workgroups per CU, 2 pairs with one 8-wave workgroup
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "G_of_E_per_energy or windowed_inverse or largest_window or singular_and_nan or config_C4 or config_C5 or spin_block_diagonal" > gpurun_out/r3b_pair_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r3b_pair_tests.log; if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" gpurun_out/r3b_pair_tests.log | head -20; exit $rc; fi
for pr in ${PAIRS:-0 2 1}; do
echo "== NEGF_GJ_PAIR=$pr"
NEGF_GJ_PAIR=$pr timeout -k 10 300 python scripts/time_midsize.py ${SIZES:-300 500 800 1000 2000} 2>&1 | grep "^n="
done > gpurun_out/r3b_pair_time.log 2>&1
cat gpurun_out/r3b_pair_time.log
