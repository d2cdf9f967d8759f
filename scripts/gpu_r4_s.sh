#!/bin/bash
# window look-ahead on two streams: parity of the windowed inverse, then small / mid batch timings with and without
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "windowed or G_of_E or blocked_inverses or largest_window or singular_and_nan or C4 or C5 or full_size or imaginary" > gpurun_out/r4s_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r4s_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/fuzz_inverse.py 3 > gpurun_out/r4s_fuzz.log 2>&1; tail -2 gpurun_out/r4s_fuzz.log
CASES="800x2 800x12 800x36 800x61 800x108 800x200 1000x64 1000x128 500x12 500x64 500x200 300x50 2000x8"
echo "== look-ahead (default)"; timeout -k 10 300 python scripts/time_smallbatch.py $CASES 2>&1 | grep "^n=" | awk '{print $1, $2, $6, $7, $9, $10}'
echo "== NEGF_GJ_LOOKAHEAD_MAX=0 (stream groups)"; NEGF_GJ_LOOKAHEAD_MAX=0 timeout -k 10 300 python scripts/time_smallbatch.py $CASES 2>&1 | grep "^n=" | awk '{print $1, $2, $6, $7, $9, $10}'
