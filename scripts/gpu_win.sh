#!/bin/bash
# windowed inverse: parity tests, phase stamps of the window kernel at n = 500, timing at a few sizes
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "G_of_E or largest or singular or C3 or C4 or C5 or consistency" > gpurun_out/r2b_win_pytest.log 2>&1 || { tail -20 gpurun_out/r2b_win_pytest.log; exit 1; }
tail -2 gpurun_out/r2b_win_pytest.log
NEGF_GJ_STAMPS=1 timeout -k 10 300 python scripts/time_midsize.py 500 > gpurun_out/r2b_winstamps.log 2>&1; grep "window stamps" gpurun_out/r2b_winstamps.log | tail -2
timeout -k 10 300 python scripts/time_midsize.py ${SIZES:-300 400 500 800 1000 2000} > gpurun_out/r2b_win_time.log 2>&1 || { tail gpurun_out/r2b_win_time.log; exit 1; }
grep "^n=" gpurun_out/r2b_win_time.log
