#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_chain_rr_gpu.py tests/test_chain_cache_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "chain or C3" > gpurun_out/r4dd_tests.log 2>&1 || { tail -30 gpurun_out/r4dd_tests.log; exit 1; }
tail -2 gpurun_out/r4dd_tests.log
echo "== default policy"; timeout -k 10 200 python scripts/time_chain_first.py 2>&1 | grep fresh || exit 1
