#!/bin/bash
# chain kernel with the look-ahead synchronised by LDS counters: parity subset, then the C3 launch
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_chain_cache_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "chain or gcache or C3 or surfG" > gpurun_out/r4e_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r4e_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/time_chain.py 2000 --warm > gpurun_out/r4e_chain.log 2>&1; tail -2 gpurun_out/r4e_chain.log
timeout -k 10 300 python scripts/time_chain_uniform.py > gpurun_out/r4e_uniform.log 2>&1; tail -4 gpurun_out/r4e_uniform.log
