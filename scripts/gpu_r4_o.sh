#!/bin/bash
# interpolating order predictor: order tests, the moved-grid timing, then headline trace + PMC on the final chain source,
# then the whole gpu suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_chain_cache_gpu.py -x -q -m gpu -k "chain or C3 or gcache" > gpurun_out/r4o_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r4o_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/time_chain_order.py > gpurun_out/r4o_order.log 2>&1; grep "grid" gpurun_out/r4o_order.log
ONLY_HEAD=1 bash scripts/gpu_profiles_r4.sh || exit 1
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4o_alltests.log 2>&1; tail -5 gpurun_out/r4o_alltests.log
