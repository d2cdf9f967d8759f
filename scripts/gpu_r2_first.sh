#!/bin/bash
# round-2 first session: new config-size parity tests, chain stamps at n_c=50
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chain1d or config_C" > gpurun_out/r2_pytest_new.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/r2_pytest_new.log; tail -15 gpurun_out/r2_pytest_new.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
NEGF_CHAIN_STAMPS=1 timeout -k 10 300 python scripts/time_chain.py > gpurun_out/r2_chain_stamps.log 2>&1
rc=$?; echo "exit=$rc" >> gpurun_out/r2_chain_stamps.log; tail -8 gpurun_out/r2_chain_stamps.log
