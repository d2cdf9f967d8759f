#!/bin/bash
# chain kernel session: parity tests of the chain kernels, then timing at n_c=50 for the occupancy variants
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chain1d or config_C3 or compact or wire" > gpurun_out/r2_pytest_chain.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/r2_pytest_chain.log; tail -15 gpurun_out/r2_pytest_chain.log
if [ $rc -ne 0 ]; then exit $rc; fi
for occ in 3 2; do
NEGF_CHAIN1D_OCC=$occ NEGF_CHAIN_STAMPS=1 timeout -k 10 300 python scripts/time_chain.py > gpurun_out/r2_chain_occ$occ.log 2>&1
rc=$?; echo "exit=$rc" >> gpurun_out/r2_chain_occ$occ.log; tail -4 gpurun_out/r2_chain_occ$occ.log
if [ $rc -ne 0 ]; then exit $rc; fi
done
