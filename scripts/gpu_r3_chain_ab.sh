#!/bin/bash
# round 3: chain kernel A/B -- wave roles by wave number (0) against roles by SIMD (1); parity subset first
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chain1d or config_C3" > gpurun_out/r3_pytest_chain.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/r3_pytest_chain.log; tail -5 gpurun_out/r3_pytest_chain.log
if [ $rc -ne 0 ]; then exit $rc; fi
for roles in 0 1 0 1; do
NEGF_CHAIN1D_ROLES=$roles NEGF_CHAIN_STAMPS=1 timeout -k 10 300 python scripts/time_chain.py 768 --warm > gpurun_out/r3_chain_roles$roles.log 2>&1
rc=$?; echo "exit=$rc" >> gpurun_out/r3_chain_roles$roles.log; echo "roles=$roles"; tail -4 gpurun_out/r3_chain_roles$roles.log
if [ $rc -ne 0 ]; then exit $rc; fi
done
