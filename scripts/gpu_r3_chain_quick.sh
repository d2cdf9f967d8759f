#!/bin/bash
# round 3: chain kernel quick check -- parity subset, then the C3 launch (2000 energies x 2 contacts) with stamps
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chain1d or config_C3" > gpurun_out/r3_pytest_chain.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/r3_pytest_chain.log; tail -3 gpurun_out/r3_pytest_chain.log
if [ $rc -ne 0 ]; then exit $rc; fi
NEGF_CHAIN_STAMPS=1 timeout -k 10 300 python scripts/time_chain.py 2000 --warm > gpurun_out/r3_chain2000.log 2>&1
rc=$?; echo "exit=$rc" >> gpurun_out/r3_chain2000.log; tail -4 gpurun_out/r3_chain2000.log
exit $rc
