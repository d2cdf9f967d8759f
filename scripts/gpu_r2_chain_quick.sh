#!/bin/bash
# quick chain timing at n_c=50 (no pytest): occupancy variants with in-kernel stamps
mkdir -p gpurun_out
for occ in ${OCCS:-3 2}; do
NEGF_CHAIN1D_OCC=$occ NEGF_CHAIN_STAMPS=1 timeout -k 10 300 python scripts/time_chain.py > gpurun_out/r2_chain_occ$occ.log 2>&1
rc=$?; echo "exit=$rc" >> gpurun_out/r2_chain_occ$occ.log; tail -4 gpurun_out/r2_chain_occ$occ.log
if [ $rc -ne 0 ]; then exit $rc; fi
done
