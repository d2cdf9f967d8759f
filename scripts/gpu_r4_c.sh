#!/bin/bash
mkdir -p gpurun_out
echo "== spin wait" > gpurun_out/r4c_calls.log
timeout -k 10 300 python scripts/time_small_calls.py 60 200 >> gpurun_out/r4c_calls.log 2>&1
echo "== blocking wait" >> gpurun_out/r4c_calls.log
NEGF_SYNC_SPIN=0 timeout -k 10 300 python scripts/time_small_calls.py 60 >> gpurun_out/r4c_calls.log 2>&1
cat gpurun_out/r4c_calls.log
