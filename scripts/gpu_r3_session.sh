#!/bin/bash
# round 3: full GPU test suite, the driver's bench command, the multi-GPU configurations on one GPU and as a
# two-rank rehearsal (gloo, both ranks on device 0)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r3_tests.log; if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" gpurun_out/r3_tests.log | head -20; exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/r3_bench.log 2>&1; rc=$?; tail -1 gpurun_out/r3_bench.log | cut -c1-600; if [ $rc -ne 0 ]; then tail -20 gpurun_out/r3_bench.log; exit $rc; fi
for cfg in c4 c5; do
timeout -k 10 600 python bench.py --config $cfg --steps 2 --warmup 1 > gpurun_out/r3_bench_${cfg}_n1.log 2>&1; rc=$?; tail -1 gpurun_out/r3_bench_${cfg}_n1.log | cut -c1-400; if [ $rc -ne 0 ]; then tail -20 gpurun_out/r3_bench_${cfg}_n1.log; exit $rc; fi
NEGF_BENCH_REHEARSAL=1 timeout -k 10 600 python bench.py --config $cfg --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3_bench_${cfg}_n2reh.log 2>&1; rc=$?; tail -1 gpurun_out/r3_bench_${cfg}_n2reh.log | cut -c1-400; if [ $rc -ne 0 ]; then tail -20 gpurun_out/r3_bench_${cfg}_n2reh.log; exit $rc; fi
done
