#!/bin/bash
# round 3, end of the second session: the driver's sequence (build, pytest -m gpu, smoke, bench) + c4 / c5 lines, the
# BASELINE configs through the front-ends, small batches, then the profiles (rocprofv3 stats + PMC passes)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3b_build.log 2>&1 || { tail -20 gpurun_out/r3b_build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r3b_tests.log; if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" gpurun_out/r3b_tests.log | head -20; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3b_smoke.log 2>&1; rc=$?; tail -2 gpurun_out/r3b_smoke.log; if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/r3b_bench.log 2>&1; rc=$?; tail -1 gpurun_out/r3b_bench.log | cut -c1-300; if [ $rc -ne 0 ]; then tail -20 gpurun_out/r3b_bench.log; exit $rc; fi
for cfg in c4 c5; do
timeout -k 10 600 python bench.py --config $cfg --steps 3 --warmup 2 > gpurun_out/r3b_bench_${cfg}.log 2>&1; rc=$?; tail -1 gpurun_out/r3b_bench_${cfg}.log | cut -c1-200; if [ $rc -ne 0 ]; then tail -20 gpurun_out/r3b_bench_${cfg}.log; exit $rc; fi
done
timeout -k 10 900 python scripts/bench_configs.py > gpurun_out/r3b_configs.json 2> gpurun_out/r3b_configs.err; rc=$?; echo "configs exit=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/r3b_configs.err; exit $rc; fi
timeout -k 10 400 python scripts/time_smallbatch.py > gpurun_out/r3b_smallbatch.log 2>&1; grep "^n=" gpurun_out/r3b_smallbatch.log
bash scripts/gpu_profiles_r3.sh
