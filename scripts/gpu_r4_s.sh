#!/bin/bash
# round-robin chain launch: parity tests, then the C3 step with the queue off / on
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_chain_rr_gpu.py tests/test_chain_cache_gpu.py -x -q -m gpu > gpurun_out/r4s_tests.log 2>&1 || { tail -30 gpurun_out/r4s_tests.log; exit 1; }
tail -3 gpurun_out/r4s_tests.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k chain > gpurun_out/r4s_tests2.log 2>&1 || { tail -30 gpurun_out/r4s_tests2.log; exit 1; }
tail -3 gpurun_out/r4s_tests2.log
for q in 0 100 250 50; do
  NEGF_CHAIN_RR=$q timeout -k 10 200 python bench.py --no-cpu --no-extra --no-warm --steps 3 --warmup 2 > gpurun_out/r4s_bench_rr$q.json 2> gpurun_out/r4s_bench_rr$q.err || { echo "bench rr=$q failed"; tail -5 gpurun_out/r4s_bench_rr$q.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4s_bench_rr$q.json").read().strip().splitlines()[-1])
print("rr=$q ms/step", round(d["ms_per_step"],1), "first", d.get("first_evaluation_ms"), "kernel ms", d["roofline"].get("kernel_ms"), "frac_alg", d["roofline"].get("frac_algorithmic"))
PY
done
