#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense_hermitian or gamma or transmission or gless or C5 or C4 or explicit" > gpurun_out/r4zz_tests.log 2>&1 || { tail -30 gpurun_out/r4zz_tests.log; exit 1; }
tail -2 gpurun_out/r4zz_tests.log
for g in 0 1; do
  NEGF_ZGEMM_GT=$g timeout -k 10 300 python bench.py --config c5 --steps 2 > gpurun_out/r4zz_c5_gt$g.json 2> gpurun_out/r4zz_c5_gt$g.err || { tail -5 gpurun_out/r4zz_c5_gt$g.err; exit 1; }
  NEGF_ZGEMM_GT=$g timeout -k 10 300 python scripts/time_products.py 1000 500 400 > gpurun_out/r4zz_prod_gt$g.log 2>&1; grep "^n=" gpurun_out/r4zz_prod_gt$g.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4zz_c5_gt$g.json").read().strip().splitlines()[-1])
print("gt=$g c5 ms/step", d["ms_per_step"], d["roofline"].get("other_ms_per_step"), d["roofline"].get("avg_launch_ms"))
PY
done
