#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense_hermitian or gamma or transmission or gless or C5 or C4 or explicit or zgemm or valu" > gpurun_out/r4zz_tests.log 2>&1 || { tail -30 gpurun_out/r4zz_tests.log; exit 1; }
tail -2 gpurun_out/r4zz_tests.log
timeout -k 10 300 python bench.py --config c5 --steps 2 > gpurun_out/r4zz_c5.json 2> gpurun_out/r4zz_c5.err || { tail -5 gpurun_out/r4zz_c5.err; exit 1; }
timeout -k 10 300 python scripts/time_products.py 1000 500 400 200 > gpurun_out/r4zz_prod.log 2>&1; grep "^n=" gpurun_out/r4zz_prod.log
( cd _head_ab && timeout -k 10 300 python scripts/time_products.py 1000 500 400 200 2>&1 | grep "^n=" | sed 's/^/head: /' )
python - <<PY
import json
d=json.loads(open("gpurun_out/r4zz_c5.json").read().strip().splitlines()[-1])
print("c5 ms/step", d["ms_per_step"], d["roofline"])
PY
( cd _head_ab && timeout -k 10 300 python bench.py --config c5 --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head c5 ms/step', d['ms_per_step'])" )
