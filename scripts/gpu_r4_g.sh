#!/bin/bash
# zgemm staging change: product parity subset, then timings of both forms
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "GrLessInt or transmission or compact or gammas or spin or C5 or C2 or reference_consistency" > gpurun_out/r4g_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r4g_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/fuzz_products.py > gpurun_out/r4g_fuzz.log 2>&1; tail -3 gpurun_out/r4g_fuzz.log
echo "== default (Hermitian second product)"; timeout -k 10 300 python scripts/time_products.py 256 500 1000 2>&1 | grep "^n="
echo "== NEGF_ZGEMM_HERM=0 (A B then A B^H)"; NEGF_ZGEMM_HERM=0 timeout -k 10 300 python scripts/time_products.py 500 1000 2>&1 | grep "^n="
timeout -k 10 300 python bench.py --config c5 --steps 2 > gpurun_out/r4g_c5.json 2> gpurun_out/r4g_c5.err; python -c "
import json; d=json.loads(open('gpurun_out/r4g_c5.json').read().strip().splitlines()[-1]); print('c5 ms/step', d['ms_per_step'], d['roofline']['zgemm'], d['roofline']['family_ms_per_step'])"
