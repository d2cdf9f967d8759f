#!/bin/bash
# quick GPU check: parity tests, then bench ablations (NEGF_GJ_DEBUG) per kernel variant
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/pytest.log; tail -8 gpurun_out/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
rm -f gpurun_out/ablate.log
for v in ${VARIANTS:-0}; do
for d in ${ABLATE:-0 1 3 5 7}; do
  echo "== VARIANT=$v NEGF_GJ_DEBUG=$d" >> gpurun_out/ablate.log
  NEGF_GJ_VARIANT=$v NEGF_GJ_DEBUG=$d timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ms_per_step', round(j['ms_per_step'],3), 'inverse_ms', round(j['roofline']['avg_launch_ms'],3), 'TF', round(j['roofline']['achieved'],2))
    elif 'Error' in l or 'error' in l: print(l.strip())
" >> gpurun_out/ablate.log 2>&1 || exit 1
done
done
cat gpurun_out/ablate.log
