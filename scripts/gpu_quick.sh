#!/bin/bash
# quick check of the inverse kernel: a few parity tests, the bench line, phase stamps of workgroup 0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inverse or gr_int or C2" 2>&1 | tail -3 || exit 1
timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],3), 'inverse ms', round(d['roofline']['avg_launch_ms'],3), 'TF', round(d['roofline']['achieved'],2))" || exit 1
NEGF_GJ_STAMPS=1 timeout -k 10 120 python bench.py --steps 1 --warmup 0 --no-cpu 2>&1 | grep "gj stamps" | tail -8 | cut -c1-100
