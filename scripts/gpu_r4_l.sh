#!/bin/bash
# the column-per-wave small-system kernel: tests, call latencies against the tile layout, the N = 60 step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_small_fused_gpu.py tests/test_chain_cache_gpu.py -x -q -m gpu > gpurun_out/r4l_small.log 2>&1
rc=$?; tail -5 gpurun_out/r4l_small.log; [ $rc -ne 0 ] && exit $rc
echo "== column-per-wave"; timeout -k 10 300 python scripts/time_small_calls.py 60 2>&1 | grep "GrInt \|Segments"
echo "== tile layout"; NEGF_SMALL_KERNEL=tile timeout -k 10 300 python scripts/time_small_calls.py 60 2>&1 | grep "GrInt \|Segments"
timeout -k 10 400 python bench.py --config scf --scf-systems n60 --steps 5 > gpurun_out/r4l_scf.json 2> gpurun_out/r4l_scf.err; python -c "
import json; d=json.loads(open('gpurun_out/r4l_scf.json').read().strip().splitlines()[-1])
for r in d['config']['systems']: print(r['system'], r['wall_ms_per_step'], r['kernel_ms_per_step'], r['integrals_per_step'], r.get('parity_rel_fro_P_vs_oracle_replay'))"
