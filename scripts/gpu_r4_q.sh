#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_small_fused_gpu.py tests/test_fermi_search.py -x -q -m gpu > gpurun_out/r4q_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r4q_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4q_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4q_smoke.log
timeout -k 10 600 python bench.py --config scf --scf-systems n60,n200,n800 --steps 3 > gpurun_out/r4q_scf.json 2> gpurun_out/r4q_scf.err; python -c "
import json; d=json.loads(open('gpurun_out/r4q_scf.json').read().strip().splitlines()[-1])
for r in d['config']['systems']: print(r['system'], round(r['wall_ms_per_step'],1), round(r['kernel_ms_per_step'],1), r['integrals_per_step'], r.get('parity_rel_fro_P_vs_oracle_replay'))"
