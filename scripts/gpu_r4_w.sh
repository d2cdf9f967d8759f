#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_chain_rr_gpu.py tests/test_chain_cache_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "chain or C3" > gpurun_out/r4w_tests.log 2>&1 || { tail -30 gpurun_out/r4w_tests.log; exit 1; }
tail -2 gpurun_out/r4w_tests.log
timeout -k 10 600 python bench.py --config scf --scf-systems chain --steps 3 > gpurun_out/r4w_scf.json 2> gpurun_out/r4w_scf.err || { tail -5 gpurun_out/r4w_scf.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4w_scf.json").read().strip().splitlines()[-1])
for s in d["config"]["systems"]:
    print(s["system"], "wall %.1f kernel %.1f integrals %.0f points %.0f" % (s["wall_ms_per_step"], s["kernel_ms_per_step"], s["integrals_per_step"], s["energy_points_per_step"]))
PY
