#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4ee_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4ee_tests.log
timeout -k 10 600 python bench.py --config scf --scf-systems n60,n200 --steps 3 > gpurun_out/r4ee_scf.json 2> gpurun_out/r4ee_scf.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4ee_scf.json").read().strip().splitlines()[-1])
print("; ".join("%s wall %.1f kernel %.1f parity %s" % (s["system"], s["wall_ms_per_step"], s["kernel_ms_per_step"], s.get("parity_rel_fro_P_vs_oracle_replay")) for s in d["config"]["systems"]))
PY
