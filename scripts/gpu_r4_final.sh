#!/bin/bash
# end of round 4: the driver's sequence (build check, pytest -m gpu, smoke, bench) and the lines copied into profiles/
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r4z_build.log 2>&1; echo "build rc=$?"; tail -1 gpurun_out/r4z_build.log
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4z_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4z_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4z_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4z_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err; echo "bench rc=$?"
timeout -k 10 900 python bench.py --config scf --steps 3 > gpurun_out/r4z_scf.json 2> gpurun_out/r4z_scf.err; echo "scf rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4z_bench.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("bench ms/step %.1f first %.0f alg %.3f exec %.3f busy %s traffic %s warm %.1f" % (d["ms_per_step"], d["config"]["first_evaluation_ms"], r["frac_algorithmic"], r["frac"], r["mfma_busy_frac"], r["traffic"], d["warm"]["warm_ms_per_step"]))
d = json.loads(open("gpurun_out/r4z_scf.json").read().strip().splitlines()[-1])
for s in d["config"]["systems"]:
    print(s["system"], "wall %.1f kernel %.1f integrals %.0f points %.0f parity %s" % (s["wall_ms_per_step"], s["kernel_ms_per_step"], s["integrals_per_step"], s["energy_points_per_step"], s.get("parity_rel_fro_P_vs_oracle_replay")))
PY
