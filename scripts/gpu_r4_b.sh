#!/bin/bash
# Round 4, second session: the fused small-system kernel (tests), then the WHOLE gpu suite (the host-side staging and
# the entry points changed), then the SCF call pattern again.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_small_fused_gpu.py -x -q -m gpu > gpurun_out/r4b_small.log 2>&1
rc=$?; tail -15 gpurun_out/r4b_small.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4b_tests.log 2>&1
rc=$?; tail -15 gpurun_out/r4b_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --config scf --steps 3 > gpurun_out/r4b_scf.json 2> gpurun_out/r4b_scf.err; echo "scf rc=$?"; tail -c 400 gpurun_out/r4b_scf.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4b_scf.json").read().strip().splitlines()[-1])
for r in d["config"]["systems"]:
    print(r["system"], "wall %.1f ms  kernel %.1f ms  integrals %.0f  points %.0f  parity %s" % (r["wall_ms_per_step"], r["kernel_ms_per_step"], r["integrals_per_step"], r["energy_points_per_step"], r.get("parity_rel_fro_P_vs_oracle_replay")), {k: round(v, 2) for k, v in r["family_ms_per_step"].items()}, r.get("chain_cache"))
PY
