#!/bin/bash
# Round 4, first session: the g(E) cache and unequal-contact tests, the chain / C3 parity subset, then the bench lines
# with the two-way roofline accounting (headline cold + warm, c5, c4, c4 as an 8-way share).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_chain_cache_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "chain or gcache or C3 or surfG" > gpurun_out/r4a_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r4a_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
rc=$?; echo "bench rc=$rc"; tail -c 600 gpurun_out/r4a_bench.err; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --config c5 --steps 2 > gpurun_out/r4a_c5.json 2> gpurun_out/r4a_c5.err; echo "c5 rc=$?"
timeout -k 10 300 python bench.py --config c4 --steps 3 > gpurun_out/r4a_c4.json 2> gpurun_out/r4a_c4.err; echo "c4 rc=$?"
timeout -k 10 300 python bench.py --config c4 --steps 3 --emulate-share 8 > gpurun_out/r4a_c4_share8.json 2> gpurun_out/r4a_c4_share8.err; echo "c4 share rc=$?"
timeout -k 10 600 python bench.py --config scf --steps 3 > gpurun_out/r4a_scf.json 2> gpurun_out/r4a_scf.err; echo "scf rc=$?"; tail -c 400 gpurun_out/r4a_scf.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4a_scf.json").read().strip().splitlines()[-1])
    for r in d["config"]["systems"]:
        print(r["system"], "wall %.1f ms  kernel %.1f ms  integrals %.0f  points %.0f  parity %s" % (r["wall_ms_per_step"], r["kernel_ms_per_step"], r["integrals_per_step"], r["energy_points_per_step"], r.get("parity_rel_fro_P_vs_oracle_replay")), r["family_ms_per_step"])
except Exception as e:
    print("scf unreadable:", e)
for f in ("r4a_bench", "r4a_c5", "r4a_c4", "r4a_c4_share8"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "ms/step %.1f" % d["ms_per_step"], "alg %.3f exec %.3f" % (r["frac_algorithmic"], r["frac"]), d.get("warm", {}).get("warm_ms_per_step"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
