#!/bin/bash
# the two instantiations of the chain kernel (plain / round robin): chain tests, fuzz, C3 first / repeated evaluation, SCF chain pattern
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_chain_rr_gpu.py tests/test_chain_cache_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "chain or C3" > gpurun_out/r4cc_tests.log 2>&1 || { tail -30 gpurun_out/r4cc_tests.log; exit 1; }
tail -2 gpurun_out/r4cc_tests.log
timeout -k 10 200 python scripts/fuzz_chain_rr.py 21 60 2>&1 | tail -2 || exit 1
for q in 0 100; do echo "== NEGF_CHAIN_RR=$q"; NEGF_CHAIN_RR=$q timeout -k 10 200 python scripts/time_chain_first.py 2>&1 | grep fresh || exit 1; done
timeout -k 10 600 python bench.py --config scf --scf-systems chain --steps 3 > gpurun_out/r4cc_scf.json 2> gpurun_out/r4cc_scf.err || { tail -5 gpurun_out/r4cc_scf.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/r4cc_scf.json").read().strip().splitlines()[-1])
print("; ".join("%s wall %.1f kernel %.1f" % (s["system"], s["wall_ms_per_step"], s["kernel_ms_per_step"]) for s in d["config"]["systems"]))
PY
