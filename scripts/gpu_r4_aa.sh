#!/bin/bash
# the SCF call pattern with the host algebra of the density step left to numpy's default thread count / limited (_hostblas.py)
mkdir -p gpurun_out
for t in 0 default; do
  if [ $t = 0 ]; then export NEGF_HOST_BLAS_THREADS=0; else unset NEGF_HOST_BLAS_THREADS; fi
  timeout -k 10 800 python bench.py --config scf --steps 3 > gpurun_out/r4aa_scf_$t.json 2> gpurun_out/r4aa_scf_$t.err || { tail -5 gpurun_out/r4aa_scf_$t.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4aa_scf_$t.json").read().strip().splitlines()[-1])
print("host threads $t:", "; ".join("%s wall %.1f kernel %.1f parity %s" % (s["system"], s["wall_ms_per_step"], s["kernel_ms_per_step"], s.get("parity_rel_fro_P_vs_oracle_replay")) for s in d["config"]["systems"]))
PY
done
