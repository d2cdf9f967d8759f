#!/bin/bash
# host BLAS threads of the density step (NEGF_HOST_BLAS_THREADS): section times of one n200 step, then the SCF lines
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for t in 1 4 8; do echo "threads $t"; NEGF_HOST_BLAS_THREADS=$t timeout -k 10 300 python scripts/probe_focktop_host.py n200 2>&1 | grep " step "; done
for t in 1 4 8; do
  NEGF_HOST_BLAS_THREADS=$t timeout -k 10 600 python bench.py --config scf --scf-systems n60,n200,n800 --steps 3 > gpurun_out/r4bb_$t.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4bb_$t.json").read().strip().splitlines()[-1])
print("threads $t:", "; ".join("%s wall %.1f kernel %.1f" % (s["system"], s["wall_ms_per_step"], s["kernel_ms_per_step"]) for s in d["config"]["systems"]))
PY
done
