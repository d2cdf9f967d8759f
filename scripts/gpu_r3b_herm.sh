#!/bin/bash
# round 3 (second session): Hermitian form of G Gamma G^H -- parity subset, then C5 / C2 timings with and without it
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "GrLessInt or compact_gamma or reference_consistency or config_C5 or spin_block or full_size_properties or golden or density_front or fock_to_p" > gpurun_out/r3b_herm_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r3b_herm_tests.log; if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" gpurun_out/r3b_herm_tests.log | head -20; exit $rc; fi
for h in 0 1; do
echo "== NEGF_ZGEMM_HERM=$h"
NEGF_ZGEMM_HERM=$h timeout -k 10 300 python bench.py --config c5 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c1-200
NEGF_ZGEMM_HERM=$h timeout -k 10 300 python scripts/bench_configs.py c2 2>/dev/null | tail -3 | cut -c1-400
done > gpurun_out/r3b_herm_time.log 2>&1
cat gpurun_out/r3b_herm_time.log
