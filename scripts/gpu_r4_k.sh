#!/bin/bash
# the N > 1 control flow of bench.py on the one-GPU box (gloo rehearsal: both ranks on device 0), after this round's edits
mkdir -p gpurun_out
NEGF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu --energies 300 > gpurun_out/r4k_c3_2ranks.json 2> gpurun_out/r4k_c3_2ranks.err; echo "c3 x2 rc=$?"; tail -c 300 gpurun_out/r4k_c3_2ranks.err; tail -c 700 gpurun_out/r4k_c3_2ranks.json; echo
NEGF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --config c4 --steps 2 --warmup 1 > gpurun_out/r4k_c4_2ranks.json 2> gpurun_out/r4k_c4_2ranks.err; echo "c4 x2 rc=$?"; tail -c 300 gpurun_out/r4k_c4_2ranks.err; python -c "
import json; d=json.loads(open('gpurun_out/r4k_c4_2ranks.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d.get('comm_ms'), d.get('collectives_per_step'))"
NEGF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --config c5 --steps 1 --warmup 1 > gpurun_out/r4k_c5_2ranks.json 2> gpurun_out/r4k_c5_2ranks.err; echo "c5 x2 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r4k_c5_2ranks.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d.get('comm_ms'), d.get('collectives_per_step'))"
timeout -k 10 600 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -3
