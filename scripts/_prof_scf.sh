set -e
timeout -k 10 400 python -m cProfile -o gpurun_out/r5_scf.prof bench.py --config scf --scf-systems n60,n200 --steps 5 --no-cpu > gpurun_out/r5_scf_prof.json 2> gpurun_out/r5_scf_prof.err
python - <<PY > gpurun_out/r5_scf_hostprof.log 2>&1
import pstats
p = pstats.Stats("gpurun_out/r5_scf.prof"); p.sort_stats("tottime").print_stats(60)
p.sort_stats("cumulative").print_stats(70)
PY
