#!/bin/bash
# Round-5 GPU session, parametrised: scripts/gpu_r5.sh <step> [<step> ...]   (run through gpurun; output under gpurun_out/r5_*)
# steps: inv_tests | inv_time | inv_ab | small | fuzz | tests | bench | c4 | c5 | share | scf | kstats | pmc
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for step in "$@"; do
  echo "=== $step"
  case "$step" in
    inv_tests) timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "G_of_E or window or blocked or singular or inverse or imaginary" > gpurun_out/r5_inv_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r5_inv_tests.log; [ $rc -eq 0 ] || exit $rc ;;
    inv_time) timeout -k 10 600 python scripts/time_midsize.py 200 256 300 400 500 650 800 1000 > gpurun_out/r5_inv_time.log 2>&1 || exit 1; cat gpurun_out/r5_inv_time.log ;;
    inv_ab) for v in 0 1; do echo "NEGF_GJ_STRIP=$v"; NEGF_GJ_STRIP=$v timeout -k 10 600 python scripts/time_midsize.py ${AB_N:-300 500 800 1000} || exit 1; done > gpurun_out/r5_inv_ab.log 2>&1; cat gpurun_out/r5_inv_ab.log ;;
    small) for v in ${SMALL_STRIP:-1}; do echo "NEGF_GJ_STRIP=$v"; NEGF_GJ_STRIP=$v timeout -k 10 600 python scripts/time_smallbatch.py ${SMALL_CASES:-800x12 800x61 800x93 800x486 1000x64 1000x128 500x64 500x250} || exit 1; done > gpurun_out/r5_small.log 2>&1; grep -E "^n=|STRIP" gpurun_out/r5_small.log ;;
    fuzz) timeout -k 10 900 python scripts/fuzz_inverse.py > gpurun_out/r5_fuzz.log 2>&1; rc=$?; tail -8 gpurun_out/r5_fuzz.log; [ $rc -eq 0 ] || exit $rc ;;
    tests) timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r5_tests.log; [ $rc -eq 0 ] || exit $rc ;;
    bench) timeout -k 10 600 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || exit 1; cat gpurun_out/r5_bench.json ;;
    c4) timeout -k 10 600 python bench.py --config c4 --no-cpu > gpurun_out/r5_c4.json 2> gpurun_out/r5_c4.err || exit 1; cat gpurun_out/r5_c4.json ;;
    c5) timeout -k 10 600 python bench.py --config c5 --no-cpu > gpurun_out/r5_c5.json 2> gpurun_out/r5_c5.err || exit 1; cat gpurun_out/r5_c5.json ;;
    share) for c in c4 c5; do timeout -k 10 600 python bench.py --config $c --emulate-share 8 --no-cpu > gpurun_out/r5_${c}_share8.json 2> gpurun_out/r5_${c}_share8.err || exit 1; cat gpurun_out/r5_${c}_share8.json; done ;;
    scf) timeout -k 10 900 python bench.py --config scf --no-cpu > gpurun_out/r5_scf.json 2> gpurun_out/r5_scf.err || exit 1; cat gpurun_out/r5_scf.json ;;
    kstats) # per-kernel times of the inverse pass (KS_N sizes, default 500)
       rm -rf gpurun_out/r5_kstats; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5_kstats -o ks -- python3 $GRAFT_REPO_ROOT/scripts/time_midsize.py ${KS_N:-500} > $GRAFT_REPO_ROOT/gpurun_out/r5_kstats.log 2>&1; rc=$?; cd $GRAFT_REPO_ROOT; [ $rc -eq 0 ] || { tail -20 gpurun_out/r5_kstats.log; exit $rc; }
       f=$(find gpurun_out/r5_kstats -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
       find gpurun_out/r5_kstats -name "*.db" -delete; find gpurun_out/r5_kstats -name "*trace.csv" -delete ;;
    ablate) for d in 0 1 2 3 4 8 12 15; do echo "NEGF_GJ_STRIP_DBG=$d"; NEGF_GJ_SPLIT_MAX=0 NEGF_GJ_PAIR=0 NEGF_GJ_STRIP_DBG=$d KS_N=${KS_N:-500} bash scripts/gpu_r5.sh kstats 2>&1 | grep "strip_kernel"; done ;;
    stamps) NEGF_GJ_STAMPS=1 timeout -k 10 300 python scripts/time_midsize.py ${KS_N:-500} 2>&1 | grep -E "stamps|^n=" | tail -3 | tee gpurun_out/r5_stamps.log ;;
    cfg_ab) for v in 0 1; do echo "NEGF_GJ_STRIP_CFG=$v"; NEGF_GJ_STRIP_CFG=$v KS_N="300 500" bash scripts/gpu_r5.sh stamps; NEGF_GJ_STRIP_CFG=$v timeout -k 10 300 python scripts/time_midsize.py 300 400 500 2>&1 | grep "^n="; done 2>&1 | tee gpurun_out/r5_cfg_ab.log ;;
    chain_ab) # A/B of chain-kernel builds: the tree's library and the variants under gaunegf_amd/lib/libnegf_hip_<name>.so (CHAIN_LIBS)
       for L in ${CHAIN_LIBS:-default}; do
         if [ "$L" = default ]; then unset NEGF_LIB_PATH; else export NEGF_LIB_PATH=$GRAFT_REPO_ROOT/gaunegf_amd/lib/libnegf_hip_$L.so; fi
         echo "--- lib $L"; timeout -k 10 300 python scripts/time_chain_uniform.py 2>&1 | grep "energies" | head -2; timeout -k 10 300 python scripts/time_chain.py 2>&1 | tail -3
       done 2>&1 | tee gpurun_out/r5_chain_ab.log; unset NEGF_LIB_PATH ;;
    chain_tests) timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "chain or C3 or surfG or sigma" > gpurun_out/r5_chain_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5_chain_tests.log; [ $rc -eq 0 ] || exit $rc ;;
    small_stamps) for v in 1 0; do echo "NEGF_GJ_STRIP=$v"; NEGF_GJ_STRIP=$v NEGF_GJ_STAMPS=1 timeout -k 10 300 python scripts/time_smallbatch.py ${SMALL_CASES:-800x12} 2>&1 | grep -E "stamps|^n=" | tail -3; done | tee gpurun_out/r5_small_stamps.log ;;
    small_kstats) rm -rf gpurun_out/r5_skstats; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5_skstats -o ks -- python3 $GRAFT_REPO_ROOT/scripts/time_smallbatch.py ${SMALL_CASES:-800x12} > $GRAFT_REPO_ROOT/gpurun_out/r5_skstats.log 2>&1; rc=$?; cd $GRAFT_REPO_ROOT; [ $rc -eq 0 ] || { tail -20 gpurun_out/r5_skstats.log; exit $rc; }
       grep "^n=" gpurun_out/r5_skstats.log; f=$(find gpurun_out/r5_skstats -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
       find gpurun_out/r5_skstats -name "*.db" -delete; find gpurun_out/r5_skstats -name "*trace.csv" -delete ;;
    hostprof) for c in c4 c5; do timeout -k 10 600 python -m cProfile -o gpurun_out/r5_${c}_share8.prof bench.py --config $c --emulate-share 8 --steps 5 --no-cpu > gpurun_out/r5_${c}_share8_prof.json 2> gpurun_out/r5_${c}_share8_prof.err || exit 1
         python - <<PY
import pstats
p = pstats.Stats("gpurun_out/r5_${c}_share8.prof"); p.sort_stats("cumulative").print_stats(45)
PY
       done > gpurun_out/r5_hostprof.log 2>&1; grep -E "ms_per_step" gpurun_out/r5_c4_share8_prof.json | cut -c1-200; tail -130 gpurun_out/r5_hostprof.log ;;
    seg_tests) timeout -k 10 900 python -m pytest tests/test_small_fused_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r5_seg_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r5_seg_tests.log; [ $rc -eq 0 ] || exit $rc ;;
    scf_small) timeout -k 10 600 python bench.py --config scf --scf-systems ${SCF_SYSTEMS:-n60,n200} --no-cpu --steps 5 > gpurun_out/r5_scf_small.json 2> gpurun_out/r5_scf_small.err || exit 1
       python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5_scf_small.json"))
for s in d["config"]["systems"]:
    print(s["system"], "wall %.2f ms  kernels %.2f ms  calls %.0f  points %.0f" % (s["wall_ms_per_step"], s["kernel_ms_per_step"], s["integrals_per_step"], s["energy_points_per_step"]), {k: round(v, 2) for k, v in s["family_ms_per_step"].items()})
PY
       ;;
    scf_kstats) rm -rf gpurun_out/r5_scfks; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5_scfks -o ks -- python3 $GRAFT_REPO_ROOT/scripts/prof_scf_host.py ${SCF_SYSTEMS:-n60} 5 > $GRAFT_REPO_ROOT/gpurun_out/r5_scfks.log 2>&1; rc=$?; cd $GRAFT_REPO_ROOT; [ $rc -eq 0 ] || { tail -20 gpurun_out/r5_scfks.log; exit $rc; }
       f=$(find gpurun_out/r5_scfks -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
       find gpurun_out/r5_scfks -name "*.db" -delete; find gpurun_out/r5_scfks -name "*trace.csv" -delete ;;
    small_tests) timeout -k 10 900 python -m pytest tests/test_small_fused_gpu.py -x -q -m gpu > gpurun_out/r5_small_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5_small_tests.log; [ $rc -eq 0 ] || exit $rc
       timeout -k 10 600 python scripts/fuzz_small.py > gpurun_out/r5_fuzz_small.log 2>&1; rc=$?; tail -3 gpurun_out/r5_fuzz_small.log; [ $rc -eq 0 ] || exit $rc ;;
    small_calls) timeout -k 10 600 python scripts/time_small_calls.py ${SMALL_N:-60 96} > gpurun_out/r5_small_calls.log 2>&1 || exit 1; grep -E "GrInt " gpurun_out/r5_small_calls.log; grep Segments gpurun_out/r5_small_calls.log ;;
    small_ab) for k in cw panel; do echo "NEGF_SMALL_KERNEL=$k"; NEGF_SMALL_KERNEL=$k timeout -k 10 300 python scripts/time_small_calls.py ${SMALL_N:-60 96} 2>&1 | grep -E "GrInt +m= +(2|324)|972"; done | tee gpurun_out/r5_small_ab.log ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
