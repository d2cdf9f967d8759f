#!/bin/bash
# Round-3 profiles: rocprofv3 kernel trace + stats of the driver's bench command (C3 headline with the
# N_orb=500 x 1000 and C2 secondary lines), of the C5 configuration, and separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, SQ groups) on the headline alone.  Outputs under gpurun_out/prof_r3/.
mkdir -p gpurun_out/prof_r3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3/bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/prof_r3/bench.json 2> $R/gpurun_out/prof_r3/bench.err
rc=$?; echo "bench trace exit=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r3/bench.err; exit $rc; fi
if [ -z "$SKIP_C5" ]; then
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3/c5 -- python $R/bench.py --config c5 --steps 1 --warmup 1 > $R/gpurun_out/prof_r3/c5.json 2> $R/gpurun_out/prof_r3/c5.err
rc=$?; echo "c5 trace exit=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r3/c5.err; exit $rc; fi
fi
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/prof_r3/pmc/$1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-extra > $R/gpurun_out/prof_r3/pmc_$1.log 2>&1
  rc=$?; echo "pmc $1 exit=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r3/pmc_$1.log; exit $rc; fi
}
run fetch "FETCH_SIZE" || exit 1
run write "WRITE_SIZE" || exit 1
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" || exit 1
run sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" || exit 1
python $R/scripts/pmc_summarize.py $R/gpurun_out/prof_r3/pmc $R/gpurun_out/prof_r3/pmc_per_launch_avg.json
find $R/gpurun_out/prof_r3 -name "*kernel_stats.csv"
