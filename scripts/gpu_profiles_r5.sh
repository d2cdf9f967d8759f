#!/bin/bash
# Round-5 profiles (outputs under gpurun_out/prof_r5/, summaries copied to profiles/r05_* afterwards):
#   1. rocprofv3 kernel trace + stats of the driver's bench command (C3 headline with its secondary lines) and its JSON
#   2. separate PMC passes on the headline alone (FETCH_SIZE; WRITE_SIZE; two SQ groups)
#   3. the N = 500 x 1000 windowed inverse: FETCH / WRITE per kernel (the traffic of a pass) + kernel stats
#   4. kernel trace + stats of the C5 configuration; the C4 / C5 lines, their 8-way shares, the SCF call pattern
# The profiled program stands directly after "--" (no env / bash -c hop).
mkdir -p gpurun_out/prof_r5
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_r5
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
step "bench trace"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warm > $P/bench.json 2> $P/bench.err
rc=$?; echo "bench trace exit=$rc"; [ $rc -ne 0 ] && { tail -5 $P/bench.err; exit $rc; }
run() { # dir name, counters, program...
  local name=$1 ctr=$2; shift 2
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $P/$name -- "$@" > $P/$(echo $name | tr / _).log 2>&1
  rc=$?; echo "pmc $name exit=$rc"; [ $rc -ne 0 ] && { tail -5 $P/$(echo $name | tr / _).log; exit $rc; }
  return 0
}
HEAD="python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-extra --no-warm"
run pmc/fetch "FETCH_SIZE" $HEAD || exit 1
run pmc/write "WRITE_SIZE" $HEAD || exit 1
run pmc/sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $HEAD || exit 1
run pmc/sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $HEAD || exit 1
python $R/scripts/pmc_summarize.py $P/pmc $P/pmc_c3_per_launch_avg.json
[ -n "$ONLY_HEAD" ] && { find $P -name "*kernel_stats.csv"; exit 0; }
step "N = 500 inverse"
N5="python $R/scripts/time_midsize.py 500"
run pmc_n500/fetch "FETCH_SIZE" $N5 || exit 1
run pmc_n500/write "WRITE_SIZE" $N5 || exit 1
run pmc_n500/sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" $N5 || exit 1
python $R/scripts/pmc_summarize.py $P/pmc_n500 $P/pmc_n500_per_launch_avg.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/n500 -- python $R/scripts/time_midsize.py 500 > $P/n500.log 2>&1; echo "n500 trace exit=$?"
step "c5 trace, scf, c4, shares"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/c5 -- python $R/bench.py --config c5 --steps 1 --warmup 1 > $P/c5.json 2> $P/c5.err; echo "c5 trace exit=$?"
cd $R
timeout -k 10 900 python bench.py --config scf --steps 5 > $P/scf.json 2> $P/scf.err; echo "scf exit=$?"
timeout -k 10 300 python bench.py --config c4 --steps 5 > $P/c4.json 2> $P/c4.err; echo "c4 exit=$?"
timeout -k 10 300 python bench.py --config c5 --steps 3 --no-cpu > $P/c5_plain.json 2> $P/c5_plain.err; echo "c5 exit=$?"
timeout -k 10 300 python bench.py --config c4 --steps 5 --emulate-share 8 > $P/c4_share8.json 2> $P/c4_share8.err; echo "c4 share exit=$?"
timeout -k 10 300 python bench.py --config c5 --steps 3 --emulate-share 8 > $P/c5_share8.json 2> $P/c5_share8.err; echo "c5 share exit=$?"
find $P -name "*.db" -delete; find $P -name "*_kernel_trace.csv" -delete; find $P -name "*counter_collection.csv" -size +20M -delete
find $P -name "*kernel_stats.csv"
