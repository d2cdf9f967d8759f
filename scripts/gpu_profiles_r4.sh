#!/bin/bash
# Round-4 profiles (outputs under gpurun_out/prof_r4/, summaries copied to profiles/ afterwards):
#   1. rocprofv3 kernel trace + stats of the driver's bench command (C3 headline with its secondary lines) and its JSON
#   2. separate PMC passes on the headline alone (FETCH_SIZE; WRITE_SIZE; two SQ groups)
#   3. C2's single-workgroup inverse (gj_blocked_kernel, 1000 x N = 200): FETCH / WRITE / TCC_EA requests / MFMA busy
#   4. the dense product forms A B and A B^H at n = 1000 (NEGF_ZGEMM_HERM=0: the second product of G Gamma G^H runs as a
#      full A B^H): LDS conflict and matrix-pipe counters per dispatch
#   5. kernel trace + stats of the C5 configuration, and the SCF call pattern's JSON
# The profiled program stands directly after "--" (no env / bash -c hop).
mkdir -p gpurun_out/prof_r4
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_r4
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
step "C2 inverse"
C2="python $R/scripts/time_midsize.py 200"
run pmc_c2/fetch "FETCH_SIZE" $C2 || exit 1
run pmc_c2/write "WRITE_SIZE" $C2 || exit 1
run pmc_c2/tcc "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" $C2 || exit 1
run pmc_c2/sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $C2 || exit 1
run pmc_c2/sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $C2 || exit 1
python $R/scripts/pmc_summarize.py $P/pmc_c2 $P/pmc_c2_per_launch_avg.json
step "zgemm forms"
export NEGF_ZGEMM_HERM=0
run pmc_zgemm/sq "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" python $R/scripts/time_products.py 1000 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/zgemm_trace -- python $R/scripts/time_products.py 1000 > $P/zgemm_trace.log 2>&1; echo "zgemm trace exit=$?"
unset NEGF_ZGEMM_HERM
python $R/scripts/zgemm_forms.py $P > $P/zgemm_forms.txt 2>&1; cat $P/zgemm_forms.txt
step "c5 trace, scf"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/c5 -- python $R/bench.py --config c5 --steps 1 --warmup 1 > $P/c5.json 2> $P/c5.err; echo "c5 trace exit=$?"
cd $R
timeout -k 10 900 python bench.py --config scf --steps 3 > $P/scf.json 2> $P/scf.err; echo "scf exit=$?"
timeout -k 10 300 python bench.py --config c4 --steps 3 > $P/c4.json 2> $P/c4.err; echo "c4 exit=$?"
timeout -k 10 300 python bench.py --config c4 --steps 3 --emulate-share 8 > $P/c4_share8.json 2> $P/c4_share8.err; echo "c4 share exit=$?"
timeout -k 10 300 python bench.py --config c5 --steps 2 --emulate-share 8 > $P/c5_share8.json 2> $P/c5_share8.err; echo "c5 share exit=$?"
find $P -name "*kernel_stats.csv"
