#!/bin/bash
# instruction counters of the chain launch: round robin off / on (compare with profiles/r04_pmc_c3_per_launch_avg.json)
mkdir -p gpurun_out/prof_r4v
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_r4v
cd /tmp && export TMPDIR=/tmp
for q in 0 100; do
  export NEGF_CHAIN_RR=$q
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $P/rr$q/sq1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-extra --no-warm > $P/rr$q.log 2>&1 || { tail -5 $P/rr$q.log; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $P/rr$q/sq2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-extra --no-warm > $P/rr${q}b.log 2>&1 || { tail -5 $P/rr${q}b.log; exit 1; }
  python $R/scripts/pmc_summarize.py $P/rr$q $P/pmc_rr$q.json
done
