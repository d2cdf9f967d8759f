#!/bin/bash
# PMC passes on the chain kernel at n_c=50 (scripts/time_chain.py): separate runs per counter group
mkdir -p gpurun_out/pmc_chain
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc_chain/$1 -- python $R/scripts/time_chain.py 384 > $R/gpurun_out/pmc_chain/$1.log 2>&1
  rc=$?; echo "$1 exit=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/pmc_chain/$1.log; exit $rc; fi
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" || exit 1
run sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" || exit 1
run sq3 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" || exit 1
run mem "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" || exit 1
python $R/scripts/pmc_summarize.py $R/gpurun_out/pmc_chain $R/gpurun_out/pmc_chain/summary.json
tail -60 $R/gpurun_out/pmc_chain/summary.json
