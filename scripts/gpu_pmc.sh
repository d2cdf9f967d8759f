#!/bin/bash
# PMC passes on the bench (separate runs per counter group; kernel-trace only alongside)
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc/$1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/pmc/$1.log 2>&1
  rc=$?; echo "$1 exit=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/pmc/$1.log; exit $rc; fi
}
run fetch "FETCH_SIZE" || exit 1
run write "WRITE_SIZE" || exit 1
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" || exit 1
run sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" || exit 1
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" || exit 1
ls $R/gpurun_out/pmc/*/* | head -30
