#!/bin/bash
# staging of zgemm_mfma_kernel: old (4 consecutive elements per lane) / new (256 consecutive bytes per 16 lanes) --
# time per form from a kernel trace, L2 requests per form from one PMC pass each
mkdir -p gpurun_out/prof_r4y
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_r4y
cd /tmp && export TMPDIR=/tmp
export NEGF_ZGEMM_HERM=0
for s in 0 1; do
  export NEGF_ZGEMM_STAGE=$s
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/trace$s -- python $R/scripts/time_products.py 1000 500 200 > $P/trace$s.log 2>&1 || { tail -5 $P/trace$s.log; exit 1; }
  grep "^n=" $P/trace$s.log
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $P/tcc$s -- python $R/scripts/time_products.py 1000 > $P/tcc$s.log 2>&1 || { tail -5 $P/tcc$s.log; exit 1; }
done
unset NEGF_ZGEMM_HERM
cd $R
for s in 0 1; do
  NEGF_ZGEMM_STAGE=$s timeout -k 10 300 python bench.py --config c5 --steps 2 > gpurun_out/prof_r4y/c5_stage$s.json 2> gpurun_out/prof_r4y/c5_stage$s.err || { tail -5 gpurun_out/prof_r4y/c5_stage$s.err; exit 1; }
done
timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "gless or transmission or zgemm or product or C5 or herm" > gpurun_out/prof_r4y/tests.log 2>&1; tail -2 gpurun_out/prof_r4y/tests.log
