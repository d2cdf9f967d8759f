#!/bin/bash
# PMC passes on the chain kernel at n_c=50 (scripts/time_chain.py): separate runs per counter
mkdir -p gpurun_out/pmc_chain
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc_chain/$1 -- python $R/scripts/time_chain.py ${NE:-768} > $R/gpurun_out/pmc_chain/$1.log 2>&1
  rc=$?; echo "$1 exit=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/pmc_chain/$1.log; exit $rc; fi
}
run fetch "FETCH_SIZE" || exit 1
run write "WRITE_SIZE" || exit 1
python $R/scripts/pmc_summarize.py $R/gpurun_out/pmc_chain $R/gpurun_out/pmc_chain/summary.json
