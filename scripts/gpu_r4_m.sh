#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4m_cw -- python $R/scripts/time_small_calls.py 60 > $R/gpurun_out/r4m_cw.log 2>&1
cat $(ls -t $R/gpurun_out/r4m_cw/*/*_kernel_stats.csv | head -1) | cut -c1-200 | head -8
