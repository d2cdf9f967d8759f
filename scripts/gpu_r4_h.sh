#!/bin/bash
# where a tiny batch of large matrices spends its time: 4, 12, 36 x N = 800 and 2 x N = 200 under a kernel trace
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/time_smallbatch.py 800x2 800x4 800x12 800x36 800x108 200x2 200x12 200x108 500x4 500x36 > gpurun_out/r4h_small.log 2>&1; grep "^n=" gpurun_out/r4h_small.log
cd /tmp && export TMPDIR=/tmp
for c in 800x12 200x12; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4h_trace_$c -- python $R/scripts/time_smallbatch.py $c > $R/gpurun_out/r4h_trace_$c.log 2>&1
  echo "== $c"; python $R/scripts/trace_gaps.py $R/gpurun_out/r4h_trace_$c 0.85
done
