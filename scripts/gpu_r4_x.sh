#!/bin/bash
# the chain launches of one SCF step (bench.py --config scf, chain system): jobs per launch and kernel duration,
# round robin off (order predicted) / on
mkdir -p gpurun_out/prof_r4x
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_r4x
cd /tmp && export TMPDIR=/tmp
export NEGF_CHAIN_LOG=1
for q in 0 100; do
  export NEGF_CHAIN_RR=$q
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $P/rr$q -- python $R/bench.py --config scf --scf-systems chain --steps 1 > $P/rr$q.json 2> $P/rr$q.err || { tail -5 $P/rr$q.err; exit 1; }
done
