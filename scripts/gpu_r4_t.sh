#!/bin/bash
# same-box A/B: the committed kernel (a built copy of HEAD under _head_ab/) against the tree's, round robin off / on
mkdir -p gpurun_out
ROOT=$PWD
run() {  # label dir env
  ( cd $2 && env $3 timeout -k 10 200 python bench.py --no-cpu --no-extra --no-warm --steps 3 --warmup 2 > $ROOT/gpurun_out/r4t_$1.json 2> $ROOT/gpurun_out/r4t_$1.err ) || { echo "$1 failed"; tail -5 gpurun_out/r4t_$1.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4t_$1.json").read().strip().splitlines()[-1])
print("$1 ms/step", round(d["ms_per_step"],1), "first", round(d["config"]["first_evaluation_ms"],1), "chain launch", round(d["roofline"]["avg_launch_ms"],1))
PY
}
for rep in 1 2; do
  run head_$rep _head_ab A=1
  run rr0_$rep . NEGF_CHAIN_RR=0
  run rr100_$rep . NEGF_CHAIN_RR=100
done
run rr30 . NEGF_CHAIN_RR=30
run rr200 . NEGF_CHAIN_RR=200
