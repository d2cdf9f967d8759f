#!/bin/bash
mkdir -p gpurun_out
echo "== committed kernel"; ( cd _head_ab && timeout -k 10 200 python scripts/time_chain_first.py ) || exit 1
for q in 0 100 60 150; do echo "== tree, NEGF_CHAIN_RR=$q"; NEGF_CHAIN_RR=$q timeout -k 10 200 python scripts/time_chain_first.py || exit 1; done
