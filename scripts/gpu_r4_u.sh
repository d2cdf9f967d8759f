#!/bin/bash
mkdir -p gpurun_out
if [ -d _head_ab ]; then echo "== committed kernel"; ( cd _head_ab && timeout -k 10 200 python scripts/time_chain_first.py ) || exit 1; fi   # (_head_ab: git archive HEAD | tar -x -C _head_ab, built there)
for q in ${RR_LIST:-0 100}; do echo "== tree, NEGF_CHAIN_RR=$q"; NEGF_CHAIN_RR=$q timeout -k 10 200 python scripts/time_chain_first.py || exit 1; done
