#!/bin/bash
mkdir -p gpurun_out
CASES="800x12 800x36 800x61 800x108 1000x64 1000x128 500x64 500x250"
for cfg in "4 12" "8 12" "8 6" "8 4" "8 8"; do
  set -- $cfg
  echo "== NEGF_GJ_GROUP_MAX=$1 NEGF_GJ_GROUP_MIN=$2"
  NEGF_GJ_GROUP_MAX=$1 NEGF_GJ_GROUP_MIN=$2 timeout -k 10 200 python scripts/time_smallbatch.py $CASES 2>&1 | grep "^n=" | awk '{print $1, $2, $6, $7, $9, $10}'
done
