#!/bin/bash
# per-call durations of the dense products of one C5 step (Hermitian second product of G Gamma G^H vs plain products)
mkdir -p gpurun_out/hermtrace
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/hermtrace/t -- python $R/bench.py --config c5 --steps 1 --warmup 1 > $R/gpurun_out/hermtrace/run.log 2>&1
echo "exit=$?"
f=$(ls -t $(find $R/gpurun_out/hermtrace/t -name "*kernel_trace.csv") | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'zgemm' in r['Kernel_Name']:
        print(r['Kernel_Name'][:20], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
PY
