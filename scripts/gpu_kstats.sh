#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the midsize inverse timing script
mkdir -p gpurun_out/kstats
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats/t -- python $R/scripts/time_midsize.py ${SIZES:-500} > $R/gpurun_out/kstats/run.log 2>&1
rc=$?; echo "exit=$rc"; grep "^n=" $R/gpurun_out/kstats/run.log
f=$(ls -t $(find $R/gpurun_out/kstats/t -name "*kernel_stats.csv") | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {r["Percentage"]}%')
PY
