"""Timeline of one launch sequence from a rocprofv3 kernel trace: per kernel name the count, the summed duration and
the summed idle gap in front of it (start minus the previous kernel's end on the same queue order), for the LAST
`gr_int_dev` pass of a `time_smallbatch.py`-style run.  usage: trace_gaps.py <dir> [min_start_fraction]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
f = max(glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0 + frac * (t1 - t0)]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    dur[name] += (e - s) / 1e3; cnt[name] += 1
    if prev_end is not None:
        gap[name] += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end or e, e)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{len(rows)} kernels over {span:.0f} us; busy {sum(dur.values()):.0f} us (overlap counted twice), gaps {sum(gap.values()):.0f} us")
for n in sorted(dur, key=lambda k: -dur[k]):
    print(f"{n:60s} x{cnt[n]:4d}  {dur[n]:9.1f} us  ({dur[n]/cnt[n]:7.1f} each)  gaps in front {gap[n]:8.1f} us")
