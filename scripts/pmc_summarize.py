#!/usr/bin/env python3
"""Average the rocprofv3 --pmc passes written by scripts/gpu_pmc.sh per kernel launch.

    python scripts/pmc_summarize.py gpurun_out/pmc profiles/r01_pmc/c2_bench_pmc_per_launch_avg.json

One counter group per sub-directory (fetch, write, sq1, sq2, tcc); the newest
*_counter_collection.csv of each group is used.  Output: {kernel name: {counter: mean per launch}}.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def main(src, dst):
    out = defaultdict(dict)
    for group in sorted(os.listdir(src)):
        files = glob.glob(os.path.join(src, group, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        newest = max(files, key=os.path.getmtime)
        acc = defaultdict(lambda: defaultdict(list))
        with open(newest) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for kern, counters in acc.items():
            if kern.startswith("__amd_rocclr"):
                continue
            for name, vals in counters.items():
                out[kern][name] = sum(vals) / len(vals)
                out[kern].setdefault("_launches", {})[name] = len(vals)
    # which kernel sources the counters belong to: bench.py only quotes a profile taken on the tree's kernels
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaunegf_amd", "csrc")
    out["_kernel_source_sha16"] = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16]
                                   for f in sorted(os.listdir(csrc)) if f.endswith(".hip")}
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for kern, c in out.items():
        if kern.startswith("_"):
            continue
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd, wr = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
            print(f"{kern[:70]:70s} read {rd/1e9:8.3f} GB  write {wr/1e9:8.3f} GB per launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
