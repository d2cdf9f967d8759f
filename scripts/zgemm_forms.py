"""Per-dispatch comparison of the two product forms of zgemm_mfma_kernel in a `time_products.py 1000` run with
NEGF_ZGEMM_HERM=0: dispatches alternate  X = G Gamma (A B)  and  X G^H (A B^H, the conjugate-transposed operand read in
place).  Reads the rocprofv3 outputs of scripts/gpu_profiles_r4.sh (PMC pass and kernel trace) under the given directory
and prints durations and LDS / matrix-pipe counters per form."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]


def newest(pattern):
    f = glob.glob(os.path.join(root, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


pmc = newest("pmc_zgemm/sq/**/*_counter_collection.csv")
trace = newest("zgemm_trace/**/*_kernel_trace.csv")
forms = ("A*B (X = G Gamma)", "A*B^H (X G^H)")
if pmc:
    per = defaultdict(lambda: defaultdict(float))           # dispatch id -> counter -> value
    for row in csv.DictReader(open(pmc)):
        if "zgemm_mfma_kernel" in row["Kernel_Name"]:
            per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(per)
    for form in (0, 1):
        sel = ids[form::2]
        if not sel:
            continue
        keys = sorted(per[sel[0]])
        avg = {k: sum(per[i][k] for i in sel) / len(sel) for k in keys}
        line = ", ".join(f"{k} {avg[k]:.3e}" for k in keys)
        extra = ""
        if avg.get("SQ_ACTIVE_INST_LDS"):
            extra = f"; bank-conflict cycles / LDS-active cycles = {avg.get('SQ_LDS_BANK_CONFLICT', 0) / avg['SQ_ACTIVE_INST_LDS']:.3f}"
        print(f"{forms[form]}: {len(sel)} dispatches: {line}{extra}")
if trace:
    rows = [r for r in csv.DictReader(open(trace)) if "zgemm_mfma_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for form in (0, 1):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[form::2]]
        if d:
            print(f"{forms[form]}: {len(d)} dispatches, {sum(d) / len(d):.2f} ms average (256 products of n = 1000)")
