"""Summarise rocprofv3 --pmc CSV output (per-kernel mean of each counter) and, with --kernel-trace
rows alongside, the mean kernel duration.  Usage: python tools/pmc_summary.py gpurun_out/pmc"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    if "rows_kernel" in name:
        return "rows_kernel"
    if "rows_single_kernel" in name:
        return "rows_single"
    if "fir_block_kernel" in name:
        return "fir_block"
    if "row_first_peak_chunked_kernel" in name:
        return "peak_search"
    if "cols_kernel" in name or "cols_mixed_kernel" in name or "cols_small_kernel" in name:
        return "cols_fwd" if ", -1, " in name else "cols_inv"
    return name[:40]


def main(root):
    out = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            out[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            dur[short(row["Kernel_Name"])].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    res = {}
    for k in sorted(out):
        res[k] = {c: sum(v) / len(v) for c, v in sorted(out[k].items())}
        if k in dur:
            res[k]["_mean_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
    return res


if __name__ == "__main__":
    import json
    r = main(sys.argv[1])
    print(json.dumps(r, indent=1))
