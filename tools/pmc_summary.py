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


def totals_from(root, first_kernel_substr):
    """{counter: (sum of Counter_Value over every dispatch from the first one whose kernel name contains the substring,
    number of such dispatches)} - the counters of the calls a child run makes after its set-up."""
    out = {}
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        started = False
        for row in rows:
            started = started or first_kernel_substr in row["Kernel_Name"]
            if started:
                tot, cnt = out.get(row["Counter_Name"], (0.0, 0))
                out[row["Counter_Name"]] = (tot + float(row["Counter_Value"]), cnt + 1)
    return out


if __name__ == "__main__":
    import json
    r = main(sys.argv[1])
    print(json.dumps(r, indent=1))
