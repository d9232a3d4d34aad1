#!/usr/bin/env python3
"""Turn `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...` output into the
summary committed under profiles/: per kernel, mean duration over the overlapped timed region and over
the strictly serial launches bench.py makes afterwards for the isolated roofline.

    python tools/profile_summary.py <dir with *_kernel_trace.csv> <steps> > profiles/rNN_kernel_trace_summary.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    if "rows_kernel" in name:
        return "rows_kernel"
    if "cols_kernel" in name or "cols_mixed_kernel" in name:
        return "cols_fwd" if ", -1, " in name else "cols_inv"
    return None


def main(root, n_timed, n_iso=40):
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    rows = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                rows[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out = {}
    for k, v in rows.items():
        v.sort()
        d = [(e - s) / 1e3 for s, e in v]
        # bench.py ends with n_iso strictly serial launch groups; the n_timed groups before them are the timed
        # region; everything earlier is clock ramp and warm-up
        iso = d[-n_iso:] if n_iso else []
        ov = d[-(n_iso + n_timed):-n_iso] if n_iso else d[-n_timed:]
        out[k] = {"launches": len(d),
                  "timed_region_launches": len(ov),
                  "overlapped_avg_us": sum(ov) / max(len(ov), 1),
                  "isolated_avg_us": (sum(iso) / len(iso)) if iso else None,
                  "isolated_min_us": min(iso) if iso else None}
    return out


if __name__ == "__main__":
    print(json.dumps({"kernels": main(sys.argv[1], int(sys.argv[2]))}, indent=1))
