#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 kernel trace: python tools/timeline.py <dir with *_kernel_trace.csv> [skip_fraction=0.3]
For the steady-state part of the run: wall time, time with 0 / 1 / 2 / 3+ kernels running, and per kernel name the number
of launches, mean duration and the share of the wall time it was on the chip."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("rows_kernel", "rows_single_kernel", "row_first_peak_chunked_kernel", "cols_small_kernel", "cols_mixed_kernel",
                "cols_kernel", "fir_block_kernel"):
        if key in name:
            tag = key
            if "cols_" in key:
                tag += "<fwd>" if ", -1, " in name or "ELin1E" in name else "<inv>"
                if "LoadCropAtPeak" in name:
                    tag += "K5A"
            return tag
    return name[:40]


def main(root, skip=0.3):
    ev = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    ev.sort()
    t_lo = ev[0][0] + skip * (ev[-1][1] - ev[0][0])
    ev = [e for e in ev if e[0] >= t_lo]
    wall = ev[-1][1] - ev[0][0]
    pts = []
    for s, e, _ in ev:
        pts.append((s, 1))
        pts.append((e, -1))
    pts.sort()
    depth, last, hist = 0, pts[0][0], defaultdict(int)
    for t, d in pts:
        hist[min(depth, 4)] += t - last
        last, depth = t, depth + d
    print(f"steady-state wall {wall / 1e3:.1f} us, {len(ev)} kernels")
    for k in sorted(hist):
        print(f"  {k}{'+' if k == 4 else ' '} kernels running: {hist[k] / wall * 100:5.1f} %")
    per = defaultdict(list)
    for s, e, n in ev:
        per[n].append(e - s)
    for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {n:40s} {len(d):5d} launches, mean {sum(d) / len(d) / 1e3:7.1f} us, on chip {sum(d) / wall * 100:5.1f} % of wall")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
