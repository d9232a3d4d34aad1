#!/usr/bin/env python3
"""per-kernel table of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [reps]"""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
reps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in csv.DictReader(open(f)):
    n, t = int(r["Calls"]), float(r["TotalDurationNs"])
    tot += t
    print(f"{r['Name'][:90]:90s} calls/rep {n / reps:6.1f} avg {float(r['AverageNs']) / 1e3:7.1f} us  per rep {t / reps / 1e3:8.1f} us")
print(f"total per rep {tot / reps / 1e3:.1f} us")
