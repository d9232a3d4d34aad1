#!/usr/bin/env python3
"""Per-kernel duration summary from a rocprofv3 rocpd database (results.db).
    python tools/rocpd_kernels.py gpurun_out/prof/x_results.db [skip_first_n]"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
by = defaultdict(list)
for name, s, e in rows:
    by[name.split("(")[0][:90]].append((s, e))
print(f"{'kernel':90s} {'n':>6s} {'mean_us':>9s} {'median':>8s} {'min':>8s}")
for name, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d = np.array([(e - s) / 1e3 for s, e in v[skip:]])
    if len(d):
        print(f"{name:90s} {len(d):6d} {d.mean():9.2f} {np.median(d):8.2f} {d.min():8.2f}")
