#!/usr/bin/env python3
"""Per-kernel VGPR / scratch / occupancy table from `python impulcifer-pip313_amd/build.py --force --verbose 2> log`:
python tools/kernel_resources.py log [substring ...]  (no substrings: every kernel with scratch or spills)"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    v, sc, sp, occ, lds = g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if (want and any(w in dn for w in want)) or (not want and (sc > 0 or sp > 0)):
        print(f"{dn[:120]:120s} VGPR {v:3d} scratch {sc:4d} spill {sp:3d} occ {occ} lds {lds}")
