#!/usr/bin/env python3
"""Resident slice against the staged path with every scalar printed (debug aid): python tools/slice_debug.py [small|c2|c3] [M]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_resident_slice import synth_firs, synth_frames  # noqa: E402

from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator  # noqa: E402
from impulse_hip.pipeline_slice import run_slice  # noqa: E402
from impulse_hip.resident_slice import Layout, ResidentSlice  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "small"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if config == "small":
    fs, dur, files_spk = 48000, 1.0, [["FL", "FR"], ["FC"]]
elif config == "c2":
    fs, dur, files_spk = 48000, 5.0, [["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"]]
else:
    from impulse_hip.constants import TRUEHD_13CH_ORDER
    fs, dur, files_spk = 96000, 5.0, [list(TRUEHD_13CH_ORDER)]
e = ImpulseResponseEstimator(min_duration=dur, fs=fs)
meas = [[synth_frames(e, spk, 1000 * m + 17 * k + (0xC2 if config != "small" else 0), rt60=0.2 + 0.05 * m) for k, spk in enumerate(files_spk)]
        for m in range(M)]
layout = Layout(e, [(fr.shape[0], 2, spk) for fr, spk in zip(meas[0], files_spk)])
rs = ResidentSlice(e, layout, max_measurements=M, keep_cap=int(2.5 * fs))
print("plan paired", rs.plan.paired, "rows", rs.plan.n1, "keep_cap", rs.keep_cap, "taps", rs.taps, "norm fft", rs.slice.norm_fft_len,
      "column", layout.column_len)
firs = synth_firs(layout.tasks, rs.taps, 5)
rs.set_firs(firs)
ctx = rs.ctx
item = layout.dtype.itemsize
d_rec = ctx.malloc(M * layout.samples * item)
for m, recs in enumerate(meas):
    ctx.h2d(d_rec + m * layout.samples * item, layout.pack(recs))
block = rs.execute_device(d_rec, M)
rows, res = rs.slice.results()
print("rows (peak, cut, len, knee, knee_flags, why):")
for r in rows:
    print("  ", r)
print("meas (keep, out_len, peak_db, gain_db, gain, flags):")
for r in res:
    print("  ", r)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for m in range(M):
        stages = {}
        stats = {}
        h, g = run_slice(e, [((fs, fr), sp) for fr, sp in zip(meas[m], files_spk)], firs=firs, stages=stages)
        print("staged", m, "gain", g, "keep", len(next(iter(stages["crop_tails"].values()))), "out", len(next(iter(stages["equalize"].values()))))
        out = block.host().reshape(-1, rs.out_pitch)
        for q, sp in enumerate(layout.speakers):
            for s, sd in enumerate(("left", "right")):
                b = m * rs.slice.rows + 2 * q + s
                n = int(res["out_len"][m])
                want = h.irs[sp][sd].peek()
                got = out[b, :n].astype(np.float64)
                same = got.shape == want.shape and np.array_equal(got, want)
                k = min(len(got), len(want))
                print("   ", sp, sd, "identical" if same else f"DIFFERENT shapes {got.shape} {want.shape} max|d| "
                      f"{np.max(np.abs(got[:k] - want[:k])) if k else None}")
