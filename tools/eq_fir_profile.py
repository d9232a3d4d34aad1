#!/usr/bin/env python3
"""EQ curves -> minimum-phase FIRs (K12 -> K6) for the 16 channels of a 7.1 measurement, timed; run under
rocprofv3 --kernel-trace --stats for the per-kernel split: python tools/eq_fir_profile.py [fs=48000] [reps=20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip.frequency_response import FrequencyResponse  # noqa: E402
from impulse_hip.parallel_workers import process_equalization_batch  # noqa: E402

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
flat = len(sys.argv) > 3 and sys.argv[3] == "flat"        # no room curves: the error is minus the target (what bench.py's slice legs use)
common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
target = FrequencyResponse(name="target", frequency=common.copy(), raw=0)
rng = np.random.default_rng(0)
speakers = ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"]
tasks = [(sp, sd) for sp in speakers for sd in ("left", "right")]
room = {sp: {sd: FrequencyResponse("r", frequency=common.copy(), raw=0, error=np.cumsum(rng.standard_normal(len(common))) * 0.3)
             for sd in ("left", "right")} for sp in speakers}
if flat:
    room = None
for _ in range(3):
    process_equalization_batch(tasks, room, None, None, None, None, target, common, fs)
t0 = time.perf_counter()
for _ in range(reps):
    out = process_equalization_batch(tasks, room, None, None, None, None, target, common, fs)
dt = (time.perf_counter() - t0) / reps
print(f"{len(tasks)} FIRs of {len(out[0][2])} taps: {dt * 1e3:.3f} ms per call")
