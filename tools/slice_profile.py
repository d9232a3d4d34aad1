"""Where the end-to-end slice (bench.py `slice`) spends its time: python tools/slice_profile.py [reps=5]
cProfile of run_slice(...).to_host() on one C2 measurement (host-side view: the GPU work shows up as the waits)."""
import cProfile
import os
import pstats
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402
from impulse_hip.pipeline_slice import run_slice  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
est = bench.make_estimator("c2")
rec, L, pitch, _ = bench.synth_recordings(est, 16, 0xC2)
fs = est.fs
speakers = ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"]
tracks = np.zeros((2, 2 * fs + L * 8), dtype=np.float64)
for i in range(8):
    for ear in range(2):
        tracks[ear, 2 * fs + i * L: 2 * fs + (i + 1) * L] = rec[2 * i + ear, :L]
frames = np.ascontiguousarray(np.clip(np.rint(tracks.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32))
job = [((fs, frames), speakers)]
warnings.simplefilter("ignore")
run_slice(est, job)
run_slice(est, job)[0].to_host()
t0 = time.perf_counter()
for _ in range(reps):
    run_slice(est, job)[0].to_host()
print(f"{(time.perf_counter() - t0) / reps * 1e3:.2f} ms per measurement", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    run_slice(est, job)[0].to_host()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
