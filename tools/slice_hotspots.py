"""cProfile of the slice's host side: python tools/slice_hotspots.py [stage=crop_tails|all] - where the Python time of a stage goes"""
import cProfile
import os
import pstats
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402
from impulse_hip.hrir import HRIR  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "crop_tails"
est = bench.make_estimator("c2")
rec, L, pitch, _ = bench.synth_recordings(est, 16, 0xC2)
fs = est.fs
speakers = ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"]
tracks = np.zeros((2, 2 * fs + L * 8), dtype=np.float64)
for i in range(8):
    for ear in range(2):
        tracks[ear, 2 * fs + i * L: 2 * fs + (i + 1) * L] = rec[2 * i + ear, :L]
frames = np.ascontiguousarray(np.clip(np.rint(tracks.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32))
warnings.simplefilter("ignore")
if stage == "all":                                            # the whole slice, as bench.py's `slice` times it
    from impulse_hip.pipeline_slice import run_slice
    for _ in range(3):
        run_slice(est, [((fs, frames), speakers)])[0].to_host()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        run_slice(est, [((fs, frames), speakers)])[0].to_host()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(40)
    sys.exit(0)
hs = []
for _ in range(12):
    h = HRIR(est)
    h.open_recording_frames(fs, frames, speakers)
    h.crop_heads(head_ms=1)
    hs.append(h)
hs[0].crop_tails()
hs[1].crop_tails()
pr = cProfile.Profile()
pr.enable()
for h in hs[2:]:
    h.crop_tails()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
