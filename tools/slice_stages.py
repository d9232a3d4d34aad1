"""Wall time of each stage of the hot-path slice (bench.py `slice`), device drained between stages:
python tools/slice_stages.py [reps=10]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402
from impulse_hip import _native  # noqa: E402
from impulse_hip.frequency_response import FrequencyResponse  # noqa: E402
from impulse_hip.hrir import HRIR  # noqa: E402
from impulse_hip.parallel_workers import process_equalization_batch  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
ctx = _native.default_context()
acc = {}


def lap(name, t0):
    ctx.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


for r in range(reps + 2):
    if r == 2:
        acc.clear()
    t = time.perf_counter()
    h = HRIR(est)
    h.open_recording_frames(fs, frames, speakers)
    t = lap("ingest (H2D + K1)", t)
    h.crop_heads(head_ms=1)
    t = lap("crop_heads (K3 + K4)", t)
    h.crop_tails()
    t = lap("crop_tails (K7 + K4)", t)
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    target = FrequencyResponse(name="target", frequency=common.copy(), raw=0)
    tasks = [(sp, sd) for sp, pair in h.irs.items() for sd in pair]
    firs = process_equalization_batch(tasks, None, None, None, None, None, target, common, fs, on_device=True)   # as run_slice does
    t = lap("EQ curves + FIR design (K12, K6; FIRs stay on the device)", t)
    h.equalize_channels({(sp, sd): fir for sp, sd, fir in firs})
    t = lap("equalize (K5)", t)
    h.normalize(peak_target=-0.1)
    t = lap("normalize (K2 + K4)", t)
    h.to_host()
    t = lap("to_host (D2H)", t)
    del h
    t = lap("release", t)
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:58s} {v / reps * 1e3:7.3f} ms")
print(f"{'total':58s} {tot / reps * 1e3:7.3f} ms")
from impulse_hip.impulse_response import _k5_plans  # noqa: E402
print("K5 plan shapes (ctx, n, taps, channels):", [k[1:] for k in _k5_plans.plans])
