"""Flag statistics and timing of the device knee search (imp_decay_knees_device) against the host search:
python tools/knee_stats.py [rows=512] [fs=48000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import _native  # noqa: E402
from impulse_hip.decay import decay_params_rows, knee_indices_rows  # noqa: E402
from impulse_hip.device_rows import DeviceBlock, Row, span  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
rng = np.random.default_rng(3)
ctx = _native.default_context()
n = int(0.68 * fs)
t = np.arange(n) / fs
flat = np.zeros((B, (n + 63) // 64 * 64), dtype=np.float32)
for b in range(B):
    x = rng.standard_normal(n) * 10 ** (-3.0 * t / rng.uniform(0.1, 0.6)) + rng.standard_normal(n) * 10 ** (rng.uniform(-90, -50) / 20)
    x[48] = 2.0
    flat[b, :n] = x
block = DeviceBlock(ctx, flat.size)
ctx.h2d(block.ptr, flat)
rows = [Row(block, b * flat.shape[1], n) for b in range(B)]
base, offs, lens = span(rows)
peak, knee, floor, win, flags = ctx.decay_knees_device(base, offs, lens, fs)
host = decay_params_rows(rows, fs)
bad = [b for b in range(B) if not flags[b] and (int(knee[b]) != int(host[b][1]) or int(win[b]) != int(host[b][3]))]
dfl = max(abs(float(floor[b]) - float(host[b][2])) for b in range(B) if not flags[b])
print(f"{B} rows: flags {np.bincount(flags, minlength=3).tolist()} (0 clear, 1 guard band, 2 range), mismatches {len(bad)}, "
      f"max |floor - host floor| {dfl:.2e} dB")
for name, fn, nb in (("device (16 rows)", lambda: knee_indices_rows(rows[:16], fs), 16),
                     ("host   (16 rows)", lambda: decay_params_rows(rows[:16], fs), 16),
                     (f"device ({B} rows)", lambda: knee_indices_rows(rows, fs), B),
                     (f"host   ({B} rows)", lambda: decay_params_rows(rows, fs), B)):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:22s} {dt * 1e3:8.3f} ms per call, {dt / nb * 1e6:7.1f} us per row")
