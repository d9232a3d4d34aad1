"""Time of ConvPlan.set_filters (per-channel FIR spectra: pack, fp64 FFT, alpha/beta) and of the K2 / K6 calls of a slice:
python tools/setfilters_time.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import Context, ConvPlan  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
for n, K, B in ((32640, 9600, 16), (24000, 9600, 16), (65280, 19200, 26), (20000, 4800, 16), (10000, 4800, 16), (5000, 4800, 16), (3000, 1000, 16)):
    firs = rng.standard_normal((B, K))
    plan = ConvPlan(ctx, firs, n, "full", ws_channels=B)
    for _ in range(3):
        plan.set_filters(firs)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.set_filters(firs)
    ctx.synchronize()
    print(f"set_filters {B} x {K} taps, n = {n}, rows {plan.n1}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
    plan.close()
x = rng.standard_normal((2, 42239))
for _ in range(3):
    ctx.magnitude_db(x[0])
t0 = time.perf_counter()
for _ in range(20):
    ctx.magnitude_db(x[0])
print(f"magnitude_db n = 42239: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
ctx.close()
