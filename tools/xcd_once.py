"""One resident call of B channels at the 7.1 / 6.15 s size (for rocprofv3 --pmc / diag builds).
    python3 tools/xcd_once.py B [reps] [three]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import Context, ConvPlan  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 640
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
three = len(sys.argv) > 3
ctx = Context(0)
rng = np.random.default_rng(1)
L, M = 420000, 295270
pitch = (L + 63) // 64 * 64
h = rng.standard_normal(M) * np.exp(-np.arange(M) / (M / 5.0))
host = rng.standard_normal((min(B, 64), pitch)).astype(np.float32)
d_x, d_y = ctx.malloc(B * pitch * 4), ctx.malloc(B * pitch * 4)
for c0 in range(0, B, host.shape[0]):
    ctx.h2d(d_x + c0 * pitch * 4, host[:min(host.shape[0], B - c0)])
plan = ConvPlan(ctx, h, L, "same", ws_channels=16)
if not three:
    plan.set_resident(True)
import time
for i in range(reps):
    ctx.synchronize()
    t0 = time.perf_counter()
    plan.execute_device(d_x, B, pitch, d_y, pitch)
    ctx.synchronize()
    print(f"call {i}: {(time.perf_counter() - t0) * 1e6:.0f} us", flush=True)
if not three:
    print(plan.resident_status())
plan.close()
ctx.free(d_x)
ctx.free(d_y)
ctx.close()
