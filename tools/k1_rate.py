"""K1 throughput at an arbitrary (L, M): python tools/k1_rate.py L M [B=16] [lanes=3] [sets=40] [mono|pair|frames]
mono: one channel per transform, planar rows; pair: pair mode on the same planar rows; frames: pair mode on interleaved
stereo frames [L][2] (one 8-byte load per frame).
Prints IR/s, the plan's rows and the algorithmic GB/s (8 L bytes per IR)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import Context, ConvPlan  # noqa: E402

L, M = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sets = int(sys.argv[5]) if len(sys.argv) > 5 else 40
layout = sys.argv[6] if len(sys.argv) > 6 else "mono"
ctx = Context(0)
rng = np.random.default_rng(0)
h = rng.standard_normal(M) * np.exp(-np.arange(M) / (M / 5.0))
pitch = (L + 63) // 64 * 64
host = rng.standard_normal((B, pitch)).astype(np.float32)
d_x = [ctx.malloc(B * pitch * 4) for _ in range(sets)]
d_y = [ctx.malloc(B * pitch * 4 + 256) for _ in range(lanes)]
for d in d_x:
    ctx.h2d(d, host)
plan = ConvPlan(ctx, h, L, "same", ws_channels=lanes * B, paired=layout != "mono")
plan.set_overlap(lanes)
skew = ((M - 1) // 2) % 32
n = 0


def step():
    global n
    for s in range(sets):
        if layout == "frames":      # the same bytes read as B / 2 blocks of stereo frames [pitch][2]
            plan.execute_device_pairs(d_x[s], 0, B // 2, 2 * pitch, 1, 2, d_y[n % lanes] + 4 * skew, pitch)
        else:
            plan.execute_device(d_x[s], B, pitch, d_y[n % lanes] + 4 * skew, pitch)
        n += 1


for _ in range(5):
    step()
ctx.synchronize()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    step()
ctx.synchronize()
dt = (time.perf_counter() - t0) / (reps * sets)
print(f"L={L} M={M} B={B} lanes={lanes} {layout}: rows {plan.n1} nfft {plan.nfft}: {dt * 1e6:.1f} us per group = {B / dt / 1e3:.1f} k IR/s, "
      f"{B * 8 * L / dt / 1e9:.0f} GB/s algorithmic ({B * 8 * L / dt / 8e12 * 100:.1f} % of 8 TB/s)", flush=True)
plan.close()
for d in d_x + d_y:
    ctx.free(d)
ctx.close()
