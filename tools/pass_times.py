"""Per-pass HIP-event times of K1 with strictly serial launch groups: python tools/pass_times.py L M B [B ...]
One line per group size: average pass A / B / C time and the number of row workgroups (B x N1 / 2)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import Context, ConvPlan  # noqa: E402

PAIRED = os.environ.get("PAIRED", "0") == "1"
L, M = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(0)
rng = np.random.default_rng(0)
h = rng.standard_normal(M) * np.exp(-np.arange(M) / (M / 5.0))
pitch = (L + 63) // 64 * 64
for B in [int(v) for v in sys.argv[3:]]:
    host = rng.standard_normal((B, pitch)).astype(np.float32)
    d_x = [ctx.malloc(B * pitch * 4) for _ in range(8)]
    d_y = ctx.malloc(B * pitch * 4 + 256)
    for d in d_x:
        ctx.h2d(d, host)
    plan = ConvPlan(ctx, h, L, "same", ws_channels=B, paired=PAIRED)
    for i in range(40):
        plan.execute_device(d_x[i % 8], B, pitch, d_y, pitch)
    ctx.synchronize()
    plan.set_timing(1)
    for i in range(200):
        plan.execute_device(d_x[i % 8], B, pitch, d_y, pitch)
    ms, n = plan.get_timing()
    print(f"B={B:3d} rows {plan.n1} row workgroups {B * plan.n1 // 2:4d}: A {ms[0] / n * 1e3:6.1f}  B {ms[1] / n * 1e3:6.1f}  C {ms[2] / n * 1e3:6.1f} us "
          f"({n} groups)", flush=True)
    plan.close()
    for d in d_x + [d_y]:
        ctx.free(d)
ctx.close()
