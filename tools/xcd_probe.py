"""First-contact and timing probe of the XCD-resident path (tools only; the tests cover parity).

    python tools/xcd_probe.py [--time]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, ROOT)
from impulse_hip import Context, ConvPlan  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b)) / np.max(np.abs(b)))


def main():
    ctx = Context(0)
    rng = np.random.default_rng(5)
    for L, M, mode, B in ((420000, 295270, "same", 16), (420000, 295270, "same", 1), (420000, 295270, "same", 23),
                          (300000, 400000, "same", 9), (300000, 280000, "full", 5)):
        x = rng.standard_normal((B, L)).astype(np.float32)
        h = rng.standard_normal(M) * np.exp(-np.arange(M) / (M / 5.0))
        plan = ConvPlan(ctx, h, L, mode, ws_channels=16)
        ref = plan.execute(x)
        print(f"L={L} M={M} {mode} B={B}: n1={plan.n1} available={plan.resident_available()}", flush=True)
        plan.set_resident(True)
        t0 = time.perf_counter()
        y = plan.execute(x)
        dt = time.perf_counter() - t0
        ab, seen, ticks = plan.resident_status()
        print(f"   resident: {dt*1e3:.2f} ms, aborted={ab} xcc_seen={seen:#x} wait_ticks={ticks} "
              f"max rel diff vs three-launch path {rel(y, ref.astype(np.float64)):.3e} finite={np.isfinite(y).all()}", flush=True)
        y2 = plan.execute(x)
        print(f"   rerun bit-identical: {np.array_equal(y, y2)}", flush=True)
        plan.close()
    if "--time" in sys.argv:
        L, M = 420000, 295270
        pitch = (L + 63) // 64 * 64
        h = rng.standard_normal(M) * np.exp(-np.arange(M) / (M / 5.0))
        for B in (16, 64, 640):
            host = rng.standard_normal((min(B, 64), pitch)).astype(np.float32)
            d_x = ctx.malloc(B * pitch * 4)
            d_y = ctx.malloc(B * pitch * 4)
            for c0 in range(0, B, host.shape[0]):
                ctx.h2d(d_x + c0 * pitch * 4, host[:min(host.shape[0], B - c0)])
            plan = ConvPlan(ctx, h, L, "same", ws_channels=48)
            for resident in (False, True):
                if resident:
                    plan.set_resident(True)
                else:
                    plan.set_overlap(3)
                for _ in range(3):
                    plan.execute_device(d_x, B, pitch, d_y, pitch)
                ctx.synchronize()
                reps = max(3, 1280 // B)
                t0 = time.perf_counter()
                for _ in range(reps):
                    plan.execute_device(d_x, B, pitch, d_y, pitch)
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / reps
                extra = plan.resident_status() if resident else ""
                print(f"B={B} resident={resident}: {dt*1e6:.1f} us per call = {B/dt/1e3:.1f} k IR/s {extra}", flush=True)
            plan.close()
            ctx.free(d_x)
            ctx.free(d_y)
    ctx.close()


if __name__ == "__main__":
    main()
