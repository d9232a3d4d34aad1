#!/usr/bin/env python3
"""Wall time of the auxiliary device kernels at C2/C3 shapes: K6 (minimum-phase FIR), K2 (magnitude
response), K3 (peak index).  python tools/bench_aux.py [repeats]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import _native  # noqa: E402


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = _native.default_context()
    rng = np.random.default_rng(0)
    for fs, n, B in ((48000, 9600, 16), (96000, 19200, 26)):
        gain = np.abs(1.0 + 0.3 * rng.standard_normal((B, n)))
        gain[:, -1] = 0.0
        ms = timeit(lambda: ctx.minphase_fir(gain, fs), reps)
        print(f"K6 minphase_fir  fs={fs} B={B} n={n}: {ms:.3f} ms per batch = {B / ms * 1e3:.0f} FIR/s")
    for n, B in ((32640, 16), (65280, 26)):
        x = rng.standard_normal((B, n))
        ms = timeit(lambda: ctx.magnitude_db(x), reps)
        print(f"K2 magnitude_db  B={B} n={n}: {ms:.3f} ms per batch")
    # K5: 16 x (32 640-sample response (*) 9 600-tap FIR) at C2, 26 x (65 280 (*) 19 200) at C3: device time of one
    # launch group (HIP events of the plan) and the wall time of the host-array call around it; IMPULSE_HIP_MIN_ROWS=16
    # reproduces the round-1 plan (smallest transform 131 072 points)
    from impulse_hip import ConvPlan
    for n, k, B in ((32640, 9600, 16), (65280, 19200, 26)):
        x = rng.standard_normal((B, n)).astype(np.float32)
        firs = rng.standard_normal((B, k)) * np.exp(-np.arange(k) / 800.0)
        plan = ConvPlan(ctx, firs, n, "full", ws_channels=B)
        plan.set_timing(1)
        ms = timeit(lambda: plan.execute(x), reps)
        kms, launches = plan.get_timing()
        ms_set = timeit(lambda: plan.set_filters(firs), reps)
        print(f"K5 equalize      B={B} n={n} taps={k}: nfft {plan.nfft} ({plan.n1} rows), device {sum(kms) / launches * 1e3:.1f} us "
              f"per launch group (A/B/C {kms[0] / launches * 1e3:.1f}/{kms[1] / launches * 1e3:.1f}/{kms[2] / launches * 1e3:.1f}), "
              f"{ms:.3f} ms per host call, set_filters {ms_set:.3f} ms")
        plan.close()
    for n, B in ((391270, 16),):
        rows = [rng.standard_normal(n).astype(np.float32) for _ in range(B)]
        ms = timeit(lambda: ctx.peak_index(rows), reps)
        print(f"K3 peak_index    B={B} n={n}: {ms:.3f} ms per batch (host rows uploaded each call)")


if __name__ == "__main__":
    main()
