"""GPU bring-up diagnostic for the convolution passes (run on the MI355X box via gpurun).

Compares each pass of the HIP path with the NumPy dataflow model (tests/model) and the final
result with scipy.signal.fftconvolve in float64.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "model"))

from scipy.signal import fftconvolve  # noqa: E402

import fourstep_model as fm  # noqa: E402
from impulse_hip import Context, ConvPlan  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    ctx = Context(0)
    rng = np.random.default_rng(1)
    ok = True
    # (L, M) chosen to hit every column radix: nfft 2^17 (R2=1) .. 2^21 (R2=16)
    cases = [(70001, 61000), (150000, 100001), (243635, 147635), (391270, 295270), (500000, 400000),
             (827965, 635965), (1048576, 1048576)]
    stage_checks = {(70001, 61000), (243635, 147635), (391270, 295270)}
    for (L, M) in cases:
        x = rng.standard_normal((3, L)).astype(np.float32)
        h = rng.standard_normal(M) * np.exp(-np.arange(M) / (M / 6.0))
        t0 = time.time()
        plan = ConvPlan(ctx, h, L, "same")
        t_plan = time.time() - t0
        print(f"L={L} M={M} nfft={plan.nfft} N1={plan.n1} ws_channels={plan.ws_channels} plan {t_plan:.2f}s", flush=True)
        if (L, M) in stage_checks:
            wsA = plan.debug_stage(x[:1], 0)[0]
            refA = fm.pass_a(x[0].astype(np.float64), plan.nfft)
            eA = rel(wsA, refA)
            alpha, beta = fm.plan_alpha_beta(h, plan.nfft)
            wsB = plan.debug_stage(x[:1], 1)[0]
            refB = fm.pass_b(refA, alpha, beta)
            eB = rel(wsB, refB)
            print(f"   pass A rel err {eA:.3e}   pass B rel err {eB:.3e}", flush=True)
            ok &= eA < 1e-5 and eB < 1e-5
        y = plan.execute(x)
        for b in range(x.shape[0]):
            ref = fftconvolve(x[b].astype(np.float64), h, "same")
            e = rel(y[b], ref)
            print(f"   ch{b} same rel err {e:.3e}", flush=True)
            ok &= e < 2e-6
        plan.close()
        if L <= 400000:
            planf = ConvPlan(ctx, h, L, "full")
            yf = planf.execute(x[:1])
            ref = fftconvolve(x[0].astype(np.float64), h, "full")
            e = rel(yf[0], ref)
            print(f"   full rel err {e:.3e} (out_len {planf.out_len})", flush=True)
            ok &= e < 2e-6
            planf.close()
    # interleaved frames + per-channel filters + odd crop offset
    L, M = 100000, 9600
    frames = rng.standard_normal((L, 4)).astype(np.float32)
    firs = rng.standard_normal((4, M)) * 0.1
    p = ConvPlan(ctx, firs[0], L, "same")
    yi = p.execute_interleaved(frames)
    for c in range(4):
        e = rel(yi[c], fftconvolve(frames[:, c].astype(np.float64), firs[0], "same"))
        print(f"   interleaved ch{c} rel err {e:.3e}", flush=True)
        ok &= e < 2e-6
    p.close()
    p = ConvPlan(ctx, firs, L, "full")
    yp = p.execute(frames.T.copy())
    for c in range(4):
        e = rel(yp[c], fftconvolve(frames[:, c].astype(np.float64), firs[c], "full"))
        print(f"   per-channel FIR ch{c} rel err {e:.3e}", flush=True)
        ok &= e < 2e-6
    p.close()
    # peak index + window smoke
    rows = [rng.standard_normal(5000).astype(np.float32), np.zeros(100, np.float32)]
    rows[0][1000] = 30.0
    idx, mx = ctx.peak_index(rows)
    print("   peak idx", idx, mx)
    ok &= idx[0] == 1000 and idx[1] == 0
    out = ctx.apply_window([np.ones(64, np.float32)], [dict(gain=2.0, fade_in=8, fade_out=8)])
    print("   window head", out[0][:9], "tail", out[0][-9:])
    # throughput sketch, device resident
    L, M, B = 391270, 295270, 16
    h = rng.standard_normal(M)
    plan = ConvPlan(ctx, h, L, "same")
    pin = (L + 1) & ~1
    d_x = ctx.malloc(B * pin * 4)
    d_y = ctx.malloc(B * pin * 4)
    ctx.h2d(d_x, rng.standard_normal((B, pin)).astype(np.float32))
    for _ in range(3):
        plan.execute_device(d_x, B, pin, d_y, pin)
    ctx.synchronize()
    plan.set_timing(True)
    t0 = time.time()
    K = 20
    for _ in range(K):
        plan.execute_device(d_x, B, pin, d_y, pin)
    ctx.synchronize()
    dt = time.time() - t0
    ms, n = plan.get_timing()
    print(f"   C2 B=16: {B * K / dt:.0f} IR/s wall; per group ms A/B/C = {[m / n for m in ms]}", flush=True)
    print("DIAG", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
