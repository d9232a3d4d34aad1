#!/usr/bin/env python3
"""Wide seeded fuzz of the convolution plans against the oracle: every plan kind (three launches, fused
overlap-save, pair mode), every column shape, both modes, all input
layouts (planar with odd pitches, interleaved frames, PCM16/32 columns), per-channel filters, lanes, and a few
overlap-add sizes.  python tools/fuzz_conv.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, ROOT)
from impulse_hip import _native  # noqa: E402
from oracle.scipy_restated import fft_convolve  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = _native.default_context()
    worst = 0.0
    kinds = {"pairs": 0, "fused": 0, "three-launch": 0}
    sizes = [(1, 50), (50, 3000), (3000, 60000), (60000, 250000), (250000, 700000), (700000, 1100000)]
    for case in range(cases):
        lo, hi = sizes[int(rng.integers(0, len(sizes)))]
        L = int(rng.integers(lo, hi))
        lo, hi = sizes[int(rng.integers(0, len(sizes) - 1))]
        M = int(rng.integers(lo, hi))
        if case % 40 == 39:                                     # an overlap-add size now and then
            L, M = int(rng.integers(2_200_000, 2_600_000)), int(rng.integers(1, 400_000))
        mode = "same" if rng.random() < 0.5 else "full"
        B = int(rng.integers(1, 5))
        per_channel = rng.random() < 0.25
        h = rng.standard_normal((B if per_channel else 1, M)) * np.exp(-np.arange(M) / max(M / 5.0, 1.0))
        layout = ("planar", "frames", "pcm16", "pcm32", "device")[case % 5]
        filt = h if per_channel else h[0]
        ws = B if layout in ("pcm16", "pcm32", "frames", "device") else int(rng.integers(1, B + 1))
        kind = ("default", "three-launch", "pairs")[int(rng.integers(0, 3))]    # default fuses short filters
        plan = _native.ConvPlan(ctx, filt, L, mode, ws_channels=ws, fused=kind != "three-launch",
                                paired="auto" if kind == "pairs" and not per_channel else False)
        kinds[("pairs" if plan.paired else "fused" if plan.fused else "three-launch")] += 1
        if layout == "planar":
            x = rng.standard_normal((B, L)).astype(np.float32)
            y = plan.execute(x)
            ref_in = x.astype(np.float64)
        elif layout == "frames":
            fr = rng.standard_normal((L, B)).astype(np.float32)
            y = plan.execute_interleaved(fr)
            ref_in = fr.T.astype(np.float64)
        elif layout == "device":                                 # odd pitch, unaligned base, overlapped lanes
            pitch_in, pitch_out = L + int(rng.integers(0, 7)), plan.out_len + int(rng.integers(0, 7))
            x = rng.standard_normal((B, pitch_in)).astype(np.float32)
            skew = int(rng.integers(0, 5))
            d_x = ctx.malloc(x.nbytes + 64)
            d_y = ctx.malloc(B * pitch_out * 4 + 64)
            ctx.h2d(d_x + 4 * skew, x)
            plan.execute_device(d_x + 4 * skew, B, pitch_in, d_y + 4 * skew, pitch_out)
            ctx.synchronize()
            out = np.empty((B, pitch_out), dtype=np.float32)
            ctx.d2h(out, d_y + 4 * skew)
            ctx.free(d_x)
            ctx.free(d_y)
            y = out[:, :plan.out_len]
            ref_in = x[:, :L].astype(np.float64)
        else:
            bits = 16 if layout == "pcm16" else 32
            lead = int(rng.integers(0, 40))
            fr = rng.integers(-2 ** (bits - 1), 2 ** (bits - 1), size=(lead + L + 3, B), dtype=np.int64).astype(
                np.int16 if bits == 16 else np.int32)
            y = plan.execute_pcm_columns(fr, [lead])[0]
            ref_in = fr[lead:lead + L].T.astype(np.float64) / 2.0 ** (bits - 1)
        nfft = plan.nfft
        plan.close()
        for b in range(B):
            ref = fft_convolve(ref_in[b], h[b if per_channel else 0], mode)
            # a window of a few samples out of a long filter's output can sit far below the transform's rounding
            # noise, which scales with |x| |h|: the error is measured against the larger of the two scales
            scale = max(np.max(np.abs(ref)), 0.05 * np.max(np.abs(ref_in[b])) * np.max(np.abs(h[b if per_channel else 0])))
            e = float(np.max(np.abs(np.asarray(y[b], np.float64) - ref)) / max(scale, 1e-300))
            tol = 1e-6 * (2 if layout == "pcm32" else 1) * (1.5 if (mode == "same" and M > 4 * L) else 1) * (2 if L > 2_000_000 else 1)
            worst = max(worst, e / tol)
            if e > tol:
                print(f"FAIL case {case}: {kind} {layout} L={L} M={M} {mode} B={B} per_channel={per_channel} nfft={nfft} b={b}: {e:.3e} > {tol:.1e}")
                sys.exit(1)
    print(f"{cases} cases ok ({kinds}); worst error / tolerance = {worst:.2f}")


if __name__ == "__main__":
    main()
