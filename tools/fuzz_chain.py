#!/usr/bin/env python3
"""Seeded fuzz of the deconvolution + FIR chain (imp_chain) against the oracle: random lengths (odd crop starts, every
column family up to 96 rows), peaks at the very start, in the middle and too close to the end, silent channels, heads
and fades of any size.  python tools/fuzz_chain.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, ROOT)
from impulse_hip import _native  # noqa: E402
from impulse_hip._native import ConvPlan, FirChain  # noqa: E402
from oracle.impulse_response import peak_index  # noqa: E402
from oracle.scipy_restated import fft_convolve, hann  # noqa: E402


def main(cases=60, seed=1):
    rng = np.random.default_rng(seed)
    ctx = _native.default_context()
    worst = 0.0
    for case in range(cases):
        L = int(rng.integers(3000, 700000))
        M = int(rng.integers(1, min(L, 400000)))
        n = int(rng.integers(50, min(L, 120000)))
        K = int(rng.integers(1, 20000))
        B = int(rng.integers(1, 5))
        head = int(rng.integers(0, min(n, 500)))
        fade_in = int(rng.integers(0, min(n, 600)))
        fade_out = int(rng.integers(0, min(n, 3000)))
        h = rng.standard_normal(M) * 1e-3 * np.exp(-np.arange(M) / max(M / 6.0, 1.0))
        h[int(rng.integers(0, M))] += 1.0
        x = (rng.standard_normal((B, L)) * 1e-4).astype(np.float32)
        for c in range(B):
            kind = int(rng.integers(0, 5))
            if kind == 0:
                x[c] = 0.0                                           # silent channel: peak 0
                continue
            at = {1: int(rng.integers(0, 20)), 2: int(rng.integers(0, L)), 3: L - 1 - int(rng.integers(0, 50)),
                  4: int(rng.integers(0, L))}[kind]
            x[c, at] += 1.0 if rng.random() < 0.5 else -1.0
            if kind == 4 and at + 7 < L:
                x[c, at + 1:at + 6] = x[c, at]                       # a plateau
        # unit-gain FIRs (a leading tap plus a tail of unit energy): fp32 noise of the response passes through with gain ~1
        firs = rng.standard_normal((B, K)) * np.exp(-np.arange(K) / max(K / 5.0, 1.0))
        firs /= np.maximum(np.linalg.norm(firs, axis=1, keepdims=True), 1e-30)
        firs[:, 0] += 1.0
        plan1 = ConvPlan(ctx, h, L, "same", ws_channels=B, fused=False)          # the deconvolution stage leaves chunk maxima
        plan5 = ConvPlan(ctx, firs, n, "full", ws_channels=B, fused=bool(case % 2))   # both K5 forms
        chain = FirChain(plan1, plan5, B, head, fade_in, fade_out)
        po = n + K - 1 + int(rng.integers(0, 9))
        d_x, d_out, d_pk = ctx.malloc(x.nbytes), ctx.malloc(B * po * 4), ctx.malloc(B * 8)
        ctx.h2d(d_x, x)
        chain.execute_device(d_x, L, d_out, po, d_pk)
        ctx.synchronize()
        y = np.empty((B, po), dtype=np.float32)
        pk = np.empty(B, dtype=np.int64)
        ctx.d2h(y, d_out)
        ctx.d2h(pk, d_pk)
        w = np.ones(n)
        if fade_in:
            w[:fade_in] *= hann(2 * fade_in)[:fade_in]
        if fade_out:
            w[n - fade_out:] *= hann(2 * fade_out)[fade_out:]
        for c in range(B):
            ir32 = fft_convolve(x[c].astype(np.float64), h, "same")
            # the peak search runs on the fp32 result of K1: compare it with the oracle ON THAT ROW when the fp64 answer is
            # a near tie (noise-level differences can reorder candidates), else with the fp64 oracle
            want_pk = peak_index(ir32)
            if int(pk[c]) != want_pk:
                a = np.abs(ir32)
                thr = 0.12589 * a.max()
                near = np.any(np.abs(a[max(want_pk - 2, 0):want_pk + 3] - thr) < 1e-5 * a.max()) or \
                    np.any(np.abs(a[max(int(pk[c]) - 2, 0):int(pk[c]) + 3] - thr) < 1e-5 * a.max())
                assert near, (case, c, int(pk[c]), want_pk, L, M)
                continue
            s0 = min(max(want_pk - head, 0), L - n)
            seg = ir32[s0:s0 + n]
            ref = fft_convolve(seg * w, firs[c], "full")
            # fp32 rounding of K1 scales with the largest value of the WHOLE circular convolution (the unit spike times the
            # unit tap of h), also when the 'same' window or the crop keeps only a quiet stretch of it
            scale = max(np.max(np.abs(ref)), 1.0)
            err = float(np.max(np.abs(y[c, :n + K - 1] - ref)) / scale)
            worst = max(worst, err)
            if err > 2e-6:
                d = np.abs(y[c, :n + K - 1] - ref)
                i = int(np.argmax(d))
                print("FAIL", dict(case=case, c=c, err=err, L=L, M=M, n=n, K=K, head=head, fade_in=fade_in, fade_out=fade_out,
                                   pk=int(pk[c]), s0=s0, rows=plan1.n1, rows5=plan5.n1, at=i, got=float(y[c, i]), want=float(ref[i]),
                                   scale=float(scale), seg_max=float(np.max(np.abs(seg)))), flush=True)
                raise SystemExit(1)
        chain.close()
        plan1.close()
        plan5.close()
        for p in (d_x, d_out, d_pk):
            ctx.free(p)
    print(f"{cases} chain cases ok; worst error {worst:.2e}")
    return worst


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
