"""Whole-column accuracy of K1 for one synthetic channel of a workload: python tools/column_error.py c2|c3|c5 [pair]
Prints max |dA| / max |A| on the un-cropped column (and on the cropped window) against the float64 oracle, the same
figures for pocketfft run in single precision, and the plan's rows.  Plan variants through the environment:
IMPULSE_HIP_POW2_ONLY=1, IMPULSE_HIP_NO_WRAP=1, IMPULSE_HIP_MIN_ROWS=n."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402
from impulse_hip import Context, ConvPlan  # noqa: E402
from oracle.estimator import estimate  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
paired = len(sys.argv) > 2 and sys.argv[2] == "pair"
est = bench.make_estimator(wl)
fs = est.fs
rec, L, pitch, delays = bench.synth_recordings(est, 2, seed0=0xC2, column=(len(est) if wl in ("c4", "c5") else None))
inv = np.asarray(est.inverse_filter, dtype=np.float64)
ctx = Context(0)
plan = ConvPlan(ctx, inv, L, "same", fused=False, paired=paired)
y = plan.execute(rec[:, :L])
rows, nfft = plan.n1, plan.nfft
plan.close()
ctx.close()
floor = bench.fp32_fft_floor(est, rec[0], L)
for c in range(2):
    ref = estimate(rec[c, :L].astype(np.float64), inv)
    pk = int(np.argmax(np.abs(ref)))
    sl = slice(pk - fs // 1000, pk - fs // 1000 + int(0.68 * fs))
    out = []
    for s_ in (slice(None), sl):
        A, R = np.abs(np.fft.rfft(y[c][s_].astype(np.float64))), np.abs(np.fft.rfft(ref[s_]))
        out.append(float(np.max(np.abs(A - R)) / np.max(R)))
    t = float(np.max(np.abs(y[c] - ref)) / np.max(np.abs(ref)))
    rms = float(np.sqrt(np.mean((y[c] - ref) ** 2)) / np.max(np.abs(ref)))
    print(f"{wl} ch{c} rows {rows} nfft {nfft}{' pair' if paired else ''}: whole column {out[0]:.3e}  cropped {out[1]:.3e}  time max {t:.3e} rms {rms:.3e}"
          f"  (pocketfft fp32 whole column {floor:.3e})", flush=True)

if os.environ.get("COLUMN_ERROR_BINS") == "1":
    c = 0
    ref = estimate(rec[c, :L].astype(np.float64), inv)
    Y, Rf = np.fft.rfft(y[c].astype(np.float64)), np.fft.rfft(ref)
    A, R = np.abs(Y), np.abs(Rf)
    err = np.abs(A - R) / np.max(R)
    cerr = np.abs(Y - Rf) / np.max(R)
    top = np.argsort(err)[-12:][::-1]
    print("median |dA|", float(np.median(err)), "median |dY|", float(np.median(cerr)), "max |dY|", float(cerr.max()))
    for k in top:
        print(f"  bin {int(k):8d}  f {k * fs / L:10.2f} Hz  |dA| {err[k]:.3e}  |dY| {cerr[k]:.3e}  A/max {A[k] / R.max():.3e}")
    # time-domain error: where is it, and is it periodic?
    d = y[c].astype(np.float64) - ref
    top_t = np.argsort(np.abs(d))[-8:][::-1]
    print("largest time-domain errors at samples", [int(i) for i in top_t], "values", [float(f"{d[i] / np.max(np.abs(ref)):.2e}") for i in top_t])
    seg = np.abs(d).reshape(-1)[: (L // 4096) * 4096].reshape(-1, 4096)
    print("rms error per 4096-sample block (first 12, x1e-10):", np.round(np.sqrt((seg ** 2).mean(axis=1))[:12] / np.max(np.abs(ref)) * 1e10, 2))
