#!/usr/bin/env python3
"""Phase timeline of the row pass (diagnostic build, -DIMP_PHASE_TRACE).

    python impulcifer-pip313_amd/build.py --variant trace IMP_PHASE_TRACE
    python tools/phase_trace.py [c2|c3|c5] [channels]

Every workgroup's wave 0 stamps s_memtime at the phase boundaries of rows_kernel; this prints the
median/percentile duration of each phase and the start-time histogram (dispatch rounds).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("IMPULSE_HIP_LIB", os.path.join(ROOT, "impulcifer-pip313_amd", "csrc", "libimpulse_hip_trace.so"))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, ROOT)

from impulse_hip import _native  # noqa: E402
import bench  # noqa: E402

PHASES = ["row loads land", "fft16+tw1+X1 write+barrier", "X1 read..fwd done+barrier", "partner exchange",
          "alpha/beta land + multiply", "barrier", "inverse to last barrier", "last fft16", "stores land"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else bench.GROUP_CHANNELS[wl]
    est = bench.make_estimator(wl)
    L = len(est) + 2 * est.fs if wl != "c5" else len(est)
    ctx = _native.default_context()
    plan = _native.ConvPlan(ctx, est.inverse_filter, L, "same", ws_channels=nch)
    plan.set_overlap(1)
    pitch = (L + 1) & ~1
    rng = np.random.default_rng(0)
    x = rng.standard_normal((nch, pitch)).astype(np.float32)
    d_x = ctx.malloc(x.nbytes)
    d_y = ctx.malloc(x.nbytes)
    ctx.h2d(d_x, x)
    for _ in range(int(os.environ.get("TRACE_REPEATS", "5"))):
        plan.execute_device(d_x, nch, pitch, d_y, pitch)
    ctx.synchronize()
    lib = _native.load_library()
    if not hasattr(lib, "imp_debug_phase_trace"):
        print("library without phase marks: ran the launches only (use under rocprofv3 --kernel-trace)")
        return
    n_wg = nch * plan.n1 // 2 + (nch if False else 0)
    n_wg = nch * ((plan.n1 // 2 + 8) // 8 * 8)          # grid is padded to a multiple of 8 tiles
    words = np.zeros(8192 * 16, dtype=np.uint64)
    lib.imp_debug_phase_trace.argtypes = [C.c_void_p, C.c_int64]
    assert lib.imp_debug_phase_trace(words.ctypes.data, words.size) == 0
    tr = words.reshape(8192, 16)
    live = tr[:, 9] > 0
    tr = tr[live].astype(np.int64)
    print(f"workload {wl}: N1={plan.n1} channels={nch} traced workgroups={len(tr)}")
    wall0 = tr[:, 12].min()
    start_us = (tr[:, 12] - wall0) / 100.0               # s_memrealtime: 100 MHz
    end_us = (tr[:, 13] - wall0) / 100.0
    cyc = (tr[:, 9] - tr[:, 0]).astype(float)
    dur_us = end_us - start_us
    ok = dur_us > 0
    clk_mhz = np.median(cyc[ok] / dur_us[ok])
    print(f"kernel span {end_us.max():.1f} us; s_memtime ticks per us ~ {clk_mhz:.0f}")
    print(f"workgroup duration us: p10 {np.percentile(dur_us,10):.1f}  median {np.median(dur_us):.1f}  p90 {np.percentile(dur_us,90):.1f}  max {dur_us.max():.1f}")
    hist, edges = np.histogram(start_us, bins=12)
    print("start-time histogram (us):", " ".join(f"{e:.0f}:{h}" for h, e in zip(hist, edges)))
    xcc = tr[:, 14] & 0xF
    print("workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
    for early in (True, False):
        sel = (start_us < 2.0) if early else (start_us >= 2.0)
        if sel.sum() == 0:
            continue
        print(f"--- workgroups that started {'in the first 2 us' if early else 'later'}: {sel.sum()}")
        for i, name in enumerate(PHASES):
            d = (tr[sel, i + 1] - tr[sel, i]) / clk_mhz
            print(f"  {name:32s} median {np.median(d):6.2f} us   p90 {np.percentile(d,90):6.2f}")
    ctx.free(d_x)
    ctx.free(d_y)


if __name__ == "__main__":
    main()
