#!/usr/bin/env python3
"""Leak / stability soak: plans, segment sets and the auxiliary kernels created and destroyed in a loop;
device memory before and after must agree.  python tools/soak.py [iterations]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import _native  # noqa: E402


def free_bytes():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipDeviceSynchronize()
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = _native.default_context()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((4, 60000)).astype(np.float32)
    h = rng.standard_normal((4, 9600))
    gain = np.abs(1 + 0.1 * rng.standard_normal((4, 2400)))
    gain[:, -1] = 0
    rows = [rng.standard_normal(30000) * np.exp(-np.arange(30000) / 4000.0) for _ in range(4)]

    pk1 = _native.ConvPlan(ctx, h[0], 60000, "same", paired=True)          # K1 of the slices below (kept: a slice borrows it)
    pcm = (rng.standard_normal((240000, 2)) * 2 ** 20).astype(np.int32)
    pcm[5000, :] = 2 ** 29
    pcm[125000, :] = 2 ** 29
    hfir = rng.standard_normal((4, 9600)) * 0.01

    def once():
        p = _native.ConvPlan(ctx, h, 60000, "full")
        y = p.execute(x)
        p.close()
        p = _native.ConvPlan(ctx, h[0], 60000, "same")
        p.execute_interleaved(np.ascontiguousarray(x.T))
        p.close()
        s = _native.SegSet(ctx, rows)
        s.range_means([(i, 0, 1000) for i in range(4)])
        s.close()
        ctx.minphase_fir(gain, 48000)
        ctx.magnitude_db(np.stack(rows))
        ctx.peak_index(rows)
        ctx.xcorr_argmax([r[:1440] for r in rows[:2]], [r[:1440] for r in rows[2:]])
        ctx.decay_times(rows, [0] * 4, [20000] * 4, [-80.0] * 4, [1440] * 4, 48000)
        # round 3: pair-mode and three-launch plans, a chain with its events and buffer sets on a second context
        p = _native.ConvPlan(ctx, h[0], 60000, "same", paired=True)
        p.execute(x)
        p.close()
        p1 = _native.ConvPlan(ctx, np.concatenate([h[0], np.zeros(30000)]), 60000, "same", ws_channels=8, fused=False)
        p1.set_overlap(2)
        tail = _native.Context(0)
        p5 = _native.ConvPlan(tail, h, 20000, "full", ws_channels=4)
        ch = _native.FirChain(p1, p5, 4, 48, 48, 200)
        d_x, d_o = ctx.malloc(x.nbytes), ctx.malloc(4 * 29600 * 4)
        ctx.h2d(d_x, x)
        for _ in range(3):
            ch.execute_device(d_x, 60000, d_o, 29600)
        ctx.synchronize()
        tail.synchronize()
        ch.close()
        p1.close()
        p5.close()
        ctx.free(d_x)
        ctx.free(d_o)
        tail.close()
        # the device knee search (its segment block comes from the pool), the staged window tables, the peak-only spectra
        from impulse_hip.device_rows import DeviceBlock, Row, span
        blk = DeviceBlock(ctx, 4 * 30016)
        flat = np.zeros((4, 30016), dtype=np.float32)
        for i, r in enumerate(rows):
            flat[i, :30000] = r
        ctx.h2d(blk.ptr, flat)
        drows = [Row(blk, i * 30016, 30000) for i in range(4)]
        base, offs, lens = span(drows)
        ctx.decay_knees_device(base, offs, lens, 48000)
        ctx.apply_window_device(base, offs, base, offs, lens, [dict(fade_out=100)] * 4)
        ctx.magnitude_db_sum_peak_device(base, offs, lens, [0, 1, 0, 1], 2, 30000)
        blk.close()
        # round 4: a resident slice (imp_slice) made, run and destroyed; FIRs left on the device; the fp64 tile transform
        sl = _native.Slice(pk1, [0, 120000], [48, 48], 2, 32, 48, 200, 9600, 20000, 48000, max_measurements=2)
        sl.set_firs(hfir)
        d_rec, d_out = ctx.malloc(pcm.nbytes * 2), ctx.malloc(2 * 4 * sl.out_len_max * 4 + 1024)
        ctx.h2d(d_rec, pcm)
        ctx.h2d(d_rec + pcm.nbytes, pcm)
        sl.execute_device(d_rec, pcm.size, 2, d_out, sl.out_len_max)
        sl.results()
        # the decay stage's tables, the float64 pack and a page-locked result block: made, used and released per iteration
        sl.set_decay([0.3, np.nan, 0.3, 0.3])
        sl.set_alignment([(0, 1)], [-1, 0], 0, 1440)
        sl.execute_device(d_rec, pcm.size, 2, d_out, sl.out_len_max)
        d_packed = ctx.malloc(2 * 4 * sl.out_len_max * 8)
        sl.pack_f64(d_out, sl.out_len_max, 2, d_packed, 4 * sl.out_len_max)
        _, meas = sl.results()
        pool = _native.PinnedPool(64)
        blk = pool.take(ctx, 4 * sl.out_len_max)
        ctx.d2h(np.asarray(blk)[:4 * max(int(meas["out_len"][0]), 1)], d_packed)
        del blk
        pool.close()
        sl.close()
        ctx.free(d_rec)
        ctx.free(d_out)
        ctx.free(d_packed)
        ctx.fft64(np.ones((2, 19200), dtype=np.complex128))
        return float(y[0, 0])

    for _ in range(10):
        once()
    before = free_bytes()
    import time
    t0 = time.perf_counter()
    import faulthandler
    for it in range(iters):
        # a call that does not come back within a minute: print where the host is waiting and leave
        faulthandler.dump_traceback_later(60, exit=True)
        once()
        faulthandler.cancel_dump_traceback_later()
        if it % 50 == 49:
            print(f"  {it + 1} iterations, {time.perf_counter() - t0:.0f} s", flush=True)
    after = free_bytes()
    print(f"{iters} iterations: free device memory {before / 2**20:.1f} MiB -> {after / 2**20:.1f} MiB (delta {(before - after) / 2**20:.2f} MiB)")
    assert before - after < 8 * 2 ** 20, "device memory leak"
    print("soak ok")


if __name__ == "__main__":
    main()
