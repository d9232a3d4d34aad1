#!/usr/bin/env python3
"""What the host link gives the slice's two transfers: 25.8 MB of pageable PCM up (imp_memcpy_h2d), 8.4 MB of float64
responses down into page-locked memory (imp_memcpy_d2h), alone and at the same time from two threads with a stream each.
python tools/probes/link_probe.py"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import _native  # noqa: E402

UP, DOWN, N = 25_809_280, 8_400_000, 40
up_ctx, down_ctx = _native.Context(_native.default_device()), _native.Context(_native.default_device())
src = np.random.default_rng(0).integers(-2 ** 31, 2 ** 31 - 1, UP // 4, dtype=np.int32)
pool = _native.PinnedPool(1024)
blk = pool.take(down_ctx, DOWN // 8)
dst_pinned = np.asarray(blk)[:DOWN // 8]
dst_paged = np.empty(DOWN // 8)
blk_up = pool.take(up_ctx, UP // 8)
src_pinned = np.asarray(blk_up)[:UP // 8].view(np.int32)
src_pinned[:] = src
d_up, d_down = up_ctx.malloc(UP), down_ctx.malloc(DOWN)
down_ctx.memset(d_down, 0, DOWN)
down_ctx.synchronize()


def up(n=N, a=src):
    t0 = time.perf_counter()
    for _ in range(n):
        up_ctx.h2d(d_up, a)
    return (time.perf_counter() - t0) / n * 1e3


def down(n=N, a=dst_pinned):
    t0 = time.perf_counter()
    for _ in range(n):
        down_ctx.d2h(a, d_down)
    return (time.perf_counter() - t0) / n * 1e3


up(3), down(3), down(3, dst_paged), up(3, src_pinned)
print(f"up, pageable source   {up():.3f} ms  ({UP / up() / 1e6:.1f} GB/s)")
print(f"up, pinned source     {up(N, src_pinned):.3f} ms")
print(f"down, pinned target   {down():.3f} ms  ({DOWN / down() / 1e6:.1f} GB/s)")
print(f"down, pageable target {down(N, dst_paged):.3f} ms")
out = {}
for label, a_up in (("pageable up", src), ("pinned up", src_pinned)):
    th = [threading.Thread(target=lambda: out.__setitem__("up", up(N, a_up))), threading.Thread(target=lambda: out.__setitem__("down", down(3 * N)))]
    [t.start() for t in th]
    [t.join() for t in th]
    print(f"both at once ({label}): up {out['up']:.3f} ms, down {out['down']:.3f} ms")

# the upload cut in two halves on two streams (two host threads)
up2_ctx = _native.Context(_native.default_device())
half = (UP // 8) * 4


def up_half(ctx, lo, hi, a, n):
    for _ in range(n):
        ctx.h2d(d_up + lo, a.view(np.uint8)[lo:hi])


def up_split(a, n=N):
    t0 = time.perf_counter()
    th = [threading.Thread(target=up_half, args=(up_ctx, 0, half, a, n)), threading.Thread(target=up_half, args=(up2_ctx, half, UP, a, n))]
    [t.start() for t in th]
    [t.join() for t in th]
    return (time.perf_counter() - t0) / n * 1e3


up_split(src, 3)
print(f"up in two halves on two streams, pageable: {up_split(src):.3f} ms; page-locked: {up_split(src_pinned):.3f} ms")
th = threading.Thread(target=lambda: out.__setitem__("down", down(3 * N)))
th.start()
both = up_split(src)
th.join()
print(f"the same with the download running: up {both:.3f} ms, down {out['down']:.3f} ms")
