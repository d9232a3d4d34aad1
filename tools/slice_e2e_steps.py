#!/usr/bin/env python3
"""Where a one-measurement resident slice call spends its host time (upload, launch, wait, collect, copy-out)."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402
from impulse_hip import _native  # noqa: E402
from impulse_hip.resident_slice import Layout, ResidentSlice  # noqa: E402

est = bench.make_estimator("c2")
rec, L, pitch, _ = bench.synth_recordings(est, 16, seed0=0xC2)
speakers = bench.SLICE_SPEAKERS["c2"]
frames = bench.measurement_frames(est, rec, L, speakers)
layout = Layout(est, [(frames.shape[0], 2, speakers)])
rs = ResidentSlice(est, layout, max_measurements=1)
firs = bench.synth_firs(16, rs.taps)
rs.set_firs(firs)
ctx = rs.ctx
d_rec = ctx.malloc(frames.nbytes)
acc = {}


def tick(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for it in range(12):
        if it == 2:
            acc.clear()
        t = time.perf_counter()
        rs.upload(d_rec, [frames])
        t = tick("upload", t)
        block = rs.execute_device(d_rec, 1)
        t = tick("launch", t)
        rows, meas = rs.slice.results()
        t = tick("wait", t)
        res = rs.collect(block, [[frames]], None)[0]
        t = tick("collect", t)
        res[0].to_host()
        t = tick("to_host", t)
print({k: round(v / 10 * 1e3, 3) for k, v in acc.items()}, "ms per measurement; total", round(sum(acc.values()) / 10 * 1e3, 3))
# pinned upload for comparison
import ctypes as C
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
assert hip.hipHostMalloc(C.byref(p), C.c_size_t(frames.nbytes), 0) == 0
C.memmove(p, frames.ctypes.data, frames.nbytes)
pinned = np.ctypeslib.as_array((C.c_int32 * frames.size).from_address(p.value)).reshape(frames.shape)
t0 = time.perf_counter()
for _ in range(10):
    ctx.h2d(d_rec, pinned)
print("pinned h2d", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms;", "pageable:")
t0 = time.perf_counter()
for _ in range(10):
    ctx.h2d(d_rec, frames)
print("pageable h2d", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms")
t0 = time.perf_counter()
for _ in range(10):
    C.memmove(p, frames.ctypes.data, frames.nbytes)
print("host memcpy to pinned", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms")
