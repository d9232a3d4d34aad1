#!/usr/bin/env python3
"""Seeded fuzz of the device knee search (K7c, imp_decay_knees_device) against the host search over sampling rates,
row lengths and decay shapes the tests do not enumerate: every unflagged row must carry the host's knee and window.
python tools/fuzz_knees.py [batches=40] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import _native  # noqa: E402
from impulse_hip.decay import decay_params_rows  # noqa: E402
from impulse_hip.device_rows import DeviceBlock, Row, span  # noqa: E402


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = _native.default_context()
    tot = np.zeros(3, dtype=np.int64)
    rows_total = 0
    for batch in range(batches):
        fs = float(rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 44100.5]))
        B = int(rng.integers(1, 96))
        datas = []
        for _ in range(B):
            n = int(rng.integers(1, int(rng.choice([200, 5000, 3.5 * fs, 3.5 * fs, 8 * fs]))))
            t = np.arange(n) / fs
            kind = int(rng.integers(0, 8))
            if kind == 0:
                x = rng.standard_normal(n)                                  # no decay at all
            elif kind == 1:
                x = np.zeros(n)
                x[int(rng.integers(0, n))] = float(rng.uniform(0.01, 3))    # a spike
            else:
                env = np.zeros(n)
                for _ in range(int(rng.integers(1, 4))):
                    env += 10 ** (float(rng.uniform(-50, 0)) / 20) * 10 ** (-3.0 * t / float(10 ** rng.uniform(-2.5, 0.6)))
                x = rng.standard_normal(n) * env + rng.standard_normal(n) * 10 ** (float(rng.uniform(-140, -20)) / 20)
                if kind == 2:
                    x = np.concatenate((rng.standard_normal(int(rng.integers(1, n + 1))) * 1e-4, x))[:n]
                if kind == 3:
                    x *= 10 ** float(rng.uniform(-6, 3))                    # absolute level far from 1
            datas.append(x)
        pitch = (max(len(d) for d in datas) + 63) // 64 * 64
        flat = np.zeros((B, pitch), dtype=np.float32)
        for i, d in enumerate(datas):
            flat[i, :len(d)] = d
        block = DeviceBlock(ctx, flat.size)
        ctx.h2d(block.ptr, flat)
        rows = [Row(block, i * pitch, len(d)) for i, d in enumerate(datas)]
        host = decay_params_rows(rows, fs)
        base, offs, lens = span(rows)
        peak, knee, floor, win, flags = ctx.decay_knees_device(base, offs, lens, fs)
        for k, h in enumerate(host):
            if int(peak[k]) != int(h[0]):
                print(f"FAIL batch {batch} row {k} fs {fs} n {len(datas[k])}: peak {int(peak[k])} != {h[0]}")
                sys.exit(1)
            if flags[k] == 0 and (int(knee[k]), int(win[k])) != (int(h[1]), int(h[3])):
                print(f"FAIL batch {batch} row {k} fs {fs} n {len(datas[k])}: device ({int(knee[k])}, {int(win[k])}) host {h}")
                sys.exit(1)
            if flags[k] == 0 and not abs(float(floor[k]) - float(h[2])) <= 1e-9 * max(1.0, abs(float(h[2]))):
                print(f"FAIL batch {batch} row {k} fs {fs}: floor {float(floor[k])!r} host {h[2]!r}")
                sys.exit(1)
        tot += np.bincount(np.minimum(flags, 2), minlength=3)[:3]
        rows_total += B
        block.close()
        if batch % 10 == 9:
            print(f"  {batch + 1} batches, {rows_total} rows, flags {tot.tolist()}", flush=True)
    print(f"{rows_total} rows ok: {int(tot[0])} decided on the device, {int(tot[1])} inside a guard band, {int(tot[2])} outside the device path's shapes")


if __name__ == "__main__":
    main()
