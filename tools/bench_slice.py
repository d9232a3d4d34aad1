#!/usr/bin/env python3
"""Wall time of the whole hot-path slice (SURVEY 8d secondary metric) on one synthetic 7.1 measurement:
ingest (K1) -> crop_heads (K3/K4) -> crop_tails (host Lundeby + K4) -> FIR design (K2/K6) + equalize (K5)
-> decay (K8) -> normalize (K2).   python tools/bench_slice.py [repeats] [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from impulse_hip.pipeline_slice import run_slice  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
    est = bench.make_estimator("c2")
    fs, N = est.fs, len(est)
    speakers = ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"]
    cols, L, pitch, _ = bench.synth_recordings(est, 16, seed0=0xC2)
    # sweep_sequence geometry (core/impulse_response_estimator.py:153-232): 2 s lead, then one column per speaker
    tracks = np.zeros((2, 2 * fs + L * 8), dtype=np.float64)
    for i in range(8):
        for ear in range(2):
            tracks[ear, 2 * fs + i * L: 2 * fs + (i + 1) * L] = cols[2 * i + ear, :L]
    # the recording as a capture buffer / WAV data chunk holds it: interleaved 32-bit PCM frames
    frames = np.ascontiguousarray(np.clip(np.rint(tracks.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32))
    rec = [((fs, frames), speakers)]
    run_slice(est, rec)                                  # warm-up: plans, twiddles, K6 roots
    stages = {}
    t0 = time.perf_counter()
    for _ in range(reps):
        run_slice(est, rec)[0].to_host()                     # float64 host arrays out
    dt = (time.perf_counter() - t0) / reps
    print(f"slice: {dt * 1e3:.1f} ms per 16-channel measurement = {16 / dt:.0f} IR/s end to end (host arrays in, host arrays out)")
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        run_slice(est, rec)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
