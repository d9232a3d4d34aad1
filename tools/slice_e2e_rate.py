#!/usr/bin/env python3
"""End-to-end rate of the slice over a job of measurements (host PCM frames -> host float64 responses):
python tools/slice_e2e_rate.py [workers=0] [measurements=24]   (workers 0: the three-stage pipeline; n: n lanes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402

workers = (int(sys.argv[1]) if len(sys.argv) > 1 else 0) or None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
est = bench.make_estimator("c2")
rec, L, pitch, _ = bench.synth_recordings(est, 16, seed0=0xC2)
print(bench.slice_rate(est, rec, L, reps=reps, workers=workers))
