#!/usr/bin/env python3
"""Throughput of the device-resident slice (imp_slice) at BASELINE C2 / C3 with the recordings resident in HBM:
python tools/slice_resident_rate.py [c2|c3] [streams=2] [M=8] [calls=20] [ring=3] [decay target RT60 in s, 0 = stage off] [alignment 1|0]
One host thread feeds `streams` slices (one context = one stream each) round robin; every call = M measurements."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
M = int(sys.argv[3]) if len(sys.argv) > 3 else 8
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ring = int(sys.argv[5]) if len(sys.argv) > 5 else 3
decay = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
align = bool(int(sys.argv[7])) if len(sys.argv) > 7 else True
est = bench.make_estimator(workload)
B_meas = bench.WORKLOADS[workload][2]
rec, L, pitch, _ = bench.synth_recordings(est, B_meas, seed0=0xC2)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    team = bench.SliceTeam(est, rec, L, n_streams=n_streams, M=M, ring=ring, align=align)
    print(f"{workload}: {team.describe()}" + (f", decay stage on every row (target {decay} s)" if decay > 0 else ""))
    if decay > 0:
        for ln in team.lanes:
            ln["rs"].set_decay(decay)
    for _ in range(2):
        team.step()
    team.sync()
    t0 = time.perf_counter()
    for _ in range(calls):
        team.step()
    issued = time.perf_counter() - t0
    team.sync()
    dt = time.perf_counter() - t0
    irs = calls * n_streams * M * team.rows
    print(f"{irs} IRs in {dt * 1e3:.1f} ms = {irs / dt / 1e3:.1f} k IR/s; {dt / (calls * n_streams) * 1e3:.3f} ms per call of {M} measurements; "
          f"flags {team.flags()}; host thread busy issuing {issued / dt * 100:.0f} % of that ({issued / (calls * n_streams) * 1e3:.3f} ms per call)")
    team.release()
