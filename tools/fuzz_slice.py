#!/usr/bin/env python3
"""Seeded fuzz of the resident slice against the staged class path: random speaker sets (ipsilateral pairs complete and
broken, top layers, FC / WL alone), decay times, noise floors, levels, PCM width, alignment and decay stage on or off - every
result must be the staged path's, bit for bit, whatever the device flagged.
python tools/fuzz_slice.py [cases=40] [seed=0]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "impulcifer-pip313_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, d))
from test_resident_slice import assert_same_as_staged, synth_firs, synth_frames  # noqa: E402
from impulse_hip import _native  # noqa: E402
from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator  # noqa: E402
from impulse_hip.pipeline_slice import run_slice  # noqa: E402
from impulse_hip.resident_slice import Layout, ResidentSlice, _fir_taps  # noqa: E402

ALL = ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL", "WR", "TFL", "TFR", "TSL", "TSR", "TBL", "TBR"]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    warnings.simplefilter("ignore")
    flagged = staged = 0
    kinds = {}
    for c in range(cases):
        n_spk = int(rng.integers(1, 7))
        others = [s for s in ALL if s != "FL"]
        spk = ["FL"] + [others[i] for i in rng.permutation(len(others))[:n_spk - 1]]
        rng.shuffle(spk)
        M = int(rng.integers(1, 4))
        pcm16 = bool(rng.integers(0, 4) == 0)
        align = bool(rng.integers(0, 4) != 0)
        decay = None
        if rng.integers(0, 2):
            decay = {sp: float(rng.choice([0.15, 0.4, 1.0, 3.0])) for sp in spk if rng.integers(0, 2)} or None
        head_ms = float(rng.choice([1, 1, 1, 0.5, 2]))
        meas = []
        for m in range(M):
            fr = synth_frames(e, spk, int(rng.integers(1 << 30)), rt60=float(rng.uniform(0.1, 0.5)), level=float(rng.uniform(0.05, 0.8)),
                              noise_db=float(rng.uniform(-110, -60)))
            meas.append([(fr >> 16).astype(np.int16) if pcm16 else fr])
        layout = Layout(e, [(meas[0][0].shape[0], 2, spk)], dtype=np.int16 if pcm16 else np.int32)
        firs = synth_firs(layout.tasks, _fir_taps(fs), int(rng.integers(1 << 30)))
        rs = ResidentSlice(e, layout, max_measurements=M, head_ms=head_ms)
        rs.set_firs(firs)
        rs.set_alignment(align)
        rs.set_decay(decay)
        try:
            got = rs.run(meas)
            _, res = rs.slice.results()
            for m in range(M):
                want = run_slice(e, [((fs, meas[m][0]), spk)], firs=firs, head_ms=head_ms, decay=decay, align=align)
                assert_same_as_staged(got[m], want)
        except TypeError:
            # the reference raises when no decay time is defined (core/decay.py:367): the staged fallback did - count it
            kinds["TypeError"] = kinds.get("TypeError", 0) + 1
            rs.close()
            continue
        for f in res["flags"]:
            for bit, name in ((1, "knee"), (2, "knee_range"), (4, "keep_cap"), (8, "fade"), (16, "gain"), (32, "nonfinite"), (128, "decay"), (256, "align")):
                if int(f) & bit:
                    kinds[name] = kinds.get(name, 0) + 1
        flagged += int(np.count_nonzero(res["flags"] & _native.SLICE_REDO))
        staged += rs.stats["staged"]
        rs.close()
        if c % 10 == 9:
            print(f"  {c + 1} cases", flush=True)
    print(f"{cases} cases: every measurement identical to the staged path; measurements the last call of a case flagged: {flagged}, "
          f"run by the staged path in all: {staged}; flag kinds {kinds}")


if __name__ == "__main__":
    main()
