"""Headphone compensation: the error curves of the headphone measurement that the EQ worker adds to every channel
(surface of reference core/pipeline_stages.py:353-482; logging and the plot left out).

The stereo sweep recording (FL to the left cup, FR to the right one) is deconvolved on the GPU like any other
recording; the two responses that matter - FL-left and FR-right - become one [2, n] matrix of dB curves (one batched
spectrum launch, one re-gridding map); the right curve is levelled with the gain that centres the left one between
100 Hz and 10 kHz, and since the target is flat zero the error of each ear is its levelled raw curve.
"""
import os

import numpy as np

from .frequency_response import FrequencyResponse, center_shifts
from .hrir import HRIR

HEADPHONES_FILENAME = 'headphones.wav'
HEADPHONES_FALLBACK_FILENAMES = (HEADPHONES_FILENAME, 'headphone.wav', 'hp.wav', 'compensation.wav')


def resolve_headphone_file(dir_path, headphone_file_path=None):
    """Which file the reference would open (:369-424): an explicit file (relative paths are taken from dir_path), a
    directory searched for the usual names and then for any WAV, else <dir_path>/headphones.wav; None if nothing exists."""
    chosen = None
    if headphone_file_path:
        path = os.path.normpath(headphone_file_path)
        if os.path.isdir(path):
            for name in HEADPHONES_FALLBACK_FILENAMES:
                if os.path.isfile(os.path.join(path, name)):
                    chosen = os.path.join(path, name)
                    break
            if chosen is None:
                wavs = [f for f in os.listdir(path) if f.lower().endswith('.wav')]
                chosen = os.path.join(path, wavs[0]) if wavs else None
        else:
            chosen = path if os.path.isabs(path) else os.path.join(dir_path, path)
    else:
        chosen = os.path.join(dir_path, HEADPHONES_FILENAME)
    if chosen is None or not os.path.exists(chosen):
        chosen = os.path.join(dir_path, HEADPHONES_FILENAME) if headphone_file_path else chosen
        if chosen is None or not os.path.exists(chosen):
            return None
    return chosen


def headphone_curves(hp_irs):
    """(left, right) FrequencyResponse of a headphone HRIR (FL-left, FR-right): raw levelled to the left ear's
    100 Hz-10 kHz mean, error = raw against a flat zero target (no min-mean recentring), target = 0."""
    from .room_correction import response_curves
    fs = hp_irs.fs
    grid, raws = response_curves([hp_irs.irs["FL"]["left"], hp_irs.irs["FR"]["right"]], fs)
    raws = raws - center_shifts(grid, raws[:1], [100, 10000])[0]        # left.center(); right.raw += gain
    out = []
    for name, raw in zip(("left", "right"), raws):
        fr = FrequencyResponse(name="Frequency response", frequency=grid.copy(), raw=raw, error=raw.copy(),
                               target=np.zeros(len(grid)))
        out.append(fr)
    return out[0], out[1]


def headphone_compensation(estimator, dir_path, headphone_file_path=None):
    """(left, right) error curves of the headphone measurement, or (None, None) when there is no file.  Writes
    headphone-responses.wav beside the measurement like the reference."""
    path = resolve_headphone_file(dir_path, headphone_file_path)
    if path is None:
        return None, None
    hp_irs = HRIR(estimator)
    hp_irs.open_recording(path, speakers=["FL", "FR"])
    hp_irs.write_wav(os.path.join(dir_path, "headphone-responses.wav"))
    return headphone_curves(hp_irs)
