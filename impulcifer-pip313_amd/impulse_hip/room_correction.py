"""Room correction: per speaker-ear equalisation error curves from room measurements
(surface of reference core/room_correction.py:36-461; plotting left out).

The measurements are deconvolved on the GPU through HRIR.open_recording (all tracks/columns of a
file in one batch); cropping uses the device peak search; the responses of all channels become one [B, n]
matrix of dB curves (one batched spectrum launch, one re-gridding map), levelled, compared with the target and masked
row-wise; the fractional-octave smoothing of the generic path runs on the device (K12).
"""
import os
import re
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .audio_io import read_wav
from .constants import SPEAKER_NAMES
from .frequency_response import FrequencyResponse
from .hrir import HRIR, _hann
from .impulse_response import ImpulseResponse

SPEAKER_LIST_PATTERN = r'([A-Z]{2,3}(,[A-Z]{2,3})*)'
IR_ROOM_SPL = {sp: {'left': 0.0, 'right': 0.0} for sp in SPEAKER_NAMES}


@dataclass(frozen=True)
class RoomMeasurement:
    file_path: str
    speakers: tuple
    side: Optional[str]


@dataclass(frozen=True)
class RoomMeasurementDiscovery:
    measurements: tuple
    generic_path: Optional[str]
    mic_calibration_path: Optional[str]
    target_path: Optional[str]
    responses_path: str


def discover_room_measurements(dir_path):
    """room-<SPK[,SPK...]>[-left|-right].wav files plus the optional room.wav, mic calibration and
    target files of a measurement folder; nothing is opened."""
    pattern = rf'^room-{SPEAKER_LIST_PATTERN}(-(left|right))?\.wav$'
    found = []
    for file_name in os.listdir(dir_path):
        if re.match(pattern, file_name) is None:
            continue
        speakers = re.search(SPEAKER_LIST_PATTERN, file_name)
        if speakers is None:
            continue
        side = re.search(r'(left|right)', file_name)
        found.append(RoomMeasurement(os.path.join(dir_path, file_name), tuple(speakers[0].split(',')),
                                     side[0] if side is not None else None))

    def existing(*names):
        for nm in names:
            p = os.path.join(dir_path, nm)
            if os.path.isfile(p):
                return p
        return None

    return RoomMeasurementDiscovery(tuple(found), existing('room.wav'),
                                    existing('room-mic-calibration.csv', 'room-mic-calibration.txt'),
                                    existing('room-target.csv'), os.path.join(dir_path, 'room-responses.wav'))


def _correction_limit_mask(frequency, limit):
    start = np.argmax(frequency > limit / 2)
    end = np.argmax(frequency > limit)
    return np.concatenate([np.ones(start if start > 0 else 0), _hann(end - start), np.zeros(len(frequency) - end)])


def _open_curve(estimator, path, default_flat):
    if path is not None and os.path.isfile(path):
        fr = FrequencyResponse.read_csv(path)
        fr.interpolate(f_step=1.01, f_min=10, f_max=estimator.fs / 2)
        fr.center()
        return fr
    if not default_flat:
        return None
    fr = FrequencyResponse(name='room-target')
    fr.raw = np.zeros(fr.frequency.shape)
    fr.interpolate(f_step=1.01, f_min=10, f_max=estimator.fs / 2)
    return fr


def open_room_target(estimator, dir_path, target=None):
    if target is None:
        target = discover_room_measurements(dir_path).target_path or os.path.join(dir_path, 'room-target.csv')
    return _open_curve(estimator, target, default_flat=True)


def open_mic_calibration(estimator, dir_path, mic_calibration=None):
    if mic_calibration is None:
        mic_calibration = discover_room_measurements(dir_path).mic_calibration_path
    elif not os.path.isfile(mic_calibration):
        raise FileNotFoundError(f'Room mic calibration file doesn\'t exist at "{mic_calibration}"')
    return _open_curve(estimator, mic_calibration, default_flat=False)


def open_room_measurements(estimator, dir_path, debug=False):
    rir = HRIR(estimator)
    for m in discover_room_measurements(dir_path).measurements:
        rir.open_recording(m.file_path, list(m.speakers), side=m.side, debug=debug)
    return rir


def response_curves(irs, fs):
    """ImpulseResponse.frequency_response() of several responses at once -> (grid 10 Hz..fs/2, raw [B, n]).  Equal-length
    responses share one batched spectrum launch (K2) and one re-gridding map."""
    from .audio_io import magnitude_responses
    from .frequency_response import _Regrid, generate_frequencies
    grid = generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    raws = np.zeros((len(irs), len(grid)))
    by_len = {}
    for i, ir in enumerate(irs):
        by_len.setdefault(len(ir.data), []).append(i)
    for n, idx in by_len.items():
        if n < 2:
            continue                                           # reference: flat curve for degenerate responses
        f, m = magnitude_responses(np.stack([irs[i].data for i in idx]), fs)
        wanted = (fs / 2) / 4.0
        step = 1 if (wanted < 2 or len(f) < 2) else (int(round(len(f) / wanted)) or 1)
        sel = slice(1, None, step) if len(f[1::step]) else slice(1, None)
        if len(f[sel]) < 2:
            continue
        raws[idx] = _Regrid(f[sel], grid)(m[:, sel])
    return grid, raws


def _curve_objects(names, grid, raws, errors, targets=None):
    out = []
    for k, name in enumerate(names):
        fr = FrequencyResponse(name=name, frequency=grid.copy(), raw=raws[k], error=errors[k])
        if targets is not None:
            fr.target = np.array(targets[k] if np.ndim(targets) > 1 else targets, dtype=np.float64)
        out.append(fr)
    return out


def calculate_specific_room_corrections(rir, target, mic_calibration=None, limit=400):
    """{speaker: {side: FrequencyResponse}} with ``error`` = measured - target, every channel levelled with the gain
    that centres the FIRST channel's 100 Hz-10 kHz mean (core/room_correction.py:185-210), faded out between limit/2
    and limit Hz.  All channels are handled as one [B, n] matrix: one batched spectrum, one re-gridding, row-wise
    arithmetic."""
    from .frequency_response import center_shifts
    keys = [(sp, sd) for sp, pair in rir.irs.items() for sd in pair]
    if not keys:
        return dict()
    fs = rir.irs[keys[0][0]][keys[0][1]].fs
    grid, raws = response_curves([rir.irs[sp][sd] for sp, sd in keys], fs)
    if mic_calibration is not None:
        raws = raws - mic_calibration.raw
    raws = raws - center_shifts(grid, raws[:1], [100, 10000])[0]              # the first channel's gain for everyone
    wanted = target.raw + np.array([IR_ROOM_SPL[sp][sd] for sp, sd in keys])[:, None]
    targets = wanted - center_shifts(target.frequency, wanted, 1000)[:, None]  # compensate() centres a copy at 1 kHz
    errors = raws - targets
    if limit > 0:
        errors = errors * _correction_limit_mask(grid, limit)
    frs = dict()
    for (sp, sd), fr in zip(keys, _curve_objects(["Frequency response"] * len(keys), grid, raws, errors, targets)):
        frs.setdefault(sp, dict())[sd] = fr
    return frs


def calculate_generic_room_correction(irs, target, mic_calibration=None, method='average', limit=1000):
    """One correction curve from the positions of a generic room measurement (core/room_correction.py:231-292):
    'average' = mean of the per-position errors, 'conservative' = the smallest correction every position agrees on."""
    from .frequency_response import center_shifts, smooth_curves
    if len(irs) > 1 and method not in ('average', 'conservative'):
        raise ValueError(f'Invalid value "{method}" for method. Supported values are "conservative" and "average"')
    grid, raws = response_curves(irs, irs[0].fs)
    if mic_calibration is not None:
        raws = raws - mic_calibration.raw
    raws = raws - center_shifts(grid, raws, [100, 10000])[:, None]
    target_c = target.raw - center_shifts(target.frequency, target.raw, 1000)[0]
    errors = raws - target_c
    band = np.logical_and(grid >= 100, grid <= 10000)
    errors = errors - np.mean(errors[:, band], axis=1)[:, None]                # min_mean_error=True, per position
    several = len(irs) > 1
    if method == 'conservative' and several:
        errors = smooth_curves(grid, errors, 1 / 3, 1 / 3)
        share = np.mean(errors > 0, axis=0)
        error = np.zeros(len(grid))
        pos, neg = share == 1, share == 0
        error[pos] = np.min(errors[:, pos], axis=0)
        error[neg] = np.max(errors[:, neg], axis=0)
        error = smooth_curves(grid, error, 1 / 6, 1 / 6)
        error_smoothed = error.copy()
    else:
        error = np.mean(errors, axis=0) if several else errors[0]
        error_smoothed = smooth_curves(grid, error, 1 / 3, 1 / 3)
    raw = np.sum(raws, axis=0) / len(irs)
    room_fr = FrequencyResponse(name='generic_room', frequency=grid, raw=raw, error=error, target=target.raw)
    room_fr.smoothed = smooth_curves(grid, raw, 1 / 3 if not (method == 'conservative' and several) else 1 / 6,
                                     1 / 3 if not (method == 'conservative' and several) else 1 / 6)
    room_fr.error_smoothed = error_smoothed
    if limit > 0:
        mask = _correction_limit_mask(grid, limit)
        room_fr.error = room_fr.error * mask
        room_fr.error_smoothed = room_fr.error_smoothed * mask
    return room_fr


def open_generic_room_measurement(estimator, dir_path, mic_calibration, target, method='average', limit=1000):
    path = discover_room_measurements(dir_path).generic_path
    if path is None:
        return None
    fs, data = read_wav(path, expand=True)
    if fs != estimator.fs:
        raise ValueError(f'Sampling rate of "{path}" doesn\'t match!')
    sweeps = []
    for track in data:
        n_cols = int(round((len(track) / estimator.fs - 2) / (estimator.duration + 2)))
        for i in range(n_cols):
            start = int(2 * estimator.fs + i * (2 * estimator.fs + len(estimator)))
            sweeps.append(track[start:min(int(start + 2 * estimator.fs + len(estimator)), len(track))])
    # equal-length columns go to the GPU as one batch
    irs = [None] * len(sweeps)
    by_len = {}
    for i, s in enumerate(sweeps):
        by_len.setdefault(len(s), []).append(i)
    for idxs in by_len.values():
        est = estimator.estimate_batch(np.stack([sweeps[i] for i in idxs]))
        for i, y in zip(idxs, est):
            irs[i] = ImpulseResponse(y, estimator.fs, sweeps[i])
            irs[i].crop_head(head_ms=1)
    return calculate_generic_room_correction(irs, target, mic_calibration=mic_calibration, method=method, limit=limit)


def room_correction(estimator, dir_path, target=None, mic_calibration=None, fr_combination_method='average',
                    specific_limit=400, generic_limit=300, plot=False):
    """(room HRIR or None, {speaker: {side: FrequencyResponse}} or None) for a measurement folder."""
    target = open_room_target(estimator, dir_path, target)
    mic_calibration = open_mic_calibration(estimator, dir_path, mic_calibration)
    rir = open_room_measurements(estimator, dir_path)
    missing = [ch for ch in SPEAKER_NAMES if ch not in rir.irs]
    room_fr = open_generic_room_measurement(estimator, dir_path, mic_calibration, target,
                                            method=fr_combination_method, limit=generic_limit)
    if not len(rir.irs) and room_fr is None:
        return None, None
    frs = dict()
    if len(rir.irs):
        for pair in rir.irs.values():
            for ir in pair.values():
                ir.crop_head()
        rir.crop_tails()
        rir.write_wav(discover_room_measurements(dir_path).responses_path)
        frs = calculate_specific_room_corrections(rir, target, mic_calibration=mic_calibration, limit=specific_limit)
    if len(missing) > 0 and room_fr is not None:
        for speaker in missing:
            frs[speaker] = {'left': room_fr.copy(), 'right': room_fr.copy()}
    return rir, frs
