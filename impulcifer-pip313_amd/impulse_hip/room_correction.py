"""Room correction: per speaker-ear equalisation error curves from room measurements
(surface of reference core/room_correction.py:36-461; plotting left out).

The measurements are deconvolved on the GPU through HRIR.open_recording (all tracks/columns of a
file in one batch); cropping uses the device peak search; the curve arithmetic on the resulting
~800-point responses is host NumPy.
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


def _apply_correction_limit(fr, limit):
    fr.error *= _correction_limit_mask(fr.frequency, limit)


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


def calculate_specific_room_corrections(rir, target, mic_calibration=None, limit=400):
    """{speaker: {side: FrequencyResponse}} with ``error`` = measured - target, levelled to the first
    channel's 100 Hz-10 kHz mean and faded out between limit/2 and limit Hz."""
    frs = dict()
    reference_gain = None
    for speaker, pair in rir.irs.items():
        frs[speaker] = dict()
        for side, ir in pair.items():
            fr = ir.frequency_response()
            if mic_calibration is not None:
                fr.raw -= mic_calibration.raw
            if reference_gain is None:
                reference_gain = fr.center([100, 10000])
            else:
                fr.raw += reference_gain
            wanted = target.copy()
            wanted.raw += IR_ROOM_SPL[speaker][side]
            fr.compensate(wanted, min_mean_error=False)
            if limit > 0:
                _apply_correction_limit(fr, limit)
            frs[speaker][side] = fr
    return frs


def calculate_generic_room_correction(irs, target, mic_calibration=None, method='average', limit=1000):
    """One correction curve from several positions of a generic room measurement."""
    room_fr = FrequencyResponse(name='generic_room',
                                frequency=FrequencyResponse.generate_frequencies(f_min=10, f_max=irs[0].fs / 2,
                                                                                 f_step=1.01),
                                raw=0, error=0, target=target.raw)
    errors = []
    for ir in irs:
        fr = ir.frequency_response()
        if mic_calibration is not None:
            fr.raw -= mic_calibration.raw
        fr.center([100, 10000])
        room_fr.raw += fr.raw
        fr.compensate(target, min_mean_error=True)
        if method == 'conservative' and len(irs) > 1:
            fr.smoothen(window_size=1 / 3, treble_window_size=1 / 3)
            errors.append(fr.error_smoothed)
        else:
            errors.append(fr.error)
    room_fr.raw /= len(irs)
    errors = np.vstack(errors)
    if errors.shape[0] > 1:
        if method == 'conservative':
            share = np.mean(errors > 0, axis=0)
            pos, neg = share == 1, share == 0
            room_fr.error[pos] = np.min(errors[:, pos], axis=0)
            room_fr.error[neg] = np.max(errors[:, neg], axis=0)
            room_fr.smoothen(window_size=1 / 6, treble_window_size=1 / 6)
            room_fr.error = room_fr.error_smoothed.copy()
        elif method == 'average':
            room_fr.error = np.mean(errors, axis=0)
            room_fr.smoothen(window_size=1 / 3, treble_window_size=1 / 3)
        else:
            raise ValueError(f'Invalid value "{method}" for method. Supported values are "conservative" and "average"')
    else:
        room_fr.error = errors[0, :]
        room_fr.smoothen(window_size=1 / 3, treble_window_size=1 / 3)
    if limit > 0:
        _apply_correction_limit(room_fr, limit)
        room_fr.error_smoothed *= _correction_limit_mask(room_fr.frequency, limit)
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
