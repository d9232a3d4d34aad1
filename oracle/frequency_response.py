"""Oracle: the FrequencyResponse operations on the hot path (reference autoeq/frequency_response.py)
and the specific room-correction chain (reference core/room_correction.py:185-306), NumPy only.

Curves are plain arrays on a shared log grid.  SciPy's savgol_filter(mode='interp', polyorder=2),
scipy.special.expit and the FITPACK quadratic interpolating spline that `equalize` uses to bridge
gain-clipping kinks are restated (scipy_restated.spline2_interp).
"""
import math

import numpy as np

from .impulse_response import frequency_response as ir_frequency_response
from .impulse_response import generate_frequencies, interpolate_log
from .scipy_restated import hann, spline2_interp


def savgol_interp(x, window, order=2):
    """scipy.signal.savgol_filter(x, window, order) with the default mode='interp': least-squares
    polynomial smoothing; the first/last window//2 samples come from a polynomial fitted to the
    first/last `window` samples (scipy/signal/_savitzky_golay.py)."""
    x = np.asarray(x, dtype=np.float64)
    half = window // 2
    t = np.arange(-half, half + 1, dtype=np.float64)
    A = np.vander(t, order + 1, increasing=True)
    weights = np.linalg.pinv(A)[0]                    # value of the fitted polynomial at t = 0
    y = np.convolve(x, weights[::-1], mode="same")
    pos = np.arange(window, dtype=np.float64)
    head = np.polyfit(pos, x[:window], order)
    y[:half] = np.polyval(head, pos[:half])
    tail = np.polyfit(pos, x[-window:], order)
    y[-half:] = np.polyval(tail, pos[window - half:])
    return y


def window_size(frequency, octaves):
    """autoeq :1033-1050."""
    steps = [frequency[i] / frequency[i - 1] for i in range(1, len(frequency))]
    step = sum(steps) / len(steps)
    w = round(math.log(2 ** octaves) / math.log(step))
    return w + 1 if not w % 2 else w


def sigmoid(frequency, f_lower, f_upper, a_normal=0.0, a_treble=1.0):
    """autoeq :1052-1058 (expit(x) = 1 / (1 + exp(-x)))."""
    f_center = np.sqrt(f_upper / f_lower) * f_lower
    half_range = np.log10(f_upper) - np.log10(f_center)
    a = 1.0 / (1.0 + np.exp(-((np.log10(frequency) - np.log10(f_center)) / (half_range / 4))))
    return a * -(a_normal - a_treble) + a_normal


def smoothen_fractional_octave(frequency, data, window_size_oct, treble_window_size_oct, treble_f_lower, treble_f_upper):
    """autoeq :1060-1105 with one iteration each."""
    y_n = savgol_interp(data, window_size(frequency, window_size_oct))
    y_t = savgol_interp(data, window_size(frequency, treble_window_size_oct))
    k_t = sigmoid(frequency, treble_f_lower, treble_f_upper)
    return y_n * (k_t * -1 + 1) + y_t * k_t


def smoothen_heavy_light(frequency, error):
    """error_smoothed of autoeq :1181-1239."""
    light = smoothen_fractional_octave(frequency, error, 1 / 6, 1 / 3, 100, 10000)
    heavy = smoothen_fractional_octave(frequency, error, 1 / 3, 1.3, 1000, 6000)
    combo = np.max(np.vstack([light, heavy]), axis=0)
    return smoothen_fractional_octave(frequency, combo, 1 / 3, 1 / 3, 100, 10000)


def equalize(frequency, error_smoothed, max_gain, treble_f_lower, treble_f_upper, treble_max_gain=6.0,
             treble_gain_k=1.0):
    """equalization curve of autoeq :1241-1310: -error clipped at the (treble-dependent) maximum gain,
    then the samples within a 1/12-octave window of every clip on/off transition are dropped and the
    rest is re-interpolated with FITPACK's quadratic spline in log-frequency."""
    frequency = np.asarray(frequency, dtype=np.float64)
    limit = sigmoid(frequency, treble_f_lower, treble_f_upper, a_normal=max_gain, a_treble=treble_max_gain)
    gain = -error_smoothed * sigmoid(frequency, treble_f_lower, treble_f_upper, a_normal=1.0, a_treble=treble_gain_k)
    clipped = gain > limit
    kinks = np.flatnonzero(np.concatenate(([clipped[0]], clipped[1:] != clipped[:-1])))
    if len(kinks) and kinks[0] == 0:
        kinks = kinks[1:]
    eq = np.where(clipped, limit, gain)
    half = (window_size(frequency, 1 / 12) - 1) // 2
    n = len(eq)
    doomed = set()
    for i in kinks:
        doomed.update(range(i - min(i, half), i + 1 + min(n - i - 1, half)))
    doomed -= {n - 1, n - 2}
    keep = np.ones(n, dtype=bool)
    keep[sorted(doomed)] = False
    return spline2_interp(np.log10(frequency[keep]), eq[keep], np.log10(frequency))


def equalization_worker_curve(common_freq, room_error, target_raw, fs):
    """The curve part of core/parallel_workers.py:69-126: error = room (+hp +eq) - target ->
    smoothen_heavy_light -> equalize(max_gain=40, treble 10 kHz..fs/2)."""
    error = np.zeros(len(common_freq)) + room_error - target_raw
    return equalize(common_freq, smoothen_heavy_light(common_freq, error), 40, 10000, fs / 2)


def center_shift(frequency, raw, at=1000):
    """The value `center` subtracts (autoeq :903-940): read on the default 20..20 000 Hz grid."""
    grid = generate_frequencies(20, 20000, 1.01)
    on_grid = interpolate_log(frequency, raw, grid)
    if isinstance(at, (list, tuple, np.ndarray)) and len(at) > 1:
        return np.mean(on_grid[np.logical_and(grid >= at[0], grid <= at[1])])
    if isinstance(at, (list, tuple, np.ndarray)):
        at = at[0]
    return float(interpolate_log(grid, on_grid, np.array([at]))[0])


def correction_limit_mask(frequency, limit):
    """core/room_correction.py:295-302: ones | falling Hann between limit/2 and limit | zeros."""
    start = int(np.argmax(frequency > limit / 2))
    end = int(np.argmax(frequency > limit))
    full = hann(end - start)
    return np.concatenate([np.ones(start if start > 0 else 0), full, np.zeros(len(frequency) - end)])


def specific_room_correction(ir_data, fs, target_raw, mic_calibration_raw=None, limit=400, reference_gain=None):
    """One iteration of core/room_correction.py:185-210.  target_raw / mic_calibration_raw live on the
    10 Hz..fs/2 grid (already interpolated and centred, as _open_room_target/_open_mic_calibration
    leave them).  Returns (frequency, raw, error, reference_gain)."""
    freq, raw = ir_frequency_response(ir_data, fs)
    if mic_calibration_raw is not None:
        raw = raw - mic_calibration_raw
    if reference_gain is None:
        reference_gain = -center_shift(freq, raw, [100, 10000])
    raw = raw + reference_gain
    target = target_raw - center_shift(freq, target_raw, 1000)        # compensate() centres a copy at 1 kHz
    error = raw - target
    if limit > 0:
        error = error * correction_limit_mask(freq, limit)
    return freq, raw, error, reference_gain


def smoothen(frequency, data, window_size_oct=1 / 3, treble_window_size_oct=1 / 3, treble_f_lower=100.0,
             treble_f_upper=10000.0):
    """FrequencyResponse.smoothen_fractional_octave on one curve with the defaults room_correction passes."""
    return smoothen_fractional_octave(frequency, data, window_size_oct, treble_window_size_oct, treble_f_lower,
                                      treble_f_upper)


def generic_room_correction(ir_datas, fs, target_raw, mic_calibration_raw=None, method="average", limit=1000):
    """core/room_correction.py:231-292 (_calculate_generic_room_correction).  ir_datas: head-cropped responses of the
    positions of room.wav; target_raw / mic_calibration_raw on the 10 Hz..fs/2 grid.  Returns (frequency, raw, error,
    error_smoothed) of the combined curve."""
    freq = generate_frequencies(10, fs / 2, 1.01)
    raw_sum = np.zeros(len(freq))
    errors = []
    target_c = target_raw - center_shift(freq, target_raw, 1000)          # compensate() centres a copy at 1 kHz
    for d in ir_datas:
        f, raw = ir_frequency_response(d, fs)
        if mic_calibration_raw is not None:
            raw = raw - mic_calibration_raw
        raw = raw - center_shift(f, raw, [100, 10000])
        raw_sum = raw_sum + raw
        err = raw - target_c
        err = err - np.mean(err[np.logical_and(f >= 100, f <= 10000)])    # min_mean_error=True
        if method == "conservative" and len(ir_datas) > 1:
            err = smoothen(f, err)
        errors.append(err)
    raw = raw_sum / len(ir_datas)
    errors = np.vstack(errors)
    if errors.shape[0] > 1:
        if method == "conservative":
            share = np.mean(errors > 0, axis=0)
            error = np.zeros(len(freq))
            pos, neg = share == 1, share == 0
            error[pos] = np.min(errors[:, pos], axis=0)
            error[neg] = np.max(errors[:, neg], axis=0)
            error = smoothen(freq, error, 1 / 6, 1 / 6)
            error_smoothed = error.copy()
        elif method == "average":
            error = np.mean(errors, axis=0)
            error_smoothed = smoothen(freq, error)
        else:
            raise ValueError(f'Invalid value "{method}" for method. Supported values are "conservative" and "average"')
    else:
        error = errors[0]
        error_smoothed = smoothen(freq, error)
    if limit > 0:
        mask = correction_limit_mask(freq, limit)
        error = error * mask
        error_smoothed = error_smoothed * mask
    return freq, raw, error, error_smoothed


def headphone_curves(ir_left, ir_right, fs):
    """core/pipeline_stages.py:424-443: frequency responses of FL-left and FR-right (un-cropped responses), both shifted
    by the gain that centres the LEFT one between 100 Hz and 10 kHz; compensate(zero target, min_mean_error=False)
    leaves error = raw.  Returns (frequency, raw_left, raw_right)."""
    f, left = ir_frequency_response(ir_left, fs)
    _, right = ir_frequency_response(ir_right, fs)
    shift = center_shift(f, left, [100, 10000])
    return f, left - shift, right - shift
