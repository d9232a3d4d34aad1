"""Oracle: Lundeby knee / Schroeder decay analysis and decay-window adjustment.

Restates reference core/decay.py (decay_params :44-260, decay_times :263-352,
decay_adjustment_params :355-380, apply_decay_window :383-403) in float64 NumPy.
"""
import numpy as np

from .impulse_response import peak_index
from .scipy_restated import hann, linregress

EPS = 1e-20


def _db_power(v):
    return 10 * np.log10(np.maximum(v, EPS))


def _first_at_or_below(values, level):
    """Index of the first element <= level, or None (np.argwhere(...)[0, 0] / IndexError)."""
    hits = np.nonzero(values <= level)[0]
    return int(hits[0]) if len(hits) else None


def _nearest(t, value):
    return int(np.argmin(np.abs(t - value))) if len(t) else 0


def _window_levels(sq, n, w):
    """Mean power (dB) of n consecutive windows of w samples (core/decay.py:103-105, 163-165)."""
    return _db_power(np.mean(np.reshape(sq[: n * w], (n, w)), axis=1))


def decay_params(data, fs):
    """Returns (peak_index, knee_point_index, noise_floor_dB, window_size) -- core/decay.py:44-260."""
    ir = np.asarray(data, dtype=np.float64)
    if len(ir) < 10:
        return 0, len(ir), -200.0, len(ir) if len(ir) > 0 else 1

    pk = peak_index(ir)
    stop = min(pk + int(2 * fs), len(ir))                      # analyse at most 2 s after the peak
    if pk >= stop:
        pk = min(max(pk, 0), len(ir) - 1)
        seg = ir[pk:pk + 1].copy()
    else:
        seg = ir[pk:stop].copy()
    if len(seg) == 0:
        seg = np.array([EPS])
    mx = np.max(np.abs(seg))
    if mx >= EPS:
        seg = seg / mx
    sq = seg ** 2
    n_sq = len(sq)
    t_sq = np.linspace(0, n_sq / fs, n_sq)                     # endpoint included: dt != 1/fs

    def bail(knee_rel, floor, win):
        return pk, pk + knee_rel, floor, win

    # --- first pass: fixed 30 ms windows
    wd = 0.03
    n = int(n_sq / fs / wd) if fs > 0 else 0
    if n == 0:
        return bail(n_sq, 10 * np.log10(max(np.mean(sq), EPS)), max(1, n_sq))
    w = max(int(n_sq / n), 1)
    w_first = w
    t_win = np.arange(n) * wd + wd / 2
    levels = _window_levels(sq, n, w)

    tail = sq[int(n_sq * 0.9):]
    if len(tail) == 0:
        tail = sq
    floor = 10 * np.log10(np.maximum(np.mean(tail), EPS))

    near = np.nonzero(levels <= floor + 10.0)[0]
    fit_end = len(levels)
    if len(near) > 0 and near[0] > 0:
        fit_end = int(near[0])
    if fit_end < 2:
        if len(levels) >= 2:
            fit_end = len(levels)
        else:
            return bail(n_sq, floor, w_first)
    slope, icpt = linregress(t_win[:fit_end], levels[:fit_end])
    if np.isnan(slope) or abs(slope) < EPS:
        return bail(n_sq, floor, w_first)
    knee_t = np.clip((floor - icpt) / slope, t_sq[0], t_sq[-1])

    # --- second pass: 3 windows per 10 dB of decay
    denom = abs(slope) * 3
    wd = (t_sq[-1] / 3.0) if denom < EPS else 10 / denom
    n = 1 if (fs <= 0 or wd <= EPS) else int(n_sq / fs / wd)
    if n == 0:
        n = 1
    w = max(int(n_sq / n), 1)
    t_win = np.arange(n) * wd + wd / 2
    levels = _window_levels(sq, n, w)

    hits = np.nonzero(t_win >= knee_t)[0]
    if len(hits):
        knee_i = int(hits[0])
        knee_level = levels[knee_i]
    else:
        knee_t = t_win[-1]
        knee_i = len(t_win) - 1
        knee_level = levels[-1]

    floor_it, knee_t_it, knee_level_it, knee_i_it = floor, knee_t, knee_level, knee_i
    total = t_sq[-1]
    for _ in range(5):
        i0 = _first_at_or_below(levels, knee_level_it - 5)
        if i0 is None:
            break
        t0 = max(t_win[i0], 0.1 * total)
        if t0 > t_win[-1]:
            break
        t1 = min(t0 + knee_t_it, total)
        a, b = _nearest(t_sq, t0), _nearest(t_sq, t1)
        if a >= b:
            break
        floor_it = 10 * np.log10(np.maximum(np.mean(sq[a:b]), EPS))

        e = _first_at_or_below(levels, floor_it + 8)
        s = _first_at_or_below(levels, floor_it + 8 + 20)
        if e is None or s is None:
            break
        e, s = e - 1, max(s - 1, 0)
        if e <= s + 1 or len(t_win[s:e]) < 2:
            break
        late_slope, late_icpt = linregress(t_win[s:e], levels[s:e])
        if np.isnan(late_slope) or abs(late_slope) < EPS:
            break
        new_t = np.clip((floor_it - late_icpt) / late_slope, t_win[0], t_win[-1])
        hits = np.nonzero(t_win >= new_t)[0]
        new_i = int(hits[0]) if len(hits) else len(t_win) - 1
        if new_i == knee_i_it:
            knee_t_it = t_win[knee_i_it]
            break
        knee_i_it = new_i
        knee_t_it = t_win[new_i]
        knee_level_it = levels[new_i]

    return pk, pk + _nearest(t_sq, knee_t_it), floor_it, w


def running_mean(x, N):
    """core/audio_io.py:116-118."""
    c = np.cumsum(np.insert(x, 0, 0))
    return (c[N:] - c[:-N]) / float(N)


def decay_times(data, fs, peak_ind=None, knee_point_ind=None, noise_floor=None, window_size=None):
    """(EDT, RT20, RT30, RT60), None where undefined -- core/decay.py:263-352."""
    ir = np.asarray(data, dtype=np.float64)
    if peak_ind is None or knee_point_ind is None or noise_floor is None:
        peak_ind, knee_point_ind, noise_floor, window_size = decay_params(ir, fs)
    t = np.linspace(0, len(ir) / fs, len(ir))
    knee = knee_point_ind - peak_ind
    env = ir[peak_ind:].copy()
    env /= np.max(np.abs(env))
    env = np.abs(env)
    sch = np.cumsum(env[knee::-1] ** 2 / np.sum(env[:knee] ** 2))[:0:-1]   # Schroeder backward integral
    sch = 10 * np.log10(sch)

    head = min(window_size // 2, peak_ind)
    tail = min(window_size // 2, len(ir) - (peak_ind + knee))
    shift = window_size // 2 - head
    avg = ir[peak_ind - head: peak_ind + knee + tail].copy()
    avg /= np.max(np.abs(avg))
    avg = 10 * np.log10(running_mean(avg ** 2, window_size) + 1e-18)
    a = max(int(len(sch) * 0.1), shift)
    b = min(int(len(sch) * 0.9), shift + len(avg))
    offset = np.mean(sch[a:b] - avg[a - shift: b - shift])

    out = []
    for start_db, end_db, span_db in ((-1, -10, -10), (-5, -25, -20), (-5, -35, -30), (-5, -65, -60)):
        val = None
        if not (end_db < noise_floor + offset + 10):
            s = _first_at_or_below(sch, start_db)
            e = _first_at_or_below(sch, end_db)
            if s is not None and e is not None and s < e and (e - s) >= 2:
                slope, _ = linregress(t[s:e], sch[s:e])
                val = span_db / slope
        out.append(val)
    return tuple(out)


def decay_adjustment_params(data, fs, target):
    """(window_start, half_window, knee_point_index, window_level) or None -- core/decay.py:355-380.
    Like the reference this raises TypeError when not even EDT is defined."""
    pk, knee, _, _ = decay_params(data, fs)
    slope_db_s = None
    for rt, level in zip(decay_times(data, fs), (-10, -20, -30, -60)):
        if not rt:
            break
        slope_db_s = level / rt
    target_slope = -60 / target
    if target_slope > slope_db_s:
        return None
    knee_time = knee / fs
    level = target_slope * knee_time - slope_db_s * knee_time
    start = pk + 2 * (fs // 1000)
    return start, knee - start, knee, level


def apply_decay_window(data, params):
    """core/decay.py:383-403 (in place on a float64 array)."""
    if params is None:
        return data
    start, half, knee, level = params
    win = np.concatenate([np.ones(start), hann(half * 2)[half:], np.zeros(len(data) - knee)]) - 1.0
    data *= 10 ** (win * -level / 20)
    return data
