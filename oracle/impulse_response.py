"""Oracle: single impulse response operations (reference core/impulse_response.py, core/audio_io.py)."""
import numpy as np

from .scipy_restated import fft_convolve, find_peaks_height, spline1_eval

EPSILON = 1e-20


def peak_index(data, start=0, end=None, peak_height=0.12589):
    """core/impulse_response.py:32-70 (twin core/decay.py:12-41)."""
    data = np.asarray(data, dtype=np.float64)
    if len(data) == 0:
        return 0
    if end is None:
        end = len(data)
    seg = data[start:end].copy()
    if len(seg) == 0:
        return start
    mx = np.max(np.abs(seg))
    if mx < EPSILON:
        return start
    seg /= mx
    peaks = np.concatenate([find_peaks_height(seg, peak_height), find_peaks_height(seg * -1.0, peak_height)])
    if len(peaks) == 0:
        return int(np.argmax(np.abs(seg)) + start)
    return int(np.min(peaks + start))


def crop_head(data, fs, head_ms=1):
    """core/impulse_response.py:82-90."""
    if len(data) == 0:
        return data
    i0 = peak_index(data) - int(fs * head_ms / 1000)
    return data[max(i0, 0):]


def magnitude_response(x, fs):
    """core/audio_io.py:100-113: first ceil(n/2) bins of 20 log10 |rfft(x)|, no epsilon."""
    n = len(x)
    half = int(np.ceil(n / 2))
    X = np.fft.rfft(x)
    with np.errstate(divide="ignore"):
        mag = 20 * np.log10(np.abs(X[:half]))
    return np.arange(half) * (fs / n), mag


def equalize(data, fir):
    """core/impulse_response.py:110-119: full linear convolution."""
    return fft_convolve(data, fir, "full")


def generate_frequencies(f_min=20.0, f_max=20000.0, f_step=1.01):
    """autoeq/frequency_response.py:850-857: iterative geometric grid (f *= step)."""
    out = []
    f = f_min
    while f <= f_max:
        out.append(f)
        f *= f_step
    return np.array(out)


def interpolate_log(frequency, raw, f_new):
    """autoeq/frequency_response.py:859-901 with pol_order=1: linear in log10 f, zero frequencies
    temporarily replaced by 0.001 Hz, linear extrapolation outside the data."""
    frequency = np.asarray(frequency, dtype=np.float64).copy()
    f_new = np.asarray(f_new, dtype=np.float64).copy()
    if frequency[0] == 0:
        frequency[0] = 0.001
    z = f_new == 0
    f_new[z] = 0.001
    return spline1_eval(np.log10(frequency), raw, np.log10(f_new))


def frequency_response(data, fs):
    """core/impulse_response.py:157-188: decimated magnitude response on the 1.01-step log grid."""
    f, m = magnitude_response(data, fs)
    step = int(round(len(f) / ((fs / 2) / 4.0)))
    if step == 0:
        step = 1
    fd, md = f[1::step], m[1::step]
    grid = generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    return grid, interpolate_log(fd, md, grid)
