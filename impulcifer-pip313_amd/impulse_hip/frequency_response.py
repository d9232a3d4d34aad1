"""Frequency-response helpers on the hot path (subset of the reference's vendored AutoEQ class,
autoeq/frequency_response.py): the log grid, log-linear interpolation, and the equalisation curve ->
minimum-phase FIR step (the FIR design itself runs on the GPU in fp64, kernel K6).
"""
import numpy as np

from . import _native

DEFAULT_F_MIN = 20.0
DEFAULT_F_MAX = 20000.0
DEFAULT_STEP = 1.01


def generate_frequencies(f_min=DEFAULT_F_MIN, f_max=DEFAULT_F_MAX, f_step=DEFAULT_STEP):
    """Geometric grid built by repeated multiplication (autoeq/frequency_response.py:850-857)."""
    out = []
    f = f_min
    while f <= f_max:
        out.append(f)
        f *= f_step
    return np.array(out)


def next_fast_len(n):
    """Smallest 2^a 3^b 5^c >= n (scipy.fftpack.next_fast_len)."""
    from .hrir import next_fast_len as _nfl
    return _nfl(n)


def log_interp(frequency, values, f_new):
    """Piecewise linear in log10(f) with linear extrapolation beyond the ends - what
    InterpolatedUnivariateSpline(log10 f, y, k=1) evaluates (autoeq :859-901); a zero target
    frequency is evaluated at 0.001 Hz."""
    xk = np.log10(np.asarray(frequency, dtype=np.float64))
    yk = np.asarray(values, dtype=np.float64)
    fq = np.array(f_new, dtype=np.float64)
    fq[fq == 0] = 0.001
    xq = np.log10(fq)
    idx = np.clip(np.searchsorted(xk, xq, side="right") - 1, 0, len(xk) - 2)
    t = (xq - xk[idx]) / (xk[idx + 1] - xk[idx])
    return yk[idx] + t * (yk[idx + 1] - yk[idx])


def fir_design_gain(frequency, equalization, fs, f_res=5.0, normalize=True):
    """Linear gain on linspace(0, fs//2, n) that the reference hands to firwin2
    (autoeq/frequency_response.py:651-674): dB doubled because the homomorphic step halves them,
    flat below the first grid frequency, zero at Nyquist."""
    frequency = np.asarray(frequency, dtype=np.float64)
    eq = np.asarray(equalization, dtype=np.float64)
    f_res = f_res / 2
    f_min = np.max([frequency[0], f_res])
    gain_f_min = float(log_interp(frequency, eq, [f_min])[0])
    n = next_fast_len(round(fs // 2 / f_res))
    f = np.linspace(0.0, fs // 2, n)
    raw = log_interp(frequency, eq, f)
    raw[f <= f_min] = gain_f_min
    if normalize:
        raw -= np.max(raw)
        raw -= 0.5
    raw *= 2
    lin = 10 ** (raw / 20)
    lin[-1] = 0.0
    return lin


def minimum_phase_impulse_response(frequency, equalization, fs, f_res=5.0, normalize=True):
    """Minimum-phase FIR (n = next_fast_len(fs//2 / (f_res/2)) taps) of one equalisation curve."""
    return _native.default_context().minphase_fir(fir_design_gain(frequency, equalization, fs, f_res, normalize), fs)


def minimum_phase_impulse_responses(frequency, equalizations, fs, f_res=5.0, normalize=True):
    """Batched form: one launch chain for all speaker-ear curves [B, len(frequency)] -> [B, n]."""
    gains = np.stack([fir_design_gain(frequency, eq, fs, f_res, normalize) for eq in equalizations])
    return _native.default_context().minphase_fir(gains, fs)


class FrequencyResponse(object):
    """Minimal carrier with the attributes the hot path reads (frequency, raw, error, equalization)
    and the FIR method the equalisation worker calls (core/parallel_workers.py:129)."""

    def __init__(self, name, frequency=None, raw=None, error=None, equalization=None, target=None):
        self.name = name
        self.frequency = generate_frequencies() if frequency is None else np.array(frequency, dtype=np.float64)
        n = len(self.frequency)

        def arr(v):
            if v is None:
                return np.array([])
            if np.isscalar(v):
                return np.ones(n) * v
            return np.array(v, dtype=np.float64)

        self.raw, self.error, self.equalization, self.target = arr(raw), arr(error), arr(equalization), arr(target)

    @staticmethod
    def generate_frequencies(f_min=DEFAULT_F_MIN, f_max=DEFAULT_F_MAX, f_step=DEFAULT_STEP):
        return generate_frequencies(f_min, f_max, f_step)

    def copy(self, name=None):
        return FrequencyResponse(name or self.name + "_copy", self.frequency, self.raw, self.error,
                                 self.equalization, self.target)

    def minimum_phase_impulse_response(self, fs=44100, f_res=10, normalize=True):
        return minimum_phase_impulse_response(self.frequency, self.equalization, fs, f_res, normalize)
