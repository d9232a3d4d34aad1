"""FrequencyResponse: the part of the reference's vendored AutoEQ class that the hot path touches
(autoeq/frequency_response.py): the log grid, log-linear interpolation, centring, target
compensation, fractional-octave smoothing, gain-limited equalisation and the equalisation curve ->
minimum-phase FIR step.

Curve conditioning works on ~800-point dB curves and stays on the host (NumPy, plus SciPy's
Savitzky-Golay filter and quadratic spline, which the reference itself calls); the FIR design - four
19 200/38 400-point transforms per channel - runs on the GPU in fp64 (kernel K6), batched over all
speaker-ear channels.  The parametric-EQ optimiser, file writers and plots of the AutoEQ class are
outside the path and not provided.
"""
import math
import warnings

import numpy as np

from . import _native

DEFAULT_F_MIN = 20
DEFAULT_F_MAX = 20000
DEFAULT_STEP = 1.01
DEFAULT_MAX_GAIN = 6.0
DEFAULT_TREBLE_F_LOWER = 6000.0
DEFAULT_TREBLE_F_UPPER = 8000.0
DEFAULT_TREBLE_MAX_GAIN = 6.0
DEFAULT_TREBLE_GAIN_K = 1.0
DEFAULT_SMOOTHING_WINDOW_SIZE = 1 / 3
DEFAULT_SMOOTHING_ITERATIONS = 1
DEFAULT_TREBLE_SMOOTHING_F_LOWER = 100.0
DEFAULT_TREBLE_SMOOTHING_F_UPPER = 10000.0
DEFAULT_TREBLE_SMOOTHING_WINDOW_SIZE = 1 / 3
DEFAULT_TREBLE_SMOOTHING_ITERATIONS = 1
DEFAULT_FS = 44100
DEFAULT_F_RES = 10

_CURVES = ("raw", "smoothed", "error", "error_smoothed", "equalization", "equalized_raw", "equalized_smoothed",
           "target")


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
    def __init__(self, name, frequency=None, raw=None, error=None, smoothed=None, error_smoothed=None,
                 equalization=None, equalized_raw=None, equalized_smoothed=None, target=None):
        self.name = name.strip()
        self.frequency = self._curve(frequency)
        if not len(self.frequency):
            self.frequency = generate_frequencies()
        self.raw = self._curve(raw)
        self.smoothed = self._curve(smoothed)
        self.error = self._curve(error)
        self.error_smoothed = self._curve(error_smoothed)
        self.equalization = self._curve(equalization)
        self.equalized_raw = self._curve(equalized_raw)
        self.equalized_smoothed = self._curve(equalized_smoothed)
        self.target = self._curve(target)
        order = self.frequency.argsort()
        if np.any(order[1:] < order[:-1]):
            self.frequency = self.frequency[order]
            for key in _CURVES:
                cur = getattr(self, key)
                if len(cur):
                    setattr(self, key, cur[order])
        if len(self.frequency) > 1 and np.any(self.frequency[1:] == self.frequency[:-1]):
            dup = self.frequency[1:][self.frequency[1:] == self.frequency[:-1]][0]
            raise ValueError('Duplicate values found at frequency {}. Remove duplicates manually.'.format(dup))

    def _curve(self, data):
        """None -> empty, scalar -> constant curve on the grid, sequence -> float array (None -> NaN)."""
        if data is None:
            return np.array([])
        if isinstance(data, (int, float, np.integer, np.floating)):
            return np.ones(self.frequency.shape) * float(data)
        return np.array([np.nan if v is None else v for v in data], dtype=np.float64) \
            if isinstance(data, (list, tuple)) else np.array(data, dtype=np.float64)

    @staticmethod
    def generate_frequencies(f_min=DEFAULT_F_MIN, f_max=DEFAULT_F_MAX, f_step=DEFAULT_STEP):
        return generate_frequencies(f_min, f_max, f_step)

    @classmethod
    def read_csv(cls, file_path):
        """Two-or-more-column CSV/TXT with a header line: frequency, raw (autoeq :200-260 subset;
        also accepts whitespace-separated REW-style text with '*' comment lines)."""
        name = ".".join(file_path.replace("\\", "/").split("/")[-1].split(".")[:-1])
        freq, raw = [], []
        with open(file_path, "r", encoding="utf-8") as fh:
            for line in fh:
                line = line.strip()
                if not line or line[0] in "*#;":
                    continue
                parts = [p for p in line.replace(",", " ").replace("\t", " ").split(" ") if p]
                try:
                    f, v = float(parts[0]), float(parts[1])
                except (ValueError, IndexError):
                    continue                                   # header
                freq.append(f)
                raw.append(v)
        return cls(name=name, frequency=freq, raw=raw)

    def copy(self, name=None):
        return FrequencyResponse(name=(name or self.name + "_copy"), frequency=self.frequency.copy(), raw=self.raw,
                                 error=self.error, smoothed=self.smoothed, error_smoothed=self.error_smoothed,
                                 equalization=self.equalization, equalized_raw=self.equalized_raw,
                                 equalized_smoothed=self.equalized_smoothed, target=self.target)

    def reset(self, raw=False, smoothed=True, error=True, error_smoothed=True, equalization=True,
              fixed_band_eq=True, parametric_eq=True, equalized_raw=True, equalized_smoothed=True, target=True):
        """Empties the named curves (autoeq :143-180); the parametric/fixed-band fields do not exist here."""
        flags = dict(raw=raw, smoothed=smoothed, error=error, error_smoothed=error_smoothed,
                     equalization=equalization, equalized_raw=equalized_raw, equalized_smoothed=equalized_smoothed,
                     target=target)
        for key, clear in flags.items():
            if clear:
                setattr(self, key, np.array([]))

    # ---- grid -----------------------------------------------------------------------------
    def interpolate(self, f=None, f_step=DEFAULT_STEP, pol_order=1, f_min=DEFAULT_F_MIN, f_max=DEFAULT_F_MAX):
        """Re-samples every populated curve onto a new grid, linearly in log-frequency."""
        if pol_order != 1:
            raise NotImplementedError("only pol_order=1 is on the path")
        if len(self.raw):
            ok = ~np.isnan(self.raw)
            if not ok.all():
                self.raw, self.frequency = self.raw[ok], self.frequency[ok]
        old = self.frequency
        new = generate_frequencies(f_min=f_min, f_max=f_max, f_step=f_step) if f is None else np.array(f, dtype=np.float64)
        for key in ("raw", "error", "error_smoothed", "equalization", "equalized_raw", "equalized_smoothed", "target"):
            cur = getattr(self, key)
            if len(cur):
                setattr(self, key, log_interp(old, cur, new))
        self.frequency = new
        self.reset(raw=False, error=False, error_smoothed=False, equalization=False, equalized_raw=False,
                   equalized_smoothed=False, target=False)       # i.e. only `smoothed` is dropped

    def center(self, frequency=1000):
        """Shift so that the level at ``frequency`` (or the mean between two frequencies), read on the
        default 20 Hz-20 kHz grid, becomes 0 dB.  Returns the applied shift."""
        probe = FrequencyResponse(name="equal_energy", frequency=self.frequency.copy(), raw=self.raw.copy())
        probe.interpolate()
        if isinstance(frequency, (list, np.ndarray)) and len(frequency) > 1:
            band = np.logical_and(probe.frequency >= frequency[0], probe.frequency <= frequency[1])
            diff = np.mean(probe.raw[band])
        else:
            if isinstance(frequency, (list, np.ndarray)):
                frequency = frequency[0]
            diff = float(log_interp(probe.frequency, probe.raw, [frequency])[0])
        self.raw -= diff
        if len(self.smoothed):
            self.smoothed -= diff
        if len(self.error):
            self.error += diff
        if len(self.error_smoothed):
            self.error_smoothed += diff
        self.reset(raw=False, smoothed=False, error=False, error_smoothed=False, target=False)
        return -diff

    def compensate(self, compensation, bass_boost_gain=0.0, tilt=None, sound_signature=None, min_mean_error=False):
        """target = centred compensation curve; error = raw - target (autoeq :984-1031).  The bass-boost
        shelf and tilt of the full AutoEQ method are not on the path (the reference calls this with
        their defaults, 0 dB and None)."""
        if bass_boost_gain != 0.0 or tilt is not None or sound_signature is not None:
            raise NotImplementedError("bass boost / tilt / sound signature are outside the path")
        comp = FrequencyResponse(name="compensation", frequency=compensation.frequency, raw=compensation.raw)
        comp.center()
        self.target = comp.raw + np.zeros(len(self.frequency))
        self.error = self.raw - self.target
        if min_mean_error:
            delta = np.mean(self.error[np.logical_and(self.frequency >= 100, self.frequency <= 10000)])
            self.error -= delta
            self.target += delta
        self.reset(raw=False, smoothed=False, error=False, error_smoothed=True, target=False)

    # ---- smoothing ------------------------------------------------------------------------
    def _window_size(self, octaves):
        """Odd Savitzky-Golay window covering ``octaves`` on this grid."""
        f = np.asarray(self.frequency, dtype=np.float64)
        # mean neighbour ratio; cumsum adds left to right, i.e. the same roundings as a running Python sum
        step = float(np.cumsum(f[1:] / f[:-1])[-1]) / (len(f) - 1)
        size = round(math.log(2 ** octaves) / math.log(step))
        return size + 1 if size % 2 == 0 else size

    def _sigmoid(self, f_lower, f_upper, a_normal=0.0, a_treble=1.0):
        from scipy.special import expit
        centre = np.sqrt(f_upper / f_lower) * f_lower
        half = np.log10(f_upper) - np.log10(centre)
        a = expit((np.log10(self.frequency) - np.log10(centre)) / (half / 4))
        return a * -(a_normal - a_treble) + a_normal

    def _smoothen_fractional_octave(self, data, window_size=DEFAULT_SMOOTHING_WINDOW_SIZE,
                                    iterations=DEFAULT_SMOOTHING_ITERATIONS, treble_window_size=None,
                                    treble_iterations=None, treble_f_lower=DEFAULT_TREBLE_SMOOTHING_F_LOWER,
                                    treble_f_upper=DEFAULT_TREBLE_SMOOTHING_F_UPPER):
        from scipy.signal import savgol_filter
        if np.any(np.isnan(self.frequency)) or np.any(np.isnan(np.asarray(data, dtype=float))):
            raise ValueError('NaN values present, cannot smoothen!')
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y_normal = data
            w = self._window_size(window_size)
            for _ in range(iterations):
                y_normal = savgol_filter(y_normal, w, 2)
            y_treble = data
            w = self._window_size(treble_window_size)
            for _ in range(treble_iterations):
                y_treble = savgol_filter(y_treble, w, 2)
        k_treble = self._sigmoid(treble_f_lower, treble_f_upper)
        return y_normal * (k_treble * -1 + 1) + y_treble * k_treble

    def smoothen_fractional_octave(self, window_size=DEFAULT_SMOOTHING_WINDOW_SIZE,
                                   iterations=DEFAULT_SMOOTHING_ITERATIONS,
                                   treble_window_size=DEFAULT_TREBLE_SMOOTHING_WINDOW_SIZE,
                                   treble_iterations=DEFAULT_TREBLE_SMOOTHING_ITERATIONS,
                                   treble_f_lower=DEFAULT_TREBLE_SMOOTHING_F_LOWER,
                                   treble_f_upper=DEFAULT_TREBLE_SMOOTHING_F_UPPER):
        kw = dict(window_size=window_size, iterations=iterations, treble_window_size=treble_window_size,
                  treble_iterations=treble_iterations, treble_f_lower=treble_f_lower, treble_f_upper=treble_f_upper)
        if len(self.raw):
            self.smoothed = self._smoothen_fractional_octave(self.raw, **kw)
        if len(self.error):
            self.error_smoothed = self._smoothen_fractional_octave(self.error, **kw)
        self.reset(raw=False, smoothed=False, error=False, error_smoothed=False, equalization=True,
                   equalized_raw=True, equalized_smoothed=True, target=False)

    smoothen = smoothen_fractional_octave

    def smoothen_heavy_light(self):
        """Error curve smoothed as max(light, heavy) then once more at 1/3 octave (autoeq :1181-1239)."""
        light = self._smoothen_fractional_octave(self.error, window_size=1 / 6, iterations=1, treble_f_lower=100,
                                                 treble_f_upper=10000, treble_window_size=1 / 3, treble_iterations=1)
        heavy = self._smoothen_fractional_octave(self.error, window_size=1 / 3, iterations=1, treble_f_lower=1000,
                                                 treble_f_upper=6000, treble_window_size=1.3, treble_iterations=1)
        third = dict(window_size=1 / 3, iterations=1, treble_f_lower=100, treble_f_upper=10000,
                     treble_window_size=1 / 3, treble_iterations=1)
        self.smoothed = self._smoothen_fractional_octave(self.raw, **third)
        self.error_smoothed = self._smoothen_fractional_octave(np.max(np.vstack([light, heavy]), axis=0), **third)
        self.reset(raw=False, smoothed=False, error=False, error_smoothed=False, equalization=True,
                   equalized_raw=True, equalized_smoothed=True, target=False)

    # ---- equalisation ---------------------------------------------------------------------
    def equalize(self, max_gain=DEFAULT_MAX_GAIN, smoothen=True, treble_f_lower=DEFAULT_TREBLE_F_LOWER,
                 treble_f_upper=DEFAULT_TREBLE_F_UPPER, treble_max_gain=DEFAULT_TREBLE_MAX_GAIN,
                 treble_gain_k=DEFAULT_TREBLE_GAIN_K):
        """equalization = -error, clipped at a (treble-dependent) maximum gain; the kinks that clipping
        leaves are cut out and bridged with a quadratic spline (autoeq :1241-1310)."""
        if len(self.error_smoothed):
            error = self.error_smoothed
        elif len(self.error):
            error = self.error
        else:
            raise ValueError('Error data is missing. Call FrequencyResponse.compensate().')
        if np.any(np.isnan(np.asarray(error, dtype=float))):
            raise ValueError('NaN values detected during equalization, interpolating data with default parameters.')
        limit = self._sigmoid(treble_f_lower, treble_f_upper, a_normal=max_gain, a_treble=treble_max_gain)
        gain = -error * self._sigmoid(treble_f_lower, treble_f_upper, a_normal=1.0, a_treble=treble_gain_k)
        clipped = gain > limit
        edges = np.flatnonzero(np.concatenate(([clipped[0]], clipped[1:] != clipped[:-1])))
        if len(edges) and edges[0] == 0:
            edges = edges[1:]
        self.equalization = np.where(clipped, limit, gain)
        if smoothen:
            from scipy.interpolate import InterpolatedUnivariateSpline
            half = (self._window_size(1 / 12) - 1) // 2
            n = len(self.equalization)
            doomed = set()
            for i in edges:
                doomed.update(range(i - min(i, half), i + 1 + min(n - i - 1, half)))
            doomed.discard(n - 1)
            doomed.discard(n - 2)
            keep = np.ones(n, dtype=bool)
            keep[sorted(doomed)] = False
            spline = InterpolatedUnivariateSpline(np.log10(self.frequency[keep]), self.equalization[keep], k=2)
            self.equalization = spline(np.log10(self.frequency))
        self.equalized_raw = self.raw + self.equalization
        if len(self.smoothed):
            self.equalized_smoothed = self.smoothed + self.equalization

    def minimum_phase_impulse_response(self, fs=DEFAULT_FS, f_res=DEFAULT_F_RES, normalize=True):
        """FIR taps realising ``self.equalization`` with minimum phase (designed on the GPU)."""
        return minimum_phase_impulse_response(self.frequency, self.equalization, fs, f_res, normalize)
