"""FrequencyResponse: the part of the reference's vendored AutoEQ class that the hot path touches
(autoeq/frequency_response.py): the log grid, log-linear interpolation, centring, target compensation,
fractional-octave smoothing, gain-limited equalisation and the equalisation curve -> minimum-phase FIR step.

Curves are handled as [B, n] float64 matrices on a shared grid: the module-level functions below take every
speaker-ear curve of a measurement at once and run the arithmetic on the device (kernel K12, csrc/curves.hip:
Savitzky-Golay smoothing as a fixed linear operator per window, logistic blends, the gain-limited inversion with its
quadratic-spline kink bridge, the FIR design grid) chained into the minimum-phase FIR design (K6).  The class keeps the
reference's per-object surface and calls the same functions with B = 1.  Re-gridding and centring are linear maps
that depend on the two grids only; they are evaluated as one gather + lerp over the whole matrix.  The parametric-EQ
optimiser, file writers and plots of the AutoEQ class are outside the path and not provided.
"""
import math
import threading

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


_GRIDS = {}


def generate_frequencies(f_min=DEFAULT_F_MIN, f_max=DEFAULT_F_MAX, f_step=DEFAULT_STEP):
    """Geometric grid built by repeated multiplication (autoeq/frequency_response.py:850-857).  Every measurement asks
    for the same grid: the loop runs once per (f_min, f_max, f_step) and callers get their own copy."""
    key = (float(f_min), float(f_max), float(f_step))
    grid = _GRIDS.get(key)
    if grid is None:
        out = []
        f = f_min
        while f <= f_max:
            out.append(f)
            f *= f_step
        if len(_GRIDS) > 64:
            _GRIDS.clear()
        grid = _GRIDS[key] = np.array(out)
    return grid.copy()


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


class _Regrid:
    """log_interp from one grid onto another as (index, weight) pairs: built once per pair of grids, applied to any
    [B, n] matrix of curves with one gather and one lerp."""

    def __init__(self, frequency, f_new):
        xk = np.log10(np.asarray(frequency, dtype=np.float64))
        fq = np.array(f_new, dtype=np.float64)
        fq[fq == 0] = 0.001
        xq = np.log10(fq)
        self.idx = np.clip(np.searchsorted(xk, xq, side="right") - 1, 0, len(xk) - 2)
        self.t = (xq - xk[self.idx]) / (xk[self.idx + 1] - xk[self.idx])

    def __call__(self, curves):
        y = np.asarray(curves, dtype=np.float64)
        lo, hi = y[..., self.idx], y[..., self.idx + 1]
        return lo + self.t * (hi - lo)


_DEFAULT_GRID = None


def center_shifts(frequency, raws, at=1000):
    """What FrequencyResponse.center subtracts from each row of ``raws`` [B, n] (autoeq :903-940): the curve is read
    on the default 20 Hz..20 kHz grid; ``at`` = one frequency (value there) or [f0, f1] (mean between them)."""
    global _DEFAULT_GRID
    if _DEFAULT_GRID is None:
        _DEFAULT_GRID = generate_frequencies()
    on_grid = _Regrid(frequency, _DEFAULT_GRID)(np.atleast_2d(raws))
    if isinstance(at, (list, tuple, np.ndarray)) and len(at) > 1:
        band = np.logical_and(_DEFAULT_GRID >= at[0], _DEFAULT_GRID <= at[1])
        return np.mean(on_grid[:, band], axis=1)
    if isinstance(at, (list, tuple, np.ndarray)):
        at = at[0]
    return _Regrid(_DEFAULT_GRID, [at])(on_grid)[:, 0]


_curves_lock = threading.Lock()
_curves_by_grid = {}


def curves_for(frequency):
    """The device handle (K12) of a frequency grid; grids are few (one per sampling rate), handles are kept."""
    f = np.ascontiguousarray(frequency, dtype=np.float64)
    ctx = _native.default_context()
    key = (id(ctx), f.tobytes())
    with _curves_lock:
        h = _curves_by_grid.get(key)
        if h is None or not h._h:
            h = _native.Curves(ctx, f)
            _curves_by_grid[key] = h
            while len(_curves_by_grid) > 8:
                _curves_by_grid.pop(next(iter(_curves_by_grid))).close()
        return h


def smooth_curves(frequency, curves, window_size=DEFAULT_SMOOTHING_WINDOW_SIZE,
                  treble_window_size=DEFAULT_TREBLE_SMOOTHING_WINDOW_SIZE,
                  treble_f_lower=DEFAULT_TREBLE_SMOOTHING_F_LOWER, treble_f_upper=DEFAULT_TREBLE_SMOOTHING_F_UPPER):
    """_smoothen_fractional_octave (autoeq :1060-1105, one iteration per window) of every row of ``curves``."""
    return curves_for(frequency).smooth(curves, window_size, treble_window_size, treble_f_lower, treble_f_upper)


def equalization_curves(frequency, errors, smoothen_first=True, max_gain=DEFAULT_MAX_GAIN,
                        treble_f_lower=DEFAULT_TREBLE_F_LOWER, treble_f_upper=DEFAULT_TREBLE_F_UPPER,
                        treble_max_gain=DEFAULT_TREBLE_MAX_GAIN, treble_gain_k=DEFAULT_TREBLE_GAIN_K, smoothen=True):
    """(error_smoothed, equalization) for every row of ``errors``: smoothen_heavy_light (when ``smoothen_first``)
    followed by equalize (autoeq :1181-1310)."""
    es, eq, _ = curves_for(frequency).equalization(errors, smoothen_first, max_gain, treble_f_lower, treble_f_upper,
                                                   treble_max_gain, treble_gain_k, smoothen)
    return es, eq


def equalization_firs(frequency, errors, fs, smoothen_first=True, max_gain=DEFAULT_MAX_GAIN,
                      treble_f_lower=DEFAULT_TREBLE_F_LOWER, treble_f_upper=DEFAULT_TREBLE_F_UPPER,
                      treble_max_gain=DEFAULT_TREBLE_MAX_GAIN, treble_gain_k=DEFAULT_TREBLE_GAIN_K, f_res=DEFAULT_F_RES,
                      normalize=True, on_device=False):
    """error curves [B, n] -> (equalization [B, n], minimum-phase FIRs [B, taps]); nothing but the inputs and the
    results crosses the PCIe bus.  on_device: the FIRs stay on the device - (None, _native.DeviceFirs): their consumer
    (HRIR.equalize_channels, the resident slice) takes them there, and they visit the host only if somebody reads them."""
    if on_device:
        return curves_for(frequency).equalization_fir_device(errors, smoothen_first, max_gain, treble_f_lower, treble_f_upper,
                                                             treble_max_gain, treble_gain_k, fs, f_res, normalize)
    return curves_for(frequency).equalization_fir(errors, smoothen_first, max_gain, treble_f_lower, treble_f_upper,
                                                  treble_max_gain, treble_gain_k, fs, f_res, normalize)


def fir_design_gain(frequency, equalization, fs, f_res=5.0, normalize=True):
    """Linear gain on linspace(0, fs//2, n) that the reference hands to firwin2 (autoeq :651-674): dB doubled because
    the homomorphic step halves them, flat below the first grid frequency, zero at Nyquist (computed on the device)."""
    return curves_for(frequency).fir(equalization, fs, f_res, normalize, want_gain=True)[1]


def minimum_phase_impulse_response(frequency, equalization, fs, f_res=5.0, normalize=True):
    """Minimum-phase FIR (n = next_fast_len(fs//2 / (f_res/2)) taps) of one equalisation curve."""
    return curves_for(frequency).fir(equalization, fs, f_res, normalize)


def minimum_phase_impulse_responses(frequency, equalizations, fs, f_res=5.0, normalize=True):
    """Batched form: one launch chain for all speaker-ear curves [B, len(frequency)] -> [B, n]."""
    return curves_for(frequency).fir(np.asarray(equalizations, dtype=np.float64), fs, f_res, normalize)


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
        centre = np.sqrt(f_upper / f_lower) * f_lower
        half = np.log10(f_upper) - np.log10(centre)
        a = 1.0 / (1.0 + np.exp(-((np.log10(self.frequency) - np.log10(centre)) / (half / 4))))
        return a * -(a_normal - a_treble) + a_normal

    def _smoothen_fractional_octave(self, data, window_size=DEFAULT_SMOOTHING_WINDOW_SIZE,
                                    iterations=DEFAULT_SMOOTHING_ITERATIONS, treble_window_size=None,
                                    treble_iterations=None, treble_f_lower=DEFAULT_TREBLE_SMOOTHING_F_LOWER,
                                    treble_f_upper=DEFAULT_TREBLE_SMOOTHING_F_UPPER):
        if iterations != 1 or treble_iterations != 1:
            raise NotImplementedError("every caller on the path smooths once per window")
        if np.any(np.isnan(self.frequency)) or np.any(np.isnan(np.asarray(data, dtype=float))):
            raise ValueError('NaN values present, cannot smoothen!')
        return smooth_curves(self.frequency, data, window_size, treble_window_size, treble_f_lower, treble_f_upper)

    def smoothen_fractional_octave(self, window_size=DEFAULT_SMOOTHING_WINDOW_SIZE,
                                   iterations=DEFAULT_SMOOTHING_ITERATIONS,
                                   treble_window_size=DEFAULT_TREBLE_SMOOTHING_WINDOW_SIZE,
                                   treble_iterations=DEFAULT_TREBLE_SMOOTHING_ITERATIONS,
                                   treble_f_lower=DEFAULT_TREBLE_SMOOTHING_F_LOWER,
                                   treble_f_upper=DEFAULT_TREBLE_SMOOTHING_F_UPPER):
        kw = dict(window_size=window_size, iterations=iterations, treble_window_size=treble_window_size,
                  treble_iterations=treble_iterations, treble_f_lower=treble_f_lower, treble_f_upper=treble_f_upper)
        rows = [key for key in ("raw", "error") if len(getattr(self, key))]
        if rows:
            out = self._smoothen_fractional_octave(np.stack([getattr(self, key) for key in rows]), **kw)
            for key, y in zip(rows, out):
                setattr(self, "smoothed" if key == "raw" else "error_smoothed", y)
        self.reset(raw=False, smoothed=False, error=False, error_smoothed=False, equalization=True,
                   equalized_raw=True, equalized_smoothed=True, target=False)

    smoothen = smoothen_fractional_octave

    def smoothen_heavy_light(self):
        """Error curve smoothed as max(light, heavy) then once more at 1/3 octave (autoeq :1181-1239)."""
        if np.any(np.isnan(np.asarray(self.error, dtype=float))):
            raise ValueError('NaN values present, cannot smoothen!')
        self.smoothed = smooth_curves(self.frequency, self.raw, 1 / 3, 1 / 3, 100, 10000)
        self.error_smoothed = equalization_curves(self.frequency, self.error, smoothen_first=True)[0]
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
        self.equalization = equalization_curves(self.frequency, error, smoothen_first=False, max_gain=max_gain,
                                                treble_f_lower=treble_f_lower, treble_f_upper=treble_f_upper,
                                                treble_max_gain=treble_max_gain, treble_gain_k=treble_gain_k,
                                                smoothen=smoothen)[1]
        self.equalized_raw = self.raw + self.equalization
        if len(self.smoothed):
            self.equalized_smoothed = self.smoothed + self.equalization

    def minimum_phase_impulse_response(self, fs=DEFAULT_FS, f_res=DEFAULT_F_RES, normalize=True):
        """FIR taps realising ``self.equalization`` with minimum phase (designed on the GPU)."""
        return minimum_phase_impulse_response(self.frequency, self.equalization, fs, f_res, normalize)
