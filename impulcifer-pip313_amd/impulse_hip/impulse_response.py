"""ImpulseResponse: one measured channel (class surface of reference core/impulse_response.py:15-188).

``data`` stays a public, writable NumPy array at all times (callers mutate and re-bind it); the
array work - first-peak search, FIR filtering, decay windows - is sent to the device per call.
"""
from copy import deepcopy

import numpy as np

from . import _native, decay
from .audio_io import magnitude_response

EPSILON = 1e-20


class _PlanCache:
    """K5 plans kept per shape (signal length, taps, channels): the FIRs change with every measurement, the shapes do
    not, so a plan - workspace, tables, staging buffers - is made once per shape and only its filter spectra are
    recomputed (imp_plan_set_filters).  A handful of shapes live at a time; the least recently used plan goes first."""

    def __init__(self, capacity=8):
        import threading
        self.capacity = capacity
        self.lock = threading.Lock()          # set_filters + execute of one plan is one critical section
        self.plans = {}                       # key -> plan, in LRU order

    def run(self, signals, taps):
        ctx = _native.default_context()
        key = (id(ctx), signals.shape[1], taps.shape[1], signals.shape[0])
        with self.lock:
            plan = self.plans.pop(key, None)
            if plan is not None and not plan._h:                  # closed behind our back (context torn down)
                plan = None
            if plan is None:
                plan = _native.ConvPlan(ctx, taps, signals.shape[1], "full", ws_channels=signals.shape[0])
            else:
                plan.set_filters(taps)
            self.plans[key] = plan
            while len(self.plans) > self.capacity:
                self.plans.pop(next(iter(self.plans))).close()
            return plan.execute(signals)

    def run_device(self, d_in, B, stride_in, n, taps, d_out, stride_out):
        """Device rows in (B rows of n samples, stride_in apart), device rows out (n + taps - 1 samples, stride_out
        apart); taps: host [B, K], or a _native.DeviceFirs batch of B FIRs (no upload then).  Asynchronous on the context
        stream."""
        ctx = _native.default_context()
        on_device = isinstance(taps, _native.DeviceFirs)
        K = taps.taps if on_device else taps.shape[1]
        key = (id(ctx), int(n), K, int(B))
        with self.lock:
            plan = self.plans.pop(key, None)
            if plan is not None and not plan._h:
                plan = None
            if on_device:
                if plan is None:
                    plan = _native.ConvPlan(ctx, None, int(n), "full", ws_channels=int(B), empty_M=K, n_filters=int(B))
                if not plan.fused and plan.n_filters != B:
                    raise ValueError("device FIRs: one filter per channel")
                taps.ready()                                # the design ran on a stream of its own
                plan.set_filters_device(taps.ptr, K)
                plan._fir_batch = taps                       # the spectra are formed in stream order: keep the taps alive
            elif plan is None:
                plan = _native.ConvPlan(ctx, taps, int(n), "full", ws_channels=int(B))
            else:
                plan.set_filters(taps)
            self.plans[key] = plan
            while len(self.plans) > self.capacity:
                self.plans.pop(next(iter(self.plans))).close()
            plan.execute_device(d_in, B, stride_in, d_out, stride_out)
            return plan.out_len


_k5_plans = _PlanCache()


def fir_convolve_full(x, taps):
    """scipy.signal.convolve(x, taps, 'full') on the device (K5)."""
    x = np.asarray(x)
    taps = np.asarray(taps, dtype=np.float64)
    if len(x) == 0 or len(taps) == 0:
        return np.zeros(0)
    return _k5_plans.run(x[None, :], taps[None, :])[0].astype(np.float64)


def fir_convolve_full_batch(signals, taps):
    """[scipy.signal.convolve(x_i, taps_i, 'full')] for equally long signals and equally long FIRs, as ONE
    plan with per-channel filters and one launch group (K5) instead of a plan per channel."""
    xs = [np.asarray(x) for x in signals]
    hs = [np.asarray(h, dtype=np.float64) for h in taps]
    if len(xs) != len(hs):
        raise ValueError("fir_convolve_full_batch: one FIR per signal")
    out = [None] * len(xs)
    groups = {}
    for i, (x, h) in enumerate(zip(xs, hs)):
        if len(x) == 0 or len(h) == 0:
            out[i] = np.zeros(0)
        else:
            groups.setdefault((len(x), len(h)), []).append(i)
    for idx in groups.values():
        y = _k5_plans.run(np.stack([xs[i] for i in idx]), np.stack([hs[i] for i in idx]))
        for row, i in zip(y, idx):
            out[i] = row.astype(np.float64)
    return out


class ImpulseResponse(object):
    def __init__(self, data, fs, recording=None):
        self.fs = fs
        self._row = None              # device_rows.Row while the samples live on the GPU (then _data is None)
        self._data = data
        self._recording = recording   # array, or a callable producing it on first use

    @classmethod
    def on_device(cls, row, fs, recording=None):
        """A response whose samples are a row of a device block (see device_rows.py)."""
        ir = cls(None, fs, recording)
        ir._row = row
        return ir

    @property
    def data(self):
        """The samples as a writable float64 array.  Reading this from a device-resident response brings the row to the
        host; from then on the host array is the truth (callers mutate and re-bind it, as with the reference)."""
        if self._data is None and self._row is not None:
            self._data = self._row.to_host()
            self._row = None
        return self._data

    @data.setter
    def data(self, value):
        self._data = value
        self._row = None

    def peek(self):
        """A copy of the samples that leaves a device-resident response on the device (tests, snapshots)."""
        return self._row.to_host() if (self._data is None and self._row is not None) else np.array(self._data, copy=True)

    @property
    def recording(self):
        if callable(self._recording):
            self._recording = self._recording()
        return self._recording

    @recording.setter
    def recording(self, value):
        self._recording = value

    def __getstate__(self):
        return {"fs": self.fs, "_row": None, "_data": self.data, "_recording": self.recording}

    def copy(self):
        return deepcopy(self)

    def __deepcopy__(self, memo):
        # peek(): copying a device-resident response must not move the original to the host
        other = ImpulseResponse(self.peek(), self.fs, deepcopy(self.recording, memo))
        for k, v in self.__dict__.items():
            if k not in ("fs", "_row", "_data", "_recording"):
                other.__dict__[k] = deepcopy(v, memo)
        return other

    def __len__(self):
        return self._row.n if (self._data is None and self._row is not None) else len(self._data)

    def duration(self):
        return len(self) / self.fs

    def peak_index(self, start=0, end=None, peak_height=0.12589):
        """Index of the first positive or negative peak within -18 dB of the largest sample."""
        return decay._peak_index(self.data, start, end, peak_height)

    def decay_params(self):
        return decay.decay_params(self.data, self.fs)

    def decay_times(self, peak_ind=None, knee_point_ind=None, noise_floor=None, window_size=None):
        return decay.decay_times(self.data, self.fs, peak_ind, knee_point_ind, noise_floor, window_size)

    def crop_head(self, head_ms=1):
        if len(self.data) == 0:
            return
        first = self.peak_index() - int(self.fs * head_ms / 1000)
        self.data = self.data[max(first, 0):]

    def shift(self, samples):
        """Delay (samples > 0) or advance (samples < 0) keeping the length.  A response that is on the device stays there
        (the shifted copy is a new device row)."""
        if self._data is None and self._row is not None:
            if samples != 0:
                from .device_rows import shift_rows
                self._row = shift_rows([self._row], [int(samples)])[0]
            return
        n = len(self.data)
        if samples > 0:
            self.data = np.concatenate((np.zeros(samples), self.data))[:n]
        elif samples < 0:
            rest = self.data[-samples:]
            self.data = np.pad(rest, (0, n - len(rest))) if len(rest) < n else rest

    def equalize(self, fir):
        """Filter with a FIR: data becomes the full convolution (length n + taps - 1)."""
        self.data = fir_convolve_full(self.data, fir)

    def resample(self, fs):
        raise NotImplementedError(
            "resample depends on nnresample, which has no oracle in this build (parity unpinned); "
            "out of scope for the device path")

    def convolve(self, x):
        return fir_convolve_full(x, self.data)

    def decay_adjustment_params(self, target):
        return decay.decay_adjustment_params(self.data, self.fs, target)

    def adjust_decay(self, target):
        decay.apply_decay_window(self.data, self.decay_adjustment_params(target))

    def magnitude_response(self):
        return magnitude_response(self.data, self.fs)

    def frequency_response(self):
        """Magnitude response decimated to ~4 Hz steps and re-sampled on the 1 % log grid from 10 Hz
        to fs/2 (reference :157-188)."""
        from .frequency_response import FrequencyResponse, generate_frequencies

        def flat(label):
            f = generate_frequencies(f_step=1.01, f_min=10, f_max=self.fs / 2)
            return FrequencyResponse(name=label, frequency=f, raw=np.zeros_like(f))

        if len(self.data) < 2:
            return flat("Frequency response (short IR)")
        f, m = self.magnitude_response()
        if len(f) == 0:
            return flat("Frequency response (empty FFT)")
        wanted = (self.fs / 2) / 4.0
        step = 1 if (wanted < 2 or len(f) < 2) else (int(round(len(f) / wanted)) or 1)
        if len(f[1::step]) == 0:
            if len(f[1:]) == 0:
                return flat("Frequency response (FFT too short)")
            frequency, raw = f[1:], m[1:]
        else:
            frequency, raw = f[1::step], m[1::step]
        fr = FrequencyResponse(name="Frequency response", frequency=frequency, raw=raw)
        fr.interpolate(f_step=1.01, f_min=10, f_max=self.fs / 2)
        return fr
