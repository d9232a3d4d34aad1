"""ImpulseResponseEstimator with the deconvolution running on the MI355X.

Class surface of reference core/impulse_response_estimator.py:26-273.  Sweep and inverse-filter
generation is one-time float64 host set-up; ``estimate`` (the reference's serial per-channel
hot loop, core/hrir.py:307-355) goes through libimpulse_hip.so and has a batched form that the
reference lacks: ``estimate_batch`` deconvolves every column/track of a recording in one launch
group.  There is no CPU fallback: without the library or a gfx950 device, estimate raises.
"""
import threading

import numpy as np

from . import _native
from .audio_io import read_wav
from .constants import SEQUENCE_TRACK_ORDERS, SPEAKER_NAMES


def _hann_sym(M):
    if M <= 0:
        return np.zeros(0)
    if M == 1:
        return np.ones(1)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(M) / (M - 1))


class ImpulseResponseEstimator(object):
    """Farina exponential sine sweep: probe generation + recording -> impulse response."""

    def __init__(self, min_duration=5.0, fs=44100):
        if fs != int(fs):
            raise ValueError('Sampling rate "fs" must be an integer.')
        self.fs = int(fs)
        self.high = self.fs / 2                      # sweep ends at Nyquist
        self.low = 5
        self.n_octaves = np.ceil(np.log2(self.high / self.low))   # P, integer-valued float
        self.low = self.high / 2 ** self.n_octaves
        self.w1 = self.low / self.fs * 2 * np.pi
        self.w2 = self.high / self.fs * 2 * np.pi
        self._plans = {}
        self._plan_lock = threading.Lock()
        self.test_signal = self.generate_test_signal(min_duration)
        self.duration = len(self.test_signal) / self.fs
        self.inverse_filter = self.generate_inverse_filter()

    # -- plumbing so instances survive pickle / deepcopy (process pools, HRIR.copy) ------------
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_plans"] = {}
        st["_plan_lock"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._plans = {}
        self._plan_lock = threading.Lock()

    def __setattr__(self, name, value):
        # a new inverse filter invalidates the cached device spectra (from_wav re-binds it)
        if name == "inverse_filter" and getattr(self, "_plans", None):
            self._drop_plans()
        object.__setattr__(self, name, value)

    def _drop_plans(self):
        with self._plan_lock:
            for p in self._plans.values():
                p.close()
            self._plans = {}

    def _forget_context(self, ctx):
        """close the plans made on `ctx` (a worker's context that is about to be closed)"""
        with self._plan_lock:
            for key in [k for k in self._plans if k[2] == id(ctx)]:
                self._plans.pop(key).close()

    def __len__(self):
        return len(self.test_signal)

    # -- set-up (host, float64) ---------------------------------------------------------------
    def generate_test_signal(self, min_duration, fade_in=1 / 2, fade_out=None):
        """ESS on the Garai-Guidorzi length grid with a Hann fade-in of ``fade_in`` octaves
        (reference :86-147)."""
        P = self.n_octaves
        octaves_ln = np.log(2 ** P)
        mult = np.ceil(min_duration * self.fs * (np.pi / 2 ** P) / (np.pi * 2 * octaves_ln))
        L = mult * np.pi * 2 * octaves_ln / (np.pi / 2 ** P)
        N = np.round(L)
        n = np.arange(N)
        signal = np.sin(np.pi / 2 ** P * L / octaves_ln * np.exp(n / N * octaves_ln))
        per_octave = N / self.fs / P

        def fade(octaves, rising):
            if octaves is None:
                return np.zeros(0)
            m = 2 * int(self.fs * per_octave * octaves)
            m += m % 2
            w = _hann_sym(m)
            return w[: m // 2] if rising else w[m // 2:]

        head, tail = fade(fade_in, True), fade(fade_out, False)
        signal[: len(head)] *= head
        if len(tail):
            signal[len(signal) - len(tail):] *= tail
        return signal

    def generate_inverse_filter(self):
        """Time-reversed sweep with a -6 dB/octave envelope, scaled so that the sweep*filter
        spectrum has unit magnitude at a quarter of the (2N-1)-point band (reference :73-84).

        The reference evaluates ``abs(fft(convolve(inv, sweep))[round((2N-1)/4)])``; the full
        convolution has exactly 2N-1 samples, so that bin equals the product of the two
        zero-padded single-bin DFTs, which is what is computed here (O(N), float64)."""
        P = self.n_octaves
        x = np.asarray(self.test_signal, dtype=np.float64)
        N = len(x)
        inv = np.flip(x) * (2 ** (P / N)) ** (np.arange(N) * -1) * P * np.log(2) / (1 - 2 ** -P)
        n_full = 2 * N - 1
        k0 = round(n_full / 4)
        # exp(-2 pi i k0 n / n_full) with the phase reduced in integers to keep float64 exact
        ph = (k0 * np.arange(N, dtype=np.int64)) % n_full
        w = np.exp(-2j * np.pi * ph / n_full)
        scale = np.abs(np.dot(inv, w) * np.dot(x, w))
        return inv / scale

    # -- hot path (device) --------------------------------------------------------------------
    def _plan(self, L, paired=False):
        """The deconvolution plan for columns of L samples on the calling thread's context.  paired: the pair-mode plan
        (two ears of a speaker as one complex signal, core/hrir.py:326-341 hands estimate() exactly such pairs) where the
        lengths allow it, else the one-channel-per-transform plan - a pair plan has no even/odd unpack, whose rounding
        error the un-cropped column shows near Nyquist."""
        L = int(L)
        ctx = _native.default_context()
        with self._plan_lock:
            return self._plan_on(ctx, L, bool(paired))

    def _plan_on(self, ctx, L, paired):
        """(under the plan lock)  The inverse-sweep spectrum is prepared once, on the process's root context; the plan of
        any other context - another device of IMPULSE_HIP_DEVICES, a worker's stream - is made empty and receives a copy
        (hipMemcpyPeer: the one datum the GPUs of a process share)."""
        key = (L, paired, id(ctx))
        plan = self._plans.get(key)
        if plan is not None and not plan._h:               # its context was closed
            plan = None
        if plan is None:
            root = _native.root_context()
            inv = np.asarray(self.inverse_filter, dtype=np.float64)
            if ctx is root:
                plan = _native.ConvPlan(ctx, inv, L, "same", paired="auto" if paired else False)
            else:
                src = self._plan_on(root, L, paired)
                plan = _native.ConvPlan(ctx, None, L, "same", empty_M=len(inv), n_filters=1, paired=src.paired)
                plan.copy_spectrum_from(src)
            self._plans[key] = plan
        return plan

    def estimate(self, recording):
        """Impulse response of one recorded channel: convolve(recording, inverse_filter, 'same')."""
        rec = np.asarray(recording)
        if rec.ndim != 1:
            raise ValueError("estimate() takes a 1-D recording; use estimate_batch for [B, L]")
        if len(rec) == 0:
            return np.zeros(0)
        return self._plan(len(rec)).execute(rec).astype(np.float64)

    def estimate_batch(self, recordings, dtype=np.float64):
        """Deconvolve B equally long channels at once: [B, L] -> [B, L]."""
        rec = np.asarray(recordings)
        if rec.ndim != 2:
            raise ValueError("estimate_batch() takes [B, L]")
        if rec.shape[0] == 0 or rec.shape[1] == 0:
            return np.zeros(rec.shape, dtype=dtype)
        ctxs = _native.device_contexts() if getattr(_native._thread_ctx, "ctx", None) is None else None
        if ctxs is not None and len(ctxs) > 1 and rec.shape[0] > 1:
            # several devices in this process (IMPULSE_HIP_DEVICES): contiguous channel blocks, pairs kept together, one host
            # thread per device - the reference's pool over channels (core/parallel_utils.py:97-152) with a GPU per worker
            from .sharding import device_shards, run_sharded
            shards = device_shards(rec.shape[0], len(ctxs))
            parts = run_sharded(ctxs, shards, lambda ctx, lo, hi: self._plan(rec.shape[1]).execute(rec[lo:hi]))
            out = np.concatenate(parts, axis=0)
        else:
            out = self._plan(rec.shape[1]).execute(rec)
        return out if dtype == np.float32 else out.astype(dtype)

    def estimate_frames(self, frames, dtype=np.float64):
        """Deconvolve interleaved frames [L, C] (WAV wire order) -> [C, L]; the de-interleave is
        done by the device loader, not by a host transpose."""
        fr = np.asarray(frames)
        if fr.ndim != 2:
            raise ValueError("estimate_frames() takes [L, C]")
        out = self._plan(fr.shape[0]).execute_interleaved(fr)
        return out if dtype == np.float32 else out.astype(dtype)

    # -- recording-side helpers ---------------------------------------------------------------
    def sweep_sequence(self, speakers, tracks):
        """Multi-track playback sequence: 2 s silence, then per speaker a sweep + 2 s silence on
        that speaker's track (reference :153-232)."""
        # (the reference's uniqueness loop, core/impulse_response_estimator.py:186-189, never appends to its list and so
        # never raises: duplicate speakers pass, and so they do here)
        if tracks in SEQUENCE_TRACK_ORDERS:
            order = SEQUENCE_TRACK_ORDERS[tracks]
            n_tracks = len(order)
        elif tracks == 'stereo':
            if not 1 <= len(speakers) <= 2:
                raise ValueError('"stereo" track configuration requires one or two speakers.')
            for sp in speakers:
                if sp not in SPEAKER_NAMES:
                    raise ValueError(f'Speaker name "{sp}" is not a recognised speaker.')
            order = list(speakers)
            n_tracks = 2
        elif tracks == 'mono':
            order = ['FL']
            speakers = ['FL']
            n_tracks = 1
        else:
            raise ValueError(f'Unsupported track configuration "{tracks}".')
        for sp in speakers:
            if sp not in order:
                raise ValueError(f'Speaker name "{sp}" not supported with track configuration "{tracks}"')
        gap = self.fs * 2.0
        slot = gap + len(self)
        data = np.zeros((n_tracks, int(slot * len(speakers) + gap)))
        for i, sp in enumerate(speakers):
            start = int(slot * i + gap)
            data[order.index(sp), start:start + len(self)] = self.test_signal
        return data

    @classmethod
    def from_wav(cls, file_path):
        """Estimator for a sweep stored in a WAV file (reference :234-262): the file's samples
        replace the generated sweep when the lengths differ or the values differ by > 1e-4."""
        fs, data = read_wav(file_path)
        ref = data[0, :] if data.ndim > 1 else data
        ire = cls(min_duration=(len(ref) - 1) / fs, fs=fs)
        if len(ire.test_signal) != len(ref):
            ire.test_signal = np.array(ref, dtype=np.float64)
            ire.duration = len(ref) / fs
            ire.inverse_filter = ire.generate_inverse_filter()
        elif np.max(np.abs(ire.test_signal - ref)) > 1e-4:
            print("Warning: Loaded WAV differs slightly from generated signal. "
                  "Re-calculating inverse filter based on WAV.")
            ire.test_signal = np.array(ref, dtype=np.float64)
            ire.inverse_filter = ire.generate_inverse_filter()
        return ire

    def file_name(self, bit_depth):
        return f'{self.duration:.2f}s-{self.fs:d}Hz-{bit_depth:d}bit-{self.low:.2f}Hz-{self.high:.0f}Hz'
