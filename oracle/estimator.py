"""Oracle: Farina exponential-sine-sweep estimator (reference core/impulse_response_estimator.py)."""
import numpy as np

from .scipy_restated import fft_convolve, hann


def sweep_constants(fs, low_limit=5.0):
    """core/impulse_response_estimator.py:33-46 -- P octaves (float), low/high edge, w1/w2."""
    if fs != int(fs):
        raise ValueError('Sampling rate "fs" must be an integer.')
    fs = int(fs)
    high = fs / 2
    P = np.ceil(np.log2(high / low_limit))
    low = high / 2 ** P
    return dict(fs=fs, high=high, low=low, P=P, w1=low / fs * 2 * np.pi, w2=high / fs * 2 * np.pi)


def generate_test_signal(fs, min_duration, fade_in=0.5, fade_out=None):
    """core/impulse_response_estimator.py:86-147 (Garai & Guidorzi 2015 eq. 2 length grid)."""
    c = sweep_constants(fs)
    P = c["P"]
    ln2P = np.log(2 ** P)
    Mmul = np.ceil(min_duration * fs * (np.pi / 2 ** P) / (np.pi * 2 * ln2P))
    Lreal = Mmul * np.pi * 2 * ln2P / (np.pi / 2 ** P)
    N = np.round(Lreal)
    phase = np.pi / 2 ** P * Lreal / ln2P * np.exp(np.arange(N) / N * ln2P)
    sig = np.sin(phase)
    sec_per_oct = N / fs / P

    def half_window(octaves, rising):
        if octaves is None:
            return np.zeros(0)
        n = 2 * int(fs * sec_per_oct * octaves)
        if n % 2:
            n += 1
        w = hann(n)
        return w[: n // 2] if rising else w[n // 2:]

    wi, wo = half_window(fade_in, True), half_window(fade_out, False)
    win = np.concatenate([wi, np.ones(len(sig) - len(wi) - len(wo)), wo])
    return sig * win


def generate_inverse_filter(test_signal, P):
    """core/impulse_response_estimator.py:73-84: time-reversed sweep with the +6 dB/oct envelope,
    then one scalar normalisation by |FFT(full_conv(inv, sweep))[round(len/4)]|."""
    N = len(test_signal)
    inv = np.flip(test_signal) * (2 ** (P / N)) ** (np.arange(N) * -1) * P * np.log(2) / (1 - 2 ** -P)
    frp = np.fft.fft(fft_convolve(inv, test_signal, "full"))
    return inv / np.abs(frp[round(frp.shape[0] / 4)])


def estimate(recording, inverse_filter):
    """core/impulse_response_estimator.py:149-151: convolve(recording, inverse_filter, 'same')."""
    return fft_convolve(recording, inverse_filter, "same")


class Estimator:
    """Minimal stand-in with the attributes the rest of the oracle needs."""

    def __init__(self, min_duration=5.0, fs=44100, test_signal=None):
        c = sweep_constants(fs)
        self.fs = c["fs"]
        self.high, self.low, self.n_octaves, self.w1, self.w2 = c["high"], c["low"], c["P"], c["w1"], c["w2"]
        if test_signal is None:
            self.test_signal = generate_test_signal(self.fs, min_duration)
        else:
            # from_wav mismatch branch, core/impulse_response_estimator.py:250-254
            self.test_signal = np.asarray(test_signal, dtype=np.float64)
        self.duration = len(self.test_signal) / self.fs
        self.inverse_filter = generate_inverse_filter(self.test_signal, self.n_octaves)

    def __len__(self):
        return len(self.test_signal)

    def estimate(self, recording):
        return estimate(recording, self.inverse_filter)


def sweep_sequence_layout(n_speakers, N, fs, silence=2.0):
    """core/impulse_response_estimator.py:220-229: total length and sweep start offsets."""
    step = int(fs * silence + N)
    total = int((fs * silence + N) * n_speakers + fs * silence)
    starts = [int(step * i + fs * silence) for i in range(n_speakers)]
    return total, starts


def from_wav_samples(samples, fs):
    """core/impulse_response_estimator.py:234-262 after the file has been read (track 0 of the WAV, float64 in
    [-1, 1)): min_duration = (len - 1) / fs; the file's samples replace the generated sweep when the LENGTH differs
    (:250-254, duration := len / fs) or when any sample differs by more than 1e-4 (:256-260); the inverse filter is
    regenerated in both cases."""
    samples = np.asarray(samples, dtype=np.float64)
    e = Estimator(min_duration=(len(samples) - 1) / fs, fs=fs)
    if len(e.test_signal) != len(samples):
        e.test_signal = samples
        e.duration = len(samples) / fs
        e.inverse_filter = generate_inverse_filter(samples, e.n_octaves)
    elif np.max(np.abs(e.test_signal - samples)) > 1e-4:
        e.test_signal = samples
        e.inverse_filter = generate_inverse_filter(samples, e.n_octaves)
    return e
