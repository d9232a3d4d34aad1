"""Synthetic FL,FR measurement shared by the golden generator and the tests (inputs only).

Four tracks (FL-left, FL-right, FR-left, FR-right; the track layout of the reference's
tests/test_pipeline_direct.py:182-204) of a sweep played through a small synthetic room: a direct
path (delay, gain) plus an exponentially decaying reverberant tail (RT60 0.25 s, -26 dB) and a
-80 dBFS noise floor, so that the Lundeby knee and the tail crop are decided by the signal and not
by arithmetic noise.  NumPy only; deterministic for a given test signal.
"""
import numpy as np

SPECS = ((0, 1.0), (12, 0.6), (12, 0.6), (0, 0.9))      # (direct-path delay, gain) per track
ROOM_LEN = 16000


def room_ir(channel, delay, gain, fs):
    rng = np.random.default_rng(0x51CE + channel)
    t = np.arange(ROOM_LEN) / fs
    h = rng.standard_normal(ROOM_LEN) * 0.05 * gain * 10 ** (-3.0 * t / 0.25)
    h[: delay + 24] = 0.0
    h[delay] = gain
    return h


def make_tracks(test_signal, fs):
    """float64 [4, 2 fs + N + 2 fs] before PCM rounding."""
    N = len(test_signal)
    total = 2 * fs + N + 2 * fs
    tracks = np.zeros((4, total))
    nfft = 1 << int(np.ceil(np.log2(N + ROOM_LEN)))
    S = np.fft.rfft(0.5 * np.asarray(test_signal, dtype=np.float64), nfft)
    for c, (delay, gain) in enumerate(SPECS):
        y = np.fft.irfft(S * np.fft.rfft(room_ir(c, delay, gain, fs), nfft), nfft)[: N + ROOM_LEN - 1]
        tracks[c, 2 * fs: 2 * fs + len(y)] = y
        tracks[c] += np.random.default_rng(0xF100 + c).standard_normal(total) * 10 ** (-80 / 20)
    return tracks


def to_pcm32(tracks):
    return np.clip(np.rint(tracks * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32)
