"""Oracle for core/virtual_bass.py (test infrastructure only)."""
import numpy as np


def sosfilt(sos, x):
    """scipy.signal.sosfilt(sos, x) with zero initial state, restating scipy/signal/_sosfilt.pyx (_sosfilt_float):
    direct form II transposed, per sample and section y = b0 x + z0; z0 = b1 x - a1 y + z1; z1 = b2 x - a2 y.
    Vectorised over rows (x: [rows, n]); every operation is a separately rounded float64 multiply/add."""
    sos = np.asarray(sos, dtype=np.float64).reshape(-1, 6)
    x = np.atleast_2d(np.array(x, dtype=np.float64))
    z = np.zeros((len(sos), 2, x.shape[0]))
    for n in range(x.shape[1]):
        cur = x[:, n].copy()
        for s in range(len(sos)):
            b0, b1, b2, _, a1, a2 = sos[s]
            out = b0 * cur + z[s, 0]
            z[s, 0] = b1 * cur - a1 * out + z[s, 1]
            z[s, 1] = b2 * cur - a2 * out
            cur = out
        x[:, n] = cur
    return x


def delay_signal(sig, delay, length):
    """core/virtual_bass.py:30-43."""
    out = np.zeros(length)
    if delay >= 0:
        if delay < length:
            n = min(length - delay, len(sig))
            out[delay: delay + n] = sig[:n]
    else:
        if -delay < len(sig):
            n = min(length, len(sig) + delay)
            out[:n] = sig[-delay: -delay + n]
    return out


def mag_at(ir, fs, freq_hz):
    """core/virtual_bass.py:46-57."""
    mag = np.abs(np.fft.rfft(ir))
    freqs = np.fft.rfftfreq(len(ir), 1.0 / fs)
    return float(mag[np.argmin(np.abs(freqs - freq_hz))])
