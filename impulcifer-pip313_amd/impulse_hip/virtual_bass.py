"""Virtual bass synthesis (surface of reference core/virtual_bass.py:19-200).

One shared band-limited bass impulse is gain-matched to the measured responses at the crossover and added
to every speaker/ear pair after the measured response has been high-passed there.  The filter DESIGNS
(Butterworth sections, RBJ shelves: a few dozen scalars) come from SciPy on the host like in the reference;
every application of a cascade to signal data - scipy.signal.sosfilt in the reference - runs on the device
(K11, fp64, bit-identical), all responses of the set in one batch.
"""
import numpy as np

from . import _native
from .constants import speaker_side


def _detect_polarity(ir):
    return 1.0 if ir[np.argmax(np.abs(ir))] >= 0 else -1.0


def _shift(ir, n_samples):
    """Circular shift (core/virtual_bass.py:26-28; kept for the module's surface, the synthesis uses _delay_signal)."""
    return np.roll(ir, n_samples)


def _rfft_magnitude(ir, fs):
    """(|rfft(ir)|, bin frequencies) (core/virtual_bass.py:46-50)."""
    return np.abs(np.fft.rfft(ir)), np.fft.rfftfreq(len(ir), 1.0 / fs)


def _delay_signal(sig, delay, length):
    """Delay (>= 0) or advance (< 0) with zero padding, never wrapping."""
    out = np.zeros(length, dtype=sig.dtype)
    if delay >= 0:
        if delay < length:
            n = min(length - delay, len(sig))
            out[delay: delay + n] = sig[:n]
    else:
        adv = -delay
        if adv < len(sig):
            n = min(length, len(sig) - adv)
            out[:n] = sig[adv: adv + n]
    return out


def _mag_at(ir, fs, freq_hz):
    """|rfft(ir)| at the bin nearest freq_hz (a single bin: evaluated as one dot product)."""
    n = len(ir)
    freqs = np.fft.rfftfreq(n, 1.0 / fs)
    k = int(np.argmin(np.abs(freqs - freq_hz)))
    return float(np.abs(np.fft.rfft(ir)[k]))


def _duplicate_sos(sos, times):
    return np.vstack([sos for _ in range(times)])


def _rbj_high_shelf(fc, fs, gain_db, q):
    from scipy import signal
    a = 10 ** (gain_db / 40.0)
    w0 = 2 * np.pi * fc / fs
    alpha = np.sin(w0) / (2 * q)
    cw = np.cos(w0)
    b = [a * ((a + 1) + (a - 1) * cw + 2 * np.sqrt(a) * alpha), -2 * a * ((a - 1) + (a + 1) * cw),
         a * ((a + 1) + (a - 1) * cw - 2 * np.sqrt(a) * alpha)]
    den = [(a + 1) - (a - 1) * cw + 2 * np.sqrt(a) * alpha, 2 * ((a - 1) - (a + 1) * cw),
           (a + 1) - (a - 1) * cw - 2 * np.sqrt(a) * alpha]
    return signal.tf2sos(b, den)


def synthesize_virtual_bass(irs, fs, crossover_freq=250, head_ms=1.0, hp_freq=15.0, invert_polarity=None):
    """In place on {speaker: {side: ImpulseResponse}} (reference :82-176)."""
    from scipy import signal
    if crossover_freq >= fs / 2:
        return
    ctx = _native.default_context()
    n_ir = max(len(ir.data) for pair in irs.values() for ir in pair.values())
    for pair in irs.values():
        for side in ("left", "right"):
            if side in pair and len(pair[side].data) < n_ir:
                pair[side].data = np.pad(pair[side].data, (0, n_ir - len(pair[side].data)))

    imp = np.zeros(n_ir)
    imp[0] = 1.0
    sos_hp4_sub = signal.butter(4, hp_freq / (fs / 2), btype="high", output="sos")
    sos_lp8_xo = _duplicate_sos(signal.butter(4, crossover_freq / (fs / 2), btype="low", output="sos"), 2)
    mpbass = ctx.sosfilt(sos_lp8_xo, ctx.sosfilt(sos_hp4_sub, [imp]))[0]
    sos_ild = np.vstack([_rbj_high_shelf(fc, fs, g, q) for fc, g, q in ((150.0, -1.5, 0.760), (400.0, -3.0, 0.660),
                                                                         (800.0, -3.5, 0.610))])
    sos_hp8_xo = _duplicate_sos(signal.butter(4, crossover_freq / (fs / 2), btype="high", output="sos"), 2)

    pairs = [(sp, pair) for sp, pair in irs.items() if "left" in pair and "right" in pair]
    if not pairs:
        return
    # every measured response through the crossover high-pass, one batch
    highs = ctx.sosfilt(sos_hp8_xo, [pair[sd].data for _, pair in pairs for sd in ("left", "right")])
    mean_xo = float(np.mean([_mag_at(h, fs, crossover_freq) for h in highs]))
    gain = mean_xo / (_mag_at(mpbass, fs, crossover_freq) + 1e-20)
    head = int(round(head_ms * 1e-3 * fs))
    direct_undelayed = mpbass * gain * (-1.0 if invert_polarity else 1.0)
    cross_undelayed = ctx.sosfilt(sos_ild, [direct_undelayed])[0]
    for i, (speaker, pair) in enumerate(pairs):
        on_left = speaker_side(speaker) == "left"
        itd = int(pair["right"].peak_index()) - int(pair["left"].peak_index())
        direct = _delay_signal(direct_undelayed, head, n_ir)
        cross = _delay_signal(cross_undelayed, head + (itd if on_left else -itd), n_ir)
        pair["left"].data = highs[2 * i] + (direct if on_left else cross)
        pair["right"].data = highs[2 * i + 1] + (cross if on_left else direct)


def apply_virtual_bass_to_hrir(hrir, crossover_freq=250, head_ms=1.0, hp_freq=15.0, invert_polarity=None):
    synthesize_virtual_bass(hrir.irs, hrir.fs, crossover_freq, head_ms, hp_freq, invert_polarity)
