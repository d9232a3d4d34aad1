"""Oracle: equalization curve -> minimum-phase FIR (reference autoeq/frequency_response.py:637-681)."""
import numpy as np

from .impulse_response import interpolate_log
from .scipy_restated import firwin2_hamming, minimum_phase_homomorphic, next_fast_len_real, spline1_eval


def minimum_phase_impulse_response(frequency, equalization, fs, f_res=5.0, normalize=False):
    frequency = np.asarray(frequency, dtype=np.float64)
    eq = np.asarray(equalization, dtype=np.float64)
    f_res = f_res / 2                                           # halved again by the homomorphic step
    f_min = np.max([frequency[0], f_res])
    gain_f_min = spline1_eval(np.log10(frequency), eq, np.log10(f_min))
    n = next_fast_len_real(round(fs // 2 / f_res))
    f = np.linspace(0.0, fs // 2, n)
    raw = interpolate_log(frequency, eq, f)
    raw[f <= f_min] = gain_f_min
    if normalize:
        raw -= np.max(raw)
        raw -= 0.5
    raw *= 2                                                    # minimum_phase(half=True) halves dB gain
    lin = 10 ** (raw / 20)
    lin[-1] = 0.0
    ir = firwin2_hamming(len(f) * 2, f, lin, fs)
    return minimum_phase_homomorphic(ir, n_fft=len(ir))
