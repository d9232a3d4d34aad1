import sys, math
sys.path.insert(0, "impulcifer-pip313_amd"); sys.path.insert(0, ".")
import numpy as np
from impulse_hip import Context
from impulse_hip.frequency_response import fir_design_gain
from oracle import scipy_restated as sr
g = np.load("tests/golden/minphase.npz"); ctx = Context(0)
fs = 48000; freq = g["fs48000_freq"]; eq = g["fs48000_tilt_eq"]
lin = fir_design_gain(freq, eq, fs, 5, False); n = len(lin); f = np.linspace(0.0, fs // 2, n)
ir_ref = sr.firwin2_hamming(2 * n, f, lin, fs)
ir = ctx.minphase_debug_stage(lin, fs, 0)[0]
print("firwin2 taps max abs diff", np.abs(ir - ir_ref).max(), "peak", np.abs(ir_ref).max())
alt = ((-1.0) ** np.arange(2 * n))
print("alt sum ref", math.fsum(alt * ir_ref), "dev", math.fsum(alt * ir))
print("asym ref", np.abs(ir_ref - ir_ref[::-1]).max(), "dev", np.abs(ir - ir[::-1]).max())
mag = ctx.minphase_debug_stage(lin, fs, 1)[0]
mref = np.abs(np.fft.fft(ir_ref))
print("mag diff max", np.abs(mag - mref).max(), "nyq dev", mag[n], "ref", mref[n], " fft of dev taps by numpy", np.abs(np.fft.fft(ir))[n])
# spectrum stage pieces: fx interpolation check
nfreqs = 1 + 2 ** int(np.ceil(np.log2(2 * n)))
x = np.linspace(0.0, fs / 2, nfreqs)
fx = np.interp(x, f, lin)
print("nfreqs", nfreqs)
