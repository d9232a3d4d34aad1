"""NumPy model of the device dataflow used by the HIP deconvolution kernels.

This is NOT the oracle and NOT a product path: it mirrors, thread for thread,
the index/twiddle algebra of csrc/conv_kernels.hip.h (pass A column FFT,
pass B fused row FFT * spectrum * row IFFT, pass C column IFFT + crop) so the
algebra can be unit-tested on a CPU-only box.  All arithmetic is float64 here;
the kernels do the same steps in fp32.

Layout conventions (identical to the kernels):
  nfft = 2*Nc, Nc = N1*N2, N2 = 4096 (row length), N1 = F*R2 (column length, F = 16 or 8)
  z[n] = x[2n] + i x[2n+1]            (even/odd packing of one real channel)
  n  = N2*n1 + n2   -> workspace[k1][n2] after pass A   (k1 = ka + 16*kb)
  k  = k1 + N1*k2   -> workspace[k1][k2] inside pass B  (transposed spectrum order)
"""
import numpy as np

N2 = 4096


def w(n, k, N):
    return np.exp(-2j * np.pi * ((n * k) % N) / N)


def plan_alpha_beta(h, nfft):
    """alpha/beta of the fused middle step, in workspace [k1][k2] order.

    W[k] = alpha[k]*Z[k] + beta[k]*conj(Z[Nc-k]); includes the 1/Nc of the inverse.
    Bin 0 packs (H[0], H[Nc]) real gains instead (special-cased in pass B).
    """
    Nc = nfft // 2
    N1 = Nc // N2
    H = np.fft.rfft(h, nfft)                      # H[0..Nc]
    k = np.arange(Nc)
    Hk = H[:Nc]
    Gk = np.conj(H[Nc - k])                       # conj(H[Nc-k]); k=0 -> conj(H[Nc])
    ang = -2 * np.pi * k / nfft
    s, c = np.sin(ang), np.cos(ang)
    alpha = (Hk * (1 + s) + Gk * (1 - s)) / 2 / Nc
    beta = 1j * c * (Hk - Gk) / 2 / Nc
    alpha[0] = H[0].real / Nc                     # special bin: (H0, HN) real
    beta[0] = H[Nc].real / Nc
    k1 = np.arange(N1)[:, None]
    k2 = np.arange(N2)[None, :]
    idx = k1 + N1 * k2
    return alpha[idx], beta[idx]


def pass_a(x, nfft):
    """Column FFTs (over n1) + four-step twiddle. x real [L] -> ws [N1][N2] complex."""
    Nc = nfft // 2
    N1 = Nc // N2
    R2 = N1 // 16
    xp = np.zeros(nfft)
    xp[: len(x)] = x
    z = (xp[0::2] + 1j * xp[1::2]).reshape(N1, N2)          # [n1][n2]
    if N1 % 16:
        # N1 = 8*R2 shapes (24, 40, 72), the short plans (4, 8) and 66 = 11 x 6: the staging differs but the result is the same
        # column DFT times the four-step twiddle, whatever the factorisation
        k1 = np.arange(N1)[:, None]
        return np.fft.fft(z, axis=0) * w(np.arange(N2)[None, :], k1, Nc)
    ws = np.zeros((N1, N2), complex)
    g = np.arange(R2)
    for c in range(0, N2, 1024):                              # model a few columns vectorised
        cols = slice(c, c + 1024)
        # thread (g, col) holds rows n1 = g + R2*j, j=0..15
        v = z[:, cols].reshape(16, R2, -1)                     # [j][g][col]
        a = np.fft.fft(v, axis=0)                              # FFT16 over j -> ka
        ka = np.arange(16)[:, None, None]
        a = a * w(g[None, :, None], ka, N1)                    # w_N1^(g*ka)
        b = np.fft.fft(a, axis=1)                              # FFT_R2 over g -> kb ; a is [ka][g]
        # k1 = ka + 16*kb
        out = np.transpose(b, (1, 0, 2)).reshape(N1, -1)       # [kb][ka] -> k1 = 16*kb + ka
        k1 = np.arange(N1)[:, None]
        n2 = np.arange(c, c + 1024)[None, :]
        ws[:, cols] = out * w(n2, k1, Nc)
    return ws


def fft4096_fwd(row):
    """Forward 4096 FFT exactly as the row kernel stages it. Returns regs[u'][kb2], k2 = ka+16kb1+256kb2, u'=16ka+kb1."""
    t = np.arange(256)
    v = row.reshape(16, 256).T                                  # v[t][j] = row[t + 256 j]
    a = np.fft.fft(v, axis=1)                                   # over j -> ka
    a = a * w(t[:, None], np.arange(16)[None, :], 4096)         # w4096^(t*ka)
    # exchange 1: thread (ka,t2) gets j2=0..15, t = 16*j2+t2
    a = a.reshape(16, 16, 16)                                   # [j2][t2][ka]
    b = np.transpose(a, (2, 1, 0))                              # [ka][t2][j2]
    b = np.fft.fft(b, axis=2)                                   # over j2 -> kb1
    t2 = np.arange(16)[None, :, None]
    kb1 = np.arange(16)[None, None, :]
    b = b * w(t2, kb1, 256)                                     # w256^(t2*kb1)
    # exchange 2: thread (ka,kb1) gets t2=0..15
    cc = np.transpose(b, (0, 2, 1))                             # [ka][kb1][t2]
    cc = np.fft.fft(cc, axis=2)                                 # over t2 -> kb2
    return cc.reshape(256, 16)                                  # [u'=16ka+kb1][kb2]


def regs_to_natural(regs):
    out = np.zeros(4096, complex)
    r = regs.reshape(16, 16, 16)                                # [ka][kb1][kb2]
    ka, kb1, kb2 = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
    out[(ka + 16 * kb1 + 256 * kb2).ravel()] = r.ravel()
    return out


def natural_to_regs(nat):
    ka, kb1, kb2 = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
    return nat[(ka + 16 * kb1 + 256 * kb2)].reshape(256, 16)


def ifft4096_inv(regs):
    """Unnormalised inverse 4096 FFT from the (ka,kb1)[kb2] register layout back to row order."""
    cc = regs.reshape(16, 16, 16)                               # [ka][kb1][kb2]
    cc = np.fft.ifft(cc, axis=2) * 16                           # over kb2 -> t2
    kb1 = np.arange(16)[None, :, None]
    t2 = np.arange(16)[None, None, :]
    cc = cc * np.conj(w(t2, kb1, 256))
    b = np.transpose(cc, (0, 2, 1))                             # [ka][t2][kb1]
    b = np.fft.ifft(b, axis=2) * 16                             # over kb1 -> j2
    ka = np.arange(16)[:, None, None]
    t2 = np.arange(16)[None, :, None]
    j2 = np.arange(16)[None, None, :]
    b = b * np.conj(w(16 * j2 + t2, ka, 4096))
    a = np.transpose(b, (2, 1, 0))                              # [j2][t2][ka]
    a = np.fft.ifft(a, axis=2) * 16                             # over ka -> j
    a = a.reshape(256, 16)                                      # [t][j]
    return a.T.reshape(4096)                                    # row[t + 256 j]


def pass_b(ws, alpha, beta):
    """Row FFT -> middle (alpha/beta with partner bins) -> row IFFT, in place."""
    N1 = ws.shape[0]
    Nc = N1 * N2
    Z = np.stack([regs_to_natural(fft4096_fwd(ws[k1])) for k1 in range(N1)])   # [k1][k2]
    k1 = np.arange(N1)[:, None]
    k2 = np.arange(N2)[None, :]
    kk = k1 + N1 * k2
    kp = (Nc - kk) % Nc
    Zp = np.conj(Z[kp % N1, kp // N1])
    W = alpha * Z + beta * Zp
    z0 = Z[0, 0]
    X0, XN = z0.real + z0.imag, z0.real - z0.imag
    a0, b0 = alpha[0, 0].real, beta[0, 0].real
    W[0, 0] = ((X0 * a0 + XN * b0) + 1j * (X0 * a0 - XN * b0)) / 2
    return np.stack([ifft4096_inv(natural_to_regs(W[r])) for r in range(N1)])


def pass_c(ws, nfft, start, length):
    """Inverse four-step twiddle, column IFFT (over k1 -> n1), unpack + crop."""
    Nc = nfft // 2
    N1 = Nc // N2
    R2 = N1 // 16
    k1 = np.arange(N1)[:, None]
    n2 = np.arange(N2)[None, :]
    v = ws * np.conj(w(n2, k1, Nc))
    if N1 % 16:
        # short (4, 8 rows) and 8 x R2 shapes: whatever the staging, the result is the column IDFT
        y = (np.fft.ifft(v, axis=0) * N1).reshape(-1)
        full = np.empty(nfft)
        full[0::2] = y.real
        full[1::2] = y.imag
        return full[start:start + length]
    # thread (g, col) holds rows k1 = g + R2*j
    v = v.reshape(16, R2, N2)                                   # [j][g][col]
    a = np.fft.ifft(v, axis=0) * 16                             # over j -> oa
    g = np.arange(R2)[None, :, None]
    oa = np.arange(16)[:, None, None]
    a = a * np.conj(w(g, oa, N1))
    b = np.fft.ifft(a, axis=1) * R2                             # over g -> ob ; n1 = oa + 16*ob
    y = np.transpose(b, (1, 0, 2)).reshape(N1, N2)              # [ob][oa] -> n1 = 16*ob + oa
    y = y.reshape(-1)
    full = np.empty(nfft)
    full[0::2] = y.real
    full[1::2] = y.imag
    return full[start:start + length]


# N1 = F*R2 rows of 4096 complex points (F = rows per thread in the column passes) -> nfft = 8192*N1
SUPPORTED_N1 = (4, 8, 16, 24, 32, 40, 48, 64, 66, 72, 80, 96, 128, 144, 160, 192, 256)


def pick_nfft(L, M, mode="same"):
    """Same rule as plan_geometry() in csrc/impulse_hip.hip: 'same' only needs L + M/2 points
    because wrap-around may fall into the part of the linear convolution the window discards."""
    need = max(L + M // 2, M) if mode == "same" else L + M - 1
    for n1 in SUPPORTED_N1:
        if 8192 * n1 >= need:
            return 8192 * n1
    raise ValueError("too long")


def alpha_beta_register_order(alpha, beta):
    """[k1][k2] complex planes -> the float32 [k1][kb2*256 + u][4] layout of the row kernel."""
    k2 = np.arange(N2)
    pos = (k2 >> 8) * 256 + 16 * (k2 & 15) + ((k2 >> 4) & 15)
    out = np.zeros(alpha.shape + (4,), dtype=np.float64)
    out[:, pos, 0], out[:, pos, 1] = alpha.real, alpha.imag
    out[:, pos, 2], out[:, pos, 3] = beta.real, beta.imag
    out[0, 0, 1] = 0.0
    out[0, 0, 3] = 0.0
    return out


def convolve_same_model(x, h, nfft=None):
    L, M = len(x), len(h)
    if nfft is None:
        nfft = pick_nfft(L, M, "same")
    alpha, beta = plan_alpha_beta(h, nfft)
    ws = pass_a(x, nfft)
    ws = pass_b(ws, alpha, beta)
    return pass_c(ws, nfft, (M - 1) // 2, L)
