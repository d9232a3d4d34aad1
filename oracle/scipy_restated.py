"""NumPy restatements of the SciPy 1.15.3 routines the reference calls on the hot path."""
import numpy as np


def next_fast_len_real(n):
    """Smallest 5-smooth integer >= n (scipy.fft.next_fast_len(n, real=True) and
    scipy.fftpack.next_fast_len: pocketfft good_size_real, factors 2, 3, 5)."""
    n = int(n)
    if n <= 6:
        return max(n, 0) if n >= 1 else 0
    best = None
    p5 = 1
    while p5 < 2 * n:
        p35 = p5
        while p35 < 2 * n:
            q = p35
            while q < n:
                q *= 2
            if best is None or q < best:
                best = q
            p35 *= 3
        p5 *= 5
    return best


def fft_convolve(x, h, mode="full"):
    """scipy.signal.convolve(x, h, mode, method='auto') for sizes where 'auto' selects the FFT
    method (scipy/signal/_signaltools.py fftconvolve -> _freq_domain_conv -> _centered):
    rfft at nfft = next_fast_len(L+M-1, real=True), product, irfft, then the mode window."""
    x = np.asarray(x, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    L, M = len(x), len(h)
    if L == 0 or M == 0:
        return np.zeros(0)
    full = L + M - 1
    nfft = next_fast_len_real(full)
    y = np.fft.irfft(np.fft.rfft(x, nfft) * np.fft.rfft(h, nfft), nfft)[:full]
    if mode == "full":
        return y
    if mode == "same":
        # _centered(ret, s1): start = (full - L) // 2
        start = (full - L) // 2
        return y[start:start + L]
    raise ValueError(mode)


def local_maxima_1d(x):
    """scipy.signal._peak_finding_utils._local_maxima_1d: midpoints of strict local maxima,
    plateaus included (midpoint = (left+right)//2); first and last samples never qualify."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    mids = []
    i = 1
    i_max = n - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            ahead = i + 1
            while ahead < i_max and x[ahead] == x[i]:
                ahead += 1
            if x[ahead] < x[i]:
                left, right = i, ahead - 1
                mids.append((left + right) // 2)
                i = ahead
        i += 1
    return np.asarray(mids, dtype=np.intp)


def find_peaks_height(x, height):
    """scipy.signal.find_peaks(x, height=height)[0]: local maxima with height <= x[peak]."""
    pk = local_maxima_1d(x)
    if len(pk) == 0:
        return pk
    return pk[height <= np.asarray(x)[pk]]


def _cosine_window(M, a0, a1):
    if M <= 0:
        return np.zeros(0)
    if M == 1:
        return np.ones(1)
    n = np.arange(M)
    return a0 - a1 * np.cos(2.0 * np.pi * n / (M - 1))


def hann(M):
    """scipy.signal.windows.hann(M, sym=True)."""
    return _cosine_window(int(M), 0.5, 0.5)


def hamming(M):
    """scipy.signal.windows.hamming(M, sym=True)."""
    return _cosine_window(int(M), 0.54, 0.46)


def linregress(x, y):
    """slope, intercept of scipy.stats.linregress (means + biased covariances)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.size == 0 or y.size == 0:
        raise ValueError("Inputs must not be empty.")
    if len(x) > 1 and np.amax(x) == np.amin(x):
        raise ValueError("Cannot calculate a linear regression if all x values are identical")
    xm, ym = np.mean(x), np.mean(y)
    ssxm, ssxym, _, _ = np.cov(x, y, bias=1).flat
    slope = ssxym / ssxm
    return slope, ym - slope * xm


def firwin2_hamming(numtaps, freq, gain, fs):
    """scipy.signal.firwin2(numtaps, freq, gain, fs=fs) with the default Hamming window and
    default nfreqs, symmetric (type I/II) case only."""
    freq = np.asarray(freq, dtype=np.float64)
    gain = np.asarray(gain, dtype=np.float64)
    nyq = 0.5 * fs
    if freq[0] != 0 or freq[-1] != nyq:
        raise ValueError("freq must start with 0 and end with fs/2.")
    if numtaps % 2 == 0 and gain[-1] != 0.0:
        raise ValueError("A Type II filter must have zero gain at the Nyquist frequency.")
    nfreqs = 1 + 2 ** int(np.ceil(np.log2(numtaps)))
    x = np.linspace(0.0, nyq, nfreqs)
    fx = np.interp(x, freq, gain)
    shift = np.exp(-(numtaps - 1) / 2.0 * 1.0j * np.pi * x / nyq)
    out_full = np.fft.irfft(fx * shift)
    return out_full[:numtaps] * hamming(numtaps)


def minimum_phase_homomorphic(h, n_fft):
    """scipy.signal.minimum_phase(h, method='homomorphic', n_fft=n_fft, half=True)."""
    h = np.asarray(h, dtype=np.float64)
    n_fft = int(n_fft)
    if n_fft < len(h):
        raise ValueError("n_fft must be at least len(h)")
    mag = np.abs(np.fft.fft(h, n_fft))
    mag += 1e-7 * mag[mag > 0].min()
    cep = np.fft.ifft(0.5 * np.log(mag)).real
    win = np.zeros(n_fft)
    win[0] = 1.0
    stop = n_fft // 2
    win[1:stop] = 2.0
    if n_fft % 2:
        win[stop] = 1.0
    h_min = np.fft.ifft(np.exp(np.fft.fft(cep * win))).real
    return h_min[: len(h) // 2 + len(h) % 2]


def spline1_eval(xk, yk, xq):
    """InterpolatedUnivariateSpline(xk, yk, k=1)(xq): piecewise linear through the knots with
    LINEAR EXTRAPOLATION from the end segments (FITPACK ext=0), not clamping."""
    xk = np.asarray(xk, dtype=np.float64)
    yk = np.asarray(yk, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    idx = np.clip(np.searchsorted(xk, xq, side="right") - 1, 0, len(xk) - 2)
    x0, x1 = xk[idx], xk[idx + 1]
    t = (xq - x0) / (x1 - x0)
    return yk[idx] + t * (yk[idx + 1] - yk[idx])


def _bspline_basis_all(t, k, x):
    """All B-spline basis functions of degree k on knot vector t evaluated at x (Cox-de Boor).
    Returns [len(x), len(t) - k - 1]."""
    t = np.asarray(t, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    n = len(t) - k - 1
    # degree 0: indicator of [t_i, t_{i+1}), with the last non-empty interval closed on the right
    B = np.zeros((len(x), len(t) - 1))
    last = np.max(np.nonzero(t[1:] > t[:-1])[0])
    for i in range(len(t) - 1):
        if t[i + 1] > t[i]:
            if i == last:
                B[:, i] = (x >= t[i]) & (x <= t[i + 1])
            else:
                B[:, i] = (x >= t[i]) & (x < t[i + 1])
    for d in range(1, k + 1):
        Bn = np.zeros((len(x), len(t) - 1 - d))
        for i in range(len(t) - 1 - d):
            left = t[i + d] - t[i]
            right = t[i + d + 1] - t[i + 1]
            term = 0.0
            if left > 0:
                term = term + (x - t[i]) / left * B[:, i]
            if right > 0:
                term = term + (t[i + d + 1] - x) / right * B[:, i + 1]
            Bn[:, i] = term
        B = Bn
    return B[:, :n]


def spline2_interp(xk, yk, xq):
    """InterpolatedUnivariateSpline(xk, yk, k=2)(xq): FITPACK's interpolating quadratic spline
    (fpcurf, s = 0): boundary knots of multiplicity 3 and interior knots at the midpoints
    (x[i+1] + x[i+2]) / 2, i = 0..m-4; coefficients from the collocation system; points outside
    [x0, x_{m-1}] are extrapolated with the end polynomial pieces."""
    xk = np.asarray(xk, dtype=np.float64)
    yk = np.asarray(yk, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    m = len(xk)
    interior = (xk[1:m - 2] + xk[2:m - 1]) / 2.0
    t = np.concatenate([[xk[0]] * 3, interior, [xk[-1]] * 3])
    coef = np.linalg.solve(_bspline_basis_all(t, 2, xk), yk)
    return _bspline_basis_all(t, 2, np.clip(xq, xk[0], xk[-1])) @ coef
