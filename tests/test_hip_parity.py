"""GPU parity tests: the HIP path (through the ctypes C-ABI and the product classes) against the
CPU oracle on the same seeded inputs, against the reference-generated golden fixtures, and - at
BASELINE.json's full sizes - through size-independent properties.

Tolerances (north_star): magnitude spectra within 1e-6 of the spectrum peak in fp32
(max|dA| / max|A| <= 1e-6), IR peak indices sample-exact.  Time-domain samples are held to
1e-6 of the IR peak as well.

What "magnitude spectrum" means here: the reference only ever calls magnitude_response on CROPPED
impulse responses (head-cropped at the peak, tail-cropped at the Lundeby knee: 20-66 k samples,
core/hrir.py:457-521, core/impulse_response.py:157-188).  SPEC_TOL is asserted on that window:
peak - 1 ms, 0.68 s long (32 640 samples at 48 kHz, 65 280 at 96 kHz - the tail length the
reference's demo measurements are cropped to).  On a whole un-cropped 391 270-sample column every fp32 FFT sits at
1-2.5e-6 for sweep recordings (measured: pocketfft in single precision 0.9-2.5e-6 on bench.py's
inputs, DESIGN.md "Accuracy"): the white rounding noise of the transform gains sqrt(L) in the
spectrum while the IR itself is a compact pulse.  That case is held to FULL_COLUMN_TOL.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPEC_TOL = 1e-6            # max |dA| / max |A| on |rfft(cropped ir)|  (and on whole noise-like outputs)
# The same metric on an UN-CROPPED sweep-recording column.  north_star's 1e-6 is not reachable there in fp32: measured floor
# of every fp32 transform (profiles/r04_column_error.txt): this path 1.1-2.3e-6 in pair mode (the plan the
# classes take for ear pairs) at C2, C3 and C5, pocketfft in single precision 2.1-2.5e-6 on the same inputs.  A maximum over
# 2-5 x 10^5 bins of white rounding noise (time-domain rms 3-5e-10 of the peak), it moves by up to 2x between channels of one
# plan.  Nothing the reference computes sees it: magnitude_response only ever runs on cropped responses (SPEC_TOL).  Asserted
# at all three BASELINE shapes (tests/test_pair_mode.py::test_whole_column_at_the_fp32_floor) and gated by bench.py.
FULL_COLUMN_TOL = 3e-6
TIME_TOL = 1e-6            # max |dy| / max |y|


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / np.max(np.abs(b)))


def spec_rel(y, ref):
    A = np.abs(np.fft.rfft(np.asarray(y, dtype=np.float64)))
    R = np.abs(np.fft.rfft(ref))
    return float(np.max(np.abs(A - R)) / np.max(R))


def spec_rel_cropped(y, ref, fs=48000, n=None):
    """Spectrum error on the IR as the reference would crop it before any magnitude_response."""
    from oracle.impulse_response import peak_index
    n = int(0.68 * fs) if n is None else n
    start = max(peak_index(ref) - int(fs / 1000), 0)
    return spec_rel(np.asarray(y)[start:start + n], np.asarray(ref)[start:start + n])


# ------------------------------------------------------------------------------------------------
# K1/K5: convolution plans against the oracle
# ------------------------------------------------------------------------------------------------
# sizes chosen to hit every column shape N1 = F x R2 (nfft = 8192 N1) in at least one mode:
# 'same': N1 = 4, 4, 4, 4, 8, 16, 32, 40, 66, 96, 144, 24, 48, 64, 80, 128 ; 'full': 4, 4, 4, 4, 8, 16, 32, 32, 48, 96, 128, 192, 32, 48, 80, 96, 160
@pytest.mark.parametrize("L,M", [(1, 1), (17, 5), (1000, 999), (20000, 9600), (32640, 9600), (70001, 61000), (150000, 100001),
                                 (243635, 147635), (391270, 295270), (500000, 400000), (827965, 635965),
                                 (150000, 80000), (300000, 150000), (420000, 200000), (500000, 300000), (800000, 400000)])
@pytest.mark.parametrize("mode", ["same", "full"])
def test_conv_matches_oracle(gpu_ctx, L, M, mode):
    """the three-launch transform over the whole input (filters short enough for the fused overlap-save kernel are covered
    on both paths: tests/test_fused_fir.py)"""
    from impulse_hip import ConvPlan
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(L * 31 + M)
    x = rng.standard_normal((3, L)).astype(np.float32)
    x[2] = 0.0                                                   # an all-zero (silent) channel
    h = rng.standard_normal(M) * np.exp(-np.arange(M) / max(M / 5.0, 1.0))
    plan = ConvPlan(gpu_ctx, h, L, mode, fused=False)
    from impulse_hip._native import plan_geometry
    assert plan.nfft == plan_geometry(M, L, mode)[0] and not plan.fused
    y = plan.execute(x)
    plan.close()
    assert y.shape == (3, L if mode == "same" else L + M - 1)
    assert not np.any(y[2])                                      # zero in -> exactly zero out
    for b in range(2):
        ref = fft_convolve(x[b].astype(np.float64), h, mode)
        assert rel(y[b], ref) <= TIME_TOL
        if len(ref) > 8:
            assert spec_rel(y[b], ref) <= SPEC_TOL


def test_conv_full_size_c5_properties(gpu_ctx):
    """C4/C5 shape (L = M = 2^20, circular length 3*2^19): identity filter, linearity, delay equivariance."""
    from impulse_hip import ConvPlan
    L = M = 1 << 20
    rng = np.random.default_rng(0xC5)
    x = rng.standard_normal((4, L)).astype(np.float32)
    x[0, -1000:] = 0.0                                           # room to delay channel 0 without truncation
    x[2] = (2.0 * x[0] - 0.5 * x[1])
    delta = np.zeros(M)
    delta[(M - 1) // 2] = 1.0                                    # 'same' window of a centred delta = identity
    plan = ConvPlan(gpu_ctx, delta, L, "same")
    y = plan.execute(x)
    plan.close()
    for b in range(4):
        assert rel(y[b], x[b].astype(np.float64)) <= TIME_TOL
    h = rng.standard_normal(M) * np.exp(-np.arange(M) / 50000.0)
    plan = ConvPlan(gpu_ctx, h, L, "same")
    y = plan.execute(x)
    xd = np.zeros_like(x[:1])
    xd[0, 1000:] = x[0, :-1000]                                  # delayed copy of channel 0
    yd = plan.execute(xd)
    y2 = plan.execute(x)                                         # determinism: bit-identical reruns
    plan.close()
    assert np.array_equal(y, y2)
    lin = 2.0 * y[0].astype(np.float64) - 0.5 * y[1].astype(np.float64)
    scale = np.max(np.abs(lin))
    assert np.max(np.abs(y[2] - lin)) / scale <= 3e-6            # three fp32 results combined
    assert np.max(np.abs(yd[0, 1000:] - y[0, :-1000].astype(np.float64))) / scale <= 2e-6
    assert np.max(np.abs(yd[0, :1000].astype(np.float64) - 0.0)) / scale <= 1.0   # head is just earlier filter taps


def test_conv_interleaved_and_per_channel_filters(gpu_ctx):
    from impulse_hip import ConvPlan
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(77)
    L, M = 100001, 9600                                          # odd L, even taps -> odd crop offset
    frames = rng.standard_normal((L, 4)).astype(np.float32)
    firs = rng.standard_normal((4, M)) * 0.05
    p = ConvPlan(gpu_ctx, firs[0], L, "same")
    yi = p.execute_interleaved(frames)
    yp = p.execute(np.ascontiguousarray(frames.T))
    p.close()
    assert np.array_equal(yi, yp)                                # wire-order loader == planar loader
    for c in range(4):
        assert rel(yi[c], fft_convolve(frames[:, c].astype(np.float64), firs[0], "same")) <= TIME_TOL
    p = ConvPlan(gpu_ctx, firs, L, "full")
    y = p.execute(np.ascontiguousarray(frames.T))
    p.close()
    for c in range(4):
        assert rel(y[c], fft_convolve(frames[:, c].astype(np.float64), firs[c], "full")) <= TIME_TOL


def test_conv_many_channels_cross_workspace_groups(gpu_ctx):
    """More channels than one launch group: chunking must not mix channels."""
    from impulse_hip import ConvPlan
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(5)
    L, M, B = 50000, 30000, 11
    x = rng.standard_normal((B, L)).astype(np.float32) * (1 + np.arange(B))[:, None]
    h = rng.standard_normal(M)
    p = ConvPlan(gpu_ctx, h, L, "same", ws_channels=4)
    y = p.execute(x)
    p.close()
    for b in (0, 3, 4, 7, 10):
        assert rel(y[b], fft_convolve(x[b].astype(np.float64), h, "same")) <= TIME_TOL


def test_conv_error_paths(gpu_ctx):
    from impulse_hip import ConvPlan, NativeError
    with pytest.raises(NativeError):
        ConvPlan(gpu_ctx, np.ones(4), 1 << 29, "same")           # beyond one channel's 32-bit buffer range: loud
    p = ConvPlan(gpu_ctx, np.ones(4), 100, "same")
    with pytest.raises(ValueError):
        p.execute(np.zeros((2, 99), np.float32))
    p.close()
    with pytest.raises(NativeError):
        ConvPlan(gpu_ctx, np.ones((2, 4)), 100, "same").execute(np.zeros((3, 100), np.float32))
    # the loaders reach to the END of the transform (zero padding = range check), so the guard is on
    # nfft x elem_stride x 4 B, not L x elem_stride x 4 B: 600 interleaved tracks of 2^20 samples would wrap
    p = ConvPlan(gpu_ctx, np.ones(1 << 19), 1 << 20, "same")
    assert p.nfft == 5 << 18
    d = gpu_ctx.malloc(1 << 20)
    try:
        with pytest.raises(NativeError, match="4 GiB buffer range"):
            p.execute_device(d, 1, 1, d, 1 << 20, elem_stride_in=900)      # L x 900 x 4 < 4 GiB <= nfft x 900 x 4
        with pytest.raises(NativeError, match="4 GiB buffer range"):
            p.execute_device_pcm(d, 32, 1, 1, 900, d, 1 << 20)
    finally:
        gpu_ctx.free(d)
        p.close()
    # an overlap-add plan broadcasts every filter partition, not just the first plane
    big = ConvPlan(gpu_ctx, np.ones((1 << 21) + 5), 1 << 12, "full")
    assert big.spectrum_buffer()[1] == 3 * 256 * 4096 * 16
    big.close()


# ------------------------------------------------------------------------------------------------
# estimator: the reference's own round-trip tests (tests/test_estimator_roundtrip.py) on the GPU
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def est1(gpu_ctx):
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    return ImpulseResponseEstimator(min_duration=1.0, fs=48000)


@pytest.fixture(scope="module")
def oracle_est1():
    from oracle.estimator import Estimator
    return Estimator(min_duration=1.0, fs=48000)


def test_estimator_setup_matches_oracle(est1, oracle_est1):
    assert len(est1) == len(oracle_est1) == 147635
    np.testing.assert_allclose(est1.test_signal, oracle_est1.test_signal, atol=1e-15)
    np.testing.assert_allclose(est1.inverse_filter, oracle_est1.inverse_filter,
                               atol=1e-13 * np.max(np.abs(oracle_est1.inverse_filter)))


@pytest.mark.parametrize("delay", [0, 1, 127, 1000])
def test_estimate_recovers_delayed_unit_impulse_peak(est1, golden, delay):
    from oracle.scipy_restated import fft_convolve
    g = golden("estimate")
    sysir = np.zeros(delay + 1)
    sysir[delay] = 1.0
    rec = fft_convolve(est1.test_signal, sysir, "full").astype(np.float32)
    y = est1.estimate(rec)
    assert len(y) == len(rec) == int(g[f"delay{delay}_len"])
    assert int(np.argmax(np.abs(y))) == int(g[f"delay{delay}_argmax"]) == len(est1) // 2 + delay
    assert y[np.argmax(np.abs(y))] == pytest.approx(float(g[f"delay{delay}_peakval"]), abs=1e-6)


def test_estimate_unit_impulse_has_low_noise_floor(est1):
    y = est1.estimate(est1.test_signal)
    peak = int(np.argmax(np.abs(y)))
    assert peak == len(est1) // 2
    outside = np.concatenate((y[: peak - 50], y[peak + 50:]))
    assert np.abs(y[peak]) == pytest.approx(1.0, abs=0.05)
    assert 20 * np.log10(np.abs(y[peak]) / np.max(np.abs(outside))) > 40.0


def test_estimate_golden_cases(est1, golden):
    from impulse_hip.impulse_response import ImpulseResponse
    from oracle.scipy_restated import fft_convolve
    g = golden("estimate")
    bp = int(g["baseline_peak"])
    rec = fft_convolve(est1.test_signal, g["decay_ir"], "full").astype(np.float32)
    y = est1.estimate(rec)
    pkv = np.max(np.abs(g["decay_win"]))
    assert np.max(np.abs(y[bp - 64: bp + 4096] - g["decay_win"])) / pkv <= TIME_TOL
    assert np.max(np.abs(y[::61] - g["decay_dec"])) / pkv <= TIME_TOL
    A = np.abs(np.fft.rfft(y))
    assert np.max(np.abs(A[g["decay_bins"]] - g["decay_amp"])) / float(g["decay_amax"]) <= FULL_COLUMN_TOL
    assert ImpulseResponse(y, 48000).peak_index() == int(g["decay_peak_index"])
    # recovered waveform correlates with the system IR (reference test :75-85)
    got = y[bp: bp + 600]
    corr = np.dot(got, g["decay_ir"]) / (np.linalg.norm(got) * np.linalg.norm(g["decay_ir"]))
    assert corr > 0.99 and int(np.argmax(np.abs(got))) == 31
    x = np.random.default_rng(0).standard_normal(int(g["noise_L"])).astype(np.float32)
    y = est1.estimate(x)
    pkv = np.max(np.abs(g["noise_dec"]))
    assert np.max(np.abs(y[100000:100000 + 4096] - g["noise_win"])) / pkv <= TIME_TOL
    assert np.max(np.abs(y[::61] - g["noise_dec"])) / pkv <= TIME_TOL
    A = np.abs(np.fft.rfft(y))
    assert np.max(np.abs(A[g["noise_bins"]] - g["noise_amp"])) / float(g["noise_amax"]) <= SPEC_TOL
    assert ImpulseResponse(y, 48000).peak_index() == int(g["noise_peak_index"])


def test_estimate_batch_c2_full_size_analytic_peaks(gpu_ctx):
    """BASELINE config 2: 8 speakers x 2 ears, 6.15 s sweep @48 kHz, column L = N + 2 fs.
    Analytic truth, no oracle needed: a sweep delayed by d deconvolves to a peak at N//2 + d."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    e = ImpulseResponseEstimator(min_duration=5.0, fs=48000)
    N, L = len(e), len(e) + 96000
    assert (N, L) == (295270, 391270)
    sweep = e.test_signal.astype(np.float32)
    rec = np.zeros((16, L), dtype=np.float32)
    delays = [64 + 37 * c for c in range(16)]
    for c, d in enumerate(delays):
        rec[c, d:d + N] += (0.5 + 0.03 * c) * sweep
        rec[c, d + 700:d + 700 + N] += 0.02 * sweep[: min(N, L - d - 700)]     # a weak later reflection
    y = e.estimate_batch(rec, dtype=np.float32)
    assert y.shape == (16, L)
    for c, d in enumerate(delays):
        assert int(np.argmax(np.abs(y[c]))) == N // 2 + d
    # batched == one-by-one, bit for bit; frames path == planar path
    assert np.array_equal(y[5], e.estimate(rec[5]).astype(np.float32))
    yf = e.estimate_frames(np.ascontiguousarray(rec[:4].T), dtype=np.float32)
    assert np.array_equal(yf, y[:4])


def test_estimate_batch_c3_full_size_analytic_peaks(gpu_ctx):
    """BASELINE config 3: 13 speakers x 2 ears @96 kHz (N = 635 965, column 827 965, circular length
    1 310 720 = radix-10 columns): analytic peaks, exact."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    e = ImpulseResponseEstimator(min_duration=5.0, fs=96000)
    N, L = len(e), len(e) + 2 * 96000
    assert (N, L) == (635965, 827965)
    sweep = e.test_signal.astype(np.float32)
    rec = np.zeros((26, L), dtype=np.float32)
    delays = [90 + 53 * c for c in range(26)]
    for c, d in enumerate(delays):
        rec[c, d:d + N] += (0.4 + 0.02 * c) * sweep
    y = e.estimate_batch(rec, dtype=np.float32)
    assert y.shape == (26, L)
    for c, d in enumerate(delays):
        assert int(np.argmax(np.abs(y[c]))) == N // 2 + d
    assert np.array_equal(y[17], e.estimate(rec[17]).astype(np.float32))


# ------------------------------------------------------------------------------------------------
# K3 peak index, K4/K8 windows, decay, container ops - against goldens from the reference
# ------------------------------------------------------------------------------------------------
def test_peak_index_golden_cases(gpu_ctx, golden):
    from impulse_hip.impulse_response import ImpulseResponse
    g = golden("peak_index")
    names = sorted(k[:-2] for k in g.files if k.endswith("_x"))
    rows, want = [], []
    for nm in names:
        start, end = (int(v) for v in g[nm + "_kw"])
        got = ImpulseResponse(g[nm + "_x"], 48000).peak_index(start=start, end=None if end < 0 else end)
        assert got == int(g[nm + "_idx"]), nm
        if start == 0 and end < 0 and len(g[nm + "_x"]):
            rows.append(g[nm + "_x"])
            want.append(int(g[nm + "_idx"]))
    idx, _ = gpu_ctx.peak_index(rows)                            # ragged batch in one call
    assert list(idx) == want


def test_peak_index_matches_oracle_on_long_rows(gpu_ctx):
    from oracle.impulse_response import peak_index
    rng = np.random.default_rng(9)
    rows = []
    for n in (391270, 33750, 4097):
        x = (rng.standard_normal(n) * 0.01).astype(np.float32)
        p = int(n * 0.44)
        x[p:p + 400] += (np.sin(np.arange(400) / 3.0) * np.exp(-np.arange(400) / 60.0)).astype(np.float32)
        rows.append(x)
    idx, mx = gpu_ctx.peak_index(rows)
    for r, i, m in zip(rows, idx, mx):
        assert int(i) == peak_index(r.astype(np.float64))
        assert m == np.max(np.abs(r))


def test_device_blocks_are_kept_and_reused(gpu_ctx):
    """imp_free keeps a block for the next request of about its size (no hipFree per measurement); a pointer that did not
    come from imp_malloc is refused."""
    from impulse_hip._native import NativeError
    if os.environ.get("IMPULSE_HIP_POOL_MB") == "0":
        pytest.skip("the block pool is switched off in this environment")
    a = gpu_ctx.malloc(3 << 20)
    keep = gpu_ctx.malloc(3 << 20)
    assert keep != a
    gpu_ctx.free(a)
    b = gpu_ctx.malloc((3 << 20) - 4096)                       # a little smaller: the kept block serves it
    assert b == a
    gpu_ctx.free(b)
    c = gpu_ctx.malloc(1 << 20)                                # much smaller: a block of its own
    assert c not in (a, keep)
    x = np.arange(1 << 18, dtype=np.float32)
    gpu_ctx.h2d(c, x)
    y = np.empty_like(x)
    gpu_ctx.d2h(y, c)
    assert np.array_equal(x, y)
    for p in (c, keep):
        gpu_ctx.free(p)
    with pytest.raises(NativeError, match="did not come from imp_malloc"):
        gpu_ctx.free(0x7f0000001000)


def test_peak_index_around_the_chunk_boundaries(gpu_ctx):
    """K3 reads only the 8192-sample chunks that can hold the first peak: peaks on a chunk edge, plateaus that run across
    an edge, a rising slope whose crest lies in a later chunk, rows without any peak (argmax fallback), negative peaks and
    rows that end right after an edge - all against the oracle."""
    from oracle.impulse_response import peak_index
    rng = np.random.default_rng(77)
    rows = []

    def base(n, level=1e-3):
        return (rng.standard_normal(n) * level).astype(np.float32)

    for at in (8191, 8192, 8193, 16383, 16384):                  # a lone spike on / next to an edge
        x = base(30000)
        x[at] = 1.0 if at % 2 else -1.0
        rows.append(x)
    x = base(30000)                                               # plateau across the edge: index = middle of the run
    x[8188:8197] = 0.7
    x[20000] = 1.0
    rows.append(x)
    x = base(30000)                                               # slope starts in chunk 0, crest in chunk 1
    x[8100:8300] += np.linspace(0.0, 1.0, 200, dtype=np.float32)
    x[8300:8400] += np.linspace(1.0, 0.0, 100, dtype=np.float32)
    rows.append(x)
    rows.append(np.linspace(0.0, 1.0, 20000, dtype=np.float32))   # no peak at all: first index of the maximum
    rows.append(-np.linspace(0.0, 1.0, 8193, dtype=np.float32))
    x = np.zeros(16385, dtype=np.float32)                         # the maximum is the very last sample (never a peak)
    x[-1] = 1.0
    x[5000] = 0.5
    rows.append(x)
    x = base(8192 * 3)                                            # qualifying chunk without a peak start, then the peak
    x[8192:16384] = 0.5                                           # one long plateau = chunk 1, falls at 16384
    x[20000] = -1.0
    rows.append(x)
    rows.append(np.zeros(9000, dtype=np.float32))                 # silence: index 0
    rows.append(base(5))
    idx, mx = gpu_ctx.peak_index(rows)
    for k, (r, i, m) in enumerate(zip(rows, idx, mx)):
        assert int(i) == peak_index(r.astype(np.float64)), k
        assert m == np.max(np.abs(r)), k


def _decaying_sine(fs, duration_s, rt60, freq=1000.0, floor_db=-90.0, seed=0):
    r = np.random.default_rng(seed)
    n = int(duration_s * fs)
    t = np.arange(n) / fs
    env = 10 ** ((-60.0 / rt60) * t / 20.0)
    return np.cos(2 * np.pi * freq * t) * env + r.standard_normal(n) * 10 ** (floor_db / 20.0)


@pytest.mark.parametrize("rt60", [0.3, 0.6, 1.0, 1.5])
def test_decay_analysis_and_window_golden(gpu_ctx, golden, rt60):
    from impulse_hip import decay
    from impulse_hip.parallel_workers import process_decay_worker
    g = golden("decay")
    for seed in (0, 11, 22):
        k = f"rt{int(rt60 * 10)}_s{seed}"
        data = _decaying_sine(48000, 3.0, rt60, seed=seed).astype(np.float32).astype(np.float64)
        p = decay.decay_params(data, 48000)
        exp = g[k + "_params"]
        assert (int(p[0]), int(p[1]), int(p[3])) == (int(exp[0]), int(exp[1]), int(exp[3]))
        assert p[2] == pytest.approx(exp[2], abs=1e-9)
        for got, want in zip(decay.decay_times(data, 48000), g[k + "_times"]):
            assert (got is None) if np.isnan(want) else got == pytest.approx(want, rel=1e-9)
        for target in (0.2, 0.5):
            kk = f"{k}_t{int(target * 10)}"
            want = g[kk + "_adj"]
            if np.all(np.isinf(want)):
                with pytest.raises(TypeError):
                    decay.decay_adjustment_params(data, 48000, target)
                continue
            adj = decay.decay_adjustment_params(data, 48000, target)
            if np.all(np.isnan(want)):
                assert adj is None
                continue
            assert tuple(int(v) for v in adj[:3]) == tuple(int(v) for v in want[:3])
            _, _, out = process_decay_worker(("FL", "left", data, 48000, target))
            assert np.max(np.abs(out[::37] - g[kk + "_out_dec"])) <= 2e-7     # fp32 store of a <= 1.0 signal
            assert np.sum(out ** 2) == pytest.approx(float(g[kk + "_out_energy"]), rel=1e-6)
            direct = data.copy()
            decay.apply_decay_window(direct, adj)
            assert np.array_equal(direct, out)                                  # worker == direct call


def test_hrir_container_ops_golden(gpu_ctx, golden):
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    g = golden("hrir_ops")

    class Est:
        fs = 48000
        n_octaves = 10

        def __len__(self):
            return 48000 * 6

    def mk(chs):
        h = HRIR(Est())
        h.irs = {sp: {sd: ImpulseResponse(np.asarray(v, dtype=np.float64), 48000) for sd, v in pr.items()}
                 for sp, pr in chs.items()}
        return h

    for name in ("left_first", "right_first", "tie", "near_start"):
        h = mk({"FC": {"left": g[f"ch_{name}_in_l"], "right": g[f"ch_{name}_in_r"]}})
        h.crop_heads(head_ms=1)
        for sd, key in (("left", "l"), ("right", "r")):
            want = g[f"ch_{name}_out_{key}"]
            assert h.irs["FC"][sd].data.shape == want.shape
            assert np.max(np.abs(h.irs["FC"][sd].data - want)) <= 1e-7 * np.max(np.abs(want))
    h = mk({sp: {sd: g[f"ct_in_{sp}_{sd}"] for sd in ("left", "right")} for sp in ("FL", "FR")})
    assert h.crop_tails() == int(g["ct_tail_ind"])
    for sp in ("FL", "FR"):
        for sd in ("left", "right"):
            want = g[f"ct_out_{sp}_{sd}"]
            got = h.irs[sp][sd].data
            assert got.shape == want.shape and abs(got[-1]) < 1e-6
            assert np.max(np.abs(got - want)) <= 1e-7 * np.max(np.abs(want))
    for name, seed, scales, kw in (("peak", 1, (0.3, 0.1, 0.05, 0.2), dict(peak_target=-0.1)),
                                   ("avg", 2, (0.4, 0.25, 0.15, 0.3), dict(peak_target=None, avg_target=-12.0))):
        rr = np.random.default_rng(seed)
        a = [(rr.standard_normal(4096) * s).astype(np.float32) for s in scales]
        h = mk({"FL": {"left": a[0], "right": a[1]}, "FR": {"left": a[2], "right": a[3]}})
        gain = h.normalize(**kw)
        assert gain == pytest.approx(float(g[f"nm_{name}_gain_db"]), abs=1e-9)
        np.testing.assert_allclose(h.irs["FL"]["left"].data, g[f"nm_{name}_out_FL_left"], rtol=1e-12)
    with pytest.raises(ValueError):
        mk({"FL": {"left": [1.0, 0.0], "right": [1.0, 0.0]}}).normalize(peak_target=-0.1, avg_target=-12.0)


def test_real_demo_column_against_shipped_golden(gpu_ctx, golden):
    """The one real-data case that can travel: data/demo/room-FC-left.wav column (int32 PCM) ->
    estimate -> crop_head(1 ms) -> crop to 21 600 + fade-out, against the FC-left track of the
    room-responses.wav that the reference ships."""
    from impulse_hip.hrir import _hann
    from impulse_hip.impulse_response import ImpulseResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    g = golden("demo_fc")
    N = int(g["N"])
    e = ImpulseResponseEstimator(min_duration=(N - 1) / 48000, fs=48000)
    assert len(e) == N and e.n_octaves == float(g["P"])
    col = (g["column_i32"].astype(np.float64) / 2 ** 31).astype(np.float32)
    y = e.estimate(col)
    ir = ImpulseResponse(y, 48000)
    pk = ir.peak_index()
    assert pk == int(g["peak_index"])
    assert int(np.argmax(np.abs(y))) == int(g["argmax"])
    peak_val = np.max(np.abs(g["win"]))
    assert np.max(np.abs(y[pk - 64: pk + 8192] - g["win"])) / peak_val <= TIME_TOL
    A = np.abs(np.fft.rfft(y))
    assert np.max(np.abs(A[g["bins"]] - g["amp"])) / float(g["amax"]) <= FULL_COLUMN_TOL   # un-cropped column
    ir.crop_head()
    n_out = int(g["responses_len"])
    fo = 2 * int(48000 * (N / 48000 / float(g["P"])) * (1 / 24))
    w = _hann(fo)[fo // 2:]
    d = ir.data[:n_out].copy()
    d[n_out - len(w):] *= w
    want = g["responses_fc_left_i32"].astype(np.float64)
    lsb = np.abs(np.rint(d * 2 ** 31) - want)
    assert np.max(lsb) <= max(1.0, 1e-6 * np.max(np.abs(want)))        # <= 1e-6 of the track's peak, in LSB
    # magnitude spectrum of the cropped IR against the reference's own float64 output (the shipped
    # track is PCM_32-quantised: the float64 oracle itself is 2.5e-6 away from it on this metric)
    d_ref = g["cropped_head"].copy()
    d_ref[n_out - len(w):] *= w
    assert spec_rel(d, d_ref) <= SPEC_TOL


def test_ingest_recording_matches_oracle_split(gpu_ctx, tmp_path):
    """HRIR.open_recording on a synthetic FL,FR measurement (recipe of the reference's
    tests/test_pipeline_direct.py:175-204: delays 0/12/12/0, gains 1/.6/.6/.9), through the WAV
    writer/reader, against the oracle's column split + per-channel estimate."""
    from impulse_hip.audio_io import write_wav
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle import hrir as ohrir
    from oracle.estimator import Estimator
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    oe = Estimator(min_duration=1.0, fs=48000)
    N, fs = len(e), 48000
    tracks = np.zeros((2, 2 * fs + 2 * (N + 2 * fs)))
    # track 0 = left ear, track 1 = right ear; columns FL then FR
    for col, (dl, gl, dr, gr) in enumerate(((0, 1.0, 12, 0.6), (12, 0.6, 0, 0.9))):
        base = 2 * fs + col * (N + 2 * fs)
        tracks[0, base + dl: base + dl + N] += gl * e.test_signal
        tracks[1, base + dr: base + dr + N] += gr * e.test_signal
    tracks *= 0.5
    path = str(tmp_path / "FL,FR.wav")
    write_wav(path, fs, tracks, bit_depth=32)
    h = HRIR(e)
    h.open_recording(path, ["FL", "FR"])
    assert list(h.irs) == ["FL", "FR"] and all(set(p) == {"left", "right"} for p in h.irs.values())
    from impulse_hip.audio_io import pcm_quantise
    pcm = pcm_quantise(tracks, 32) / 2.0 ** 31                      # what PCM_32 stored
    jobs = ohrir.split_recording(pcm, ["FL", "FR"], N, fs)
    assert [(sp, sd) for sp, sd, _ in jobs] == [("FL", "left"), ("FL", "right"), ("FR", "left"), ("FR", "right")]
    for sp, sd, col in jobs:
        ref = oe.estimate(col)
        got = h.irs[sp][sd]
        assert len(got.data) == len(ref) == N + 2 * fs
        assert np.array_equal(got.recording, col)
        assert rel(got.data, ref) <= TIME_TOL
        assert spec_rel_cropped(got.data, ref) <= SPEC_TOL and spec_rel(got.data, ref) <= FULL_COLUMN_TOL
    peaks = {(sp, sd): h.irs[sp][sd].peak_index() for sp in h.irs for sd in ("left", "right")}
    assert peaks[("FL", "left")] == N // 2 and peaks[("FL", "right")] == N // 2 + 12
    assert peaks[("FR", "left")] == N // 2 + 12 and peaks[("FR", "right")] == N // 2
    h2 = HRIR(e)
    h2.fs = 44100
    with pytest.raises(ValueError):
        h2.open_recording(path, ["FL", "FR"])


def test_equalize_and_plot_worker_full_convolution(gpu_ctx):
    from impulse_hip.impulse_response import ImpulseResponse
    from impulse_hip.parallel_workers import process_plot_worker
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(21)
    ir = (rng.standard_normal(33600) * np.exp(-np.arange(33600) / 4000.0)).astype(np.float32)
    fir = rng.standard_normal(9600) * np.exp(-np.arange(9600) / 800.0)
    obj = ImpulseResponse(ir.astype(np.float64), 48000)
    obj.equalize(fir)
    ref = fft_convolve(ir.astype(np.float64), fir, "full")
    assert len(obj.data) == 33600 + 9600 - 1
    assert rel(obj.data, ref) <= TIME_TOL and spec_rel(obj.data, ref) <= SPEC_TOL
    sweep = np.sin(np.arange(147635) ** 1.3 / 500.0).astype(np.float32)
    _, _, rec = process_plot_worker(("FL", "left", ir.astype(np.float64), sweep, 48000))
    assert rel(rec, fft_convolve(sweep.astype(np.float64), ir.astype(np.float64), "full")) <= TIME_TOL


# ------------------------------------------------------------------------------------------------
# K6: minimum-phase FIR design (fp64 on the device) against FIRs produced by the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fs", [48000, 96000])
def test_minimum_phase_fir_golden(gpu_ctx, golden, fs):
    from impulse_hip import NativeError
    from impulse_hip.frequency_response import (FrequencyResponse, fir_design_gain,
                                                minimum_phase_impulse_responses)
    g = golden("minphase")
    freq = g[f"fs{fs}_freq"]
    names = ("flat", "wavy", "tilt")
    firs = minimum_phase_impulse_responses(freq, [g[f"fs{fs}_{nm}_eq"] for nm in names], fs, f_res=5, normalize=False)
    for fir, nm in zip(firs, names):
        want = g[f"fs{fs}_{nm}_fir"]
        assert fir.shape == want.shape == ((9600,) if fs == 48000 else (19200,))
        # 5e-8 of the FIR peak = the reference's own reproducibility floor for this ill-conditioned
        # design (see tests/test_oracle_golden.py::test_minimum_phase_fir)
        assert np.max(np.abs(fir - want)) <= 5e-8 * np.max(np.abs(want)), nm
        assert np.all(np.isfinite(fir))
        # minimum phase: energy is front-loaded (reference tests/test_frequency_response_core.py:101-112)
        assert np.sum(fir[: len(fir) // 8] ** 2) > 0.9 * np.sum(fir ** 2)
        # magnitude response of the taps follows the requested curve (dB, away from the band edges)
        f = np.fft.rfftfreq(len(fir) * 4, 1 / fs)
        H = 20 * np.log10(np.abs(np.fft.rfft(fir, len(fir) * 4)) + 1e-30)
        from oracle.impulse_response import interpolate_log
        sel = (f > 50) & (f < fs / 2 * 0.8)
        assert np.max(np.abs(H[sel] - interpolate_log(freq, g[f"fs{fs}_{nm}_eq"], f[sel]))) < 0.5
    fr = FrequencyResponse("x", frequency=freq, equalization=g[f"fs{fs}_wavy_eq"])
    one = fr.minimum_phase_impulse_response(fs=fs, normalize=False, f_res=5)
    assert np.array_equal(one, firs[1])                          # batched == single, bit for bit
    bad = fir_design_gain(freq, g[f"fs{fs}_flat_eq"], fs, 5, False)
    bad[-1] = 1.0
    with pytest.raises(NativeError):
        gpu_ctx.minphase_fir(bad, fs)                            # type II filter needs a Nyquist zero


def test_minimum_phase_fir_then_equalize_chain(gpu_ctx, golden):
    """EQ curve -> FIR (K6) -> ir.equalize(fir) (K5): the chain of core/pipeline.py:668-691."""
    from impulse_hip.frequency_response import minimum_phase_impulse_response
    from impulse_hip.impulse_response import ImpulseResponse
    from oracle.minphase import minimum_phase_impulse_response as oracle_fir
    from oracle.scipy_restated import fft_convolve
    g = golden("minphase")
    freq, eq = g["fs48000_freq"], g["fs48000_tilt_eq"]
    fir = minimum_phase_impulse_response(freq, eq, 48000, f_res=5, normalize=False)
    ofir = oracle_fir(freq, eq, 48000, f_res=5, normalize=False)
    assert np.max(np.abs(fir - ofir)) <= 5e-8 * np.max(np.abs(ofir))
    rng = np.random.default_rng(3)
    data = (rng.standard_normal(33600) * np.exp(-np.arange(33600) / 3000.0)).astype(np.float32).astype(np.float64)
    ir = ImpulseResponse(data.copy(), 48000)
    ir.equalize(fir)
    # same taps on both sides: the two FIRs differ (within the 5e-8 design floor asserted above) by a
    # coherent tone at the ill-conditioned Nyquist bin, which would otherwise dominate a spectrum metric
    ref = fft_convolve(data, fir, "full")
    assert len(ir.data) == 33600 + 9600 - 1
    assert rel(ir.data, ref) <= TIME_TOL and spec_rel(ir.data, ref) <= SPEC_TOL
    assert rel(ir.data, fft_convolve(data, ofir, "full")) <= TIME_TOL


# ------------------------------------------------------------------------------------------------
# a13/a14/a18: room correction, EQ worker and the pipeline slice through the product classes
# ------------------------------------------------------------------------------------------------
def test_room_correction_and_worker_real_column(gpu_ctx, golden):
    """Real FC-left column: GPU estimate -> crop -> frequency_response -> mic calibration / target /
    400 Hz limit -> EQ worker (curve on host, FIR on GPU) -> equalize (GPU), against the reference."""
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.hrir import _hann
    from impulse_hip.impulse_response import ImpulseResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.parallel_workers import init_equalization_worker, process_equalization_worker
    from impulse_hip.room_correction import calculate_specific_room_corrections
    g, d = golden("room_fc"), golden("demo_fc")
    N, P, n_out = int(d["N"]), float(d["P"]), int(d["responses_len"])
    e = ImpulseResponseEstimator(min_duration=(N - 1) / 48000, fs=48000)
    ir = ImpulseResponse(e.estimate((d["column_i32"].astype(np.float64) / 2 ** 31).astype(np.float32)), 48000)
    ir.crop_head()
    fo = 2 * int(48000 * (N / 48000 / P) * (1 / 24))
    ir.data = ir.data[:n_out].copy()
    ir.data[n_out - fo // 2:] *= _hann(fo)[fo // 2:]
    freq = g["frequency"]
    target = FrequencyResponse("room-target", frequency=freq.copy(), raw=g["target_raw"])
    mic = FrequencyResponse("mic", frequency=freq.copy(), raw=g["mic_raw"])

    class Rir:
        irs = {"FC": {"left": ir}}

    # the same chain on the reference's own float64 IR isolates the curve logic + K2 from the fp32 IR
    ref_data = d["cropped_head"].copy()
    ref_data[n_out - fo // 2:] *= _hann(fo)[fo // 2:]

    class RirRef:
        irs = {"FC": {"left": ImpulseResponse(ref_data, 48000)}}

    np.testing.assert_allclose(RirRef.irs["FC"]["left"].frequency_response().raw, g["fr_raw_initial"], rtol=0, atol=1e-9)
    fr64 = calculate_specific_room_corrections(RirRef, target, mic_calibration=mic, limit=400)["FC"]["left"]
    np.testing.assert_allclose(fr64.raw, g["fr_raw"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(fr64.error, g["fr_error"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(fr64.target, g["fr_target"], rtol=0, atol=1e-9)

    frs = calculate_specific_room_corrections(Rir, target, mic_calibration=mic, limit=400)
    fr = frs["FC"]["left"]
    # dB curves from an fp32 IR: 1e-6 of the spectrum peak is ~1e-3 dB on bins 40 dB below it
    assert np.max(np.abs(fr.raw - g["fr_raw"])) < 5e-3
    assert np.max(np.abs(fr.error - g["fr_error"])) < 5e-3
    assert np.all(fr.error[freq > 400] == 0.0)                      # limit mask
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=24000, f_step=1.01)
    init_equalization_worker(frs, None, None, None, None, FrequencyResponse("t", frequency=common.copy(), raw=0),
                             common, 48000)
    sp, sd, fir = process_equalization_worker(("FC", "left"))
    want = g["worker_fir"]
    assert (sp, sd) == ("FC", "left") and fir.shape == want.shape
    # taps are a very sensitive function of the curve (see tests/test_oracle_golden.py): compare responses
    Hg, Hw = np.fft.rfft(fir, 4 * len(fir)), np.fft.rfft(want, 4 * len(want))
    f = np.fft.rfftfreq(4 * len(fir), 1 / 48000)
    sel = f < 20000
    assert np.max(np.abs(20 * np.log10(np.abs(Hg[sel]) / np.abs(Hw[sel])))) < 0.02
    assert np.max(np.abs(fir - want)) <= 1e-3 * np.max(np.abs(want))


def test_pipeline_slice_product(gpu_ctx, golden, tmp_path):
    """a18 end to end through the product: WAV -> HRIR.open_recording (batched GPU deconvolution) ->
    crop_heads -> crop_tails -> batched FIR design + equalize -> normalize, stage by stage against the
    reference's own run of the same synthetic FL,FR folder."""
    from impulse_hip.audio_io import write_wav
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.pipeline_slice import run_slice
    g = golden("pipeline_slice")
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    N, fs = len(e), 48000
    import slice_input                                   # tests/golden/slice_input.py (inputs only)
    tracks = slice_input.make_tracks(e.test_signal, fs)
    path = str(tmp_path / "FL,FR.wav")
    from scipy.io import wavfile
    wavfile.write(path, fs, slice_input.to_pcm32(tracks).T)        # the bytes the golden run read (clipped, scale 2^31)
    order = [("FL", "left"), ("FL", "right"), ("FR", "left"), ("FR", "right")]
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    room = {sp: {} for sp in ("FL", "FR")}
    for sp, sd in order:
        room[sp][sd] = FrequencyResponse("r", frequency=common.copy(), raw=0, error=g[f"room_error_{sp}_{sd}"])
    stages = {}
    hrir, gain = run_slice(e, [(path, ["FL", "FR"])], room_frs=room, stages=stages)
    assert list(hrir.irs) == ["FL", "FR"]
    from impulse_hip.impulse_response import ImpulseResponse
    assert [ImpulseResponse(stages["ingest"][k], fs).peak_index() for k in order] == list(g["ingest_peaks"])
    assert len(stages["ingest"][order[0]]) == int(g["ingest_len"])
    assert [len(stages["crop_heads"][k]) for k in order] == list(g["heads_len"])
    for k in order:
        want = g[f"heads_{k[0]}_{k[1]}"]
        assert np.max(np.abs(stages["crop_heads"][k][:512] - want)) <= TIME_TOL * 0.5     # IR peak ~0.5
    assert len(stages["crop_tails"][order[0]]) == int(g["tail_ind"])
    for k in order:
        want = g[f"tails_{k[0]}_{k[1]}"]
        assert np.max(np.abs(stages["crop_tails"][k][::3] - want)) <= TIME_TOL * np.max(np.abs(want))
    assert len(stages["equalize"][order[0]]) == int(g["eq_len"])
    assert gain == pytest.approx(float(g["norm_gain_db"]), abs=1e-4)
    for k in order:
        want = g[f"final_{k[0]}_{k[1]}"]
        got = hrir.irs[k[0]][k[1]].data
        assert got.shape == want.shape
        assert rel(got, want) <= 2e-6                     # fp32 IR through an fp32 FIR convolution and a gain
        # The FIR here is designed from curves recomputed by the product (1e-12 away from the reference's),
        # and the reference's homomorphic design turns that into a coherent tone AT the Nyquist bin (see
        # tests/test_oracle_golden.py).  Spectrum parity is therefore asserted below 0.9 Nyquist at the
        # usual scale and over the whole band at the scale of that design instability.
        A = np.abs(np.fft.rfft(got))
        R = np.abs(np.fft.rfft(want))
        band = np.fft.rfftfreq(len(want), 1 / fs) < 0.9 * fs / 2
        assert np.max(np.abs(A - R)[band]) / np.max(R) <= 2e-6
        assert np.max(np.abs(A - R)) / np.max(R) <= 5e-5


# ------------------------------------------------------------------------------------------------
# K2: magnitude response of arbitrary-length rows (fp64 Bluestein on the device)
# ------------------------------------------------------------------------------------------------
def test_magnitude_response_golden(gpu_ctx, golden):
    """The reference pins this function bit-exactly to np.fft.rfft (tests/test_magnitude_response_parity.py);
    a different transform algorithm cannot be bit-identical, so the device result is held to 1e-9 dB
    (fp64 Bluestein: ~1e-13 relative on |X|), including odd and non-smooth lengths."""
    from impulse_hip.audio_io import magnitude_response, magnitude_responses
    g = golden("magnitude")
    for seed, sizes in ((0xA110, (8, 1024, 48000)), (0xA111, (9, 1025, 48001))):
        rr = np.random.default_rng(seed)
        for n in sizes:
            x = rr.standard_normal(n).astype(np.float32).astype(np.float64)
            f, m = magnitude_response(x, 48000)
            assert np.array_equal(f, g[f"n{n}_f"])
            assert m.shape == g[f"n{n}_db"].shape
            assert np.max(np.abs(m - g[f"n{n}_db"])) <= 1e-9
    # impulse, all-zero (-inf like the reference: no epsilon), batched prime length
    x = np.zeros(2048)
    x[0] = 1.0
    assert np.max(np.abs(magnitude_response(x, 48000)[1])) <= 1e-9
    assert np.all(np.isneginf(magnitude_response(np.zeros(100), 48000)[1]))
    rows = np.random.default_rng(1).standard_normal((5, 43199))            # 43199 = 13 * 3323
    f, m = magnitude_responses(rows, 48000)
    with np.errstate(divide="ignore"):
        ref = 20 * np.log10(np.abs(np.fft.rfft(rows, axis=1)[:, : (43199 + 1) // 2]))
    assert m.shape == ref.shape and np.max(np.abs(m - ref)) <= 1e-9
    assert np.array_equal(magnitude_response(np.zeros(0), 48000)[1], np.zeros(0))


def test_magnitude_direct_transform_equals_bluestein(gpu_ctx, monkeypatch):
    """K2 takes the n-point transform itself when n is a product of the Stockham radices (every length crop_tails leaves:
    next_fast_len) and Bluestein's chirp-z otherwise: both against np.fft.rfft at 1e-9 dB and against each other, on a
    context of each kind (the switch is read when a context is made)."""
    from impulse_hip import Context
    rng = np.random.default_rng(77)
    monkeypatch.setenv("IMPULSE_HIP_K2_BLUESTEIN", "1")
    chirp = Context(0)
    monkeypatch.delenv("IMPULSE_HIP_K2_BLUESTEIN")
    try:
        for n in (2, 6, 2048, 27000, 55296, 2 * 3 * 5 * 11, 65536):
            rows = rng.standard_normal((3, n)) * np.exp(-np.arange(n) / (n / 6.0))
            with np.errstate(divide="ignore"):
                ref = 20 * np.log10(np.abs(np.fft.rfft(rows, axis=1)[:, : (n + 1) // 2]))
            direct = gpu_ctx.magnitude_db(rows)
            via_chirp = chirp.magnitude_db(rows)
            assert direct.shape == ref.shape == via_chirp.shape
            assert np.max(np.abs(direct - ref)) <= 1e-9, n
            assert np.max(np.abs(via_chirp - ref)) <= 1e-9, n
    finally:
        chirp.close()


# ------------------------------------------------------------------------------------------------
# alignment / shift (next-tier row f1; host-side correlations): the reference's own assertions
# (tests/test_dsp_stages.py:105-166)
# ------------------------------------------------------------------------------------------------
def test_alignment_and_shift_behaviour(gpu_ctx):
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse

    class Est:
        fs = 48000

    def ir(d):
        return ImpulseResponse(np.asarray(d, dtype=np.float64), 48000)

    def impulse_at(i, n=2048):
        d = np.zeros(n)
        d[i] = 1.0
        return ir(d)

    def lag(a, b, seg=1440):
        corr = np.correlate(a[:seg], b[:seg], mode="full")
        return int(np.arange(-len(a[:seg]) + 1, len(a[:seg]))[np.argmax(corr)])

    h = HRIR(Est())
    h.irs = {"FL": {"left": impulse_at(60), "right": impulse_at(60)},
             "FR": {"left": impulse_at(60), "right": impulse_at(67)}}
    assert abs(lag(h.irs["FL"]["left"].data, h.irs["FR"]["right"].data)) == 7
    h.align_ipsilateral_all(speaker_pairs=[("FL", "FR")], segment_ms=30)
    assert abs(lag(h.irs["FL"]["left"].data, h.irs["FR"]["right"].data)) <= 1
    assert all(len(x.data) == 2048 for p in h.irs.values() for x in p.values())
    # onset groups: every group's left-ear peak is moved onto FL's (device peak search)
    h = HRIR(Est())
    h.irs = {"FL": {"left": impulse_at(100), "right": impulse_at(110)},
             "SL": {"left": impulse_at(130), "right": impulse_at(150)},
             "FC": {"left": impulse_at(90), "right": impulse_at(90)}}
    h.align_onset_groups_peak_leftref()
    assert [h.irs[s]["left"].peak_index() for s in ("FL", "SL", "FC")] == [100, 100, 100]
    assert h.irs["SL"]["right"].peak_index() == 120 and h.irs["FC"]["right"].peak_index() == 100
    del h.irs["FL"]
    with pytest.raises(RuntimeError):
        h.align_onset_groups_peak_leftref()
    x = ir([1.0, 2.0, 3.0, 4.0])
    x.shift(2)
    np.testing.assert_array_equal(x.data, [0.0, 0.0, 1.0, 2.0])
    x = ir([1.0, 2.0, 3.0, 4.0])
    x.shift(-1)
    np.testing.assert_array_equal(x.data, [2.0, 3.0, 4.0, 0.0])
    x.shift(0)
    np.testing.assert_array_equal(x.data, [2.0, 3.0, 4.0, 0.0])
    x.shift(3)
    x.shift(-3)
    assert len(x.data) == 4


def test_overlapped_launch_groups_are_bit_identical(gpu_ctx):
    """imp_plan_set_overlap: launch groups on 3 lanes (private workspace slices, 3 streams) must give
    exactly the bytes of the strictly serial order, within one call and across calls."""
    from impulse_hip import ConvPlan, NativeError
    rng = np.random.default_rng(12)
    L, M, B = 60000, 40000, 13
    x = rng.standard_normal((B, L)).astype(np.float32)
    h = rng.standard_normal(M)
    plan = ConvPlan(gpu_ctx, h, L, "same", ws_channels=9)          # 3 lanes x 3 channels -> 5 groups
    serial = plan.execute(x)
    pitch = L
    d_x, d_y1, d_y2 = gpu_ctx.malloc(B * pitch * 4), gpu_ctx.malloc(B * pitch * 4), gpu_ctx.malloc(B * pitch * 4)
    gpu_ctx.h2d(d_x, x)
    plan.set_overlap(3)
    plan.execute_device(d_x, B, pitch, d_y1, pitch)
    plan.execute_device(d_x, B, pitch, d_y2, pitch)                 # a second call in flight, other output
    gpu_ctx.synchronize()
    for d in (d_y1, d_y2):
        got = np.empty((B, pitch), dtype=np.float32)
        gpu_ctx.d2h(got, d)
        assert np.array_equal(got, serial)
    with pytest.raises(NativeError):
        plan.execute(x)                                             # host-buffer path needs strict order
    with pytest.raises(NativeError):
        plan.set_overlap(5)
    plan.set_overlap(1)
    assert np.array_equal(plan.execute(x), serial)
    for d in (d_x, d_y1, d_y2):
        gpu_ctx.free(d)
    plan.close()


def test_pcm_wire_format_ingest(gpu_ctx, tmp_path):
    """HRIR.open_recording on 32- and 16-bit PCM files goes through the PCM loader (raw interleaved
    frames on the device); it must give the same IRs as the float path and as the oracle."""
    from impulse_hip.audio_io import read_wav, read_wav_pcm, write_wav
    from impulse_hip.hrir import HRIR, ingest_recording
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle.estimator import estimate
    import slice_input
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    tracks = slice_input.make_tracks(e.test_signal, 48000)
    for bits, tol in ((32, TIME_TOL), (16, TIME_TOL)):
        path = str(tmp_path / f"FL,FR-{bits}.wav")
        write_wav(path, 48000, tracks, bit_depth=bits)
        fs, frames = read_wav_pcm(path)
        assert frames.shape == (tracks.shape[1], 4) and frames.dtype == (np.int32 if bits == 32 else np.int16)
        h = HRIR(e)
        h.open_recording(path, ["FL", "FR"])
        fs2, rec = read_wav(path, expand=True)
        ref = ingest_recording(e, 48000, fs2, rec, ["FL", "FR"])           # float path of the product
        for sp in ("FL", "FR"):
            for sd in ("left", "right"):
                a, b = h.irs[sp][sd], ref[sp][sd]
                assert np.array_equal(a.recording, b.recording)            # identical float64 columns
                o = estimate(b.recording, e.inverse_filter)
                assert rel(a.data, o) <= tol and rel(b.data, o) <= tol
                assert a.peak_index() == b.peak_index()
    (tmp_path / "f24.wav").write_bytes(b"")                                # 24-bit falls back to the float reader
    write_wav(str(tmp_path / "f24.wav"), 48000, tracks[:, : 2 * 48000 + len(e) + 96000], bit_depth=24)
    assert read_wav_pcm(str(tmp_path / "f24.wav")) is None
    h = HRIR(e)
    h.open_recording(str(tmp_path / "f24.wav"), ["FL", "FR"])
    assert set(h.irs) == {"FL", "FR"}


def test_k10_lag_search_matches_scipy_and_oracle(gpu_ctx):
    """K10 (imp_xcorr_argmax): argmax of the full cross-correlation, fp64 on the device, against
    scipy.signal.correlate (what the reference calls, core/hrir.py:934) and the oracle's alignment."""
    from scipy import signal
    from impulse_hip.constants import IPSILATERAL_PAIRS
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    from oracle.hrir import align_ipsilateral_all
    rng = np.random.default_rng(10)

    def room_ir(delay, n=4096, fs=48000):
        t = np.arange(n)
        x = rng.standard_normal(n) * np.exp(-t / 600.0) * 0.2
        x[:delay] = 0.0
        x[delay] = 1.0
        return x

    a_rows, b_rows = [], []
    for na, nb in ((1440, 1440), (2880, 2880), (1000, 1440), (1440, 777), (1, 1), (5, 1), (8191, 8193)):
        a_rows.append(room_ir(int(rng.integers(0, max(na // 4, 1))), na) if na > 8 else rng.standard_normal(na))
        b_rows.append(room_ir(int(rng.integers(0, max(nb // 4, 1))), nb) if nb > 8 else rng.standard_normal(nb))
    a_rows.append(np.zeros(64)); b_rows.append(np.zeros(64))                 # all-zero: first index wins
    a_rows.append(np.ones(16)); b_rows.append(np.ones(16))                   # symmetric triangle, single max
    arg, val = gpu_ctx.xcorr_argmax(a_rows, b_rows)
    for a, b, k, v in zip(a_rows, b_rows, arg, val):
        corr = signal.correlate(a, b, mode="full")
        assert int(k) == int(np.argmax(corr)), (len(a), len(b))
        assert abs(v - corr.max()) <= 1e-12 * max(1.0, abs(corr).max())
    with pytest.raises(Exception):
        gpu_ctx.xcorr_argmax([np.zeros(9000)], [np.zeros(9000)])             # beyond the LDS-resident size

    class Est:
        fs = 48000
    speakers = ["FL", "FR", "SL", "SR", "BL", "BR", "FC", "WL", "WR"]
    irs = {sp: {"left": room_ir(40 + 9 * i), "right": room_ir(52 + 5 * i)} for i, sp in enumerate(speakers)}
    h = HRIR(Est())
    h.irs = {sp: {sd: ImpulseResponse(x.copy(), 48000) for sd, x in pair.items()} for sp, pair in irs.items()}
    h.align_ipsilateral_all()
    want = align_ipsilateral_all(irs, 48000, IPSILATERAL_PAIRS)
    for sp in speakers:
        for sd in ("left", "right"):
            np.testing.assert_array_equal(h.irs[sp][sd].data, want[sp][sd])
    # a pair list that revisits a speaker: the second search must see the first shift
    chain = [("FL", "FR"), ("FR", "SL"), ("SL", "FL")]
    h.irs = {sp: {sd: ImpulseResponse(x.copy(), 48000) for sd, x in pair.items()} for sp, pair in irs.items()}
    h.align_ipsilateral_all(speaker_pairs=chain)
    want = align_ipsilateral_all(irs, 48000, chain)
    for sp in ("FL", "FR", "SL"):
        for sd in ("left", "right"):
            np.testing.assert_array_equal(h.irs[sp][sd].data, want[sp][sd])


def test_alignment_matches_reference_run(gpu_ctx, golden):
    """HRIR.align_ipsilateral_all (device lag search) and align_onset_groups_peak_leftref (device peak
    search) against the reference's own run on the seeded nine-speaker set (fixture section 11)."""
    from make_goldens import alignment_inputs
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    g = golden("alignment")
    irs = alignment_inputs()

    class Est:
        fs = 48000
    for name, call in (("ipsi", lambda hh: hh.align_ipsilateral_all()),
                       ("chain", lambda hh: hh.align_ipsilateral_all(speaker_pairs=[("FL", "FR"), ("FR", "SL"), ("SL", "FL")])),
                       ("onset", lambda hh: hh.align_onset_groups_peak_leftref())):
        h = HRIR(Est())
        h.irs = {sp: {sd: ImpulseResponse(x.copy(), 48000) for sd, x in pair.items()} for sp, pair in irs.items()}
        call(h)
        for sp in irs:
            for sd in ("left", "right"):
                d = h.irs[sp][sd].data
                assert int(np.argmax(np.abs(d))) == int(g[f"{name}_{sp}_{sd}_peak"]), (name, sp, sd)
                np.testing.assert_array_equal(d[:96], g[f"{name}_{sp}_{sd}_head"])
                np.testing.assert_array_equal(d[-96:], g[f"{name}_{sp}_{sd}_tail"])
        # the same with the responses as rows of a device block (fp32: the inputs rounded first, so the comparison is with
        # the host-array form on the SAME rounded inputs - lags and peaks are integers, the shifted samples exact copies)
        from impulse_hip.device_rows import DeviceBlock, Row
        names = [(sp, sd) for sp in irs for sd in ("left", "right")]
        rounded = {k: irs[k[0]][k[1]].astype(np.float32) for k in names}
        pitch = max(len(x) for x in rounded.values()) + 64
        block = DeviceBlock(gpu_ctx, pitch * len(names))
        flat = np.zeros(pitch * len(names), dtype=np.float32)
        for i, k in enumerate(names):
            flat[i * pitch:i * pitch + len(rounded[k])] = rounded[k]
        gpu_ctx.h2d(block.ptr, flat)
        dev, host = HRIR(Est()), HRIR(Est())
        for i, (sp, sd) in enumerate(names):
            dev.irs.setdefault(sp, {})[sd] = ImpulseResponse.on_device(Row(block, i * pitch, len(rounded[(sp, sd)])), 48000)
            host.irs.setdefault(sp, {})[sd] = ImpulseResponse(rounded[(sp, sd)].astype(np.float64), 48000)
        call(dev)
        call(host)
        for sp, sd in names:
            assert dev.irs[sp][sd]._data is None and dev.irs[sp][sd]._row is not None, (name, sp, sd)
            np.testing.assert_array_equal(dev.irs[sp][sd].peek(), host.irs[sp][sd].data)


@pytest.mark.parametrize("method", ["mids", "trend", "left", "right", "avg", "min", "-1.5"])
def test_channel_balance_matches_reference_run(gpu_ctx, golden, method):
    """HRIR.correct_channel_balance (device K2 spectra, K6 FIRs, K5 convolutions; curve logic host)
    against the reference's run (fixture section 11).  Gains-only methods are exact to fp32; methods
    that design a minimum-phase FIR inherit K6's conditioning at the forced Nyquist zero:
    the FIR moves ~2e-4 of ITS peak per decade of a 1e-11 quantity, so chained results are held to 5e-5."""
    from make_goldens import balance_inputs, spectrum_probe
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    g = golden("alignment")

    class Est:
        fs = 48000
    h = HRIR(Est())
    h.irs = {sp: {sd: ImpulseResponse(x.astype(np.float64), 48000) for sd, x in pair.items()}
             for sp, pair in balance_inputs().items()}
    h.correct_channel_balance(method)
    tol = 2e-6 if method in ("mids", "-1.5") else 5e-5
    for sp in ("FL", "FR", "FC"):
        for sd in ("left", "right"):
            d = h.irs[sp][sd].data
            key = f"bal_{method}_{sp}_{sd}"
            assert len(d) == int(g[key + "_len"])
            scale = float(g[key + "_absmax"])
            assert np.max(np.abs(d[:768] - g[key + "_head"])) / scale <= tol, (method, sp, sd)
            bins, probe, _ = spectrum_probe(d, 256)
            err = np.abs(probe - g[key + "_probe"]) / float(g[key + "_specmax"])
            low = bins < 0.9 * (len(d) // 2)
            assert err[low].max() <= 2e-6, (method, sp, sd)               # below 0.9 Nyquist: fp32 conv accuracy
            assert err.max() <= (2e-6 if method in ("mids", "-1.5") else 5e-4)   # the Nyquist bin itself (K6 note)
    with pytest.raises(ValueError):
        h.correct_channel_balance("loud")


def test_shared_context_from_many_threads(gpu_ctx):
    """The reference calls these methods from ThreadPoolExecutor workers (core/hrir.py:529-537) and, on
    no-GIL builds, runs the EQ workers as threads (core/parallel_utils.py:53-55); ctypes drops the GIL, so
    one context really is entered concurrently.  Results must equal the single-threaded ones."""
    from concurrent.futures import ThreadPoolExecutor
    from impulse_hip import ConvPlan
    rng = np.random.default_rng(99)
    rows = [rng.standard_normal(5000 + 37 * i).astype(np.float32) for i in range(24)]
    gains = [np.abs(1.0 + 0.2 * rng.standard_normal((2, 1200))) for _ in range(6)]
    for g in gains:
        g[:, -1] = 0.0
    xs = [rng.standard_normal((2, 30000)).astype(np.float32) for _ in range(6)]
    plan = ConvPlan(gpu_ctx, rng.standard_normal(2000), 30000, "same")
    want_pk = [gpu_ctx.peak_index([r])[0][0] for r in rows]
    want_fir = [gpu_ctx.minphase_fir(g, 48000) for g in gains]
    want_mag = [gpu_ctx.magnitude_db(r[None, :4096].astype(np.float64)) for r in rows[:6]]
    want_y = [plan.execute(x) for x in xs]

    def job(i):
        kind = i % 4
        if kind == 0:
            return gpu_ctx.peak_index([rows[i % 24]])[0][0] == want_pk[i % 24]
        if kind == 1:
            return np.array_equal(gpu_ctx.minphase_fir(gains[i % 6], 48000), want_fir[i % 6])
        if kind == 2:
            return np.array_equal(gpu_ctx.magnitude_db(rows[i % 6][None, :4096].astype(np.float64)), want_mag[i % 6])
        return np.array_equal(plan.execute(xs[i % 6]), want_y[i % 6])

    with ThreadPoolExecutor(max_workers=8) as pool:
        ok = list(pool.map(job, range(96)))
    plan.close()
    assert all(ok)


def test_device_filter_spectrum_matches_host_fp64(gpu_ctx):
    """Plans prepare alpha/beta in fp64 on the device (Stockham FFT + unpack, rounded once); the host fp64
    preparation (IMPULSE_HIP_HOST_SPECTRUM=1) must give the same fp32 planes to the last bit or two."""
    from impulse_hip import ConvPlan
    rng = np.random.default_rng(123)
    for M, L, mode, nf in ((1, 1, "same", 1), (999, 1000, "same", 1), (9600, 32640, "full", 3),
                           (147635, 243635, "same", 1), (100001, 150000, "full", 1)):
        h = rng.standard_normal((nf, M)) * np.exp(-np.arange(M) / max(M / 4.0, 1.0))
        planes = []
        for host in ("0", "1"):
            os.environ["IMPULSE_HIP_HOST_SPECTRUM"] = host
            try:
                p = ConvPlan(gpu_ctx, h if nf > 1 else h[0], L, mode)
            finally:
                os.environ.pop("IMPULSE_HIP_HOST_SPECTRUM", None)
            dptr, nbytes = p.spectrum_buffer()
            ab = np.empty(nbytes // 4, dtype=np.float32)
            gpu_ctx.d2h(ab, dptr)
            planes.append(ab)
            p.close()
        dev, host = planes
        scale = np.max(np.abs(host))
        assert np.max(np.abs(dev - host)) <= 2.5e-7 * scale, (M, L, mode)      # <= 2 ulp of the largest bin
        assert np.mean(dev == host) > 0.98                                     # and almost always bit-identical


def test_equalize_channels_equals_per_channel_equalize(gpu_ctx):
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    rng = np.random.default_rng(321)

    class Est:
        fs = 48000
    data = {sp: {sd: rng.standard_normal(3000 if sp != "FC" else 2500) for sd in ("left", "right")} for sp in ("FL", "FR", "FC")}
    firs = {(sp, sd): rng.standard_normal(700) * 0.1 for sp in data for sd in ("left", "right")}
    a, b = HRIR(Est()), HRIR(Est())
    for h in (a, b):
        h.irs = {sp: {sd: ImpulseResponse(x.copy(), 48000) for sd, x in pair.items()} for sp, pair in data.items()}
    a.equalize_channels(firs)
    for (sp, sd), fir in firs.items():
        b.irs[sp][sd].equalize(fir)
    for sp in data:
        for sd in ("left", "right"):
            np.testing.assert_array_equal(a.irs[sp][sd].data, b.irs[sp][sd].data)   # same kernels, same bits


def test_k7_range_means_have_numpy_bits(gpu_ctx):
    """K7: e = (x / max|x|)^2 and np.mean over windows / ranges of it, bit for bit (NumPy's pairwise order)."""
    from impulse_hip._native import SegSet
    rng = np.random.default_rng(7)
    rows = [rng.standard_normal(n) * np.exp(-np.arange(n) / max(n / 6.0, 1.0)) for n in (96000, 12345, 130, 128, 9, 7, 1)]
    rows.append(rng.random(50000) + 0.5)                        # no decay: every piece of a long run matters
    rows.append(np.zeros(50))                                   # max < 1e-20: no normalisation
    rows.append(np.full(300, 1e-25))
    s = SegSet(gpu_ctx, rows)
    want_e = []
    for r, m in zip(rows, s.maxabs):
        top = np.max(np.abs(r))
        assert m == top
        want_e.append((r / top) ** 2 if top >= 1e-20 else r ** 2)
    queries = []
    for j, e in enumerate(want_e):
        n = len(e)
        queries += [(j, 0, n), (j, n // 10, n), (j, 0, min(n, 8)), (j, min(3, n), min(3 + 129, n)), (j, n // 2, n // 2)]
        for w in (1, 7, 8, 9, 127, 128, 129, 1000, 1440, 1441, 8192, 8193, 9000, 20000):
            if w <= n:
                k = n // w
                queries += [(j, i * w, (i + 1) * w) for i in range(0, k, max(k // 5, 1))]
    got = s.range_means(queries)
    for (j, a, b), g in zip(queries, got):
        if a == b:
            assert np.isnan(g)
            continue
        ref = np.mean(want_e[j][a:b])
        assert g == ref, (j, a, b, g, ref)
    # the (n, w) window form the reference uses must agree with the range form
    e, w = want_e[0], 1440
    k = len(e) // w
    ref = e[: k * w].reshape(k, w).mean(axis=1)
    got = s.range_means([(0, i * w, (i + 1) * w) for i in range(k)])
    np.testing.assert_array_equal(got, ref)
    with pytest.raises(Exception):
        s.range_means([(0, 5, 96001)])
    s.close()


def test_decay_params_batch_equals_single_and_goldens(gpu_ctx, golden):
    """All twelve golden decays searched in lock step (one device call per round of queries) give the
    reference's knee, window and - because the means carry NumPy's bits - its noise floor exactly."""
    from impulse_hip.decay import decay_params, decay_params_batch
    g = golden("decay")
    keys, datas = [], []
    for rt60 in (0.3, 0.6, 1.0, 1.5):
        for seed in (0, 11, 22):
            keys.append(f"rt{int(rt60 * 10)}_s{seed}")
            datas.append(_decaying_sine(48000, 3.0, rt60, seed=seed).astype(np.float32).astype(np.float64))
    datas += [np.zeros(5), np.zeros(4000), np.ones(20)]              # degenerate inputs ride along
    batch = decay_params_batch(datas, 48000)
    for k, d, bp in zip(keys, datas, batch):
        want = g[k + "_params"]
        assert (int(bp[0]), int(bp[1]), int(bp[3])) == (int(want[0]), int(want[1]), int(want[3])), k
        assert float(bp[2]) == float(want[2]), k
    for d, bp in zip(datas, batch):
        one = decay_params(d, 48000)
        assert (int(one[0]), int(one[1]), float(one[2]), int(one[3])) == (int(bp[0]), int(bp[1]), float(bp[2]), int(bp[3]))
    assert batch[12] == (0, 5, -200.0, 5)


def test_decay_times_device_vs_oracle_edges(gpu_ctx):
    """K7b against the oracle's decay_times on shapes the goldens do not reach: knee at the very end,
    peak at sample 0, windows longer than the analysed stretch, truncated responses."""
    from oracle import decay as odecay
    rng = np.random.default_rng(17)
    fs = 48000
    cases = []
    for rt60, n, lead in ((0.2, 30000, 0), (0.4, 48000, 500), (0.8, 96000, 2000), (0.3, 6000, 100), (0.5, 20000, 19000)):
        t = np.arange(n) / fs
        x = rng.standard_normal(n) * 10 ** (-3.0 * t / rt60)
        x = np.concatenate((rng.standard_normal(lead) * 1e-4, x))[:n] if lead else x
        x[lead if lead < n else 0] = 3.0
        cases.append(x)
    rows, pk, kn, nf, ws = [], [], [], [], []
    for x in cases:
        p = odecay.decay_params(x, fs)
        for knee, win in ((p[1], p[3]), (len(x), p[3]), (p[1], 7), (min(p[0] + 50, len(x)), 4000), (p[1], max(len(x) * 2, 10))):
            rows.append(x); pk.append(p[0]); kn.append(knee); nf.append(p[2]); ws.append(win)
    got = gpu_ctx.decay_times(rows, pk, kn, nf, ws, fs)
    import warnings
    for x, p0, k0, f0, w0, g in zip(rows, pk, kn, nf, ws, got):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                want = odecay.decay_times(x, fs, p0, k0, f0, w0)
            except Exception:                                    # noqa: BLE001 - the reference raises on some degenerate slices
                continue
        for gv, wv in zip(g, want):
            if wv is None:
                assert np.isnan(gv), (p0, k0, w0, g, want)
            else:
                assert gv == pytest.approx(wv, rel=1e-9), (p0, k0, w0, g, want)


def test_conv_fuzz_layouts_and_sizes(gpu_ctx):
    """Seeded fuzz over sizes (every column radix is reachable), modes, batch sizes and input layouts
    (planar with odd pitches, interleaved float frames, PCM16 / PCM32 wire format) against the oracle."""
    from impulse_hip import ConvPlan
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(20261004)
    for case in range(36):
        L = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20000), rng.integers(20000, 400000)]))
        M = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20000), rng.integers(20000, 300000)]))
        mode = "same" if rng.random() < 0.5 else "full"
        B = int(rng.integers(1, 6))
        h = rng.standard_normal(M) * np.exp(-np.arange(M) / max(M / 5.0, 1.0))
        layout = ("planar", "frames", "pcm16", "pcm32")[case % 4]
        plan = ConvPlan(gpu_ctx, h, L, mode, ws_channels=int(rng.integers(1, B + 1)))
        if layout == "planar":
            x = rng.standard_normal((B, L)).astype(np.float32)
            y = plan.execute(x)
            ref_in = x.astype(np.float64)
        elif layout == "frames":
            frames = rng.standard_normal((L, B)).astype(np.float32)
            plan.close()
            plan = ConvPlan(gpu_ctx, h, L, mode)                  # interleaved execution uses one group
            y = plan.execute_interleaved(frames)
            ref_in = frames.T.astype(np.float64)
        else:
            bits = 16 if layout == "pcm16" else 32
            dt = np.int16 if bits == 16 else np.int32
            lead = int(rng.integers(0, 50))
            frames = rng.integers(-2 ** (bits - 1), 2 ** (bits - 1), size=(lead + L + 7, B), dtype=np.int64).astype(dt)
            plan.close()
            plan = ConvPlan(gpu_ctx, h, L, mode, ws_channels=B)
            y = plan.execute_pcm_columns(frames, [lead])[0]
            ref_in = (frames[lead:lead + L].T.astype(np.float64)) / 2.0 ** (bits - 1)
        plan.close()
        for b in range(B):
            ref = fft_convolve(ref_in[b], h, mode)
            assert y[b].shape == ref.shape
            # 32-bit PCM is rounded to fp32 on load (the device dtype): 6e-8 relative input error on top
            tol = TIME_TOL if layout != "pcm32" else 2 * TIME_TOL
            if mode == "same" and M > 4 * L:
                tol *= 1.5        # a short window of a long filter's output: the window's peak is small next to
                                  # the transform's rounding noise, which scales with the whole filter's energy
            assert rel(y[b], ref) <= tol, (case, layout, L, M, mode, B, b)


def _c3_tracks(e, rng, speakers, rt60, level):
    """[2 ears][2 s lead + 13 columns] of a 13-speaker sweep measurement at the estimator's rate (SURVEY 8d recipe family):
    per speaker and ear a direct sound plus a decaying noise tail, -85 dBFS background, rounded to fp32"""
    fs, N = e.fs, len(e)
    L = N + 2 * fs
    tracks = np.zeros((2, 2 * fs + L * len(speakers)))
    S = np.fft.rfft(level * e.test_signal, 1 << 20)
    t = np.arange(30000) / fs
    for i in range(len(speakers)):
        for ear in range(2):
            h = rng.standard_normal(30000) * 0.04 * 10 ** (-3.0 * t / rt60)
            d0 = 40 + 7 * i + 23 * ear
            h[: d0 + 30] = 0.0
            h[d0] = 1.0 - 0.3 * ear
            y = np.fft.irfft(S * np.fft.rfft(h, 1 << 20), 1 << 20)[: N + 30000 - 1]
            seg = tracks[ear, 2 * fs + i * L: 2 * fs + (i + 1) * L]
            seg[: len(y)] = y
    tracks += rng.standard_normal(tracks.shape) * 10 ** (-85 / 20)
    return tracks.astype(np.float32).astype(np.float64)


def test_c3_slice_13ch_96k_against_oracle(gpu_ctx, tmp_path, monkeypatch):
    """BASELINE config C3 as it is worded - "13-ch TrueHD layout x 2 ear @96 kHz with room_correction + decay window":
    (1) room_correction() (core/room_correction.py:78-210) on a SECOND batch of 26 synthetic room recordings at 96 kHz
    (one mono file per ear, 13 columns each; flat target, limit 400) against the oracle composition ingest -> crop_heads ->
    crop_tails -> specific_room_correction; (2) ITS error curves drive the measurement batch (5 s sweep, N = 635 965, column
    827 965, circular length 1 179 648) through ingest -> crop_heads -> crop_tails -> room-error FIRs -> equalize -> decay
    window -> normalize, product (device kernels K1/K3/K4/K7/K2/K12/K6/K5/K8) against the oracle composition of the same
    stages."""
    from impulse_hip.constants import TRUEHD_13CH_ORDER
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.pipeline_slice import run_slice
    from oracle import decay as odecay, estimator as oest, frequency_response as ofr, hrir as ohrir
    from oracle import impulse_response as oir, minphase as omin
    from oracle.scipy_restated import fft_convolve
    fs = 96000
    e = ImpulseResponseEstimator(min_duration=5.0, fs=fs)
    oe = oest.Estimator(min_duration=5.0, fs=fs)
    N = len(e)
    assert N == 635965 == len(oe)
    speakers = list(TRUEHD_13CH_ORDER)
    L = N + 2 * fs
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    assert len(common) == 852
    order26 = [(sp, sd) for sp in speakers for sd in ("left", "right")]

    # ---- (1) room correction on the second batch: 26 room responses at 96 kHz
    from impulse_hip import room_correction as rc
    from impulse_hip.audio_io import write_wav
    room_tracks = _c3_tracks(e, np.random.default_rng(0xC3 + 1000), speakers, rt60=0.35, level=0.3)
    names = [f"room-{','.join(speakers)}-{sd}.wav" for sd in ("left", "right")]
    for k, nm in enumerate(names):
        write_wav(str(tmp_path / nm), fs, room_tracks[k], bit_depth=32)
    # the reference levels every channel to the FIRST one it meets in os.listdir order: fix that order
    real_listdir = os.listdir
    monkeypatch.setattr(rc.os, "listdir", lambda p: names + [f for f in real_listdir(p) if f not in names])
    rir, frs = rc.room_correction(e, str(tmp_path), specific_limit=400)
    assert sorted(frs) == sorted(speakers) and all(sorted(frs[sp]) == ["left", "right"] for sp in speakers)
    np.testing.assert_array_equal(frs["FL"]["left"].frequency, common)
    # oracle composition of the same: what the file holds (PCM_32) -> columns -> estimate -> crops -> curves
    from impulse_hip.audio_io import pcm_quantise
    o_irs = {}
    for k, sd in enumerate(("left", "right")):
        stored = pcm_quantise(room_tracks[k], 32) / 2.0 ** 31
        for sp, side, col in ohrir.split_recording(stored[None, :], speakers, N, fs, side=sd):
            o_irs.setdefault(sp, {})[side] = oe.estimate(col)
    # room_correction crops every response at ITS OWN peak - 1 ms (core/room_correction.py:160-163), not pair-wise
    o_irs = {sp: {sd: oir.crop_head(o_irs[sp][sd], fs) for sd in ("left", "right")} for sp in speakers}
    o_tail, o_irs = ohrir.crop_tails(o_irs, fs, N, oe.n_octaves)
    assert o_tail == len(rir.irs["FL"]["left"].data)
    flat = np.zeros(len(common))
    ref_gain, worst_db = None, 0.0
    for sd in ("left", "right"):                                     # the file of the left ears is met first: FL-left sets the level
        for sp in speakers:
            _, o_raw, o_err, g_ = ofr.specific_room_correction(o_irs[sp][sd], fs, flat, None, limit=400, reference_gain=ref_gain)
            if ref_gain is None:
                ref_gain = g_
            fr = frs[sp][sd]
            # dB curves from fp32 responses: 1e-6 of the spectrum peak is ~1e-3 dB on bins 40-60 dB below it
            worst_db = max(worst_db, float(np.max(np.abs(fr.raw - o_raw))), float(np.max(np.abs(fr.error - o_err))))
            assert np.all(fr.error[common > 400] == 0.0)             # the limit mask
    assert worst_db < 5e-3, worst_db
    assert max(float(np.max(np.abs(frs[sp][sd].error))) for sp, sd in order26) > 1.0    # real corrections, not zeros

    # ---- (2) the measurement batch, equalised with THOSE curves
    tracks = _c3_tracks(e, np.random.default_rng(0xC3), speakers, rt60=0.22, level=0.4)
    errs = {(sp, sd): np.array(frs[sp][sd].error, dtype=np.float64) for sp, sd in order26}
    room = {sp: {sd: FrequencyResponse("r", frequency=common.copy(), raw=0, error=errs[(sp, sd)].copy()) for sd in ("left", "right")}
            for sp in speakers}
    stages = {}
    hrir, gain = run_slice(e, [((fs, tracks), speakers)], room_frs=room, decay=0.12, stages=stages)

    # oracle composition
    jobs = ohrir.split_recording(tracks, speakers, N, fs)
    irs = {}
    for sp, sd, col in jobs:
        irs.setdefault(sp, {})[sd] = oe.estimate(col)
    order = [(sp, sd) for sp, sd, _ in jobs]
    assert len(order) == 26
    for sp, sd in order:
        assert oir.peak_index(irs[sp][sd]) == oir.peak_index(stages["ingest"][(sp, sd)].astype(np.float64))
    irs = ohrir.crop_heads(irs, fs, head_ms=1)
    assert [len(irs[sp][sd]) for sp, sd in order] == [len(stages["crop_heads"][k]) for k in order]
    tail_ind, irs = ohrir.crop_tails(irs, fs, N, oe.n_octaves)
    assert tail_ind == len(stages["crop_tails"][order[0]])
    for sp, sd in order:
        assert rel(stages["crop_tails"][(sp, sd)], irs[sp][sd]) <= TIME_TOL
    for sp, sd in order:
        eq = ofr.equalization_worker_curve(common, errs[(sp, sd)], 0.0, fs)
        fir = omin.minimum_phase_impulse_response(common, eq, fs, f_res=5, normalize=False)
        assert len(fir) == 19200
        irs[sp][sd] = fft_convolve(irs[sp][sd], fir, "full")
    for sp, sd in order:
        assert rel(stages["equalize"][(sp, sd)], irs[sp][sd]) <= TIME_TOL
    # The decay stage is discontinuous in its input: the Lundeby knee of an equalised response sits in its
    # noise floor and moves by a whole analysis window (~5 ms) under a 2e-7 perturbation, which relocates
    # where the tail is cut (measured: 13 of 26 knees differ between fp64 and fp32-rounded inputs).  Each
    # remaining stage is therefore checked on the product's own previous stage: same input, same answer.
    from impulse_hip import decay as pdecay
    adjusted = 0
    decayed = {}
    for sp, sd in order:
        a = stages["equalize"][(sp, sd)].astype(np.float64)
        want = odecay.decay_adjustment_params(a, fs, 0.12)
        got = pdecay.decay_adjustment_params(a, fs, 0.12)
        assert (want is None) == (got is None)
        if want is not None:
            adjusted += 1
            assert tuple(int(v) for v in got[:3]) == tuple(int(v) for v in want[:3]), (sp, sd)   # K7 means are bit-exact
            assert got[3] == pytest.approx(want[3], rel=1e-9)
            a = odecay.apply_decay_window(a.copy(), want)
        decayed.setdefault(sp, {})[sd] = a
        assert rel(stages["decay"][(sp, sd)], a) <= 2e-7                  # fp32 store of the windowed response
    assert adjusted >= 20                                                 # the window really is exercised
    g = ohrir.normalization_gain_db(decayed, fs, peak_target=-0.1)
    assert gain == pytest.approx(g, abs=1e-5)
    for sp, sd in order:
        want = decayed[sp][sd] * 10 ** (g / 20)
        got = hrir.irs[sp][sd].data
        assert got.shape == want.shape
        assert rel(got, want) <= 1e-6, (sp, sd)


@pytest.mark.parametrize("L,M,mode", [(3_000_000, 5, "same"), (2_300_000, 700_001, "same"), (2_600_000, 1_500_000, "full"),
                                      (100, 2_500_000, "same"), (1_200_000, 1_200_000, "full")])
def test_overlap_add_beyond_one_transform(gpu_ctx, L, M, mode):
    """Convolutions longer than one two-level transform (2^21 points) run as overlap-add pieces through the
    same three passes (input blocks x filter partitions, accumulated on the device) - the reference's
    scipy.signal.convolve has no length limit."""
    from impulse_hip import ConvPlan
    from impulse_hip._native import plan_geometry
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(L % 1000 + M % 1000)
    x = rng.standard_normal((2, L)).astype(np.float32)
    h = rng.standard_normal(M) * np.exp(-np.arange(M) / max(M / 6.0, 1.0))
    assert plan_geometry(M, L, mode)[0] == 1 << 21
    plan = ConvPlan(gpu_ctx, h, L, mode)
    y = plan.execute(x)
    yi = plan.execute_interleaved(np.ascontiguousarray(x.T)) if L < 2_500_000 else None
    plan.close()
    for b in range(2):
        ref = fft_convolve(x[b].astype(np.float64), h, mode)
        assert y[b].shape == ref.shape
        assert rel(y[b], ref) <= 2 * TIME_TOL, (L, M, mode)           # sums of up to four fp32 pieces
    if yi is not None:
        assert np.array_equal(yi, y)


def test_k11_sosfilt_bits_and_virtual_bass(gpu_ctx, golden):
    """K11 (imp_sosfilt) reproduces scipy.signal.sosfilt bit for bit (golden outputs from the reference run,
    SciPy itself on ragged rows and long cascades), and the virtual-bass synthesis built on it matches the
    reference's synthesize_virtual_bass on the seeded FL/FR/FC set."""
    from scipy import signal
    from make_goldens import balance_inputs
    from impulse_hip.impulse_response import ImpulseResponse
    from impulse_hip.virtual_bass import synthesize_virtual_bass
    g = golden("virtual_bass")
    x = g["sosfilt_in"]
    got = gpu_ctx.sosfilt(g["sos_hp8_250"], list(x))
    np.testing.assert_array_equal(np.stack(got), g["sosfilt_hp8"])
    got = gpu_ctx.sosfilt(g["sos_lp8_250"], gpu_ctx.sosfilt(g["sos_hp4_15"], list(x)))
    np.testing.assert_array_equal(np.stack(got), g["sosfilt_lp8_of_hp4"])
    rng = np.random.default_rng(11)
    rows = [rng.standard_normal(n) for n in (1, 63, 64, 65, 1000, 33000)]
    for order in (2, 6, 22):                                        # 1, 3 and 11 sections (two launches)
        sos = signal.butter(order, 0.2, btype="low", output="sos")
        got = gpu_ctx.sosfilt(sos, rows)
        for r, y in zip(rows, got):
            np.testing.assert_array_equal(y, signal.sosfilt(sos, r))
    with pytest.raises(Exception):
        gpu_ctx.sosfilt(np.array([[1.0, 0, 0, 2.0, 0, 0]]), rows[:1])       # a0 != 1
    for name, kw in (("default", {}), ("inverted_300", dict(crossover_freq=300, invert_polarity=True, head_ms=1.5))):
        irs = {sp: {sd: ImpulseResponse(v.astype(np.float64), 48000) for sd, v in pair.items()}
               for sp, pair in balance_inputs(4096).items()}
        synthesize_virtual_bass(irs, 48000, **kw)
        for sp in irs:
            for sd in ("left", "right"):
                want = g[f"vb_{name}_{sp}_{sd}"]
                np.testing.assert_allclose(irs[sp][sd].data, want, rtol=0, atol=1e-13 * np.max(np.abs(want)))


def test_workspace_after_each_pass_matches_the_dataflow_model(gpu_ctx):
    """imp_plan_debug_run_stage: the workspace after pass A and after pass B against the NumPy dataflow model
    (tests/model/fourstep_model.py, the thread-for-thread algebra of the kernels) at N1 = 48 (16 rows per
    thread, radix 3), N1 = 66 (11 rows per thread, radix 11 x 6: the C2 plan) and N1 = 72 (8 rows per thread, radix 9)."""
    import fourstep_model as fm
    from impulse_hip import ConvPlan
    rng = np.random.default_rng(48)
    for L, M, n1 in ((300000, 150000, 48), (391270, 295270, 66), (420000, 295270, 72)):
        x = rng.standard_normal((2, L)).astype(np.float32)
        h = rng.standard_normal(M) * np.exp(-np.arange(M) / 30000.0)
        plan = ConvPlan(gpu_ctx, h, L, "same")
        assert plan.n1 == n1
        ws_a = plan.debug_stage(x, 0)
        ws_b = plan.debug_stage(x, 1)
        plan.close()
        alpha, beta = fm.plan_alpha_beta(h, plan.nfft)
        for b in range(2):
            want_a = fm.pass_a(x[b].astype(np.float64), plan.nfft)
            assert np.max(np.abs(ws_a[b] - want_a)) / np.max(np.abs(want_a)) <= 2e-6
            want_b = fm.pass_b(want_a, alpha, beta)
            assert np.max(np.abs(ws_b[b] - want_b)) / np.max(np.abs(want_b)) <= 3e-6


def _bench_line(res):
    import json
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract_line(gpu_ctx):
    """bench.py prints exactly one JSON line on stdout with the contract's keys, BASELINE.json's metric, a
    roofline block and (at N = 1) a cpu_baseline block; everything else goes to stderr.  `value` times the metric's own
    wording - the deconvolution + FIR chain - and K1 alone is given beside it (`deconv_only`).  A step is one pass over all
    resident measurements (10 240 IRs), so the contract's 20 steps time about half a second."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    d = _bench_line(res)
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "IR/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["higher_is_better"]) == (1, 10, 2, "weak", True)
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["ms_per_step"] * d["steps"] >= 100.0 and abs(d["timed_region_s"] * 1e3 - d["ms_per_step"] * 10) < 1e-6
    assert abs(d["value"] - d["irs_per_step"] * d["steps"] / d["timed_region_s"]) <= 1e-6 * d["value"]
    cfg = d["config"]
    assert cfg["channels_per_gpu_per_measurement"] == 16 and cfg["measurements_per_call"] == 2
    assert cfg["channels_per_launch_group"] == 32 and d["irs_per_step"] == 16 * cfg["measurements_per_step"] == 10240
    assert "FIR" in cfg["stage"] and cfg["fir_taps"] == 9600 and cfg["crop_samples"] == 32640
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert 0 < r["path_frac"] < r["deconv_only_path_frac"] < 1 and 0 < r["isolated"]["frac"] < 1
    # the counters behind `traffic` are collected by this very run (two rocprofv3 --pmc child runs); the row pass moves
    # its workspace once each way plus alpha/beta: between 1.2 and 1.7 times the algorithmic bytes of a launch group
    assert r["traffic_source"] == "live", r["traffic_note"]
    assert 1.2 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.7
    assert r["l2_fabric_traffic"]["peak_search_plus_fused_k5_bytes_per_call"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] > 1 and c["value"] > 0 and c["sample"] and "FIR" in c["work"]
    assert c["serial"]["cores"] == 1 and 0 < c["serial"]["value"] < c["value"]
    p = d["parity"]
    assert p["peak_indices_exact"] and p["spectrum_max_rel_err"] <= 1e-6 and p["chain_time_max_rel_err"] <= 1e-6
    assert p["real_demo_column"]["peak_index_equal"]
    assert d["value"] > 100 * c["value"]
    k1 = d["deconv_only"]
    assert k1["value"] > d["value"] and 0 < k1["path_frac"] < 1
    other = k1["other_plan"]
    assert k1["plan"] == "mono" and other["plan"] == "pair" and other["peak_indices_exact"] and other["rows"] == 132
    assert other["max_rel_diff_vs_headline_plan"] < 2e-6
    # the function's own output, the whole column: gated against the fp32 floor measured on the same channels
    assert p["whole_column_spectrum_max_rel_err"] <= p["whole_column_gate"] <= 1.25 * max(p["whole_column_pocketfft_fp32_err"], 3e-6)
    # the reference's real stage order, device resident (imp_slice): bit-identical to the staged class path, nothing flagged
    sr = d["slice_resident"]
    assert sr["bit_identical_to_staged_path"] and sr["no_measurement_flagged"] and sr["value"] > 100e3
    assert sr["l2_fabric_traffic"]["source"] == "live" and sr["l2_fabric_traffic"]["over_algorithmic"] > 1
    assert "ceiling_frac" in r and 0 < r["ceiling_frac"] < 0.4
    assert d["slice"]["identical_to_staged_path"] and d["slice"]["value"] > 0
    assert isinstance(cfg["environment_switches"], dict)


def test_bench_two_ranks_started_plainly(gpu_ctx):
    """`python bench.py --gpus 2 ...` invoked PLAINLY (no torchrun, no WORLD_SIZE) starts its own two ranks as a child
    before touching the GPU, and rank 0 prints one line with the weak C2 figure and the strong_c5 block.  On this
    one-GPU box the ranks are folded onto device 0 and rendezvous over gloo (IMPULSE_BENCH_BACKEND=gloo)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["IMPULSE_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--blocks", "24", "--strong-channels", "128", "--strong-passes", "2"],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    d = _bench_line(res)
    assert (d["n_gpus"], d["ranks_seen"], d["scaling"]) == (2, 2, "weak") and d["value"] > 0
    assert d["parity"]["peak_indices_exact"]
    s = d["strong_c5"]
    assert (s["channels"], s["n_gpus"], s["ranks_seen"], s["channels_per_rank"], s["scaling"]) == (128, 2, 2, 64, "strong")
    assert s["value"] > 0 and s["single_gpu_same_run"]["value"] > 0 and s["speedup_vs_single_gpu"] > 0
    assert s["peaks_exact_and_tiles_bit_equal"] and s["broadcast_bytes"] > 0


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] / configs[4] at their batch size: 256 and 1024 device-resident channels of 2^20 samples,
# launch groups of 8 round robin on 3 lanes (32 / 128 groups)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [256, 1024])
def test_full_batch_c4_c5_device_resident(gpu_ctx, B):
    import bench
    from impulse_hip import ConvPlan
    from oracle.estimator import estimate as oracle_estimate
    est = bench.make_estimator("c5")
    L = M = len(est)
    assert L == 1 << 20
    T = 16                                                        # distinct recordings, tiled over the batch
    base, L_, pitch, delays = bench.synth_recordings(est, T, seed0=0xC5, column=L)
    assert L_ == L and pitch == L
    ctx = gpu_ctx
    d_x = ctx.malloc(B * pitch * 4)
    d_y = ctx.malloc(B * pitch * 4)
    plan = None
    try:
        ctx.h2d(d_x, base)
        have = T
        while have < B:                                           # tile on the device
            n = min(have, B - have)
            ctx.d2d(d_x + have * pitch * 4, d_x, n * pitch * 4)
            have += n
        ctx.memset(d_y, 0xFF, B * pitch * 4)                      # NaN pattern: every sample must be overwritten
        plan = ConvPlan(ctx, est.inverse_filter, L, "same", ws_channels=24)
        plan.set_overlap(3)
        assert plan.nfft == 3 << 19 and plan.ws_channels // 3 == 8
        ctx.synchronize()                                         # inputs complete before the lanes start
        plan.execute_device(d_x, B, pitch, d_y, pitch)
        ctx.synchronize()
        y = np.empty((B, pitch), dtype=np.float32)
        ctx.d2h(y, d_y)
    finally:
        if plan is not None:
            plan.close()
        ctx.free(d_x)
        ctx.free(d_y)
    assert np.all(np.isfinite(y))
    # every tile bit-equals its twin, whatever launch group and lane it ran in
    first = y[:T]
    for t0 in range(T, B, T):
        assert np.array_equal(y[t0:t0 + T], first), t0
    # analytic truth: a sweep starting at offset d deconvolves to a peak at (M-1) - (M-1)//2 + d = M//2 + d
    for c in range(T):
        assert int(np.argmax(np.abs(first[c]))) == M // 2 + delays[c]
    # a sample of channels against the oracle (float64 NumPy restatement of estimate())
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    for c in (0, 7, 15):
        ref = oracle_estimate(base[c, :L].astype(np.float64), inv)
        assert rel(first[c], ref) <= TIME_TOL
        assert spec_rel_cropped(first[c], ref) <= SPEC_TOL
        from oracle.impulse_response import peak_index
        assert gpu_ctx.peak_index([first[c]])[0][0] == peak_index(ref)


# ------------------------------------------------------------------------------------------------
# Round-2 fixtures (tests/golden/round2.npz, reference run): class-level entry points that had no test
# ------------------------------------------------------------------------------------------------
def test_from_wav_estimators_deconvolve_their_own_sweep(gpu_ctx, golden, tmp_path):
    import round2_inputs as r2
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    g = golden("round2")
    e1 = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    for name, sig in (("offgrid", r2.off_grid_sweep()), ("perturbed", r2.perturbed(e1.test_signal, 3e-4))):
        path = str(tmp_path / (name + ".wav"))
        r2.write_pcm32(path, 48000, sig)
        ire = ImpulseResponseEstimator.from_wav(path)
        imp = np.zeros(len(ire) + 2 * 48000)
        imp[100: 100 + len(ire)] = ire.test_signal
        y = ire.estimate(imp)
        pk = int(np.argmax(np.abs(y)))
        assert pk == int(g[f"fw_{name}_selfpeak"]) == 100 + len(ire) // 2
        assert y[pk] == pytest.approx(float(g[f"fw_{name}_selfpeak_value"]), abs=2e-6)


def test_impulse_response_convolve_and_adjust_decay_through_the_class(gpu_ctx, golden):
    import round2_inputs as r2
    from impulse_hip.impulse_response import ImpulseResponse
    g = golden("round2")
    d = r2.decaying_ir(0x1111)
    x = np.random.default_rng(0x2222).standard_normal(1500).astype(np.float32).astype(np.float64)
    y = ImpulseResponse(d.copy(), 48000).convolve(x)
    assert y.shape == g["conv_y"].shape and rel(y, g["conv_y"]) <= TIME_TOL
    for tgt in (0.2, 0.12):
        ir = ImpulseResponse(d.copy(), 48000)
        ir.adjust_decay(tgt)                                      # in place
        assert rel(ir.data, g[f"adj_{tgt}"]) <= 1e-7
    ir = ImpulseResponse(d.copy(), 48000)
    ir.adjust_decay(5.0)                                          # slower than measured: untouched
    assert np.array_equal(ir.data, d) and bool(g["adj_noop_equal"])


def test_hrir_equalize_two_row_fir_and_write_wav_orders(gpu_ctx, golden, tmp_path):
    import round2_inputs as r2
    from impulse_hip.audio_io import pcm_quantise
    from impulse_hip.constants import HESUVI_TRACK_ORDER, HEXADECAGONAL_TRACK_ORDER
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    from scipy.io import wavfile
    g = golden("round2")

    class Est:
        fs = 48000

    base, firs = r2.hrir_set(), r2.fir_pair()

    def fresh():
        h = HRIR(Est())
        h.irs = {sp: {sd: ImpulseResponse(v.copy(), 48000) for sd, v in pair.items()} for sp, pair in base.items()}
        return h

    for name, arg in (("two_rows", firs), ("one_row", firs[:1]), ("flat", firs[1]),
                      ("ir_list", [ImpulseResponse(firs[0].copy(), 48000), ImpulseResponse(firs[1].copy(), 48000)]),
                      ("array_list", [firs[0].copy(), firs[1].copy()])):
        h = fresh()
        h.equalize(arg)
        for sp in ("FL", "SR"):
            for sd in ("left", "right"):
                want = g[f"heq_{name}_{sp}_{sd}"]
                assert h.irs[sp][sd].data.shape == want.shape and rel(h.irs[sp][sd].data, want) <= TIME_TOL
    h = fresh()
    for name, order, bits in (("hesuvi", HESUVI_TRACK_ORDER, 32), ("hexa", None, 24), ("hexa16", HEXADECAGONAL_TRACK_ORDER, 16)):
        path = str(tmp_path / (name + ".wav"))
        h.write_wav(path, track_order=order, bit_depth=bits)
        n_frames, n_tracks = (int(v) for v in g[f"ww_{name}_shape"])
        assert str(g[f"ww_{name}_subtype"]) == f"PCM_{bits}" and int(g[f"ww_{name}_rate"]) == 48000
        if bits == 24:
            from impulse_hip.audio_io import read_wav
            fs, data = read_wav(path)
            ints = np.rint(data.T * 2.0 ** 23).astype(np.int64)
        else:
            fs, ints = wavfile.read(path)
            ints = ints.astype(np.int64)
        assert fs == 48000 and ints.shape == (n_frames, n_tracks)
        # track order + samples: the first 64 frames of the matrix the reference handed to soundfile, quantised
        assert np.array_equal(ints[:64], pcm_quantise(g[f"ww_{name}_head"], bits))
        # the whole file: the oracle's frame matrix (pinned to the reference run in test_oracle_golden.py) through the
        # float -> PCM conversion that the reference's own sweep WAVs pin for PCM_32 (scale 2^31, saturating; 16 / 24 bit keep
        # the top bits: unpinned) - a direct sound a hair above 1.0 saturates at full scale
        from oracle import hrir as ohrir
        assert np.array_equal(ints, ohrir.pcm_quantise(ohrir.write_wav_frames(base, order), bits))


@pytest.mark.parametrize("name,kw,with_generic", [("avg", dict(fr_combination_method="average"), True),
                                                  ("cons", dict(fr_combination_method="conservative", specific_limit=600,
                                                                generic_limit=500), True),
                                                  ("specific_only", dict(), False)])
def test_room_correction_top_level_with_generic_measurements(gpu_ctx, golden, tmp_path, monkeypatch, name, kw, with_generic):
    """room_correction() on a folder with one file per ear (FL,FR) and a three-position generic room.wav, both
    combination methods, against the reference's run of the same folder (core/room_correction.py:78-182, 231-292)."""
    import round2_inputs as r2
    from impulse_hip import room_correction as rc
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    g = golden("round2")
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    r2.room_folder(str(tmp_path), e.test_signal, with_generic=with_generic)
    # the reference levels every channel to the first one it meets, in os.listdir order: replay the golden run's order
    wanted = [str(s) for s in g[f"rc_{name}_listdir"]]
    real_listdir = os.listdir
    monkeypatch.setattr(rc.os, "listdir", lambda p: wanted + [f for f in real_listdir(p) if f not in wanted])
    rir, frs = rc.room_correction(e, str(tmp_path), **kw)
    assert sorted(frs.keys()) == sorted(str(s) for s in g[f"rc_{name}_speakers"])
    assert [f"{sp}-{sd}" for sp, pair in rir.irs.items() for sd in pair] == [str(s) for s in g[f"rc_{name}_order"]]
    assert len(rir.irs["FL"]["left"].data) == int(g[f"rc_{name}_rir_len"])
    assert os.path.isfile(tmp_path / "room-responses.wav")
    np.testing.assert_array_equal(frs["FL"]["left"].frequency, g[f"rc_{name}_freq"])
    for sp in ("FL", "FR") + (("FC", "BL") if with_generic else ()):
        for sd in ("left", "right"):
            fr = frs[sp][sd]
            # dB curves from fp32 IRs: 1e-6 of the spectrum peak is ~1e-3 dB on bins 40-60 dB below it
            assert np.max(np.abs(fr.raw - g[f"rc_{name}_{sp}_{sd}_raw"])) < 5e-3
            assert np.max(np.abs(fr.error - g[f"rc_{name}_{sp}_{sd}_error"])) < 5e-3
            if f"rc_{name}_{sp}_{sd}_error_smoothed" in g.files:
                assert np.max(np.abs(fr.error_smoothed - g[f"rc_{name}_{sp}_{sd}_error_smoothed"])) < 5e-3
    if with_generic:
        assert frs["FC"]["left"] is not frs["FC"]["right"]
        assert np.array_equal(frs["FC"]["left"].error, frs["TBR"]["right"].error)


def test_leaked_plan_does_not_abort_interpreter_exit(tmp_path):
    """A process that exits with live plans / contexts (never closed, kept alive by a global) must end with rc 0: the
    atexit hook of impulse_hip._native releases them before the HIP runtime's own teardown (ROCm 7.2 aborts with
    std::bad_variant_access otherwise - the failure commit 8be20fa fixed)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "leak.py"
    script.write_text(f'''
import sys
sys.path.insert(0, {os.path.join(root, "impulcifer-pip313_amd")!r})
import numpy as np
from impulse_hip import Context, ConvPlan
KEEP = []
ctx = Context(0)
plan = ConvPlan(ctx, np.ones(100), 5000, "same")
y = plan.execute(np.ones((2, 5000), np.float32))
KEEP.extend([ctx, plan, y])
other = Context(0)
KEEP.append(ConvPlan(other, np.ones(7), 300, "full"))
print("leaking", len(KEEP), float(y[0, 2500]))
''')
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stderr[-1500:])
    assert "leaking 4 100.0" in res.stdout


def test_k5_short_plans_and_filter_refill(gpu_ctx):
    """K5 at its everyday size on the THREE-LAUNCH path (fused=False; the fused one-launch plan the classes use by default is
    tests/test_fused_fir.py): a 0.68 s response (*) a 9 600-tap FIR is 42 239 samples -> a 65 536-point transform (8 rows);
    per-channel FIRs refilled in place (imp_plan_set_filters) give the bits of a fresh plan; the class-level entry points
    reuse one cached plan per shape."""
    from impulse_hip import ConvPlan
    from impulse_hip import impulse_response as irm
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(55)
    n, k, B = 32640, 9600, 6
    x = rng.standard_normal((B, n)).astype(np.float32)
    fa = rng.standard_normal((B, k)) * np.exp(-np.arange(k) / 700.0)
    fb = rng.standard_normal((B, k)) * np.exp(-np.arange(k) / 300.0)
    plan = ConvPlan(gpu_ctx, fa, n, "full", ws_channels=B, fused=False)
    assert (plan.n1, plan.nfft) == (8, 65536) and not plan.fused
    ya = plan.execute(x)
    plan.set_filters(fb)
    yb = plan.execute(x)
    plan.set_filters(fa)
    assert np.array_equal(plan.execute(x), ya)
    plan.close()
    fresh = ConvPlan(gpu_ctx, fb, n, "full", ws_channels=B, fused=False)
    assert np.array_equal(fresh.execute(x), yb)
    fresh.close()
    for b in (0, B - 1):
        assert rel(ya[b], fft_convolve(x[b].astype(np.float64), fa[b], "full")) <= TIME_TOL
        assert rel(yb[b], fft_convolve(x[b].astype(np.float64), fb[b], "full")) <= TIME_TOL
    small = ConvPlan(gpu_ctx, fa[0, :3000], 20000, "full", fused=False)
    assert (small.n1, small.nfft) == (4, 32768) and not small.fused
    assert rel(small.execute(x[0, :20000]), fft_convolve(x[0, :20000].astype(np.float64), fa[0, :3000], "full")) <= TIME_TOL
    with pytest.raises(ValueError):
        small.set_filters(fa[0])
    small.close()
    # class-level calls: same shape, new taps every call -> one plan, refilled
    irm._k5_plans.plans.clear()
    for i in range(3):
        y = irm.fir_convolve_full(x[i].astype(np.float64), fa[i])
        assert rel(y, fft_convolve(x[i].astype(np.float64), fa[i], "full")) <= TIME_TOL
    assert len(irm._k5_plans.plans) == 1
    ys = irm.fir_convolve_full_batch([x[i].astype(np.float64) for i in range(B)], [fb[i] for i in range(B)])
    assert len(irm._k5_plans.plans) == 2 and all(p.fused for p in irm._k5_plans.plans.values())
    assert all(rel(ys[i], yb[i].astype(np.float64)) <= 5e-7 for i in range(B))     # fused blocks vs one 8-row transform


# ------------------------------------------------------------------------------------------------
# K12: curve conditioning on the device (csrc/curves.hip) - SURVEY 8(f)-3
# ------------------------------------------------------------------------------------------------
CURVE_TOL = 1e-12          # dB, against the reference's own curves / the oracle's restatement of SciPy


def test_k12_equalization_curves_match_reference_run(gpu_ctx, golden):
    """process_equalization_worker's curve (autoeq smoothen_heavy_light + equalize) for the six reference curves of
    minphase.npz (three shapes x two sampling rates), through the class surface (B = 1 calls) and as one batch."""
    from impulse_hip.frequency_response import FrequencyResponse, equalization_curves
    from impulse_hip.parallel_workers import equalization_curve
    mp = golden("minphase")
    worst = 0.0
    for fs in (48000, 96000):
        freq = mp[f"fs{fs}_freq"]
        flat = FrequencyResponse("t", frequency=freq.copy(), raw=0)
        names = ("flat", "wavy", "tilt")
        for nm in names:
            room = {"FL": {"left": FrequencyResponse("r", frequency=freq.copy(), raw=0, error=mp[f"fs{fs}_{nm}_error"])}}
            cur = equalization_curve("FL", "left", room, None, None, None, None, flat, freq, fs)
            worst = max(worst, float(np.max(np.abs(cur.equalization - mp[f"fs{fs}_{nm}_eq"]))))
            assert len(cur.error_smoothed) == len(freq) and len(cur.smoothed) == len(freq)
        _, eqs = equalization_curves(freq, np.stack([mp[f"fs{fs}_{nm}_error"] for nm in names]), smoothen_first=True,
                                     max_gain=40, treble_f_lower=10000, treble_f_upper=fs / 2)
        for nm, eq in zip(names, eqs):
            worst = max(worst, float(np.max(np.abs(eq - mp[f"fs{fs}_{nm}_eq"]))))
    print(f"K12 equalization curves vs the reference run: max |d| = {worst:.2e} dB")
    assert worst <= CURVE_TOL


def test_k12_smoothing_clipping_and_fir_grid_match_the_oracle(gpu_ctx):
    from impulse_hip.frequency_response import (curves_for, equalization_curves, fir_design_gain, generate_frequencies,
                                                smooth_curves)
    from oracle import frequency_response as ofr
    from oracle.impulse_response import interpolate_log
    from oracle.scipy_restated import next_fast_len_real
    rng = np.random.default_rng(12)
    for fs in (44100, 48000, 96000):
        freq = generate_frequencies(10, fs / 2, 1.01)
        n = len(freq)
        curves = np.cumsum(rng.standard_normal((5, n)), axis=1) * 0.4                      # wandering dB curves
        h = curves_for(freq)
        for octv in (1 / 12, 1 / 6, 1 / 3, 1.3, 2):
            assert h.window_size(octv) == ofr.window_size(freq, octv)
        for (wn, wt, fl, fu) in ((1 / 3, 1 / 3, 100, 10000), (1 / 6, 1 / 3, 100, 10000), (1 / 3, 1.3, 1000, 6000),
                                 (2, 1 / 3, 20000, int(round(fs / 2))), (1 / 6, 1 / 6, 100, 10000)):
            got = smooth_curves(freq, curves, wn, wt, fl, fu)
            for b in range(len(curves)):
                want = ofr.smoothen_fractional_octave(freq, curves[b], wn, wt, fl, fu)
                assert np.max(np.abs(got[b] - want)) <= CURVE_TOL
        # deep notches ask for more than the 40 dB limit: clipping, kink removal and the quadratic-spline bridge
        errors = curves.copy()
        errors[0, 300:340] -= 65.0
        errors[1, 5:40] -= 70.0                                   # a kink window that reaches the first grid point
        errors[2, n - 60:n - 20] -= 80.0
        errors[3] -= 45.0 + 10 * np.sin(np.arange(n) / 37.0)      # in and out of clipping many times
        for smoothen_first in (True, False):
            es, eq, used = h.equalization(errors, smoothen_first, 40, 10000, fs / 2, 6.0, 1.0)
            assert used[:4].all() and not used[4]
            for b in range(len(errors)):
                want_es = ofr.smoothen_heavy_light(freq, errors[b]) if smoothen_first else errors[b]
                assert np.max(np.abs(es[b] - want_es)) <= CURVE_TOL
                want = ofr.equalize(freq, want_es, 40, 10000, fs / 2)
                assert np.max(np.abs(eq[b] - want)) <= 1e-10, (fs, b, smoothen_first)
        _, eq_lim = equalization_curves(freq, errors, smoothen_first=False, max_gain=15, treble_f_lower=2000,
                                        treble_f_upper=fs / 2, smoothen=False)
        assert np.max(np.abs(eq_lim - np.minimum(-errors, ofr.sigmoid(freq, 2000, fs / 2, 15, 6.0)))) <= CURVE_TOL
        # the gain grid handed to firwin2 (autoeq :651-674), with and without normalisation
        for f_res, normalize in ((5, False), (10, True)):
            got = fir_design_gain(freq, curves[:2], fs, f_res, normalize)
            ntaps = next_fast_len_real(round(fs // 2 / (f_res / 2)))
            assert got.shape == (2, ntaps) and h.fir_taps(fs, f_res) == ntaps
            f = np.linspace(0.0, fs // 2, ntaps)
            for b in range(2):
                f_min = max(freq[0], f_res / 2)
                raw = interpolate_log(freq, curves[b], f)
                raw[f <= f_min] = interpolate_log(freq, curves[b], np.array([f_min]))[0]
                if normalize:
                    raw -= np.max(raw)
                    raw -= 0.5
                want = 10 ** (raw * 2 / 20)
                want[-1] = 0.0
                assert np.max(np.abs(got[b] - want) / np.maximum(want, 1e-300)) <= 1e-12
    with pytest.raises(Exception, match="NaN"):
        smooth_curves(freq, np.full(len(freq), np.nan))


def test_k12_worker_batch_equals_per_channel_calls_and_reference_firs(gpu_ctx, golden):
    """error -> FIR in one device chain for a whole measurement = the per-channel worker calls = the reference's FIRs."""
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.parallel_workers import (init_equalization_worker, process_equalization_batch,
                                              process_equalization_worker)
    mp = golden("minphase")
    for fs in (48000, 96000):
        freq = mp[f"fs{fs}_freq"]
        names = ("flat", "wavy", "tilt")
        room = {nm.upper()[:2]: {"left": FrequencyResponse("r", frequency=freq.copy(), raw=0, error=mp[f"fs{fs}_{nm}_error"])}
                for nm in names}
        flat = FrequencyResponse("t", frequency=freq.copy(), raw=0)
        tasks = [(nm.upper()[:2], "left") for nm in names]
        batch = process_equalization_batch(tasks, room, None, None, None, None, flat, freq, fs)
        init_equalization_worker(room, None, None, None, None, flat, freq, fs)
        for (sp, sd, fir), nm in zip(batch, names):
            want = mp[f"fs{fs}_{nm}_fir"]
            assert fir.shape == want.shape
            assert np.max(np.abs(fir - want)) <= 5e-8 * np.max(np.abs(want))
            sp1, sd1, fir1 = process_equalization_worker((sp, sd))
            assert (sp1, sd1) == (sp, sd) and np.array_equal(fir1, fir)


# ------------------------------------------------------------------------------------------------
# Device-resident responses (impulse_hip/device_rows.py): the stages between ingest and output on rows that stay on the GPU
# ------------------------------------------------------------------------------------------------
def test_device_resident_stages_equal_the_host_array_stages(gpu_ctx, golden):
    """The same measurement through HRIR twice: once from PCM frames (responses stay on the device through crop_heads,
    crop_tails, equalize_channels, normalize) and once from float arrays (every stage on host arrays, uploading per
    call).  Lengths, crops and gains must agree exactly; samples to fp32 rounding of the final gain."""
    import slice_input
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.pipeline_slice import run_slice
    g = golden("pipeline_slice")
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    fs = 48000
    pcm = slice_input.to_pcm32(slice_input.make_tracks(e.test_signal, fs))            # [4, n]
    order = [("FL", "left"), ("FL", "right"), ("FR", "left"), ("FR", "right")]
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    room = {sp: {} for sp in ("FL", "FR")}
    for sp, sd in order:
        room[sp][sd] = FrequencyResponse("r", frequency=common.copy(), raw=0, error=g[f"room_error_{sp}_{sd}"])
    st_dev, st_host = {}, {}
    h_dev, gain_dev = run_slice(e, [((fs, np.ascontiguousarray(pcm.T)), ["FL", "FR"])], room_frs=room, stages=st_dev)
    assert all(ir._data is None and ir._row is not None for pair in h_dev.irs.values() for ir in pair.values())
    h_host, gain_host = run_slice(e, [((fs, pcm.astype(np.float64) / 2 ** 31), ["FL", "FR"])], room_frs=room, stages=st_host)
    assert all(ir._row is None for pair in h_host.irs.values() for ir in pair.values())
    assert gain_dev == pytest.approx(gain_host, abs=1e-9) and gain_dev == pytest.approx(float(g["norm_gain_db"]), abs=2e-4)
    for stage in ("ingest", "crop_heads", "crop_tails", "equalize", "normalize"):
        for k in order:
            a, b = st_dev[stage][k], st_host[stage][k]
            assert a.shape == b.shape, (stage, k)
            assert rel(a, b) <= 2e-7, (stage, k, rel(a, b))
    # reading .data hands the response over to the host for good; the raw column is cut from the PCM block on demand
    ir = h_dev.irs["FL"]["left"]
    d = ir.data
    assert ir._row is None and d.dtype == np.float64 and d.flags.writeable and ir.data is d
    assert rel(d, g["final_FL_left"]) <= 2e-6
    assert np.array_equal(ir.recording, pcm[0, 2 * fs: 2 * fs + len(e) + 2 * fs].astype(np.float64) / 2 ** 31)
    d *= 0.5                                                      # in-place mutation, as the reference's callers do
    assert np.array_equal(ir.data, d)
    import copy
    import pickle
    twin = pickle.loads(pickle.dumps(h_dev.irs["FR"]["right"]))
    assert np.array_equal(twin.data, h_dev.irs["FR"]["right"].peek()) and twin.fs == fs
    assert np.array_equal(copy.deepcopy(h_dev).irs["FR"]["left"].data, h_dev.irs["FR"]["left"].peek())


def test_slice_fir_design_on_the_second_stream_changes_nothing(gpu_ctx, golden, monkeypatch):
    """run_slice designs the equalisation FIRs on a worker thread and a context of its own while the recording uploads.
    The same slice with the early design's channel list made wrong on purpose (so that the FIRs are designed where the
    reference's stage order has them, on the default context) must agree bit for bit, stage by stage; the thread-local
    context override ends with its block; the early design ran on the auxiliary context.  (The early design leaves its FIRs
    on the device - the aux context's - and equalize_channels takes them there; the late one returns host arrays: same bits.)"""
    import slice_input
    from impulse_hip import _native, pipeline_slice
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    g = golden("pipeline_slice")
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    fs = 48000
    frames = np.ascontiguousarray(slice_input.to_pcm32(slice_input.make_tracks(e.test_signal, fs)).T)
    order = [("FL", "left"), ("FL", "right"), ("FR", "left"), ("FR", "right")]
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    room = {sp: {} for sp in ("FL", "FR")}
    for sp, sd in order:
        room[sp][sd] = FrequencyResponse("r", frequency=common.copy(), raw=0, error=g[f"room_error_{sp}_{sd}"])
    assert pipeline_slice._expected_tasks([((fs, frames), ["FL", "FR"])]) == order
    assert pipeline_slice._expected_tasks([("l.wav", ["FL", "FR"], "left"), ("r.wav", ["FL", "FR"], "right")]) == \
        [("FL", "left"), ("FR", "left"), ("FL", "right"), ("FR", "right")]
    seen = []
    real = pipeline_slice.process_equalization_batch
    monkeypatch.setattr(pipeline_slice, "process_equalization_batch",
                        lambda *a, **kw: (seen.append(_native.default_context()), real(*a, **kw))[1])
    st_early, st_late = {}, {}
    h_early, gain_early = pipeline_slice.run_slice(e, [((fs, frames), ["FL", "FR"])], room_frs=room, stages=st_early)
    assert seen == [_native.aux_context()] and _native.default_context() is not _native.aux_context()
    monkeypatch.setattr(pipeline_slice, "_expected_tasks", lambda recordings: [("FC", "left")])
    h_late, gain_late = pipeline_slice.run_slice(e, [((fs, frames), ["FL", "FR"])], room_frs=room, stages=st_late)
    assert seen[1] is _native.aux_context() and seen[2] is _native.default_context() and len(seen) == 3
    assert gain_early == gain_late
    for stage in ("ingest", "crop_heads", "crop_tails", "equalize", "normalize"):
        for k in order:
            assert np.array_equal(st_early[stage][k], st_late[stage][k]), (stage, k)
    with _native.using_context(gpu_ctx):
        assert _native.default_context() is gpu_ctx
        with _native.using_context(_native.aux_context()):
            assert _native.default_context() is _native.aux_context()
        assert _native.default_context() is gpu_ctx
    assert _native.default_context() is not gpu_ctx


def test_library_rccl_broadcast_single_rank(gpu_ctx, tmp_path):
    """imp_comm_*: the spectrum broadcast by the library itself over RCCL (one rank on this one-GPU box: communicator
    creation from a unique id shared through a file, in-place broadcast of the plan's spectrum, teardown)."""
    from impulse_hip import ConvPlan
    from impulse_hip._native import Comm, comm_unique_id
    from impulse_hip.sharding import broadcast_plan_spectrum_rccl
    h = np.random.default_rng(3).standard_normal(5000)
    plan = ConvPlan(gpu_ctx, h, 20000, "same")
    x = np.random.default_rng(4).standard_normal((2, 20000)).astype(np.float32)
    before = plan.execute(x)
    n = broadcast_plan_spectrum_rccl(plan, gpu_ctx, 0, 1, rendezvous_path=str(tmp_path / "uid"))
    assert not os.path.exists(tmp_path / "uid")
    assert n == plan.spectrum_buffer()[1] == plan.n1 * 4096 * 16
    assert np.array_equal(plan.execute(x), before)              # root's buffer is unchanged by its own broadcast
    uid = comm_unique_id()
    assert len(uid) == 128
    comm = Comm(gpu_ctx, uid, 0, 1)
    d = gpu_ctx.malloc(4096)
    gpu_ctx.h2d(d, np.arange(1024, dtype=np.float32))
    comm.broadcast(d, 4096, root=0)
    back = np.empty(1024, dtype=np.float32)
    gpu_ctx.d2h(back, d)
    gpu_ctx.free(d)
    comm.close()
    plan.close()
    assert np.array_equal(back, np.arange(1024, dtype=np.float32))


def test_headphone_compensation_matches_reference_run(gpu_ctx, golden, tmp_path):
    """headphone_compensation() (core/pipeline_stages.py:353-482) on a synthetic stereo headphone measurement: the
    two error curves the EQ worker adds to every channel, and the file resolution rules."""
    import round2_inputs as r2
    from impulse_hip.headphone_compensation import headphone_compensation, resolve_headphone_file
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    g = golden("round2")
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    r2.headphone_file(str(tmp_path / "headphones.wav"), e.test_signal)
    left, right = headphone_compensation(e, str(tmp_path))
    assert os.path.isfile(tmp_path / "headphone-responses.wav")
    np.testing.assert_array_equal(left.frequency, g["hp_freq"])
    for nm, fr in (("left", left), ("right", right)):
        assert np.max(np.abs(fr.raw - g[f"hp_{nm}_raw"])) < 5e-3         # dB curves from fp32 responses
        assert np.max(np.abs(fr.error - g[f"hp_{nm}_error"])) < 5e-3
        assert not np.any(fr.target)
    assert headphone_compensation(e, str(tmp_path / "nowhere")) == (None, None)
    # resolution rules: explicit relative file, directory with a fallback name, directory with some other WAV
    (tmp_path / "sub").mkdir()
    os.rename(tmp_path / "headphones.wav", tmp_path / "sub" / "hp.wav")
    assert resolve_headphone_file(str(tmp_path), "sub/hp.wav") == str(tmp_path / "sub" / "hp.wav")
    assert resolve_headphone_file(str(tmp_path), str(tmp_path / "sub")) == str(tmp_path / "sub" / "hp.wav")
    os.rename(tmp_path / "sub" / "hp.wav", tmp_path / "sub" / "other.WAV")
    assert resolve_headphone_file(str(tmp_path), str(tmp_path / "sub")) == str(tmp_path / "sub" / "other.WAV")
    assert resolve_headphone_file(str(tmp_path), "missing.wav") is None


@pytest.mark.parametrize("bits", [16, 24, 32])
def test_write_wav_from_device_rows_equals_host_codec(gpu_ctx, tmp_path, bits):
    """HRIR.write_wav of device-resident responses (ordering, silence for absent channels and the PCM conversion on the
    device) writes the same file, byte for byte, as the host codec on the same samples."""
    import slice_input
    from impulse_hip.constants import HESUVI_TRACK_ORDER
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    pcm = slice_input.to_pcm32(slice_input.make_tracks(e.test_signal, 48000))
    h = HRIR(e)
    h.open_recording_frames(48000, np.ascontiguousarray(pcm.T), ["FL", "FR"])
    h.crop_heads()
    h.crop_tails()
    assert h._device_rows([ir for pair in h.irs.values() for ir in pair.values()]) is not None
    # (the conversion itself is pinned against the reference's files in test_write_wav_sweep_sequence_from_device_rows...)
    twin = HRIR(e)
    twin.irs = {sp: {sd: ImpulseResponse(ir.peek(), 48000) for sd, ir in pair.items()} for sp, pair in h.irs.items()}
    for order, name in ((HESUVI_TRACK_ORDER, "hesuvi"), (None, "hexa")):
        a, b = str(tmp_path / f"dev_{name}.wav"), str(tmp_path / f"host_{name}.wav")
        h.write_wav(a, track_order=order, bit_depth=bits)
        twin.write_wav(b, track_order=order, bit_depth=bits)
        assert open(a, "rb").read() == open(b, "rb").read()
    assert all(ir._row is not None for pair in h.irs.values() for ir in pair.values())      # still on the device


def test_write_wav_sweep_sequence_from_device_rows_against_shipped_files(gpu_ctx, golden, tmp_path):
    """imp_rows_to_pcm_device against REFERENCE-HELD data: sweep_sequence(['FL'], 'stereo') is put on the
    device as two fp32 rows and written through HRIR.write_wav's device path; the file is compared with
    data/sweep-seg-FL-stereo-6.15s-48000Hz-32bit-2.93Hz-24000Hz.wav as the reference ships it (tests/golden/sweep_wavs.npz:
    every 37th frame, the full-scale samples, the bounds of the silence).  Device rows are fp32, the reference writes from
    float64: a sample x differs by at most |x| 2^-24 2^31 = 128 LSB at full scale (asserted: <= 129), samples below 2^-8 of full
    scale are exact to 1 LSB, silence is exact, and the file's +2147483647 / -2147483648 samples come out saturated, not wrapped.
    The same rows through the oracle's (file-pinned) quantiser must agree bit for bit: that is the device-vs-reference-rule
    check, independent of the product's host codec."""
    from impulse_hip.device_rows import DeviceBlock, Row
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle import hrir as ohrir
    from scipy.io import wavfile
    g = golden("sweep_wavs")
    e = ImpulseResponseEstimator(min_duration=5.0, fs=48000)
    seq = e.sweep_sequence(["FL"], "stereo")
    n_frames, n_tracks = (int(v) for v in g["seg_fl_stereo_shape"])
    assert seq.shape == (n_tracks, n_frames)
    rows32 = np.ascontiguousarray(seq, dtype=np.float32)
    from impulse_hip import _native
    dctx = _native.default_context()                      # the context the classes keep their device rows on
    block = DeviceBlock(dctx, rows32.size)
    dctx.h2d(block.ptr, rows32)
    h = HRIR(e)
    h.irs = {"FL": {"left": ImpulseResponse.on_device(Row(block, 0, n_frames), 48000),
                    "right": ImpulseResponse.on_device(Row(block, n_frames, n_frames), 48000)}}
    assert h._device_rows([h.irs["FL"]["left"], h.irs["FL"]["right"]]) is not None
    path = str(tmp_path / "seg.wav")
    h.write_wav(path, track_order=["FL-left", "FL-right"], bit_depth=32)
    assert h.irs["FL"]["left"]._row is not None                                  # written from the device rows
    fs, got = wavfile.read(path)
    assert fs == 48000 and got.dtype == np.int32 and got.shape == (n_frames, n_tracks)
    got = got.astype(np.int64)
    # (a) the device conversion == the file-pinned oracle rule on the same fp32 samples, every sample
    assert np.array_equal(got, ohrir.pcm_quantise(rows32.T.astype(np.float64), 32))
    # (b) against the reference's file
    want = g["seg_fl_stereo_dec"].astype(np.int64)
    dec = got[::int(g["stride"])]
    d = np.abs(dec - want)
    assert d.max() <= 129 and np.all(d <= np.abs(want) * 2.0 ** -24 + 1.0)
    assert np.array_equal(dec[:, 1], want[:, 1]) and not want[:, 1].any()         # the silent track
    small = np.abs(want[:, 0]) < 2 ** 23
    assert small.sum() > 100 and np.max(d[small, 0]) <= 1
    nz = np.flatnonzero(got[:, 0])
    assert [int(nz[0]), int(nz[-1])] == g["seg_fl_stereo_nonzero_bounds"][0].tolist()
    for (i, t), v in zip(g["seg_fl_stereo_fullscale_idx"], g["seg_fl_stereo_fullscale_val"].astype(np.int64)):
        assert abs(int(got[i, t]) - int(v)) <= 129
        if int(v) in (2 ** 31 - 1, -2 ** 31):
            assert int(got[i, t]) == int(v)                                       # +2147483647 / -2147483648: saturated
    # (c) 16 / 24 bit from the same rows: the top bits of the 32-bit value (libsndfile's clip path; unpinned widths)
    for bits in (16, 24):
        h.write_wav(path, track_order=["FL-left", "FL-right"], bit_depth=bits)
        if bits == 16:
            ints = wavfile.read(path)[1].astype(np.int64)
        else:
            from impulse_hip.audio_io import read_wav
            ints = np.rint(read_wav(path)[1].T * 2.0 ** 23).astype(np.int64)
        assert np.array_equal(ints, got >> (32 - bits))
    block.close()


def test_fir_chain_without_host_round_trip(gpu_ctx):
    """imp_chain: recording -> estimate -> first peak -> crop at peak - head (+ fades) -> per-channel FIR as one
    stream-ordered chain; the crop offsets come from the peak search on the device.  Against the oracle, incl. a silent
    channel (peak 0), a peak too close to the end (window clamped) and a refill of the FIRs."""
    from impulse_hip import ConvPlan
    from impulse_hip._native import FirChain
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle.estimator import estimate
    from oracle.impulse_response import peak_index
    from oracle.scipy_restated import fft_convolve, hann
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    N, fs = len(e), 48000
    L, n, K, head, fade, B = N + 2 * fs, 20000, 3000, 48, 400, 5
    rng = np.random.default_rng(21)
    rec = np.zeros((B, L), dtype=np.float32)
    delays = [100, 777, 0, 60000, 155000]                      # channel 2 stays silent; channel 4 peaks near the end
    for c, d in enumerate(delays):
        if c != 2:
            rec[c, d:d + N] += (0.5 * e.test_signal).astype(np.float32)[: L - d]
            rec[c, d + 300:d + 300 + N] += (0.1 * e.test_signal).astype(np.float32)[: L - d - 300]
    rec[4] += (rng.standard_normal(L) * 1e-4).astype(np.float32)
    firs = rng.standard_normal((B, K)) * np.exp(-np.arange(K) / 200.0)
    plan1 = ConvPlan(gpu_ctx, np.asarray(e.inverse_filter), L, "same", ws_channels=B)
    plan5 = ConvPlan(gpu_ctx, firs, n, "full", ws_channels=B)
    chain = FirChain(plan1, plan5, B, head, head, fade)
    po = n + K - 1 + 5
    d_x, d_out, d_pk = gpu_ctx.malloc(rec.nbytes), gpu_ctx.malloc(B * po * 4), gpu_ctx.malloc(B * 8)
    gpu_ctx.h2d(d_x, rec)
    w = np.ones(n)
    w[:head] *= hann(2 * head)[:head]
    w[n - fade:] *= hann(2 * fade)[fade:]
    try:
        for taps in (firs, firs[::-1].copy()):
            plan5.set_filters(taps)
            chain.execute_device(d_x, L, d_out, po, d_pk)
            gpu_ctx.synchronize()
            y = np.empty((B, po), dtype=np.float32)
            pk = np.empty(B, dtype=np.int64)
            gpu_ctx.d2h(y, d_out)
            gpu_ctx.d2h(pk, d_pk)
            for c in range(B):
                ir = estimate(rec[c].astype(np.float64), e.inverse_filter)
                want_pk = peak_index(ir)
                assert int(pk[c]) == want_pk
                s0 = min(max(want_pk - head, 0), L - n)
                ref = fft_convolve(ir[s0:s0 + n] * w, taps[c], "full")
                if c == 2:
                    assert not np.any(y[c, :n + K - 1])
                else:
                    assert rel(y[c, :n + K - 1], ref) <= TIME_TOL
        assert delays[4] + N // 2 - head > L - n                  # the clamp was exercised
    finally:
        chain.close()
        plan1.close()
        plan5.close()
        for p in (d_x, d_out, d_pk):
            gpu_ctx.free(p)


@pytest.mark.parametrize("paired", [False, True])
def test_fir_chain_with_lanes_and_tail_stream(paired):
    """The arrangement the library offers beside the one-stream chain (and bench.py measured slower):
    the deconvolution plan on TWO lanes of one context, the FIR plan on a second context whose stream carries the peak search
    and K5 of every call (events order them; buffer sets rotate so that the next call's K1 runs beside the previous call's
    tail).  Seven calls in flight with distinct inputs and outputs, every one against the oracle; the deconvolution in mono
    and in pair mode.  The TAIL context is closed first: a chain is registered with both of its contexts, so whichever goes
    first closes the chain before its plans (the chain keeps raw pointers into both)."""
    from impulse_hip import Context, ConvPlan
    from impulse_hip._native import FirChain
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle.estimator import estimate
    from oracle.impulse_response import peak_index
    from oracle.scipy_restated import fft_convolve, hann
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    N, fs = len(e), 48000
    L, n, K, head, fade, B, calls = N + 2 * fs, 20000, 3000, 48, 400, 4, 7
    rng = np.random.default_rng(77)
    firs = rng.standard_normal((B, K)) * np.exp(-np.arange(K) / 200.0)
    main, tail = Context(0), Context(0)
    plan1 = ConvPlan(main, np.asarray(e.inverse_filter), L, "same", ws_channels=2 * B, fused=False, paired=paired)
    plan1.set_overlap(2)
    plan5 = ConvPlan(tail, firs, n, "full", ws_channels=B)
    assert plan1.paired == paired and plan5.fused
    chain = FirChain(plan1, plan5, B, head, head, fade)
    po = n + K - 1 + 1
    recs, bufs = [], []
    for j in range(calls):
        rec = (rng.standard_normal((B, L)) * 1e-4).astype(np.float32)
        for c in range(B):
            d = 50 + 211 * j + 37 * c
            rec[c, d:d + N] += (0.5 * e.test_signal).astype(np.float32)
        d_x, d_out, d_pk = main.malloc(rec.nbytes), main.malloc(B * po * 4), main.malloc(B * 8)
        main.h2d(d_x, rec)
        recs.append(rec)
        bufs.append((d_x, d_out, d_pk))
    w = np.ones(n)
    w[:head] *= hann(2 * head)[:head]
    w[n - fade:] *= hann(2 * fade)[fade:]
    try:
        for d_x, d_out, d_pk in bufs:                               # all in flight: nothing waits between the calls
            chain.execute_device(d_x, L, d_out, po, d_pk)
        main.synchronize()
        tail.synchronize()
        for j, (d_x, d_out, d_pk) in enumerate(bufs):
            y, pk = np.empty((B, po), dtype=np.float32), np.empty(B, dtype=np.int64)
            main.d2h(y, d_out)
            main.d2h(pk, d_pk)
            for c in range(B):
                ir = estimate(recs[j][c].astype(np.float64), e.inverse_filter)
                want_pk = peak_index(ir)
                assert int(pk[c]) == want_pk == N // 2 + 50 + 211 * j + 37 * c
                s0 = min(max(want_pk - head, 0), L - n)
                assert rel(y[c, :n + K - 1], fft_convolve(ir[s0:s0 + n] * w, firs[c], "full")) <= TIME_TOL
    finally:
        tail.close()                                                # closes the chain, then plan5, then the context
        assert chain._h is None and plan5._h is None and plan1._h
        with pytest.raises(Exception):
            chain.execute_device(bufs[0][0], L, bufs[0][1], po, bufs[0][2])
        chain.close()
        plan1.close()
        main.close()


@pytest.mark.parametrize("L,M,rows,n,K", [(52000, 20001, 8, 20000, 3000), (100000, 50001, 16, 20000, 3000),
                                          (150000, 60001, 24, 20000, 3000), (1100000, 300001, 160, 20000, 3000),
                                          (1400000, 300001, 192, 20000, 3000), (1800000, 300001, 256, 20000, 3000),
                                          # longer crops and FIRs: K5's pass A (the fused crop) on a power-of-two plan of
                                          # 16 rows, a mixed-radix one of 24 and a 32-column-tile one of 192
                                          (400000, 100001, 64, 90000, 30001), (400000, 100001, 64, 150000, 40001),
                                          (1800000, 300001, 256, 1300000, 200001)])
@pytest.mark.parametrize("k5_fused", [True, False])
def test_fir_chain_on_every_column_kernel_family(gpu_ctx, L, M, rows, n, K, k5_fused):
    """imp_chain's fused steps live in the column passes: the row maxima in pass C of the deconvolution (short plans,
    power-of-two plans with 64- and 32-column tiles, mixed-radix plans with 64- and 32-column tiles) and the crop in pass A
    of the FIR - or, for FIRs of up to 24 577 taps, in the load stage of the fused one-launch kernel (k5_fused).  Arbitrary
    (non-sweep) signals, against the oracle."""
    from impulse_hip import ConvPlan
    from impulse_hip._native import FirChain
    from oracle.impulse_response import peak_index
    from oracle.scipy_restated import fft_convolve, hann
    if k5_fused and K > 24577:
        pytest.skip("FIR too long for the fused kernel: covered by the three-launch case")
    rng = np.random.default_rng(rows)
    B, head, fade = 2, 48, 400
    h = rng.standard_normal(M) * 1e-3 * np.exp(-np.arange(M) / (M / 8.0))
    h[M // 2] += 1.0
    x = (rng.standard_normal((B, L)) * 1e-4).astype(np.float32)
    for c, at in enumerate((int(0.37 * L) if n < L // 2 else 4000, L - 9000)):   # the second one peaks too close to the end: clamped crop
        x[c, at] += 1.0
        x[c, at + 211] -= 0.6
        x[c, at - 3000] += 0.05                                     # pre-echo below -18 dB: not the first peak
    firs = rng.standard_normal((B, K)) * np.exp(-np.arange(K) / 200.0)
    plan1 = ConvPlan(gpu_ctx, h, L, "same", ws_channels=B, fused=False)
    assert plan1.n1 == rows
    plan5 = ConvPlan(gpu_ctx, firs, n, "full", ws_channels=B, fused=k5_fused)
    assert plan5.fused == k5_fused
    chain = FirChain(plan1, plan5, B, head, head, fade)
    po = n + K - 1 + 3
    d_x, d_out, d_pk = gpu_ctx.malloc(x.nbytes), gpu_ctx.malloc(B * po * 4), gpu_ctx.malloc(B * 8)
    gpu_ctx.h2d(d_x, x)
    w = np.ones(n)
    w[:head] *= hann(2 * head)[:head]
    w[n - fade:] *= hann(2 * fade)[fade:]
    try:
        for _ in range(2):                                          # twice: nothing is left over from the first run
            chain.execute_device(d_x, L, d_out, po, d_pk)
        gpu_ctx.synchronize()
        y = np.empty((B, po), dtype=np.float32)
        pk = np.empty(B, dtype=np.int64)
        gpu_ctx.d2h(y, d_out)
        gpu_ctx.d2h(pk, d_pk)
        for c in range(B):
            ir = fft_convolve(x[c].astype(np.float64), h, "same")
            want_pk = peak_index(ir)
            assert int(pk[c]) == want_pk
            s0 = min(max(want_pk - head, 0), L - n)
            ref = fft_convolve(ir[s0:s0 + n] * w, firs[c], "full")
            assert rel(y[c, :n + K - 1], ref) <= TIME_TOL
        assert peak_index(fft_convolve(x[1].astype(np.float64), h, "same")) - head > L - n      # the clamp was exercised
    finally:
        chain.close()
        plan1.close()
        plan5.close()
        for ptr in (d_x, d_out, d_pk):
            gpu_ctx.free(ptr)


def test_chain_and_peak_search_refuse_what_they_cannot_do(gpu_ctx):
    """Loud errors instead of wrong answers: a non-positive peak height (the search works on |x| against one positive
    threshold), plans whose overlap setting does not fit the chain, overlap-add plans, the wrong plan modes, a crop longer
    than the response."""
    from impulse_hip import ConvPlan
    from impulse_hip._native import FirChain, NativeError
    rng = np.random.default_rng(3)
    with pytest.raises(NativeError, match="peak_height must be positive"):
        gpu_ctx.peak_index([rng.standard_normal(100).astype(np.float32)], peak_height=0.0)
    same = ConvPlan(gpu_ctx, rng.standard_normal(500), 40000, "same", ws_channels=4, fused=False)
    full = ConvPlan(gpu_ctx, rng.standard_normal((2, 300)), 9000, "full", ws_channels=4)
    long_full = ConvPlan(gpu_ctx, rng.standard_normal(300), 50000, "full", ws_channels=4)
    ola = ConvPlan(gpu_ctx, rng.standard_normal(300), 3 << 20, "same", ws_channels=1, fused=False)
    fused_same = ConvPlan(gpu_ctx, rng.standard_normal(500), 40000, "same", ws_channels=4)
    try:
        assert fused_same.fused and full.fused and ola.nfft == 1 << 21
        with pytest.raises(NativeError, match="cannot be the deconvolution stage"):
            FirChain(fused_same, full, 2, 48, 48, 100)               # a fused plan leaves no chunk maxima
        fused_same.close()
        with pytest.raises(NativeError, match="peak_height must be positive"):
            FirChain(same, full, 2, 48, 48, 100, peak_height=-0.1)
        with pytest.raises(NativeError, match="'same' deconvolution plan and a 'full' FIR plan"):
            FirChain(full, same, 2, 48, 48, 100)
        with pytest.raises(NativeError, match="does not fit"):
            FirChain(same, long_full, 1, 48, 48, 100)              # crop of 50 000 out of 40 000 samples
        with pytest.raises(NativeError, match="fewer FIRs than channels"):
            FirChain(same, full, 3, 48, 48, 100)
        with pytest.raises(NativeError, match="cannot be chained"):
            FirChain(ola, full, 1, 48, 48, 100)
        same.set_overlap(2)
        with pytest.raises(NativeError, match="context of its own"):       # lanes need the FIR plan on its own context / stream
            FirChain(same, full, 2, 48, 48, 100)
        same.set_overlap(1)
        full.set_overlap(2)
        with pytest.raises(NativeError, match="stream order"):
            FirChain(same, full, 2, 48, 48, 100)
        full.set_overlap(1)
        chain = FirChain(same, full, 2, 48, 48, 100)
        d_x, d_out = gpu_ctx.malloc(2 * 40000 * 4), gpu_ctx.malloc(2 * 9300 * 4)
        with pytest.raises(NativeError, match="chan_stride_out"):
            chain.execute_device(d_x, 40000, d_out, 9000)           # rows of 9 299 samples do not fit a pitch of 9 000
        same.set_overlap(2)
        with pytest.raises(NativeError, match="setting changed since the chain was made"):
            chain.execute_device(d_x, 40000, d_out, 9300)
        chain.close()
        gpu_ctx.free(d_x)
        gpu_ctx.free(d_out)
    finally:
        for pl in (same, full, long_full, ola):
            pl.close()


def test_fir_chain_seeded_fuzz(gpu_ctx):
    """tools/fuzz_chain.py, 25 cases: random lengths and crop geometry through imp_chain against the oracle (peaks at the
    start, in the middle, too close to the end, plateaus, silent channels; odd crop starts; plans of 4 ... 96 rows)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_chain", os.path.join(root, "tools", "fuzz_chain.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(25, 5) <= 2e-6


def test_peak_index_seeded_fuzz(gpu_ctx):
    """K3 against the oracle on 240 random rows in ragged batches: noise, spikes of both signs, coarsely quantised rows (ties
    and plateaus everywhere, also exactly at the height threshold), ramps without any peak, rows shorter than a chunk and
    rows of many chunks.  Same fp32 data on both sides: the indices must be equal, not close."""
    from oracle.impulse_response import peak_index
    rng = np.random.default_rng(2024)
    rows = []
    for k in range(240):
        n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 9000), rng.integers(9000, 120000)]))
        kind = k % 6
        if kind == 0:
            x = rng.standard_normal(n)
        elif kind == 1:
            x = rng.standard_normal(n) * 0.01
            x[int(rng.integers(0, n))] = rng.choice([-1.0, 1.0])
        elif kind == 2:
            x = np.round(rng.standard_normal(n) * 3) / 8.0                  # values on a grid of 1/8: plateaus, exact ties
        elif kind == 3:
            x = np.linspace(-1.0, 1.0, n) * rng.choice([-1.0, 1.0])
        elif kind == 4:
            x = np.round(np.cumsum(rng.standard_normal(n)) * 2) / 2.0        # slow walk on a grid: long plateaus
        else:
            x = np.zeros(n)
            m = int(rng.integers(0, n))
            x[m:] = np.round(rng.standard_normal(n - m) * 4) * 0.12589           # multiples of the height threshold itself
        rows.append(x.astype(np.float32))
    for lo in range(0, len(rows), 48):                                       # ragged batches of 48 rows
        batch = rows[lo:lo + 48]
        idx, mx = gpu_ctx.peak_index(batch)
        for j, (r, i, m) in enumerate(zip(batch, idx, mx)):
            assert int(i) == peak_index(r.astype(np.float64)), (lo + j, len(r))
            assert m == (np.max(np.abs(r)) if len(r) else 0.0)


def test_reflection_levels_golden_from_host_arrays_and_device_rows(gpu_ctx, golden):
    """HRIR.calculate_reflection_levels (core/hrir.py:1003-1090) against the reference's own run: windows cut by the end of
    the data, a silent channel, a pre-echo as first peak; once from host arrays, once from fp32 rows that live on the device
    (K3 + K7 read them there)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import round2_inputs as r2
    from impulse_hip.device_rows import DeviceBlock, Row
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response import ImpulseResponse

    class Est:
        fs = r2.FS
    g = golden("reflection")
    data = r2.reflection_set()
    host = HRIR(Est())
    host.irs = {sp: {sd: ImpulseResponse(v.copy(), r2.FS) for sd, v in pair.items()} for sp, pair in data.items()}
    dev = HRIR(Est())
    flat = [(sp, sd, v.astype(np.float32)) for sp, pair in data.items() for sd, v in pair.items()]   # the inputs are fp32-valued
    pitch = (max(len(v) for _, _, v in flat) + 63) // 64 * 64
    from impulse_hip import _native
    dctx = _native.default_context()                              # device rows belong to the context the classes use
    block = DeviceBlock(dctx, len(flat) * pitch)
    buf = np.zeros((len(flat), pitch), dtype=np.float32)
    for i, (_, _, v) in enumerate(flat):
        buf[i, :len(v)] = v
    dctx.h2d(block.ptr, buf)
    dev.irs = {}
    for i, (sp, sd, v) in enumerate(flat):
        dev.irs.setdefault(sp, {})[sd] = ImpulseResponse.on_device(Row(block, i * pitch, len(v)), r2.FS)
    for tag, kw in (("default", {}), ("wide", dict(direct_sound_duration_ms=5, early_ref_start_ms=5, early_ref_end_ms=80,
                                                   late_ref_start_ms=80, late_ref_end_ms=400))):
        for h in (host, dev):
            got = h.calculate_reflection_levels(**kw)
            assert list(got) == list(data)
            for sp, pair in got.items():
                for sd, v in pair.items():
                    want = g[f"{tag}_{sp}_{sd}"]
                    assert abs(v["early_db"] - want[0]) <= 1e-9 and abs(v["late_db"] - want[1]) <= 1e-9, (tag, sp, sd)
    assert all(ir._row is not None for pair in dev.irs.values() for ir in pair.values())      # nothing was pulled to the host
    assert HRIR(Est()).calculate_reflection_levels() == {}


@pytest.mark.parametrize("N", [2, 3, 4, 5, 8, 11, 16, 30, 150, 528, 1000, 1024, 1056, 4096, 9600, 19200, 38400, 55296, 65536,
                               131072, 270336, 524288, 589824, 786432, 1048576])
def test_fft64_tile_transform_against_numpy(gpu_ctx, N, monkeypatch):
    """The batched fp64 transform under K6, K2 and the filter-spectrum preparation: lengths up to 1024 points in ONE launch,
    lengths that split into two factors of at most 1024 points in TWO (tiles held in LDS, decimation in frequency in place,
    csrc/fft64.hip.h) - forward and inverse against np.fft at 1e-13 of the spectrum's rms scale, over every radix (8, 4, 2,
    3, 5, 11), the sizes of the FIR design (19 200 / 38 400 / 65 536 / 131 072), of the normalisation's chirp-z transform and
    of the filter spectra (270 336 = 2^13 x 3 x 11, 589 824, 786 432), ragged batches (tiles of 16 / 8 vectors), and
    against the launch-per-radix-pass form it replaces."""
    rng = np.random.default_rng(N)
    B = 3 if N > 100000 else (21 if N <= 1024 else 5)
    x = rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))
    for inverse in (False, True):
        y, tiles = gpu_ctx.fft64(x, inverse=inverse)
        assert tiles
        ref = np.fft.ifft(x, axis=1) * N if inverse else np.fft.fft(x, axis=1)
        assert np.max(np.abs(y - ref)) <= 1e-13 * np.sqrt(N) * np.sqrt(2.0) * max(1.0, np.log2(N))
    monkeypatch.setenv("IMPULSE_HIP_FFT64_GENERIC", "1")
    # (the switch is read once per process: the generic form is reached through a length the tiles do not take)
    if N == 1048576:
        big = rng.standard_normal((1, 1 << 21)) + 0j
        yb, tiles = gpu_ctx.fft64(big)
        assert not tiles and np.max(np.abs(yb - np.fft.fft(big, axis=1))) <= 1e-13 * np.sqrt(1 << 21) * 30
