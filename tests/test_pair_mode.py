"""Pair mode of K1 (two ears of a speaker as ONE complex signal, z = x_L + i x_R; conv_kernels.hip.h rows_single_kernel):
the stereo-frame formulation of the reference's ingest loop (core/hrir.py:326-341 hands estimate() track i = left ear and
track i + 1 = right ear of the same frames; core/impulse_response_estimator.py:149-151 convolves each with the one real
inverse filter).  Same tolerances as tests/test_hip_parity.py: 1e-6 of the peak in time and on the magnitude spectrum
(cropped window for sweep recordings), peak indices exact."""
import numpy as np
import pytest

from test_hip_parity import FULL_COLUMN_TOL, SPEC_TOL, TIME_TOL, rel, spec_rel, spec_rel_cropped


def test_pair_plan_geometry_needs_no_gpu():
    from impulse_hip._native import plan_geometry, plan_geometry_paired
    # C2 (7.1, 6.15 s sweep): 538 905 samples of circular length -> 132 rows of 4 096 (11 x 12); mono: 66 rows of 8 192
    nfft, start, out_len, n1 = plan_geometry_paired(295270, 391270, "same")
    assert (nfft, start, out_len, n1) == (132 * 4096, (295270 - 1) // 2, 391270, 132)
    assert plan_geometry(295270, 391270, "same")[0] == nfft               # the same circular length in samples
    assert plan_geometry_paired(9600, 32640, "full")[3] == 16             # 42 239 samples -> 11 rows -> the 16-row plan
    assert plan_geometry_paired(5, 17, "same")[3] == 4
    assert plan_geometry_paired(400000, 648576, "full")[3] == 256         # exactly 2^20 samples
    # C3 (96 kHz): 1 145 947 samples -> 280 rows -> the 288-row plan (16 x 18), the same circular length as the mono plan's 144
    assert plan_geometry_paired(635965, 827965, "same") == (288 * 4096, (635965 - 1) // 2, 827965, 288)
    assert plan_geometry(635965, 827965, "same")[0] == 288 * 4096
    # C5 (2^20 x 2^20): 1 572 864 samples -> 384 rows (16 x 24), the same circular length as the mono plan's 192
    assert plan_geometry_paired(1 << 20, 1 << 20, "same") == (384 * 4096, ((1 << 20) - 1) // 2, 1 << 20, 384)
    assert plan_geometry(1 << 20, 1 << 20, "same")[0] == 384 * 4096
    # beyond that pair mode is not available, the caller keeps the mono plan
    assert plan_geometry_paired(1 << 20, 3 << 19, "same") is None


# every column shape of pair mode: N1 = 4, 4, 8, 16, 24, 32, 40, 48, 64, 66, 72, 80, 96, 128, 132, 144, 160, 192, 256, 288 (C3), 384 (C5)
@pytest.mark.gpu
@pytest.mark.parametrize("L,M,mode", [(1, 1, "same"), (17, 5, "full"), (20000, 9600, "full"), (32640, 9600, "full"),
                                      (70001, 61000, "same"), (100000, 30000, "full"), (150000, 20000, "same"),
                                      (150000, 40000, "full"), (243635, 30000, "same"), (243635, 147635, "same"),
                                      (270000, 60000, "same"), (300000, 50000, "same"), (300000, 150000, "same"),
                                      (391270, 200000, "same"), (391270, 295270, "same"), (500000, 150000, "same"),
                                      (500000, 300000, "same"), (500000, 250000, "full"), (800000, 400000, "same"),
                                      (827965, 635965, "same"), (1 << 20, 1 << 20, "same")])
def test_pair_conv_matches_oracle(gpu_ctx, L, M, mode):
    from impulse_hip import ConvPlan
    from impulse_hip._native import plan_geometry_paired
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(L * 31 + M)
    x = rng.standard_normal((5, L)).astype(np.float32)              # odd count: the last channel pairs with silence
    x[3] = 0.0                                                       # a silent right ear
    h = rng.standard_normal(M) * np.exp(-np.arange(M) / max(M / 5.0, 1.0))
    plan = ConvPlan(gpu_ctx, h, L, mode, paired=True)
    assert plan.paired and (plan.nfft, plan.n1) == (plan_geometry_paired(M, L, mode)[0], plan_geometry_paired(M, L, mode)[3])
    y = plan.execute(x)
    assert np.array_equal(y, plan.execute(x))                        # bit-identical reruns
    mono = ConvPlan(gpu_ctx, h, L, mode)
    ym = mono.execute(x)
    plan.close()
    mono.close()
    assert y.shape == (5, L if mode == "same" else L + M - 1)
    for b in (0, 1, 2, 4):
        ref = fft_convolve(x[b].astype(np.float64), h, mode)
        assert rel(y[b], ref) <= TIME_TOL
        if len(ref) > 8:
            assert spec_rel(y[b], ref) <= SPEC_TOL
        assert rel(y[b], ym[b].astype(np.float64)) <= 2 * TIME_TOL
    # the silent ear of a pair holds only the other ear's rounding noise (the two share a transform)
    assert np.max(np.abs(y[3])) <= 1e-6 * np.max(np.abs(y[2]))


@pytest.mark.gpu
def test_pair_loaders_planar_frames_pcm(gpu_ctx):
    """Every way a recording reaches pair mode gives the same numbers: planar rows (even and odd pitch), interleaved fp32
    frames [L][C] for C = 2 (whole-frame loads), 3 (unaligned: two loads), 4, and the WAV's own PCM_16 / PCM_32 frames."""
    from impulse_hip import ConvPlan
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(123)
    L, M = 100001, 9600                                              # odd L, even taps -> odd crop offset
    h = rng.standard_normal(M) * 0.05
    plan = ConvPlan(gpu_ctx, h, L, "same", paired=True)
    out_pitch = L + 3
    for C in (2, 3, 4):
        frames = rng.standard_normal((L, C)).astype(np.float32)
        yi = plan.execute_interleaved(frames)
        yp = plan.execute(np.ascontiguousarray(frames.T))
        assert np.array_equal(yi, yp)                                # wire-order loader == planar loader, bit for bit
        for c in range(C):
            assert rel(yi[c], fft_convolve(frames[:, c].astype(np.float64), h, "same")) <= TIME_TOL
        # device entry with an odd input pitch
        pitch = L + 1 + (C & 1)
        rows = np.zeros((C, pitch), dtype=np.float32)
        rows[:, :L] = frames.T
        d_x, d_y = gpu_ctx.malloc(rows.nbytes), gpu_ctx.malloc(C * out_pitch * 4)
        gpu_ctx.h2d(d_x, rows)
        plan.execute_device(d_x, C, pitch, d_y, out_pitch)
        gpu_ctx.synchronize()
        got = np.empty((C, out_pitch), dtype=np.float32)
        gpu_ctx.d2h(got, d_y)
        assert np.array_equal(got[:, :L], yi)
        gpu_ctx.free(d_x)
        gpu_ctx.free(d_y)
    for dtype, bits in ((np.int16, 16), (np.int32, 32)):
        for C in (2, 3):
            info = np.iinfo(dtype)
            pcm = rng.integers(info.min, info.max, size=(L, C), endpoint=True).astype(dtype)
            as_float = (pcm.astype(np.float64) / 2.0 ** (bits - 1)).astype(np.float32)
            d_x, d_y = gpu_ctx.malloc(pcm.nbytes), gpu_ctx.malloc(C * out_pitch * 4)
            gpu_ctx.h2d(d_x, pcm)
            plan.execute_device_pcm(d_x, bits, C, 1, C, d_y, out_pitch)
            gpu_ctx.synchronize()
            got = np.empty((C, out_pitch), dtype=np.float32)
            gpu_ctx.d2h(got, d_y)
            assert np.array_equal(got[:, :L], plan.execute(np.ascontiguousarray(as_float.T)))   # same fp32 samples in
            for c in range(C):
                assert rel(got[c, :L], fft_convolve(as_float[:, c].astype(np.float64), h, "same")) <= TIME_TOL
            gpu_ctx.free(d_x)
            gpu_ctx.free(d_y)
    plan.close()


@pytest.mark.gpu
def test_pair_binaural_columns_c2_size(gpu_ctx):
    """The C2 shape as the reference meets it: ONE binaural WAV (frames [n][2], PCM_32) holding the columns of several
    speakers one after the other; every column's stereo frames are one pair (imp_conv_execute_device_pairs).  Peaks
    analytic and against the oracle, spectra on the cropped window, whole column against the fp32 floor."""
    from impulse_hip import ConvPlan
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle.estimator import estimate
    from oracle.impulse_response import peak_index
    e = ImpulseResponseEstimator(min_duration=5.0, fs=48000)
    N, fs = len(e), 48000
    L = N + 2 * fs
    n_spk = 3
    rng = np.random.default_rng(0xC2)
    tracks = np.zeros((2, 2 * fs + n_spk * L))
    delays = {}
    for j in range(n_spk):
        for ear in range(2):
            d0 = 64 + 37 * (2 * j + ear)
            delays[(j, ear)] = d0
            base = 2 * fs + j * L
            tracks[ear, base + d0: base + d0 + N] += e.test_signal
            for _ in range(3):
                d = int(rng.integers(100, 24000))
                tracks[ear, base + d: base + d + N] += 0.3 * np.exp(-d / 9600.0) * rng.standard_normal() * e.test_signal
    tracks += rng.standard_normal(tracks.shape) * 10 ** (-70 / 20)
    tracks *= 0.2
    pcm = np.clip(np.rint(tracks.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32)      # [frames, 2]
    plan = ConvPlan(gpu_ctx, np.asarray(e.inverse_filter), L, "same", paired=True)
    assert plan.paired and plan.n1 == 132
    starts = [2 * fs + j * L for j in range(n_spk)]
    y = plan.execute_pcm_columns(pcm, starts)                        # [columns, tracks, L]
    mono = ConvPlan(gpu_ctx, np.asarray(e.inverse_filter), L, "same")
    ym = mono.execute_pcm_columns(pcm, starts)
    plan.close()
    mono.close()
    assert y.shape == (n_spk, 2, L)
    for j in range(n_spk):
        for ear in range(2):
            col = (pcm[starts[j]:starts[j] + L, ear].astype(np.float64) / 2.0 ** 31).astype(np.float32).astype(np.float64)
            ref = estimate(col, e.inverse_filter)
            got = y[j, ear]
            from impulse_hip.impulse_response import ImpulseResponse
            assert ImpulseResponse(got.astype(np.float64), fs).peak_index() == peak_index(ref) == N // 2 + delays[(j, ear)]
            assert rel(got, ref) <= TIME_TOL
            assert spec_rel_cropped(got, ref) <= SPEC_TOL
            assert spec_rel(got, ref) <= FULL_COLUMN_TOL
            assert rel(got, ym[j, ear].astype(np.float64)) <= 2 * TIME_TOL


@pytest.mark.gpu
def test_pair_overlapped_lanes_and_many_groups(gpu_ctx):
    """More channels than a launch group, three lanes in flight: no channel mixing, every pair bit-equal to its twin."""
    from impulse_hip import ConvPlan
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(9)
    L, M, B = 50000, 30000, 24
    h = rng.standard_normal(M) * np.exp(-np.arange(M) / 4000.0)
    x = rng.standard_normal((B, L)).astype(np.float32)
    x[B // 2:] = x[:B // 2]                                          # twins in other launch groups
    plan = ConvPlan(gpu_ctx, h, L, "same", ws_channels=12, paired=True)
    plan.set_overlap(3)                                              # 4 channels = 2 pairs per lane
    with pytest.raises(Exception):
        plan.set_overlap(4)                                          # 3 channels per lane: not whole pairs
    pitch = L + 2
    d_x, d_y = gpu_ctx.malloc(x.nbytes), gpu_ctx.malloc(B * pitch * 4)
    gpu_ctx.h2d(d_x, x)
    plan.execute_device(d_x, B, L, d_y, pitch)
    gpu_ctx.synchronize()
    y = np.empty((B, pitch), dtype=np.float32)
    gpu_ctx.d2h(y, d_y)
    plan.close()
    gpu_ctx.free(d_x)
    gpu_ctx.free(d_y)
    assert np.array_equal(y[B // 2:, :L], y[:B // 2, :L])
    for b in range(B // 2):
        assert rel(y[b, :L], fft_convolve(x[b].astype(np.float64), h, "same")) <= TIME_TOL


@pytest.mark.gpu
def test_pair_plan_refill_and_errors(gpu_ctx):
    from impulse_hip import ConvPlan
    from impulse_hip._native import NativeError
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(4)
    L, M = 32640, 9600
    h1, h2 = rng.standard_normal(M), rng.standard_normal(M) * 0.1
    x = rng.standard_normal((2, L)).astype(np.float32)
    plan = ConvPlan(gpu_ctx, h1, L, "full", paired=True)
    y1 = plan.execute(x)
    plan.set_filters(h2)
    y2 = plan.execute(x)
    plan.close()
    for y, h in ((y1, h1), (y2, h2)):
        for b in range(2):
            assert rel(y[b], fft_convolve(x[b].astype(np.float64), h, "full")) <= TIME_TOL
    with pytest.raises(ValueError):
        ConvPlan(gpu_ctx, np.stack([h1, h2]), L, "full", paired=True)          # per-channel filters cannot pair
    with pytest.raises(NativeError):
        ConvPlan(gpu_ctx, rng.standard_normal(1 << 20), 3 << 19, "same", paired=True)   # beyond 384 rows
    auto = ConvPlan(gpu_ctx, rng.standard_normal(1 << 20), 3 << 19, "same", paired="auto")
    assert not auto.paired
    auto.close()
    mono = ConvPlan(gpu_ctx, h1, L, "full")
    with pytest.raises(NativeError):
        mono.execute_device_pairs(0x1000, 0, 1, 2, 1, 2, 0x1000, L + M)
    mono.close()


@pytest.mark.gpu
def test_pair_mode_c3_size_all_loaders(gpu_ctx):
    """C3's shape (96 kHz: L = 827 965, M = 635 965, 288 rows = 16 x 18) through the three ways a recording reaches pair mode
    - planar rows, interleaved fp32 frames, the WAV's own PCM_32 frames - on a synthetic sweep recording: analytic peaks, the
    oracle in time and on the cropped spectrum, and the whole column, which in pair mode has no even/odd packing to couple
    the bins near Nyquist with the strong ones near DC."""
    from impulse_hip import ConvPlan
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle.estimator import estimate
    fs = 96000
    e = ImpulseResponseEstimator(min_duration=5.0, fs=fs)
    N = len(e)
    L = N + 2 * fs
    assert (N, L) == (635965, 827965)
    rng = np.random.default_rng(0xC3)
    x = np.zeros((2, L))
    delays = (64, 101)
    for c, d0 in enumerate(delays):
        x[c, d0:d0 + N] += 0.5 * e.test_signal
        for _ in range(3):
            d = int(rng.integers(200, 48000))
            x[c, d0 + d:d0 + d + N] += 0.15 * np.exp(-d / 19200.0) * rng.standard_normal() * e.test_signal[:min(N, L - d0 - d)]
    x += rng.standard_normal(x.shape) * 10 ** (-70 / 20)
    x32 = x.astype(np.float32)
    inv = np.asarray(e.inverse_filter, dtype=np.float64)
    plan = ConvPlan(gpu_ctx, inv, L, "same", paired=True)
    assert plan.paired and plan.n1 == 288
    y_planar = plan.execute(x32)
    y_frames = plan.execute_interleaved(np.ascontiguousarray(x32.T))
    assert np.array_equal(y_planar, y_frames)
    pcm = np.clip(np.rint(x.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32)          # [frames, 2]
    y_pcm = plan.execute_pcm_columns(pcm, [0])[0]
    plan.close()
    mono = ConvPlan(gpu_ctx, inv, L, "same")
    y_mono = mono.execute(x32)
    mono.close()
    for c in range(2):
        ref = estimate(x32[c].astype(np.float64), inv)
        assert int(np.argmax(np.abs(y_planar[c]))) == int(np.argmax(np.abs(ref))) == N // 2 + delays[c]
        assert rel(y_planar[c], ref) <= TIME_TOL and spec_rel_cropped(y_planar[c], ref, fs=fs) <= SPEC_TOL
        ref_pcm = estimate((pcm[:, c].astype(np.float64) / 2.0 ** 31).astype(np.float32).astype(np.float64), inv)
        assert rel(y_pcm[c], ref_pcm) <= TIME_TOL and spec_rel_cropped(y_pcm[c], ref_pcm, fs=fs) <= SPEC_TOL
        whole_pair, whole_mono = spec_rel(y_planar[c], ref), spec_rel(y_mono[c], ref)
        print(f"C3 whole column ch{c}: pair mode {whole_pair:.2e}, mono plan {whole_mono:.2e}")
        assert whole_pair <= FULL_COLUMN_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["c2", "c3", "c5"])
def test_whole_column_at_the_fp32_floor(gpu_ctx, config):
    """The function's own output - the whole un-cropped column of estimate() (core/impulse_response_estimator.py:149-151) -
    on BASELINE's three K1 shapes, through the plan the classes take for ear pairs (pair mode: 132 / 288 / 384 rows): the
    magnitude spectrum against the float64 oracle.  north_star's 1e-6 is not reachable on these columns in fp32: the
    transform's white rounding noise (time-domain rms 3 - 5e-10 of the peak) gains sqrt(L) in the spectrum, and the maximum over
    2 - 5 x 10^5 bins lands at 1.1 - 2.3e-6 here, 1.3 - 2.5e-6 for the reference's own pocketfft in single precision (measured
    in this test on the same channels; the statistic moves by up to 2x between channels of one transform).  Gated: every
    channel within FULL_COLUMN_TOL, and the mean over the channels within 1.25 x pocketfft-fp32's (C2 and C5 sit below
    it, C3 15 % above); the cropped window (all the reference ever transforms) stays within SPEC_TOL."""
    import bench
    from impulse_hip import ConvPlan
    from oracle.estimator import estimate
    est = bench.make_estimator(config)
    fs = est.fs
    rec, L, pitch, delays = bench.synth_recordings(est, 2, seed0=0xC2, column=(len(est) if config == "c5" else None))
    inv = np.asarray(est.inverse_filter, dtype=np.float64)
    plan = ConvPlan(gpu_ctx, inv, L, "same", paired=True)
    assert plan.paired and plan.n1 == {"c2": 132, "c3": 288, "c5": 384}[config]
    y = plan.execute(rec[:, :L])
    plan.close()
    ours, floor = [], []
    for c in range(2):
        ref = estimate(rec[c, :L].astype(np.float64), inv)
        assert int(np.argmax(np.abs(y[c]))) == int(np.argmax(np.abs(ref)))
        assert spec_rel_cropped(y[c], ref, fs=fs) <= SPEC_TOL
        ours.append(spec_rel(y[c], ref))
        floor.append(bench.fp32_fft_floor(est, rec[c], L))
    print(f"{config} whole column: this path {ours}, pocketfft fp32 {floor}")
    assert max(ours) <= FULL_COLUMN_TOL
    assert np.mean(ours) <= 1.25 * np.mean(floor)
