"""K5 as one launch: fused FIR plans (fir_block_kernel: overlap-save blocks of 32 768 samples held in one CU's registers and
LDS).  Replaces scipy.signal.convolve(x, fir, 'full') at core/impulse_response.py:110-119 (equalize), :126-135 (convolve),
core/hrir.py:858-888 and core/parallel_workers.py:9-21 for filters of up to 24 577 taps - the 9 600- / 19 200-tap
minimum-phase FIRs of the path.  Same tolerances as tests/test_hip_parity.py."""
import numpy as np
import pytest

SPEC_TOL = 1e-6
TIME_TOL = 1e-6


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / np.max(np.abs(b)))


def spec_rel(y, ref):
    A, R = np.abs(np.fft.rfft(np.asarray(y, dtype=np.float64))), np.abs(np.fft.rfft(ref))
    return float(np.max(np.abs(A - R)) / np.max(R))


def test_fused_plan_geometry_needs_no_gpu():
    from impulse_hip._native import plan_geometry, plan_geometry_fused
    # the everyday K5 job: 0.68 s response (*) 9 600-tap FIR at 48 kHz: 9 600 samples of history, 23 168 outputs per block,
    # two blocks for the 42 239 outputs; the three-launch plan for the same job is 8 rows (65 536 points)
    assert plan_geometry_fused(9600, 32640, "full") == (9600, 23168, 0, 2)
    assert plan_geometry(9600, 32640, "full")[0] == 65536
    assert plan_geometry_fused(19200, 65280, "full") == (19200, 13568, 0, 7)        # C3: 96 kHz
    assert plan_geometry_fused(1, 1, "same") == (0, 32768, 0, 1)
    assert plan_geometry_fused(4, 100, "full") == (4, 32764, 0, 1)
    # 'same' keeps [ (M-1)//2, (M-1)//2 + L ): a late window starts in a later block
    assert plan_geometry_fused(24577, 100000, "same") == (24576, 8192, 1, 13)
    assert plan_geometry_fused(24578, 1000, "full") is None and plan_geometry_fused(295270, 391270, "same") is None


@pytest.mark.gpu
@pytest.mark.parametrize("L,M", [(1, 1), (17, 5), (1000, 999), (20000, 9600), (32640, 9600), (65280, 19200), (23168, 9600),
                                 (23169, 9601), (100000, 24577), (391270, 9600), (50000, 4097), (8, 24577)])
@pytest.mark.parametrize("mode", ["same", "full"])
def test_fused_fir_matches_oracle_and_three_launch_plan(gpu_ctx, L, M, mode):
    from impulse_hip import ConvPlan
    from impulse_hip._native import plan_geometry_fused
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(L * 31 + M)
    x = rng.standard_normal((3, L)).astype(np.float32)
    x[2] = 0.0
    h = rng.standard_normal(M) * np.exp(-np.arange(M) / max(M / 5.0, 1.0))
    plan = ConvPlan(gpu_ctx, h, L, mode)
    assert plan.fused and (plan.nfft, plan.n1) == (32768, 4) and plan_geometry_fused(M, L, mode) is not None
    y = plan.execute(x)
    assert np.array_equal(y, plan.execute(x))                      # bit-identical reruns
    plan.close()
    three = ConvPlan(gpu_ctx, h, L, mode, fused=False)
    y3 = three.execute(x)
    three.close()
    assert y.shape == (3, L if mode == "same" else L + M - 1) and not np.any(y[2])
    for b in range(2):
        ref = fft_convolve(x[b].astype(np.float64), h, mode)
        assert rel(y[b], ref) <= TIME_TOL and rel(y3[b], ref) <= TIME_TOL
        if len(ref) > 8:
            assert spec_rel(y[b], ref) <= SPEC_TOL


@pytest.mark.gpu
def test_fused_fir_per_channel_filters_loaders_refill(gpu_ctx):
    """Per-channel FIRs (HRIR.equalize: one FIR per speaker-ear), more channels than eight (the XCD padding of the grid),
    interleaved frames and PCM in, odd pitches, in-place refill of the spectra."""
    from impulse_hip import ConvPlan
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(55)
    L, K, B = 32640, 9600, 19
    x = rng.standard_normal((B, L)).astype(np.float32)
    firs = rng.standard_normal((B, K)) * np.exp(-np.arange(K) / 800.0)
    plan = ConvPlan(gpu_ctx, firs, L, "full", ws_channels=B)
    assert plan.fused
    y = plan.execute(x)
    for b in range(B):
        assert rel(y[b], fft_convolve(x[b].astype(np.float64), firs[b], "full")) <= TIME_TOL
    firs2 = rng.standard_normal((B, K)) * 0.01
    plan.set_filters(firs2)
    y2 = plan.execute(x)
    for b in (0, 7, 8, B - 1):
        assert rel(y2[b], fft_convolve(x[b].astype(np.float64), firs2[b], "full")) <= TIME_TOL
    # device rows, odd pitches in and out
    pin, pout = L + 3, L + K - 1 + 5
    rows = np.zeros((B, pin), dtype=np.float32)
    rows[:, :L] = x
    d_x, d_y = gpu_ctx.malloc(rows.nbytes), gpu_ctx.malloc(B * pout * 4)
    gpu_ctx.h2d(d_x, rows)
    plan.execute_device(d_x, B, pin, d_y, pout)
    gpu_ctx.synchronize()
    got = np.empty((B, pout), dtype=np.float32)
    gpu_ctx.d2h(got, d_y)
    assert np.array_equal(got[:, :L + K - 1], y2)
    plan.close()
    gpu_ctx.free(d_x)
    gpu_ctx.free(d_y)
    # one shared FIR: interleaved frames and the WAV's own PCM
    h = firs[0]
    shared = ConvPlan(gpu_ctx, h, 50001, "same")
    assert shared.fused
    frames = rng.standard_normal((50001, 3)).astype(np.float32)
    yi = shared.execute_interleaved(frames)
    assert np.array_equal(yi, shared.execute(np.ascontiguousarray(frames.T)))
    for c in range(3):
        assert rel(yi[c], fft_convolve(frames[:, c].astype(np.float64), h, "same")) <= TIME_TOL
    pcm = rng.integers(-2 ** 15, 2 ** 15 - 1, size=(50001, 2), endpoint=True).astype(np.int16)
    d_x, d_y = gpu_ctx.malloc(pcm.nbytes), gpu_ctx.malloc(2 * 50002 * 4)
    gpu_ctx.h2d(d_x, pcm)
    shared.execute_device_pcm(d_x, 16, 2, 1, 2, d_y, 50002)
    gpu_ctx.synchronize()
    got = np.empty((2, 50002), dtype=np.float32)
    gpu_ctx.d2h(got, d_y)
    for c in range(2):
        assert rel(got[c, :50001], fft_convolve(pcm[:, c].astype(np.float64) / 2 ** 15, h, "same")) <= TIME_TOL
    shared.close()
    gpu_ctx.free(d_x)
    gpu_ctx.free(d_y)
