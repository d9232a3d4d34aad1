"""K7c: the Lundeby knee search on the device (csrc/decay_kernels.hip.h, imp_decay_knees_device) against the host search.

The host search (impulse_hip/decay.py _lundeby: NumPy control flow over device window means) is pinned by the twelve
reference-run golden decays (test_hip_parity.py::test_decay_params_batch_equals_single_and_goldens) and follows
core/decay.py:44-260 line by line.  The device search must return the SAME INTEGERS for every row it does not flag, on
every decay shape; flagged rows (a decision inside its guard band, a shape outside the device path's limits) go to the
host search inside knee_indices_rows, so HRIR.crop_tails sees the host's knees always.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows_on_device(datas):
    from impulse_hip import _native
    from impulse_hip.device_rows import DeviceBlock, Row
    ctx = _native.default_context()
    pitch = max(1, max(len(d) for d in datas))
    pitch = (pitch + 63) // 64 * 64
    flat = np.zeros((len(datas), pitch), dtype=np.float32)
    for i, d in enumerate(datas):
        flat[i, :len(d)] = d
    block = DeviceBlock(ctx, flat.size)
    ctx.h2d(block.ptr, flat)
    return ctx, block, [Row(block, i * pitch, len(d)) for i, d in enumerate(datas)]


def _decay(rng, fs, seconds, rt60, floor_db, lead=0, second_rt=None, second_db=-25.0):
    n = int(seconds * fs)
    t = np.arange(n) / fs
    x = rng.standard_normal(n) * 10 ** (-3.0 * t / rt60)
    if second_rt is not None:
        x += rng.standard_normal(n) * 10 ** (second_db / 20) * 10 ** (-3.0 * t / second_rt)
    x += rng.standard_normal(n) * 10 ** (floor_db / 20)
    if lead:
        x = np.concatenate((rng.standard_normal(lead) * 10 ** (floor_db / 20), x))[:n]
    x[min(lead, n - 1)] = 2.5
    return x


def _compare(datas, fs):
    from impulse_hip.decay import decay_params_rows
    from impulse_hip.device_rows import span
    ctx, block, rows = _rows_on_device(datas)
    host = decay_params_rows(rows, fs)
    base, offs, lens = span(rows)
    peak, knee, floor, win, flags = ctx.decay_knees_device(base, offs, lens, fs)
    clear = 0
    for k, h in enumerate(host):
        assert int(peak[k]) == int(h[0]), (k, "peak")
        if flags[k]:
            continue
        clear += 1
        assert (int(knee[k]), int(win[k])) == (int(h[1]), int(h[3])), (k, len(datas[k]), int(knee[k]), h)
        assert abs(float(floor[k]) - float(h[2])) <= 1e-9 * max(1.0, abs(float(h[2]))), (k, float(floor[k]), h[2])
    block.close()
    return clear, flags


@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_device_knees_equal_host_knees_on_room_decays(gpu_ctx, fs):
    """180 single- and double-slope decays per rate (RT60 0.1 - 2.5 s, floors -95 ... -35 dB, peaks up to 0.4 s in, spans
    cut short by the row's end): every unflagged row has the host's knee and window, and nearly all rows are unflagged."""
    rng = np.random.default_rng(fs)
    datas = []
    for i in range(180):
        rt = float(rng.uniform(0.1, 2.5))
        datas.append(_decay(rng, fs, float(rng.uniform(0.25, 3.2)), rt, float(rng.uniform(-95, -35)),
                            lead=int(rng.integers(0, int(0.4 * fs))) if i % 3 else 0,
                            second_rt=float(rng.uniform(0.5, 3.0)) if i % 4 == 1 else None,
                            second_db=float(rng.uniform(-40, -10))))
    clear, flags = _compare(datas, fs)
    assert clear >= 0.9 * len(datas), (clear, np.bincount(flags))


def test_device_knees_on_degenerate_rows(gpu_ctx):
    """Shapes where the search leaves by one of its early exits or the device path declines: empty, shorter than 10
    samples, all zeros, constants, pure noise (no decay: flat fit -> host), a peak on the last sample, spans shorter than
    one 30 ms window, decays of a few hundred microseconds (thousands of windows: host), a step, a single spike."""
    fs = 48000
    rng = np.random.default_rng(5)
    spike = np.zeros(30000)
    spike[100] = 1.0
    late = rng.standard_normal(20000) * 1e-3
    late[-1] = 5.0
    fast = rng.standard_normal(48000) * 10 ** (-3.0 * np.arange(48000) / fs / 0.002) + rng.standard_normal(48000) * 1e-6
    fast[0] = 3.0
    datas = [np.zeros(0), np.zeros(5), np.ones(9), np.zeros(4000), np.ones(20), np.ones(50000), rng.standard_normal(60000),
             rng.standard_normal(1000), late, spike, fast, np.concatenate((np.zeros(500), np.ones(30000))),
             _decay(rng, fs, 0.02, 0.01, -60), _decay(rng, fs, 0.031, 0.02, -60), _decay(rng, fs, 0.0625, 0.05, -70),
             _decay(rng, fs, 1.0, 0.3, -60, lead=47990), rng.standard_normal(10), rng.standard_normal(11) * 1e-30]
    _compare(datas, fs)
    # whatever the device flags, the entry point crop_tails uses returns the host's knees for every row
    from impulse_hip.decay import decay_params_rows, knee_indices_rows
    ctx, block, rows = _rows_on_device(datas)
    stats = {}
    assert knee_indices_rows(rows, fs, stats=stats) == [int(p[1]) for p in decay_params_rows(rows, fs)]
    assert stats["rows"] == len(datas) and 0 < stats["host_rows"] < len(datas)
    block.close()


def _decaying_sine(fs, duration_s, rt60, freq=1000.0, floor_db=-90.0, seed=0):
    """the signal behind tests/golden/decay.npz (make_goldens.py)"""
    r = np.random.default_rng(seed)
    n = int(duration_s * fs)
    t = np.arange(n) / fs
    env = 10 ** ((-60.0 / rt60) * t / 20.0)
    return np.cos(2 * np.pi * freq * t) * env + r.standard_normal(n) * 10 ** (floor_db / 20.0)


def test_device_knees_golden_decays(gpu_ctx, golden):
    """The twelve golden decays as fp32 device rows (the goldens were made from these fp32-rounded samples):
    knee_indices_rows gives the reference run's knee indices, none of them through the host."""
    from impulse_hip.decay import knee_indices_rows
    g = golden("decay")
    keys, datas = [], []
    for rt60 in (0.3, 0.6, 1.0, 1.5):
        for seed in (0, 11, 22):
            keys.append(f"rt{int(rt60 * 10)}_s{seed}")
            datas.append(_decaying_sine(48000, 3.0, rt60, seed=seed).astype(np.float32).astype(np.float64))
    ctx, block, rows = _rows_on_device(datas)
    stats = {}
    knees = knee_indices_rows(rows, 48000, stats=stats)
    assert knees == [int(g[k + "_params"][1]) for k in keys]
    assert stats["host_rows"] == 0
    block.close()


def test_device_knees_fuzz_many_rows_one_call(gpu_ctx):
    """1 500 rows in one call (the batch the C3 / C4 shapes bring), lengths 300 ... 120 000, random envelopes built from
    two to four exponential pieces: unflagged rows equal the host search, row by row."""
    fs = 48000
    rng = np.random.default_rng(99)
    datas = []
    for i in range(1500):
        n = int(rng.integers(300, 120000))
        t = np.arange(n) / fs
        env = np.zeros(n)
        for _ in range(int(rng.integers(2, 5))):
            env += 10 ** (float(rng.uniform(-60, 0)) / 20) * 10 ** (-3.0 * t / float(rng.uniform(0.02, 3.0)))
        x = rng.standard_normal(n) * env + rng.standard_normal(n) * 10 ** (float(rng.uniform(-120, -30)) / 20)
        datas.append(x)
    clear, flags = _compare(datas, fs)
    assert clear >= 0.8 * len(datas), (clear, np.bincount(flags))


def test_staging_ring_grows_and_wraps_with_many_rows(gpu_ctx):
    """The small host tables of the device-row entry points travel through a 1 MiB pinned ring mirrored on the device.
    40 000 rows make one call's tables (offsets, lengths, window parameters: 3.2 MB) larger than the ring - it is
    rebuilt - and 300 more calls make it wrap; every result must still be what NumPy gives."""
    rng = np.random.default_rng(12)
    B, n = 40000, 48
    x = rng.standard_normal((B, n)).astype(np.float32)
    d = gpu_ctx.malloc(x.nbytes)
    gpu_ctx.h2d(d, x)
    offs = np.arange(B, dtype=np.int64) * n
    lens = np.full(B, n, dtype=np.int64)
    gpu_ctx.apply_window_device(d, offs, d, offs, lens, [dict(gain=0.5)] * B)
    idx, mx = gpu_ctx.peak_index_device(d, offs, lens)
    back = np.empty_like(x)
    gpu_ctx.synchronize()
    gpu_ctx.d2h(back, d)
    assert np.array_equal(back, x * np.float32(0.5))
    assert np.array_equal(mx, np.max(np.abs(back), axis=1))
    small = 64
    for it in range(300):                                   # ~3.4 KiB of tables a call: the 1 MiB ring comes round again
        g = np.float32(2.0 if it % 2 == 0 else 0.5)           # exact in fp32, the product stays bounded
        gpu_ctx.apply_window_device(d, offs[:small], d, offs[:small], lens[:small], [dict(gain=float(g))] * small)
        back[:small] *= g
    out = np.empty((small, n), dtype=np.float32)
    gpu_ctx.synchronize()
    gpu_ctx.d2h(out, d)
    assert np.array_equal(out, back[:small])
    gpu_ctx.free(d)
