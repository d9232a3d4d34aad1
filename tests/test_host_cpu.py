"""CPU-only tests: the C-ABI library loads and exports every symbol include/*.h declares, the
host-side logic of the product (set-up, recording geometry, WAV codec, sharding) agrees with the
oracle/goldens, and the NumPy dataflow model of the kernels reproduces a linear convolution.
No compute call is made on the native library here (there is no GPU)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
    import build
    build.build_library()
    from impulse_hip import _native
    return _native.load_library()


def test_library_exports_every_declared_symbol(lib):
    from impulse_hip import _native
    declared = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            declared |= set(re.findall(r"\b(imp_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    # the ctypes table mirrors the header one to one
    assert declared == set(_native.SIGNATURES)
    assert lib.imp_version().startswith(b"impulse_hip")


def test_no_device_fails_loudly(lib):
    """Without a GPU the product path must raise - never fall back to a CPU implementation."""
    from impulse_hip import Context, NativeUnavailable
    import ctypes as C
    n = C.c_int(-1)
    lib.imp_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(NativeUnavailable):
        Context(0)
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    e = ImpulseResponseEstimator(min_duration=0.2, fs=8000)      # set-up is host-only and works
    with pytest.raises(NativeUnavailable):
        e.estimate(np.zeros(1000))
    from impulse_hip.audio_io import magnitude_response
    from impulse_hip.impulse_response import ImpulseResponse
    with pytest.raises(NativeUnavailable):
        magnitude_response(np.ones(64), 48000)
    with pytest.raises(NativeUnavailable):
        ImpulseResponse(np.ones(64), 48000).peak_index()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "impulcifer-pip313_amd", "impulse_hip")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


@pytest.mark.parametrize("fs,dur", [(48000, 1.0), (48000, 5.0), (96000, 5.0), (44100, 5.0)])
def test_product_estimator_setup_golden(golden, fs, dur):
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    g = golden("estimator")
    k = f"e{fs}_{int(dur)}"
    e = ImpulseResponseEstimator(min_duration=dur, fs=fs)
    assert len(e) == int(g[k + "_N"]) and e.n_octaves == float(g[k + "_P"])
    assert e.low == float(g[k + "_low"]) and e.duration == float(g[k + "_duration"])
    for nm, arr in (("ts", e.test_signal), ("inv", e.inverse_filter)):
        s = np.max(np.abs(arr))
        np.testing.assert_allclose(arr[:64], g[f"{k}_{nm}_head"], rtol=0, atol=1e-12 * s)
        np.testing.assert_allclose(arr[-64:], g[f"{k}_{nm}_tail"], rtol=0, atol=1e-12 * s)
        np.testing.assert_allclose(arr[::1024], g[f"{k}_{nm}_dec"], rtol=0, atol=1e-12 * s)
    with pytest.raises(ValueError):
        ImpulseResponseEstimator(fs=44100.5)


def test_sweep_sequence_and_split_roundtrip():
    from impulse_hip.hrir import split_recording
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from oracle import hrir as ohrir
    from oracle.estimator import sweep_sequence_layout
    e = ImpulseResponseEstimator(min_duration=0.25, fs=8000)
    N, fs = len(e), 8000
    seq = e.sweep_sequence(["FL", "FC", "FR"], "7.1")
    total, starts = sweep_sequence_layout(3, N, fs)
    assert seq.shape == (8, total)
    for sp_idx, (track, s) in enumerate(zip((0, 2, 1), starts)):
        assert np.array_equal(seq[track, s:s + N], e.test_signal)
        assert not seq[track, :s].any()
    for bad in (dict(speakers=["FL"], tracks="9.9"),
                dict(speakers=["FL", "FR", "FC"], tracks="stereo"), dict(speakers=["XX"], tracks="stereo")):
        with pytest.raises(ValueError):
            e.sweep_sequence(**bad)
    # binaural recording: 2 tracks (left/right ear), 3 speakers in time
    rec = np.random.default_rng(0).standard_normal((2, total))
    _, jobs = split_recording(rec, ["FL", "FC", "FR"], N, fs)
    ojobs = ohrir.split_recording(rec, ["FL", "FC", "FR"], N, fs)
    assert [(j[0], j[1]) for j in jobs] == [(o[0], o[1]) for o in ojobs]
    for (sp, sd, tr, a, b), (_, _, col) in zip(jobs, ojobs):
        assert np.array_equal(rec[:, 2 * fs:][tr, a:b], col)
    # one-sided room recording: one track, side given
    _, jobs = split_recording(rec[:1], ["FL", "FC", "FR"], N, fs, side="left")
    assert [(j[0], j[1]) for j in jobs] == [("FL", "left"), ("FC", "left"), ("FR", "left")]
    # short recording: falls back to a reduced lead silence / partial column, or raises
    short = rec[:, : 2 * fs + N + fs // 2]
    _, jobs = split_recording(short, ["FL"], N, fs)
    assert len(jobs) == 2
    with pytest.raises(ValueError):
        split_recording(rec[:, : fs], ["FL"], N, fs)
    with pytest.raises(ValueError):
        split_recording(rec, ["FL"], N, fs, silence_length=0.00001)


def test_product_sweep_files_against_the_shipped_wavs(golden, tmp_path):
    """What the reference's `python -m core.impulse_response_estimator --dir_path data --fs 48000 --speakers FL --tracks
    stereo|mono` wrote into its data/ folder (core/impulse_response_estimator.py:306-322), re-made with the product's
    estimator, sweep_sequence and host WAV codec: >= 99.9 % of the samples identical, the rest 1 LSB (libm), silence bounds
    and saturated full-scale samples as in the files (tests/golden/sweep_wavs.npz)."""
    from impulse_hip.audio_io import write_wav
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from scipy.io import wavfile
    g = golden("sweep_wavs")
    e = ImpulseResponseEstimator(min_duration=5.0, fs=48000)
    assert e.file_name(32) == "6.15s-48000Hz-32bit-2.93Hz-24000Hz"
    for key, data in (("sweep", e.test_signal), ("seg_fl_mono", e.sweep_sequence(["FL"], "mono")),
                      ("seg_fl_stereo", e.sweep_sequence(["FL"], "stereo"))):
        path = str(tmp_path / (key + ".wav"))
        write_wav(path, e.fs, data, bit_depth=32)
        fs, q = wavfile.read(path)
        q = q.reshape(len(q), -1).astype(np.int64)
        assert fs == 48000 and tuple(q.shape) == tuple(g[key + "_shape"])
        want = g[key + "_dec"].astype(np.int64)
        d = np.abs(q[::int(g["stride"])] - want)
        assert d.max() <= 1 and np.mean(d == 0) >= 0.999
        for t in range(q.shape[1]):
            nz = np.flatnonzero(q[:, t])
            assert ([int(nz[0]), int(nz[-1])] if len(nz) else [-1, -1]) == g[key + "_nonzero_bounds"][t].tolist()
        for (i, t), v in zip(g[key + "_fullscale_idx"], g[key + "_fullscale_val"].astype(np.int64)):
            assert abs(int(q[i, t]) - int(v)) <= 1 and (abs(int(v)) < 2 ** 31 - 1 or int(q[i, t]) == int(v))
        assert q.max() == 2147483647 and q.min() == -2147483648


def test_wav_codec_roundtrip(tmp_path):
    from impulse_hip.audio_io import read_wav, write_wav
    rng = np.random.default_rng(4)
    x = rng.uniform(-0.9, 0.9, size=(3, 1000))
    # written as clip(lrint(x 2^31)) >> (32 - bits) (libsndfile's clip path, which soundfile selects; PCM_32 pinned by the
    # reference's own sweep files, tests/test_oracle_golden.py::test_pcm32_conversion_against_shipped_sweep_wavs), read
    # with 1 / 2^(bits-1): a 32-bit round trip is within half an LSB, the narrower widths floor to their LSB
    for bits in (16, 24, 32):
        p = str(tmp_path / f"t{bits}.wav")
        write_wav(p, 48000, x, bit_depth=bits)
        fs, y = read_wav(p)
        assert fs == 48000 and y.shape == x.shape
        lsb = 2.0 ** -(bits - 1)
        err = y - x
        assert (np.max(np.abs(err)) <= 0.5 * lsb) if bits == 32 else (err.max() <= 2.0 ** -32 and err.min() >= -lsb)
        from impulse_hip.audio_io import pcm_quantise
        assert np.array_equal(np.rint(y * 2.0 ** (bits - 1)).astype(np.int64), pcm_quantise(x, bits))
        assert np.array_equal(pcm_quantise(y, bits), pcm_quantise(x, bits))          # re-writing a read file is lossless
        # scipy reads the same integers
        from scipy.io import wavfile
        if bits != 24:
            _, raw = wavfile.read(p)
            assert np.array_equal(raw.T.astype(np.float64) / 2.0 ** (bits - 1), y)
    p = str(tmp_path / "mono.wav")
    write_wav(p, 8000, x[0], bit_depth=32)
    fs, y = read_wav(p)
    assert y.ndim == 1
    fs, y = read_wav(p, expand=True)
    assert y.shape == (1, 1000)
    with pytest.raises(ValueError):
        write_wav(p, 8000, x, bit_depth=20)


def test_shard_channels():
    from impulse_hip.sharding import shard_channels, split_evenly
    for B in (16, 26, 256, 1024, 7):
        for W in (1, 2, 4, 8):
            spans = [shard_channels(B, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            if B % 2 == 0:
                assert all(a % 2 == 0 and b % 2 == 0 for a, b in spans)   # stereo pairs stay together
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= (2 if B % 2 == 0 else 1)
    assert list(split_evenly(10, 4)) == [3, 3, 2, 2]
    with pytest.raises(ValueError):
        shard_channels(8, 2, 2)


def test_kernel_dataflow_model_is_a_linear_convolution():
    import fourstep_model as fm
    from oracle.scipy_restated import fft_convolve
    rng = np.random.default_rng(0)
    r = rng.standard_normal(4096) + 1j * rng.standard_normal(4096)
    assert np.abs(fm.regs_to_natural(fm.fft4096_fwd(r)) - np.fft.fft(r)).max() < 1e-10
    assert np.abs(fm.ifft4096_inv(fm.natural_to_regs(np.fft.fft(r))) / 4096 - r).max() < 1e-12
    for L, M in ((70001, 61000), (150000, 100001)):
        x, h = rng.standard_normal(L), rng.standard_normal(M)
        y = fm.convolve_same_model(x, h)
        ref = fft_convolve(x, h, "same")
        assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-12


def test_plan_geometry_wraparound_rule(lib):
    """'same' plans may use a circular length shorter than L+M-1; the NumPy model proves the window
    is still exact at the chosen size, and the library picks the same size as the model."""
    import fourstep_model as fm
    from impulse_hip._native import plan_geometry
    from oracle.scipy_restated import fft_convolve
    cases = {(391270, 295270): 540672, (420000, 295270): 589824, (827965, 635965): 1179648, (1 << 20, 1 << 20): 1572864,
             (243635, 147635): 327680, (100, 4): 32768, (4, 100): 32768, (1, 1): 32768, (40000, 9600): 65536,
             (150000, 80000): 196608}                                   # N1 = 66, 72, 144, 192, 40, 4, 4, 4, 8, 24
    for (L, M), want in cases.items():
        nfft, start, n = plan_geometry(M, L, "same")
        assert nfft == want == fm.pick_nfft(L, M, "same")
        assert (start, n) == ((M - 1) // 2, L)
        nfft_f, start_f, n_f = plan_geometry(M, L, "full")
        assert nfft_f == fm.pick_nfft(L, M, "full") >= L + M - 1 and (start_f, n_f) == (0, L + M - 1)
    from impulse_hip import NativeError
    # beyond one two-level transform: overlap-add pieces of 2^21 points
    assert plan_geometry(4, 3_000_000, "same") == (1 << 21, 1, 3_000_000)
    assert plan_geometry(1_500_000, 2_600_000, "full") == (1 << 21, 0, 4_099_999)
    with pytest.raises(NativeError):
        plan_geometry(4, 1 << 29, "same")
    # aliasing lands only outside the window: model at the reduced size == linear convolution
    rng = np.random.default_rng(8)
    for L, M in ((250000, 190000), (60000, 130000)):          # nfft 393216 (N1 = 48) / 131072 with M > L
        x, h = rng.standard_normal(L), rng.standard_normal(M)
        nfft = fm.pick_nfft(L, M, "same")
        assert nfft < L + M - 1
        y = fm.convolve_same_model(x, h, nfft)
        ref = fft_convolve(x, h, "same")
        assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.parametrize("n1", [4, 8, 16, 24, 40, 48, 66, 72, 80, 96, 128, 144, 160, 192])
def test_host_spectrum_matches_model(lib, n1):
    """fp64 host FFT (radices 2/3/5) + alpha/beta packing of the library vs the NumPy model."""
    import fourstep_model as fm
    from impulse_hip._native import host_spectrum
    rng = np.random.default_rng(n1)
    M = min(50000 + n1, 2 * n1 * 4096 - 7)
    h = rng.standard_normal(M) * np.exp(-np.arange(M) / 9000.0)
    got = host_spectrum(h, n1).astype(np.float64)
    alpha, beta = fm.plan_alpha_beta(h, 2 * n1 * 4096)
    want = fm.alpha_beta_register_order(alpha, beta)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-7 * scale          # fp32 rounding of fp64 values


def test_butterflies_native_unit_test(tmp_path):
    """csrc/fft_regs.hip.h compiled for the host and checked against a naive DFT (radix 2..16)."""
    exe = tmp_path / "tb"
    src = os.path.join(ROOT, "tests", "native", "test_butterflies.cpp")
    inc = os.path.join(ROOT, "impulcifer-pip313_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-w", "-I", inc, src,
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "BUTTERFLIES OK" in r.stdout, r.stdout


def test_two_rank_gloo_broadcast_and_sharding(tmp_path):
    """world_size 2 on CPU (gloo): rank 0's prepared spectrum bytes reach rank 1 through the same
    broadcast helper bench.py uses with RCCL, and the two shards tile the channel range."""
    script = tmp_path / "rank.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {os.path.join(ROOT, "impulcifer-pip313_amd")!r})
import numpy as np, torch, torch.distributed as dist
from impulse_hip.sharding import shard_channels, broadcast_bytes
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
want = torch.from_numpy(np.random.default_rng(123).integers(0, 255, 1 << 16, dtype=np.uint8))
buf = want.clone() if rank == 0 else torch.zeros_like(want)
broadcast_bytes(buf, dist, src=0)
assert torch.equal(buf, want)
lo, hi = shard_channels(16, 2, rank)
spans = [None, None]
dist.all_gather_object(spans, (lo, hi))
assert spans == [(0, 8), (8, 16)]
t = torch.tensor([float(hi - lo)])
dist.all_reduce(t, op=dist.ReduceOp.SUM)
assert t.item() == 16
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
''')
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_product_curve_logic_matches_reference(golden, tmp_path):
    """Host-side part of the product's curve handling (file discovery, CSV parsing, interpolate, center) against the
    reference's outputs."""
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.impulse_response import ImpulseResponse
    from impulse_hip.room_correction import discover_room_measurements, open_mic_calibration, open_room_target
    g = golden("room_fc")

    class Est:
        fs = 48000

    (tmp_path / "room-target.csv").write_bytes(g["target_csv"].tobytes())
    (tmp_path / "room-mic-calibration.txt").write_bytes(g["mic_txt"].tobytes())
    for nm in ("room-FL,FR-left.wav", "room-FC-right.wav", "room-XX.wav", "room.wav", "roomFL.wav", "room-fl-left.wav"):
        (tmp_path / nm).write_bytes(b"")
    disc = discover_room_measurements(str(tmp_path))
    got = {(os.path.basename(m.file_path), m.speakers, m.side) for m in disc.measurements}
    assert got == {("room-FL,FR-left.wav", ("FL", "FR"), "left"), ("room-FC-right.wav", ("FC",), "right"),
                   ("room-XX.wav", ("XX",), None)}
    assert disc.generic_path.endswith("room.wav") and disc.target_path.endswith("room-target.csv")
    assert disc.mic_calibration_path.endswith("room-mic-calibration.txt")
    target = open_room_target(Est(), str(tmp_path))
    mic = open_mic_calibration(Est(), str(tmp_path))
    np.testing.assert_array_equal(target.frequency, g["frequency"])
    np.testing.assert_allclose(target.raw, g["target_raw"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mic.raw, g["mic_raw"], rtol=0, atol=1e-12)
    with pytest.raises(FileNotFoundError):
        open_mic_calibration(Est(), str(tmp_path), str(tmp_path / "nope.txt"))

    # (the smoothing / equalisation curves themselves are device work: tests/test_hip_parity.py)
    with pytest.raises(ValueError):
        FrequencyResponse("dup", frequency=[10, 20, 20], raw=[0, 0, 0])


def test_bench_starts_its_own_ranks_when_invoked_plainly():
    """`python bench.py --gpus 2 ...` without torchrun (no WORLD_SIZE): the parent starts two ranks as a child
    torch.distributed.run before touching any GPU API, relays rank 0's single JSON line and its exit code.
    --rehearse-launch keeps the ranks off the GPU (rendezvous, sharding, broadcast helper, MAX reduction over gloo)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--rehearse-launch"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert (d["n_gpus"], d["ranks_seen"], d["steps"], d["warmup"]) == (2, 2, 20, 5)
    assert d["broadcast_ok"] and d["shards_tile"] and d["value"] is None and "rehearsal" in d
    assert d["strong_c5"]["shards"] == [[0, 512], [512, 1024]]


def test_bench_rank_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, env=env, cwd=ROOT)
    assert res.returncode != 0 and "WORLD_SIZE=3" in res.stderr


def test_product_from_wav_branches(golden, tmp_path):
    """ImpulseResponseEstimator.from_wav (reference core/impulse_response_estimator.py:234-262), all three branches,
    against a reference run: off-grid length -> the file's samples become the sweep; on-grid but > 1e-4 away -> same,
    with a warning; on-grid and equal to PCM rounding -> the generated sweep is kept.  Host fp64 set-up: no GPU."""
    import round2_inputs as r2
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    g = golden("round2")
    e1 = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    cases = {"offgrid": r2.off_grid_sweep(), "perturbed": r2.perturbed(e1.test_signal, 3e-4), "ongrid": e1.test_signal}
    for name, sig in cases.items():
        path = str(tmp_path / (name + ".wav"))
        r2.write_pcm32(path, 48000, sig)
        ire = ImpulseResponseEstimator.from_wav(path)
        assert len(ire) == int(g[f"fw_{name}_N"]) and ire.duration == float(g[f"fw_{name}_duration"])
        assert ire.n_octaves == float(g[f"fw_{name}_P"]) and ire.fs == 48000
        is_file = np.array_equal(ire.test_signal, r2.to_pcm32(sig).astype(np.float64) / 2 ** 31)
        assert bool(is_file) == bool(g[f"fw_{name}_sig_is_file"])
        for got, key in ((ire.inverse_filter[:64], "inv_head"), (ire.inverse_filter[-64:], "inv_tail"),
                         (ire.inverse_filter[::509], "inv_dec")):
            np.testing.assert_allclose(got, g[f"fw_{name}_{key}"], rtol=1e-11, atol=0)
        assert float(np.sum(ire.inverse_filter)) == pytest.approx(float(g[f"fw_{name}_inv_sum"]), rel=1e-9)
    # a two-track WAV: track 0 is the sweep
    path = str(tmp_path / "two.wav")
    r2.write_pcm32(path, 48000, np.vstack([r2.off_grid_sweep(), np.zeros(1 << 17)]))
    assert len(ImpulseResponseEstimator.from_wav(path)) == 1 << 17


def test_unique_id_file_rendezvous(tmp_path):
    """The 128-byte RCCL id reaches the other ranks through a file published atomically by rank 0 (no torch)."""
    import threading
    from impulse_hip.sharding import share_unique_id
    path = str(tmp_path / "uid")
    want = bytes(range(128))
    got = {}

    def reader(r):
        got[r] = share_unique_id(r, path, timeout_s=30)

    threads = [threading.Thread(target=reader, args=(r,)) for r in (1, 2, 3)]
    for t in threads:
        t.start()
    import time
    time.sleep(0.05)
    assert share_unique_id(0, path, lambda: want) == want
    for t in threads:
        t.join(30)
    assert got == {1: want, 2: want, 3: want}
    with pytest.raises(TimeoutError):
        share_unique_id(1, str(tmp_path / "never"), timeout_s=0.05)
    # a file left by ANOTHER launch (other token) is skipped by readers that start before rank 0 and replaced by rank 0
    stale = bytes(reversed(range(128)))
    path2 = str(tmp_path / "uid2")
    assert share_unique_id(0, path2, lambda: stale, token="launch A") == stale
    early = {}
    t = threading.Thread(target=lambda: early.update(uid=share_unique_id(1, path2, timeout_s=30, token="launch B")))
    t.start()
    time.sleep(0.1)
    assert not early                                            # still waiting: the stale id was not taken
    assert share_unique_id(0, path2, lambda: want, token="launch B") == want
    t.join(30)
    assert early == {"uid": want}
    # the same path AND token again: indistinguishable from a stale file, refused loudly
    with pytest.raises(FileExistsError):
        share_unique_id(0, path2, lambda: want, token="launch B")


def test_line_fit_has_the_bits_of_numpy_cov():
    """decay._fit_line issues np.cov's own primitive sequence directly (the knee search of every response calls it ~20
    times): slope and intercept must be bit-identical to the np.mean / np.cov(bias=1) form scipy.stats.linregress uses."""
    from impulse_hip.decay import _fit_line
    rng = np.random.default_rng(0)
    for _ in range(3000):
        n = int(rng.integers(2, 200))
        x = np.arange(n) * rng.uniform(0.001, 0.1) + rng.uniform(0, 1)
        y = -rng.uniform(1, 100) * x + rng.standard_normal(n) * rng.uniform(0, 5) - rng.uniform(0, 80)
        lo = int(rng.integers(0, max(n - 2, 1)))
        xs, ys = x[lo:], y[lo:]                                  # views, as the search passes them
        c = np.cov(xs, ys, bias=1)
        slope = c[0, 1] / c[0, 0]
        want = (slope, np.mean(ys) - slope * np.mean(xs))
        got = _fit_line(xs, ys)
        assert got[0] == want[0] and got[1] == want[1]


def test_fir_batch_reaches_k5_without_a_copy_only_when_it_is_the_same_matrix():
    """HRIR.equalize_channels hands K5 the FIR matrix of process_equalization_batch itself when the per-channel taps are
    its consecutive rows; anything else (copies, another order, a subset with gaps, Fortran order) is stacked as before."""
    from impulse_hip.hrir import _rows_matrix
    m = np.empty((4, 6))
    m[:] = np.arange(24.0).reshape(4, 6)
    same = _rows_matrix([m[i] for i in range(4)])
    assert np.shares_memory(same, m) and np.array_equal(same, m)
    part = _rows_matrix([m[1], m[2]])
    assert np.shares_memory(part, m) and np.array_equal(part, m[1:3])
    for rows in ([m[2], m[1]], [m[0], m[2]], [m[0].copy(), m[1].copy()], [m[0], m[1][::-1]], [np.asfortranarray(m)[0], np.asfortranarray(m)[1]],
                 [m[0, :3], m[1, :3]]):
        got = _rows_matrix(rows)
        assert not np.shares_memory(got, m)
        assert np.array_equal(got, np.stack(rows))


def test_frequency_grid_is_cached_but_never_shared():
    """generate_frequencies runs its multiplication loop once per (f_min, f_max, f_step); every caller gets its own array
    (the reference's callers write into theirs)."""
    from impulse_hip.frequency_response import FrequencyResponse, generate_frequencies
    a = generate_frequencies(10, 24000, 1.01)
    b = FrequencyResponse.generate_frequencies(f_min=10, f_max=24000, f_step=1.01)
    ref, f = [], 10
    while f <= 24000:
        ref.append(f)
        f *= 1.01
    assert np.array_equal(a, np.array(ref)) and np.array_equal(a, b) and not np.shares_memory(a, b)
    a[0] = -1.0
    assert generate_frequencies(10, 24000, 1.01)[0] == 10.0


def test_surface_hygiene_refusals_and_reference_quirks():
    """Out-of-scope methods refuse clearly instead of raising AttributeError (core/pipeline.py:631 calls
    HRIR.correct_microphone_deviation); sweep_sequence follows the reference on duplicate speakers: its uniqueness loop
    (core/impulse_response_estimator.py:186-189) never appends to the list it tests and so never raises."""
    sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    e = ImpulseResponseEstimator(min_duration=0.2, fs=8000)
    h = HRIR(e)
    with pytest.raises(NotImplementedError, match="microphone"):
        h.correct_microphone_deviation(correction_strength=0.7)
    with pytest.raises(NotImplementedError, match="nnresample"):
        h.resample(44100)
    seq = e.sweep_sequence(["FL", "FL"], "7.1")                  # duplicates pass, as in the reference
    slot = int(e.fs * 2.0 + len(e))
    assert seq.shape == (8, int(slot * 2 + e.fs * 2.0))
    for i in range(2):
        start = int(slot * i + e.fs * 2.0)
        np.testing.assert_array_equal(seq[0, start:start + len(e)], e.test_signal)
    assert not seq[1:].any()
    # no optional plot mixin is looked for any more
    import impulse_hip.hrir as hrir_mod
    import impulse_hip.impulse_response as ir_mod
    for mod in (hrir_mod, ir_mod):
        assert "plotting" not in open(mod.__file__).read()


def test_device_shards_partition_and_result_order():
    """Single-process multi-device fan-out (IMPULSE_HIP_DEVICES): the partition of channels over per-device contexts -
    contiguous, pairs kept together, empty shards dropped - and run_sharded's contract: one call per shard with that
    shard's context installed for the calling thread, results in channel order whatever order the threads finish in."""
    import threading
    import time
    sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
    from impulse_hip import _native
    from impulse_hip.sharding import device_shards, run_sharded
    assert device_shards(16, 1) == [(0, 0, 16)]
    assert device_shards(16, 3) == [(0, 0, 6), (1, 6, 12), (2, 12, 16)]            # pairs stay together
    assert device_shards(5, 2, keep_pairs=False) == [(0, 0, 3), (1, 3, 5)]
    assert device_shards(2, 4) == [(0, 0, 2)]                                       # one pair: one device
    assert device_shards(7, 8, keep_pairs=False) == [(r, r, r + 1) for r in range(7)]
    for n, d in ((1024, 8), (26, 4), (3, 2)):
        sh = device_shards(n, d)
        assert sh[0][1] == 0 and sh[-1][2] == n and all(a[2] == b[1] for a, b in zip(sh, sh[1:]))
    ctxs = [object() for _ in range(4)]
    seen = {}

    def work(ctx, lo, hi):
        assert _native.default_context() is ctx                  # the shard's context is the package default in this thread
        seen[lo] = threading.current_thread().name
        time.sleep(0.02 * (4 - ctxs.index(ctx)))                 # later shards finish first
        return list(range(lo, hi))

    parts = run_sharded(ctxs, device_shards(26, 4), work)
    assert sum(parts, []) == list(range(26))
    assert len(set(seen.values())) == 4                          # one host thread per device
    assert getattr(_native._thread_ctx, "ctx", None) is None     # nothing left installed in the calling thread
    # the device list: IMPULSE_HIP_DEVICES names the devices (the first is the root), else IMPULSE_HIP_DEVICE alone
    env = dict(os.environ)
    try:
        os.environ.pop("IMPULSE_HIP_DEVICES", None)
        os.environ["IMPULSE_HIP_DEVICE"] = "3"
        assert _native.device_list() == [3] and _native.default_device() == 3
        os.environ["IMPULSE_HIP_DEVICES"] = "2, 0,5"
        assert _native.device_list() == [2, 0, 5] and _native.default_device() == 2
    finally:
        os.environ.clear()
        os.environ.update(env)


def test_alignment_tables_of_a_layout():
    """The pair tables imp_slice_set_alignment takes, from a layout's speakers: core/hrir.py:921-1001's rules - a pair of
    IPSILATERAL_PAIRS counts when both speakers are present, an onset group follows its FIRST speaker's left ear and is
    skipped when that speaker is absent, FL/FR is the reference, FL must be there."""
    from impulse_hip.resident_slice import alignment_tables
    spk = ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"]
    ipsi, leader, ref = alignment_tables(spk)
    assert ref == 0
    assert ipsi == [(0, 1), (5, 6), (3, 4), (2, 2)]                   # FL-FR, SL-SR, BL-BR, FC-FC; WL-WR lacks WR
    assert leader == [-1, -1, 2, 3, 3, 5, 5, 7]                       # FC and WL lead themselves, BR follows BL, SR follows SL
    ipsi, leader, ref = alignment_tables(["FR", "FL", "SR", "WR", "TBL", "TBR"])
    assert ref == 1 and ipsi == [(1, 0), (4, 5)]
    assert leader == [-1, -1, -1, -1, 4, 4]                           # SR / WR: their groups' first speakers are absent
    with pytest.raises(RuntimeError):
        alignment_tables(["FR", "FC"])


def test_wav_measurements_read_files_and_directories(tmp_path):
    """WavMeasurements on the CPU: lazy reads of PCM files in wire order, the layout of a job from its first measurement,
    directories as open_binaural_measurements lists them (core/pipeline_stages.py:504-522), refusals with the reason."""
    from impulse_hip.audio_io import write_wav, write_wav_frames
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import WavMeasurements
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=0.5, fs=fs)
    n = 2 * fs + 2 * (len(e) + 2 * fs)
    rng = np.random.default_rng(3)
    dirs = []
    for m in range(2):
        d = tmp_path / f"m{m}"
        d.mkdir()
        write_wav_frames(str(d / "FL,FR.wav"), fs, rng.integers(-2 ** 20, 2 ** 20, (n, 2)).astype(np.int32), 32)
        write_wav_frames(str(d / "FC.wav"), fs, rng.integers(-2 ** 20, 2 ** 20, (fs * 2 + len(e) + 2 * fs, 2)).astype(np.int32), 32)
        (d / "README.md").write_text("x")
        dirs.append(str(d))
    job, speakers = WavMeasurements.from_dirs(dirs, fs=fs)
    assert len(job) == 2 and sorted(map(tuple, speakers)) == [("FC",), ("FL", "FR")]
    first = job[0]
    assert [f.dtype for f in first] == [np.dtype("<i4")] * 2 and all(f.ndim == 2 and f.shape[1] == 2 for f in first)
    layout = job.layout(e, speakers)
    assert sorted(layout.speakers) == ["FC", "FL", "FR"] and layout.tracks == 2 and layout.dtype == np.int32
    assert layout.samples == sum(f.size for f in first)
    assert len(job[1:]) == 1 and len(list(job)) == 2
    with pytest.raises(ValueError, match="sampling rate"):
        WavMeasurements(job.files, fs=44100)[0]
    odd = str(tmp_path / "pcm24.wav")
    write_wav(odd, fs, np.zeros((2, 100)), bit_depth=24)
    with pytest.raises(ValueError, match="PCM"):
        WavMeasurements([[odd]])[0]
    (tmp_path / "m1" / "SL,SR.wav").write_bytes(b"")
    with pytest.raises(ValueError, match="differ"):
        WavMeasurements.from_dirs(dirs)
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(ValueError, match="No HRIR recordings"):
        WavMeasurements.from_dirs([str(empty)])
