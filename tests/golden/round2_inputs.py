"""Seeded inputs of the round-2 fixtures (round2.npz), shared by the golden generator and the tests.  Inputs only,
NumPy only.  Every array the HIP path will see is produced here from seeds, so that the reference (build container)
and the product (GPU box) are given bit-identical bytes."""
import os

import numpy as np

FS = 48000


def off_grid_sweep(n=1 << 17, fs=FS):
    """A sweep whose length is NOT on the reference's grid (from_wav's mismatch branch, core/impulse_response_estimator
    .py:250-254): the phase formula evaluated for N := n, P = 13, half-octave fade-in."""
    P = 13.0
    ln2p = np.log(2 ** P)
    t = np.arange(n)
    sig = np.sin(np.pi / 2 ** P * n / ln2p * np.exp(t / n * ln2p))
    m = 2 * int(fs * (n / fs / P) * 0.5)
    sig[: m // 2] *= (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(m) / (m - 1)))[: m // 2]
    return 0.8 * sig


def to_pcm32(x):
    return np.clip(np.rint(np.asarray(x) * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32)


def write_pcm32(path, fs, tracks):
    """tracks [n_tracks, n] float -> 32-bit PCM WAV (frames interleaved); returns the int32 frames [n, n_tracks]"""
    from scipy.io import wavfile
    pcm = to_pcm32(np.atleast_2d(tracks)).T
    wavfile.write(path, fs, pcm if pcm.shape[1] > 1 else pcm[:, 0])
    return pcm


def perturbed(test_signal, amp):
    n = np.arange(len(test_signal))
    return np.asarray(test_signal) + amp * np.sin(2 * np.pi * 0.013 * n)


def decaying_ir(seed, n=24000, fs=FS, rt60=0.35, delay=40):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    h = rng.standard_normal(n) * 0.2 * 10 ** (-3.0 * t / rt60)
    h[:delay] = rng.standard_normal(delay) * 1e-5
    h[delay] = 1.0
    h += rng.standard_normal(n) * 10 ** (-95 / 20)
    return h.astype(np.float32).astype(np.float64)


def hrir_set(speakers=("FL", "FR", "FC", "SL", "SR"), n=4096):
    """speaker -> side -> float32-valued response (direct sound + decaying tail), seeded per channel"""
    out = {}
    for i, sp in enumerate(speakers):
        out[sp] = {}
        for j, sd in enumerate(("left", "right")):
            out[sp][sd] = decaying_ir(0x2200 + 2 * i + j, n=n, rt60=0.12, delay=20 + 3 * i + 5 * j)
    return out


def fir_pair(n=600):
    rng = np.random.default_rng(0xF12)
    t = np.arange(n)
    left = rng.standard_normal(n) * np.exp(-t / 60.0)
    right = rng.standard_normal(n) * np.exp(-t / 45.0)
    left[0], right[0] = 1.0, 0.9
    return np.vstack([left, right])


# ---- room-correction folder: 1.0 s sweep @48 kHz, specific FL,FR (one file per ear) + generic room.wav -----------------
ROOM_LEN = 12000


def _room_ir(seed, delay, gain, tilt):
    rng = np.random.default_rng(seed)
    t = np.arange(ROOM_LEN) / FS
    h = rng.standard_normal(ROOM_LEN) * 0.06 * gain * 10 ** (-3.0 * t / 0.22)
    h = np.convolve(h, [1.0, tilt])[:ROOM_LEN]                     # a little spectral tilt per position
    h[: delay + 16] = 0.0
    h[delay] = gain
    return h


def _play(test_signal, h, total, start, noise_seed):
    N = len(test_signal)
    nfft = 1 << int(np.ceil(np.log2(N + ROOM_LEN)))
    y = np.fft.irfft(np.fft.rfft(0.5 * np.asarray(test_signal, dtype=np.float64), nfft) * np.fft.rfft(h, nfft), nfft)
    out = np.zeros(total)
    seg = y[: min(N + ROOM_LEN - 1, total - start)]
    out[start: start + len(seg)] = seg
    out += np.random.default_rng(noise_seed).standard_normal(total) * 10 ** (-82 / 20)
    return out


def room_folder(dir_path, test_signal, with_generic=True, generic_positions=3):
    """Writes room-FL,FR-left.wav, room-FL,FR-right.wav (mono, two columns each), optionally room.wav (one track,
    `generic_positions` columns), room-target.csv and room-mic-calibration.csv into dir_path."""
    N, fs = len(test_signal), FS
    col = 2 * fs + N
    for s, side in enumerate(("left", "right")):
        total = 2 * fs + 2 * col
        track = np.zeros(total)
        for i in range(2):                                          # FL then FR
            h = _room_ir(0xA0 + 2 * i + s, 30 + 7 * i + 11 * s, 1.0 - 0.2 * (i ^ s), 0.3 * (i - s))
            track += _play(test_signal, h, total, 2 * fs + i * col, 0xB0 + 2 * i + s) - 0.0
        write_pcm32(os.path.join(dir_path, f"room-FL,FR-{side}.wav"), fs, track)
    if with_generic:
        total = 2 * fs + generic_positions * col
        track = np.zeros(total)
        for i in range(generic_positions):
            h = _room_ir(0xC0 + i, 25 + 9 * i, 0.9 - 0.1 * i, 0.25 * (i - 1))
            track += _play(test_signal, h, total, 2 * fs + i * col, 0xD0 + i)
        write_pcm32(os.path.join(dir_path, "room.wav"), fs, track)
    f = np.array([10.0, 20, 50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000, 24000])
    tgt = np.array([6.0, 5.5, 4.0, 2.0, 0.5, 0.0, 0.0, -0.5, -1.5, -3.0, -6.0, -7.0])
    cal = np.array([-1.0, -0.6, -0.2, 0.0, 0.1, 0.0, 0.0, 0.3, 0.8, 1.5, 0.5, -0.5])
    with open(os.path.join(dir_path, "room-target.csv"), "w") as fh:
        fh.write("frequency,raw\n" + "".join(f"{a:.1f},{b:.2f}\n" for a, b in zip(f, tgt)))
    with open(os.path.join(dir_path, "room-mic-calibration.csv"), "w") as fh:
        fh.write("frequency,raw\n" + "".join(f"{a:.1f},{b:.2f}\n" for a, b in zip(f, cal)))


def headphone_file(path, test_signal):
    """headphones.wav: two tracks (left cup, right cup), two columns (FL then FR): FL plays into the left cup only, FR
    into the right cup only, each through its own short "headphone" response; the other cup records leakage noise."""
    N, fs = len(test_signal), FS
    col = 2 * fs + N
    total = 2 * fs + 2 * col
    tracks = np.zeros((2, total))
    for i in range(2):                                            # column i: speaker FL (0) / FR (1); loud cup = i
        rng = np.random.default_rng(0xE0 + i)
        h = np.zeros(600)
        h[20 + 3 * i] = 0.8 - 0.1 * i
        h[21 + 3 * i: 300] += rng.standard_normal(279 - 3 * i) * 0.15 * np.exp(-np.arange(279 - 3 * i) / 25.0)
        h = np.concatenate([h, np.zeros(ROOM_LEN - len(h))])
        tracks[i] += _play(test_signal, h, total, 2 * fs + i * col, 0xE8 + i)
        tracks[1 - i] += np.random.default_rng(0xEC + i).standard_normal(total) * 10 ** (-80 / 20)
    write_pcm32(path, fs, tracks)


def reflection_set():
    """speaker -> side -> response for HRIR.calculate_reflection_levels: ordinary room tails, one response so short that
    the late window is cut by its end and one that ends inside the early window, a silent channel, and one whose first
    significant peak is not its largest sample."""
    out = {"FL": {"left": decaying_ir(0x3300, n=12000, rt60=0.25, delay=31), "right": decaying_ir(0x3301, n=12000, rt60=0.30, delay=44)},
           "FR": {"left": decaying_ir(0x3302, n=5000, rt60=0.20, delay=25),            # late window (2 400 .. 7 200) cut at 5 000
                  "right": decaying_ir(0x3303, n=1600, rt60=0.15, delay=12)},          # ends inside the early window
           "FC": {"left": np.zeros(3000), "right": decaying_ir(0x3304, n=9000, rt60=0.22, delay=60)}}
    pre = decaying_ir(0x3305, n=9000, rt60=0.22, delay=300)
    pre[180] = float(np.float32(0.3))                                                  # pre-echo above -18 dB: the first peak (fp32-valued like the rest)
    out["SL"] = {"left": pre, "right": decaying_ir(0x3306, n=9000, rt60=0.18, delay=90)}
    return out
