"""Generate the golden fixtures in tests/golden/ by RUNNING the reference (Impulcifer-pip313).

Runs only in the build container, where the reference checkout is mounted read-only:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python tests/golden/make_goldens.py

Nothing from the reference is copied: this script calls its public functions on seeded inputs and
stores inputs' seeds/shapes and the expected outputs (data only).  Four third-party modules the
reference imports but that are not installed here (soundfile, nnresample, seaborn, bokeh) are
replaced by inert in-memory stand-ins; none of them carries hot-path arithmetic (WAV reading is
done with scipy.io.wavfile, PCM_32 -> /2^31).

Inputs that the HIP path will be checked on are rounded to float32 BEFORE they are given to the
reference, so both sides see bit-identical inputs.
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("IMPULCIFER_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    sf = types.ModuleType("soundfile")

    def _read(path, **_):
        from scipy.io import wavfile
        fs, d = wavfile.read(path)
        scale = {np.dtype("int32"): 2.0 ** 31, np.dtype("int16"): 2.0 ** 15}.get(d.dtype, 1.0)
        return d.astype(np.float64) / scale, fs

    sf.read = _read
    sf.write = lambda *a, **k: None
    sys.modules["soundfile"] = sf
    nn = types.ModuleType("nnresample")
    nn.resample = None
    sys.modules["nnresample"] = nn

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return lambda *a, **k: None

    for name in ("seaborn", "bokeh", "bokeh.plotting", "bokeh.models", "bokeh.palettes", "bokeh.layouts"):
        m = _Any(name)
        m.__path__ = []
        sys.modules[name] = m


def decim(a, step):
    return np.asarray(a)[::step].copy()


def spectrum_probe(y, nprobe=512):
    """|rfft(y)| at log-spaced bins + its max (for peak-normalised spectrum parity)."""
    A = np.abs(np.fft.rfft(y))
    bins = np.unique(np.round(np.geomspace(1, len(A) - 1, nprobe)).astype(np.int64))
    return bins, A[bins], float(A.max())


def main():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _install_stubs()
    from scipy.signal import convolve

    from autoeq.frequency_response import FrequencyResponse
    from core import decay as ref_decay
    from core.audio_io import magnitude_response
    from core.hrir import HRIR
    from core.impulse_response import ImpulseResponse
    from core.impulse_response_estimator import ImpulseResponseEstimator

    # ------------------------------------------------------------------ 1. estimator known answers
    est = {}
    for fs, dur in [(48000, 1.0), (48000, 5.0), (96000, 5.0), (44100, 5.0)]:
        e = ImpulseResponseEstimator(min_duration=dur, fs=fs)
        k = f"e{fs}_{int(dur)}"
        est[k + "_N"] = len(e)
        est[k + "_P"] = float(e.n_octaves)
        est[k + "_low"] = float(e.low)
        est[k + "_duration"] = float(e.duration)
        for nm, arr in (("ts", e.test_signal), ("inv", e.inverse_filter)):
            est[f"{k}_{nm}_head"] = arr[:64].copy()
            est[f"{k}_{nm}_tail"] = arr[-64:].copy()
            est[f"{k}_{nm}_dec"] = decim(arr, 1024)
        base = e.estimate(e.test_signal)
        est[k + "_baseline_peak"] = int(np.argmax(np.abs(base)))
        est[k + "_baseline_val"] = float(base[np.argmax(np.abs(base))])
    np.savez_compressed(os.path.join(OUT, "estimator.npz"), **est)

    # ------------------------------------------------------------------ 2. estimate() cases, fs 48k, 1.0 s
    e = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    N = len(e)
    g = {"N": N}
    base_peak = int(np.argmax(np.abs(e.estimate(e.test_signal))))
    g["baseline_peak"] = base_peak
    for delay in (0, 1, 127, 1000):
        sysir = np.zeros(delay + 1)
        sysir[delay] = 1.0
        rec = convolve(e.test_signal, sysir, mode="full").astype(np.float32).astype(np.float64)
        y = e.estimate(rec)
        g[f"delay{delay}_argmax"] = int(np.argmax(np.abs(y)))
        g[f"delay{delay}_len"] = len(y)
        g[f"delay{delay}_peakval"] = float(y[np.argmax(np.abs(y))])
    # decaying IR of tests/test_estimator_roundtrip.py:31-43
    samples = np.arange(600)
    dir_delay = 31
    dec_ir = np.zeros(600)
    tail = samples[dir_delay:]
    dec_ir[dir_delay:] = 0.08 * np.exp(-(tail - dir_delay) / 110.0) * np.cos(2 * np.pi * 1700 * (tail - dir_delay) / 48000)
    dec_ir[dir_delay] = 1.0
    rec = convolve(e.test_signal, dec_ir, mode="full").astype(np.float32).astype(np.float64)
    y = e.estimate(rec)
    g["decay_ir"] = dec_ir
    g["decay_len"] = len(y)
    g["decay_win"] = y[base_peak - 64: base_peak + 4096].copy()
    g["decay_dec"] = decim(y, 61)
    g["decay_peak_index"] = int(ImpulseResponse(y.copy(), 48000).peak_index())
    b, a, amax = spectrum_probe(y)
    g["decay_bins"], g["decay_amp"], g["decay_amax"] = b, a, amax
    # seeded noise column, L = N + 2 fs
    L = N + 2 * 48000
    x = np.random.default_rng(0).standard_normal(L).astype(np.float32)
    y = e.estimate(x.astype(np.float64))
    g["noise_L"] = L
    g["noise_win"] = y[100000:100000 + 4096].copy()
    g["noise_dec"] = decim(y, 61)
    b, a, amax = spectrum_probe(y)
    g["noise_bins"], g["noise_amp"], g["noise_amax"] = b, a, amax
    g["noise_peak_index"] = int(ImpulseResponse(y.copy(), 48000).peak_index())
    np.savez_compressed(os.path.join(OUT, "estimate.npz"), **g)

    # ------------------------------------------------------------------ 3. peak_index adversarial cases
    cases = {}
    rng = np.random.default_rng(3)

    def add(name, arr, **kw):
        arr = np.asarray(arr, dtype=np.float32)
        cases[name + "_x"] = arr
        cases[name + "_idx"] = int(ImpulseResponse(arr.astype(np.float64), 48000).peak_index(**kw))
        cases[name + "_kw"] = np.array([kw.get("start", 0), -1 if kw.get("end") is None else kw["end"]], dtype=np.int64)

    taps = np.zeros(2048)
    taps[1000], taps[1100], taps[1200] = 1.0, 0.5, 0.25            # tests/test_suite.py:128-148
    add("taps", taps)
    add("later_bigger", np.concatenate([np.zeros(50), [0.2], np.zeros(50), [1.0], np.zeros(20)]))
    add("neg_first", np.concatenate([np.zeros(30), [-0.5], np.zeros(10), [1.0], np.zeros(20)]))
    add("plateau", np.concatenate([np.zeros(10), [0.3, 0.9, 0.9, 0.9, 0.9, 0.2], np.zeros(10)]))
    add("plateau_even", np.concatenate([np.zeros(10), [0.9, 0.9, 0.9, 0.9], np.zeros(10)]))
    add("plateau_to_end", np.concatenate([np.zeros(10), [0.5, 0.2, 1.0, 1.0, 1.0]]))
    add("subthreshold", np.concatenate([np.zeros(10), [0.1], np.zeros(5), [1.0, 1.0]]))   # only peak < 0.12589 -> argmax path? (plateau at end)
    add("all_zero", np.zeros(64))
    add("tiny", np.full(16, 1e-25))
    add("monotone", np.linspace(0, 1, 64))
    add("first_sample_max", np.concatenate([[1.0], np.zeros(20)]))
    add("threshold_edge", np.concatenate([np.zeros(5), [0.12589, 0.0, 0.1258, 0.0, 1.0, 0.0]]))
    nz = rng.standard_normal(4096) * 0.02
    nz[700] = -1.0
    nz[650] = 0.13
    add("noise_neg", nz)
    add("noise_slice", nz, start=660, end=3000)
    add("noise_slice2", nz, start=100, end=690)
    damp = np.exp(-np.arange(4096) / 300.0) * np.sin(2 * np.pi * np.arange(4096) / 37.0)
    add("damped_sine", np.concatenate([np.zeros(333), damp]))
    add("empty", np.zeros(0))
    add("single", np.array([0.7]))
    add("two", np.array([0.1, 0.7]))
    add("three", np.array([0.1, 0.7, 0.2]))
    np.savez_compressed(os.path.join(OUT, "peak_index.npz"), **cases)

    # ------------------------------------------------------------------ 4. decay analysis
    def decaying_sine(fs, duration_s, rt60, freq=1000.0, floor_db=-90.0, seed=0):
        # same recipe as tests/test_impulse_response_decay.py:13-30 (synthetic input, not reference code)
        r = np.random.default_rng(seed)
        n = int(duration_s * fs)
        t = np.arange(n) / fs
        env = 10 ** ((-60.0 / rt60) * t / 20.0)
        return np.cos(2 * np.pi * freq * t) * env + r.standard_normal(n) * 10 ** (floor_db / 20.0)

    d = {}
    for rt60 in (0.3, 0.6, 1.0, 1.5):
        for seed in (0, 11, 22):
            k = f"rt{int(rt60 * 10)}_s{seed}"
            data = decaying_sine(48000, 3.0, rt60, seed=seed).astype(np.float32).astype(np.float64)
            params = ref_decay.decay_params(data, 48000)
            d[k + "_params"] = np.array([float(v) for v in params])
            times = ref_decay.decay_times(data, 48000)
            d[k + "_times"] = np.array([np.nan if v is None else float(v) for v in times])
            for target in (0.2, 0.5):
                try:
                    adj = ref_decay.decay_adjustment_params(data, 48000, target)
                except TypeError:
                    adj = "typeerror"
                kk = f"{k}_t{int(target * 10)}"
                if adj is None:
                    d[kk + "_adj"] = np.array([np.nan] * 4)
                elif isinstance(adj, str):
                    d[kk + "_adj"] = np.array([np.inf] * 4)
                else:
                    d[kk + "_adj"] = np.array([float(v) for v in adj])
                    out = data.copy()
                    ref_decay.apply_decay_window(out, adj)
                    d[kk + "_out_dec"] = decim(out, 37)
                    d[kk + "_out_energy"] = float(np.sum(out ** 2))
    # degenerate inputs (tests/test_impulse_response_decay.py:59-76)
    d["short4_params"] = np.array([float(v) for v in ref_decay.decay_params(np.array([1.0, 0.5, 0.25, 0.1]), 48000)])
    d["empty_params"] = np.array([float(v) for v in ref_decay.decay_params(np.array([]), 48000)])
    np.savez_compressed(os.path.join(OUT, "decay.npz"), **d)

    # ------------------------------------------------------------------ 5. HRIR container ops
    class _Est:
        fs = 48000
        n_octaves = 10

        def __len__(self):
            return 48000 * 6

    def mk(chs, est=None):
        hh = HRIR(est or _Est())
        hh.irs = {sp: {sd: ImpulseResponse(np.asarray(v, dtype=np.float64), 48000) for sd, v in pr.items()}
                  for sp, pr in chs.items()}
        return hh

    hops = {}

    def imp_at(i, n, gain=1.0):
        a = np.zeros(n)
        a[i] = gain
        return a

    # crop_heads, tests/test_dsp_characterization.py:140-156 recipe + right-first + tie
    for name, (l, r, sd_) in {"left_first": (480, 528, 41), "right_first": (600, 520, 42), "tie": (500, 500, 43),
                              "near_start": (10, 30, 44)}.items():
        rr = np.random.default_rng(sd_)
        a = (imp_at(l, 4800) + rr.standard_normal(4800) * 1e-3).astype(np.float32)
        b = (imp_at(r, 4800, 0.8) + rr.standard_normal(4800) * 1e-3).astype(np.float32)
        hops[f"ch_{name}_in_l"], hops[f"ch_{name}_in_r"] = a, b
        hh = mk({"FC": {"left": a, "right": b}})
        hh.crop_heads(head_ms=1)
        hops[f"ch_{name}_out_l"] = hh.irs["FC"]["left"].data.copy()
        hops[f"ch_{name}_out_r"] = hh.irs["FC"]["right"].data.copy()
    # crop_tails, tests/test_dsp_characterization.py:192-217 recipe
    rr = np.random.default_rng(7)
    t = np.arange(48000) / 48000

    def decaying():
        dd = rr.standard_normal(48000) * np.exp(-t * 30.0) * 0.5
        dd[100] = 1.0
        return dd.astype(np.float32)

    chs = {"FL": {"left": decaying(), "right": decaying()}, "FR": {"left": decaying(), "right": decaying()}}
    for sp in chs:
        for sd in chs[sp]:
            hops[f"ct_in_{sp}_{sd}"] = chs[sp][sd]
    hh = mk(chs)
    hops["ct_tail_ind"] = int(hh.crop_tails())
    for sp in chs:
        for sd in chs[sp]:
            hops[f"ct_out_{sp}_{sd}"] = hh.irs[sp][sd].data.copy()
            hops[f"ct_params_{sp}_{sd}"] = np.array([float(v) for v in ref_decay.decay_params(chs[sp][sd].astype(np.float64), 48000)])
    # normalize, tests/test_dsp_stages.py:58-86 recipes
    for name, seed, scales, kw in (("peak", 1, (0.3, 0.1, 0.05, 0.2), dict(peak_target=-0.1)),
                                   ("avg", 2, (0.4, 0.25, 0.15, 0.3), dict(peak_target=None, avg_target=-12.0))):
        rr = np.random.default_rng(seed)
        arrs = [(rr.standard_normal(4096) * s).astype(np.float32) for s in scales]
        hh = mk({"FL": {"left": arrs[0], "right": arrs[1]}, "FR": {"left": arrs[2], "right": arrs[3]}})
        gain = hh.normalize(**kw)
        hops[f"nm_{name}_gain_db"] = float(gain)
        hops[f"nm_{name}_out_FL_left"] = hh.irs["FL"]["left"].data.copy()
    np.savez_compressed(os.path.join(OUT, "hrir_ops.npz"), **hops)

    # ------------------------------------------------------------------ 6. magnitude_response vectors
    m = {}
    for seed, sizes in ((0xA110, (8, 1024, 48000)), (0xA111, (9, 1025, 48001))):
        rr = np.random.default_rng(seed)
        for n in sizes:
            x = rr.standard_normal(n).astype(np.float32)
            f, mag = magnitude_response(x.astype(np.float64), 48000)
            m[f"n{n}_f"] = f
            m[f"n{n}_db"] = mag
    np.savez_compressed(os.path.join(OUT, "magnitude.npz"), **m)

    # ------------------------------------------------------------------ 7. real demo column (shipped golden)
    from scipy.io import wavfile
    demo = os.path.join(REF, "data", "demo")
    e5 = ImpulseResponseEstimator.from_wav(os.path.join(REF, "data", "sweep-6.15s-48000Hz-32bit-2.93Hz-24000Hz.wav"))
    fs, rec = wavfile.read(os.path.join(demo, "room-FC-left.wav"))
    assert rec.dtype == np.int32 and rec.ndim == 1 and fs == 48000
    col = rec[2 * fs: 2 * fs + 2 * fs + len(e5)]
    fs2, resp = wavfile.read(os.path.join(demo, "room-responses.wav"))
    assert resp.shape[1] == 32 and resp.dtype == np.int32
    r = {"N": len(e5), "P": float(e5.n_octaves), "column_i32": col.copy(),
         "responses_fc_left_i32": resp[:, 4].copy(),         # HEXADECAGONAL_TRACK_ORDER index 4 = FC-left
         "responses_len": resp.shape[0]}
    y = e5.estimate(col.astype(np.float64) / 2 ** 31)
    ir = ImpulseResponse(y.copy(), fs, None)
    r["peak_index"] = int(ir.peak_index())
    r["argmax"] = int(np.argmax(np.abs(y)))
    r["win"] = y[r["peak_index"] - 64: r["peak_index"] + 8192].copy()
    b, a, amax = spectrum_probe(y)
    r["bins"], r["amp"], r["amax"] = b, a, amax
    ir.crop_head()
    r["cropped_head"] = ir.data[:21600].copy()
    r["decay_params"] = np.array([float(v) for v in ir.decay_params()])
    np.savez_compressed(os.path.join(OUT, "demo_fc.npz"), **r)

    # ------------------------------------------------------------------ 8. equalization curve -> minimum-phase FIR
    mp = {}
    for fs in (48000, 96000):
        freq = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
        mp[f"fs{fs}_freq"] = freq
        curves = {
            "flat": np.zeros(len(freq)),
            "wavy": 3.0 * np.sin(3.0 * np.log10(freq)),
            "tilt": -4.0 * np.log10(freq / 1000.0) + 2.0 * np.cos(5.0 * np.log10(freq)),
        }
        for name, err in curves.items():
            fr = FrequencyResponse(name=name, frequency=freq.copy(), raw=0, error=err.copy())
            fr.smoothen_heavy_light()
            fr.equalize(max_gain=40, treble_f_lower=10000, treble_f_upper=fs / 2)
            fir = fr.minimum_phase_impulse_response(fs=fs, normalize=False, f_res=5)
            mp[f"fs{fs}_{name}_error"] = err
            mp[f"fs{fs}_{name}_eq"] = fr.equalization.copy()
            mp[f"fs{fs}_{name}_fir"] = fir
    np.savez_compressed(os.path.join(OUT, "minphase.npz"), **mp)

    # ------------------------------------------------------------------ 9. room correction + EQ worker on the real FC-left IR
    from core.parallel_workers import init_equalization_worker, process_equalization_worker
    from core.room_correction import (_open_mic_calibration, _open_room_target, calculate_specific_room_corrections,
                                      discover_room_measurements)
    from scipy.signal.windows import hann as _hann
    fs = 48000                                   # (section 8 reused the name)
    disc = discover_room_measurements(demo)
    target = _open_room_target(e5, demo, None, disc)
    mic = _open_mic_calibration(e5, demo, None, disc)
    n_out = resp.shape[0]
    fo = 2 * int(fs * (len(e5) / fs / e5.n_octaves) * (1 / 24))
    d = r["cropped_head"].copy()
    d[n_out - fo // 2:] *= _hann(fo)[fo // 2:]
    rc = {"target_csv": np.frombuffer(open(os.path.join(demo, "room-target.csv"), "rb").read(), dtype=np.uint8),
          "mic_txt": np.frombuffer(open(os.path.join(demo, "room-mic-calibration.txt"), "rb").read(), dtype=np.uint8),
          "target_raw": target.raw.copy(), "mic_raw": mic.raw.copy(), "frequency": target.frequency.copy()}
    rir = HRIR(e5)
    rir.irs = {"FC": {"left": ImpulseResponse(d.copy(), fs)}}
    rc["fr_raw_initial"] = rir.irs["FC"]["left"].frequency_response().raw.copy()
    frs = calculate_specific_room_corrections(rir, target, mic_calibration=mic, limit=400)
    fr = frs["FC"]["left"]
    rc["fr_raw"], rc["fr_error"], rc["fr_target"] = fr.raw.copy(), fr.error.copy(), fr.target.copy()
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    flat = FrequencyResponse(name="t", frequency=common.copy(), raw=0)
    init_equalization_worker(frs, None, None, None, None, flat, common, fs)
    _, _, fir = process_equalization_worker(("FC", "left"))
    rc["worker_fir"] = fir
    ir = ImpulseResponse(d.copy(), fs)
    ir.equalize(fir)
    rc["equalized_dec"] = decim(ir.data, 5)
    rc["equalized_len"] = len(ir.data)
    np.savez_compressed(os.path.join(OUT, "room_fc.npz"), **rc)

    # ------------------------------------------------------------------ 10. pipeline slice (a18): synthetic FL,FR folder
    import tempfile
    e1 = ImpulseResponseEstimator(min_duration=1.0, fs=48000)
    N1s, fs1 = len(e1), 48000
    sys.path.insert(0, OUT)
    import slice_input
    pcm = slice_input.to_pcm32(slice_input.make_tracks(e1.test_signal, fs1))
    specs = slice_input.SPECS
    sl = {"pcm_specs": np.array(specs), "N": N1s}
    with tempfile.TemporaryDirectory() as tmp:
        wav = os.path.join(tmp, "FL,FR.wav")
        wavfile.write(wav, fs1, pcm.T)
        h = HRIR(e1)
        h.open_recording(wav, ["FL", "FR"])
    order = [("FL", "left"), ("FL", "right"), ("FR", "left"), ("FR", "right")]
    sl["ingest_peaks"] = np.array([h.irs[sp][sd].peak_index() for sp, sd in order])
    sl["ingest_len"] = len(h.irs["FL"]["left"].data)
    h.crop_heads(head_ms=1)
    sl["heads_len"] = np.array([len(h.irs[sp][sd].data) for sp, sd in order])
    for (sp, sd) in order:
        sl[f"heads_{sp}_{sd}"] = h.irs[sp][sd].data[:512].copy()
    sl["tail_ind"] = int(h.crop_tails())
    for (sp, sd) in order:
        sl[f"tails_{sp}_{sd}"] = decim(h.irs[sp][sd].data, 3)
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs1 / 2, f_step=1.01)
    room = {sp: {sd: FrequencyResponse(name="r", frequency=common.copy(), raw=0,
                                       error=2.0 * np.sin(2.5 * np.log10(common) + 0.7 * k))
                 for k, sd in enumerate(("left", "right"), start=2 * j)}
            for j, sp in enumerate(("FL", "FR"))}
    flat = FrequencyResponse(name="t", frequency=common.copy(), raw=0)
    init_equalization_worker(room, None, None, None, None, flat, common, fs1)
    for (sp, sd) in order:
        sl[f"room_error_{sp}_{sd}"] = room[sp][sd].error.copy()
        _, _, fir = process_equalization_worker((sp, sd))
        sl[f"fir_{sp}_{sd}"] = fir
        h.irs[sp][sd].equalize(fir)
    sl["eq_len"] = len(h.irs["FL"]["left"].data)
    sl["norm_gain_db"] = float(h.normalize(peak_target=-0.1))
    for (sp, sd) in order:
        sl[f"final_{sp}_{sd}"] = h.irs[sp][sd].data.copy()
    np.savez_compressed(os.path.join(OUT, "pipeline_slice.npz"), **sl)

    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(f"{fn:24s} {os.path.getsize(os.path.join(OUT, fn)) / 1024:9.1f} KiB")


if __name__ == "__main__":
    main()
