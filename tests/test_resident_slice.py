"""imp_slice: the reference's stage sequence (core/pipeline.py:565-573, 585-601, 647-692, 725-735) for M measurements per
call with no host readback between the stages, against

  * the staged class path (HRIR.open_recording_frames -> crop_heads -> crop_tails -> equalize_channels -> normalize, one
    host readback per stage): BIT FOR BIT - lengths, gains and every fp32 sample;
  * the oracle composition of the same stages (NumPy fp64 restatement of the reference): within the fp32 tolerances of
    tests/test_hip_parity.py.

Sizes: a 1 s sweep layout (fast, several files per measurement), BASELINE C2 (7.1 x 2 ears, 6.15 s sweep, M = 8) and C3
(13 speakers x 2 ears at 96 kHz)."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIME_TOL = 1e-6


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / np.max(np.abs(b)))


def synth_frames(e, speakers, seed, rt60=0.25, level=0.4, noise_db=-85.0, lead_s=2.0):
    """One binaural recording file: [n_frames, 2] int32 PCM, 2 s lead + one column (sweep + 2 s) per speaker; per speaker
    and ear a direct sound plus an exponentially decaying noise tail and a background floor, so that peaks, Lundeby knees
    and the tail crop are decided by the signal."""
    fs, N = e.fs, len(e)
    L = N + 2 * fs
    lead = int(lead_s * fs)
    rng = np.random.default_rng(seed)
    tracks = np.zeros((2, lead + L * len(speakers)))
    room = int(0.6 * fs)
    nfft = 1 << int(np.ceil(np.log2(N + room)))
    S = np.fft.rfft(level * np.asarray(e.test_signal, dtype=np.float64), nfft)
    t = np.arange(room) / fs
    for i in range(len(speakers)):
        for ear in range(2):
            h = rng.standard_normal(room) * 0.04 * 10 ** (-3.0 * t / (rt60 * (1 + 0.1 * rng.random())))
            d0 = 40 + 7 * i + int(rng.integers(0, 30)) * ear + int(rng.integers(0, 9))
            h[: d0 + 30] = 0.0
            h[d0] = 1.0 - 0.3 * ear
            y = np.fft.irfft(S * np.fft.rfft(h, nfft), nfft)[: N + room - 1]
            seg = tracks[ear, lead + i * L: lead + (i + 1) * L]
            seg[: min(len(y), L)] = y[:L]
    tracks += rng.standard_normal(tracks.shape) * 10 ** (noise_db / 20)
    return np.ascontiguousarray(np.clip(np.rint(tracks.T * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32))


def synth_firs(tasks, taps, seed):
    rng = np.random.default_rng(seed)
    firs = rng.standard_normal((len(tasks), taps)) * np.exp(-np.arange(taps) / 300.0) * 0.05
    firs[:, 0] += 1.0
    return {t: firs[i] for i, t in enumerate(tasks)}


def staged_measurement(e, files, firs, decay=None, align=False):
    """the staged class path of one measurement (one host readback per stage)"""
    from impulse_hip.pipeline_slice import run_slice
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return run_slice(e, [((e.fs, fr), sp) for fr, sp in files], firs=firs, decay=decay, align=align)


def oracle_measurement(oe, files, firs, fs, decay=None, align=False):
    """the oracle composition: estimate -> crop_heads -> crop_tails -> FIR 'full' -> [adjust decay] -> normalize (fp64)"""
    from oracle import decay as odecay
    from oracle import hrir as ohrir
    from oracle.scipy_restated import fft_convolve
    N = len(oe)
    irs = {}
    for fr, speakers in files:
        stored = fr.T.astype(np.float64) / 2.0 ** 31
        for sp, sd, col in ohrir.split_recording(stored, speakers, N, fs):
            irs.setdefault(sp, {})[sd] = oe.estimate(col)
    irs = ohrir.crop_heads(irs, fs, head_ms=1)
    if align:
        pairs = (("FL", "FR"), ("SL", "SR"), ("BL", "BR"), ("TFL", "TFR"), ("TSL", "TSR"), ("TBL", "TBR"), ("FC", "FC"), ("WL", "WR"))
        irs = ohrir.align_ipsilateral_all(irs, fs, pairs, segment_ms=30)
        irs = ohrir.align_onset_groups_peak_leftref(irs)
    tail_ind, irs = ohrir.crop_tails(irs, fs, N, oe.n_octaves)
    for sp in irs:
        for sd in irs[sp]:
            irs[sp][sd] = fft_convolve(irs[sp][sd], firs[(sp, sd)], "full")
    if decay is not None:
        for sp in irs:
            if sp in decay:
                for sd in irs[sp]:
                    x = np.array(irs[sp][sd], dtype=np.float64)
                    odecay.apply_decay_window(x, odecay.decay_adjustment_params(x, fs, decay[sp]))
                    irs[sp][sd] = x
    g = ohrir.normalization_gain_db(irs, fs, peak_target=-0.1)
    return tail_ind, g, {sp: {sd: irs[sp][sd] * 10 ** (g / 20) for sd in irs[sp]} for sp in irs}


def assert_same_as_staged(got, want):
    (h1, g1), (h2, g2) = got, want
    assert list(h1.irs) == list(h2.irs)
    # the gain in dB: two correct fp64 transforms of the ear sums (the staged path plans its chirp-z transform for the length
    # it read back, the resident one for the slice's capacity) agree to ~1e-14 dB; what is APPLIED is that gain rounded to
    # fp32, and a gain whose rounding could depend on such a difference is flagged and recomputed by the staged path
    # (IMP_SLICE_GAIN_GUARD) - so the samples below are identical to the last bit
    assert abs(g1 - g2) <= 1e-11
    for sp in h1.irs:
        for sd in ("left", "right"):
            a, b = h1.irs[sp][sd].peek(), h2.irs[sp][sd].peek()
            assert a.shape == b.shape, (sp, sd, a.shape, b.shape)
            assert np.array_equal(a, b), (sp, sd, float(np.max(np.abs(a - b))))


def test_resident_slice_small_layout_bitwise_and_oracle():
    """Three measurements of a two-file layout (FL,FR + FC; 1 s sweep at 48 kHz) in ONE call: bit-identical to the staged
    path of each measurement, and within fp32 tolerance of the oracle composition; the scalars that came back once (peaks,
    crop indices, knees, lengths) are the staged path's integers."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice
    from oracle import estimator as oest
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    oe = oest.Estimator(min_duration=1.0, fs=fs)
    files_spk = [["FL", "FR"], ["FC"]]
    meas = [[synth_frames(e, spk, 1000 * m + 17 * k, rt60=0.2 + 0.05 * m) for k, spk in enumerate(files_spk)] for m in range(3)]
    layout = Layout(e, [(fr.shape[0], 2, spk) for fr, spk in zip(meas[0], files_spk)])
    assert layout.speakers == ["FL", "FR", "FC"] and layout.tracks == 2
    rs = ResidentSlice(e, layout, max_measurements=4)
    assert rs.plan.paired
    firs = synth_firs(layout.tasks, rs.taps, 5)
    rs.set_firs(firs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run(meas)
    # nothing was left to the staged path; the synthetic rooms' knees (1.3 - 1.9 s) are past the default capacity of 1.17 s:
    # the first call came back flagged IMP_SLICE_KEEP_CAP with the knees, the slice was re-made for them and the call repeated
    assert rs.stats == dict(measurements=3, staged=0, regrown=1)
    rows, res = rs.slice.results()
    assert np.all(res["flags"] & 63 == 0)
    for m in range(3):
        files = list(zip(meas[m], files_spk))
        want = staged_measurement(e, files, firs)
        assert_same_as_staged(got[m], want)
        tail_ind, g, o_irs = oracle_measurement(oe, files, firs, fs)
        assert int(res["keep"][m]) == tail_ind
        assert got[m][1] == pytest.approx(g, abs=1e-5)
        for sp in o_irs:
            for sd in o_irs[sp]:
                y = got[m][0].irs[sp][sd].peek()
                assert y.shape == o_irs[sp][sd].shape
                assert rel(y, o_irs[sp][sd]) <= 2 * TIME_TOL, (m, sp, sd)
    rs.close()


def test_resident_slice_flagged_measurements_take_the_staged_path():
    """A decision the device does not take is flagged, never guessed: with a capacity below the crop_tails length
    (IMP_SLICE_KEEP_CAP) and with a gain guard band as wide as the fp32 grid (IMP_SLICE_GAIN_GUARD) every measurement is
    handed to the staged path - and the results are the staged path's."""
    from impulse_hip import _native
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR"]
    meas = [[synth_frames(e, spk, 77 + m)] for m in range(2)]
    layout = Layout(e, [(meas[0][0].shape[0], 2, spk)])
    for kwargs, flag in ((dict(keep_cap=2000), _native.SLICE_KEEP_CAP), (dict(), _native.SLICE_GAIN_GUARD)):
        rs = ResidentSlice(e, layout, max_measurements=2, **kwargs)
        rs.grow_for = lambda rows: False                              # (keep the capacity where the test put it)
        if flag == _native.SLICE_GAIN_GUARD:
            rs._make(rs._cap_for(3 * fs))
            rs.slice.close()
            rs.slice = _native.Slice(rs.plan, layout.pair_offsets, [rs.head] * 2, 2, 32, rs.head, rs.fade, rs.taps, rs.keep_cap, fs,
                                     max_measurements=2, gain_guard_rel=1e-6)
        firs = synth_firs(layout.tasks, rs.taps, 9)
        rs.set_firs(firs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = rs.run(meas)
        _, res = rs.slice.results()
        assert np.all(res["flags"] & flag), res["flags"]
        assert rs.stats == dict(measurements=2, staged=2, regrown=0)
        for m in range(2):
            assert_same_as_staged(got[m], staged_measurement(e, [(meas[m][0], spk)], firs))
        rs.close()


@pytest.mark.parametrize("pairs", [True, False])
def test_resident_slice_mono_plan_and_pcm16(pairs, monkeypatch):
    """The one-channel-per-transform deconvolution plan (what lengths beyond the pair plans take; forced here with
    IMPULSE_HIP_NO_PAIRS) and 16-bit PCM frames through the same resident sequence, bit-identical to the staged path on
    the same plan kind."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice
    if not pairs:
        monkeypatch.setenv("IMPULSE_HIP_NO_PAIRS", "1")
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FC", "FR"]
    fr32 = synth_frames(e, spk, 4242, noise_db=-70.0)
    fr16 = (fr32 >> 16).astype(np.int16)
    layout = Layout(e, [(fr16.shape[0], 2, spk)], dtype=np.int16)
    rs = ResidentSlice(e, layout, max_measurements=1)
    assert rs.plan.paired == pairs
    firs = synth_firs(layout.tasks, rs.taps, 11)
    rs.set_firs(firs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run([[fr16]])
        assert_same_as_staged(got[0], staged_measurement(e, [(fr16, spk)], firs))
    rs.close()


@pytest.mark.parametrize("config", ["c2", "c3"])
def test_resident_slice_full_size(config):
    """BASELINE C2 (7.1 layout x 2 ears, 6.15 s sweep at 48 kHz, M = 8 measurements per call) and C3 (13 speakers x 2 ears
    at 96 kHz, 2 measurements): every measurement bit-identical to the staged class path; one measurement against the oracle
    composition."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice
    from oracle import estimator as oest
    if config == "c2":
        fs, spk, M = 48000, ["FL", "FR", "FC", "BL", "BR", "SL", "SR", "WL"], 8
    else:
        from impulse_hip.constants import TRUEHD_13CH_ORDER
        fs, spk, M = 96000, list(TRUEHD_13CH_ORDER), 2
    e = ImpulseResponseEstimator(min_duration=5.0, fs=fs)
    base = synth_frames(e, spk, 0xC2 if config == "c2" else 0xC3, rt60=0.22)
    meas = [[base]]
    rng = np.random.default_rng(99)
    for m in range(1, M):                                             # same rooms, other noise and level: other peaks' neighbourhoods, knees, gains
        g = 1.0 - 0.07 * m
        noise = rng.standard_normal(base.shape) * (2.0 ** 31 * 10 ** (-80 / 20))
        meas.append([np.clip(np.rint(base * g + noise), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int32)])
    layout = Layout(e, [(base.shape[0], 2, spk)])
    rs = ResidentSlice(e, layout, max_measurements=M)
    assert rs.plan.paired and rs.plan.n1 == (132 if config == "c2" else 288)
    firs = synth_firs(layout.tasks, rs.taps, 21)
    rs.set_firs(firs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run(meas)
    _, res = rs.slice.results()
    assert rs.stats["measurements"] == M
    assert rs.stats["staged"] <= 1, res["flags"]                      # (a guard-band flag is legitimate, and rare)
    for m in range(M):
        assert_same_as_staged(got[m], staged_measurement(e, [(meas[m][0], spk)], firs))
    oe = oest.Estimator(min_duration=5.0, fs=fs)
    tail_ind, g, o_irs = oracle_measurement(oe, [(meas[0][0], spk)], firs, fs)
    assert int(res["keep"][0]) == tail_ind
    assert got[0][1] == pytest.approx(g, abs=1e-5)
    for sp in o_irs:
        for sd in o_irs[sp]:
            assert rel(got[0][0].irs[sp][sd].peek(), o_irs[sp][sd]) <= 2 * TIME_TOL, (sp, sd)
    # the same layout with the decay stage on every row: still the staged path's bits
    rs.set_decay(0.3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run(meas[:2])
    rows, res = rs.slice.results()
    assert set(rows["decay_state"]) <= {1, 2, 3} and np.count_nonzero(rows["decay_state"] == 1) >= 4      # adjusted / already faster
    for m in range(2):
        assert_same_as_staged(got[m], staged_measurement(e, [(meas[m][0], spk)], firs, decay=0.3))
    # ... and with the two alignments of the reference's flow between crop_heads and crop_tails as well
    rs.set_alignment(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run(meas[:2])
    rows, res = rs.slice.results()
    assert np.all(res["flags"] & 256 == 0), res["flags"]
    for m in range(2):
        assert_same_as_staged(got[m], staged_measurement(e, [(meas[m][0], spk)], firs, decay=0.3, align=True))
    rs.close()


def test_resident_slice_decay_stage():
    """The optional stage between equalize and normalize (core/pipeline.py:694-716): per-speaker target RT60s.  The
    resident sequence runs decay_params + decay_times + the window on the equalized rows on the device; the staged path
    does the same through adjust_decay_rows (the rows never visit the host); the reference's worker on host arrays
    (process_decay_worker) is the third form.  FL gets a target faster than its decay (the synthetic rooms' RT60 reads
    1.0 - 1.6 s: adjusted), FR one slower (left alone), FC none (not in the dict).  Resident = staged bit for bit; both
    agree with the oracle composition to the fp32 tolerance of the path and with the host-array form of the stage."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice, _fir_taps
    from impulse_hip.pipeline_slice import run_slice
    from oracle import estimator as oest
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    oe = oest.Estimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR", "FC"]
    decay = {"FL": 0.5, "FR": 5.0}
    meas = [[synth_frames(e, spk, 900 + m, rt60=0.22 + 0.03 * m)] for m in range(3)]
    layout = Layout(e, [(meas[0][0].shape[0], 2, spk)])
    firs = synth_firs(layout.tasks, _fir_taps(fs), 9)
    rs = ResidentSlice(e, layout, max_measurements=3)
    rs.set_firs(firs)
    rs.set_decay(decay)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run(meas)
    rows, res = rs.slice.results()
    assert rs.stats["staged"] == 0, (res["flags"], rows["decay_flags"])
    R = rs.slice.rows
    for m in range(3):
        st = rows["decay_state"][m * R:(m + 1) * R]
        assert list(st) == [1, 1, 2, 2, 0, 0], st
        assert np.all(rows["decay_level_db"][m * R:m * R + 2] < -1.0)            # a real adjustment, not a no-op
        want = staged_measurement(e, [(meas[m][0], spk)], firs, decay=decay)
        assert_same_as_staged(got[m], want)
        # without the stage the adjusted speaker differs, the others change only through the common gain
        plain = staged_measurement(e, [(meas[m][0], spk)], firs)
        tail_a, tail_b = got[m][0].irs["FL"]["left"].peek()[-2000:], plain[0].irs["FL"]["left"].peek()[-2000:]
        assert np.max(np.abs(tail_a)) < 1e-3 * np.max(np.abs(tail_b))           # the late tail is pulled down, by a lot
        tail_ind, g, o_irs = oracle_measurement(oe, [(meas[m][0], spk)], firs, fs, decay=decay)
        assert int(res["keep"][m]) == tail_ind
        assert got[m][1] == pytest.approx(g, abs=1e-5)
        for sp in o_irs:
            for sd in o_irs[sp]:
                assert rel(got[m][0].irs[sp][sd].peek(), o_irs[sp][sd]) <= 2 * TIME_TOL, (sp, sd)
    # the host-array form of the stage (the reference's worker as the classes run it when a response is on the host)
    hrir, _ = run_slice(e, [((fs, meas[0][0]), spk)], firs=firs, decay=None, peak_target=-0.1)
    from impulse_hip.parallel_workers import process_decay_worker
    from impulse_hip.decay import adjust_decay_rows
    ir = hrir.irs["FL"]["left"]
    gain_free = ir.peek()
    _, _, host_form = process_decay_worker(("FL", "left", gain_free.copy(), fs, 0.5))
    adjust_decay_rows([ir._row], fs, [0.5])
    assert np.array_equal(ir.peek(), host_form)
    # a target, then none: the stage is off again and the results are those of the plain sequence
    rs.set_decay(None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        again = rs.run(meas[:1])
    assert_same_as_staged(again[0], staged_measurement(e, [(meas[0][0], spk)], firs))
    assert np.all(rs.slice.results()[0]["decay_state"] == 0)
    rs.close()


def test_resident_slice_alignment_stage():
    """The alignments `_stage_crop_and_align` runs between crop_heads and crop_tails (core/pipeline.py:593-597):
    align_ipsilateral_all (lags from the cross-correlation of 30 ms segments, K10) and align_onset_groups_peak_leftref.  A
    5-speaker layout whose arrival times differ per speaker, so that both alignments really shift rows (delays and
    advances).  The resident sequence decides lags, leader peaks and shifts on the device; the staged path does the same
    on device rows (xcorr_argmax_device, shift_rows); both equal the oracle composition of the same stages; the shifts are
    the oracle's integers; the host-array form of the two methods (responses brought to the host first) gives the same
    samples."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice, _fir_taps
    from impulse_hip.pipeline_slice import run_slice
    from oracle import estimator as oest
    from oracle import hrir as ohrir
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    oe = oest.Estimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR", "FC", "SL", "SR"]
    meas = [[synth_frames(e, spk, 1200 + m, rt60=0.2 + 0.02 * m)] for m in range(3)]
    layout = Layout(e, [(meas[0][0].shape[0], 2, spk)])
    firs = synth_firs(layout.tasks, _fir_taps(fs), 13)
    rs = ResidentSlice(e, layout, max_measurements=3)
    rs.set_firs(firs)
    rs.set_alignment(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run(meas)
    rows, res = rs.slice.results()
    assert rs.stats["staged"] == 0, res["flags"]
    assert np.any(rows["shift_ipsilateral"] > 0) and np.any(rows["shift_onset"] != 0)
    R = rs.slice.rows
    N = len(oe)
    for m in range(3):
        want = staged_measurement(e, [(meas[m][0], spk)], firs, align=True)
        assert_same_as_staged(got[m], want)
        # the shifts against the oracle's own alignment of the oracle's cropped responses
        stored = meas[m][0].T.astype(np.float64) / 2.0 ** 31
        irs = {}
        for sp, sd, col in ohrir.split_recording(stored, spk, N, fs):
            irs.setdefault(sp, {})[sd] = oe.estimate(col)
        irs = ohrir.crop_heads(irs, fs, head_ms=1)
        seg = int(fs * 30 / 1000)
        lag_fl = ohrir.ipsilateral_lag(irs["FL"]["left"], irs["FR"]["right"], seg)
        d = rows["shift_ipsilateral"][m * R:(m + 1) * R]
        assert (int(d[2]), int(d[3])) == ((lag_fl, lag_fl) if lag_fl > 0 else (0, 0))          # FR delayed by a positive lag
        assert (int(d[0]), int(d[1])) == ((-lag_fl, -lag_fl) if lag_fl < 0 else (0, 0))
        tail_ind, g, o_irs = oracle_measurement(oe, [(meas[m][0], spk)], firs, fs, align=True)
        assert int(res["keep"][m]) == tail_ind
        assert got[m][1] == pytest.approx(g, abs=1e-5)
        for sp in o_irs:
            for sd in o_irs[sp]:
                assert rel(got[m][0].irs[sp][sd].peek(), o_irs[sp][sd]) <= 2 * TIME_TOL, (sp, sd)
    # the host-array form of the two methods on the same cropped responses
    from impulse_hip.hrir import HRIR
    from impulse_hip.constants import IPSILATERAL_PAIRS
    a, b = HRIR(e), HRIR(e)
    for h in (a, b):
        h.open_recording_frames(fs, meas[0][0], spk)
        h.crop_heads(head_ms=1)
    for sp in spk:
        for sd in ("left", "right"):
            b.irs[sp][sd].data = b.irs[sp][sd].data                          # to the host
    for h in (a, b):
        h.align_ipsilateral_all(speaker_pairs=list(IPSILATERAL_PAIRS), segment_ms=30)
        h.align_onset_groups_peak_leftref()
    for sp in spk:
        for sd in ("left", "right"):
            assert a.irs[sp][sd]._data is None and a.irs[sp][sd]._row is not None          # never left the device
            assert np.array_equal(a.irs[sp][sd].peek(), b.irs[sp][sd].data), (sp, sd)
    # alignment and decay together, then both off again
    rs.set_decay({"FL": 0.5})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        both = rs.run(meas[:1])
    assert_same_as_staged(both[0], staged_measurement(e, [(meas[0][0], spk)], firs, decay={"FL": 0.5}, align=True))
    rs.set_decay(None)
    rs.set_alignment(False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = rs.run(meas[:1])
    assert_same_as_staged(plain[0], staged_measurement(e, [(meas[0][0], spk)], firs))
    assert np.all(rs.slice.results()[0]["shift_onset"] == 0)
    rs.close()


def test_alignment_guard_flags_take_the_staged_path():
    """A delayed row whose first sample is not zero (no head fade: head_ms = 0) could gain a first peak from its new zero
    neighbour: the device flags the measurement (IMP_SLICE_ALIGN_GUARD) instead of assuming peak_index(shifted) = shift +
    peak_index, and the staged path (which searches the materialised rows) gives the result."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice, _fir_taps
    from impulse_hip import _native
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR", "SL", "SR"]
    meas = [[synth_frames(e, spk, 1300 + m)] for m in range(2)]
    layout = Layout(e, [(meas[0][0].shape[0], 2, spk)])
    firs = synth_firs(layout.tasks, _fir_taps(fs), 3)
    rs = ResidentSlice(e, layout, max_measurements=2, head_ms=0)
    rs.set_firs(firs)
    rs.set_alignment(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = rs.run(meas)
    rows, res = rs.slice.results()
    assert np.any(rows["shift_ipsilateral"] > 0)
    assert np.all(res["flags"] & _native.SLICE_ALIGN_GUARD), res["flags"]
    assert rs.stats["staged"] == 2
    from impulse_hip.pipeline_slice import run_slice
    for m in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = run_slice(e, [((fs, meas[m][0]), spk)], firs=firs, head_ms=0, align=True)
        assert_same_as_staged(got[m], want)
    rs.close()
    # a speaker in two ipsilateral pairs is refused (the searches of a measurement run as one batch)
    rs = ResidentSlice(e, layout, max_measurements=1)
    with pytest.raises(_native.NativeError):
        rs.slice.set_alignment([(0, 1), (1, 2)], [-1, -1, -1, -1], 0, 1440)
    with pytest.raises(_native.NativeError):
        rs.slice.set_alignment([(0, 1)], [-1, -1, -1, -1], 0, 9000)        # two segments must fit the lag search's LDS
    rs.close()


def test_shift_and_lag_search_of_device_rows_edge_cases():
    """imp_shift_rows_device against ImpulseResponse.shift's NumPy form (core/impulse_response.py:92-108) for delays and
    advances up to and beyond the row length and empty rows; imp_xcorr_argmax_device against imp_xcorr_argmax on the rows'
    float64 copies for equal and unequal segment lengths, the 30 ms segments of 48 - 192 kHz and lag slices that end inside
    a wave."""
    from impulse_hip import _native
    ctx = _native.default_context()
    rng = np.random.default_rng(5)
    lens = np.array([1, 2, 63, 64, 65, 1000, 4097, 0, 777], dtype=np.int64)
    shifts = np.array([0, 1, -1, 70, -64, 999, -5000, 3, -776], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens + 5)[:-1]]).astype(np.int64)
    flat = rng.standard_normal(int(offs[-1] + lens[-1] + 5)).astype(np.float32)
    d_src, d_dst = ctx.malloc(flat.nbytes), ctx.malloc(flat.nbytes)
    ctx.h2d(d_src, flat)
    ctx.memset(d_dst, 0xFF, flat.nbytes)
    ctx.shift_rows_device(d_src, offs, lens, shifts, d_dst, offs)
    out = np.empty_like(flat)
    ctx.synchronize()
    ctx.d2h(out, d_dst)
    for o, n, sh in zip(offs, lens, shifts):
        x = flat[o:o + n].astype(np.float64)
        if sh > 0:
            want = np.concatenate((np.zeros(sh), x))[:n]
        elif sh < 0:
            t = x[-sh:]
            want = np.pad(t, (0, n - len(t))) if len(t) < n else t
        else:
            want = x
        assert np.array_equal(out[o:o + n].astype(np.float64), want), (int(n), int(sh))
    ctx.free(d_dst)
    # lag searches
    segs = [(1440, 1440), (5760, 5760), (1, 1), (7, 300), (300, 7), (1025, 1024), (255, 257), (2880, 2880), (64, 64)]
    a_len = np.array([p[0] for p in segs], dtype=np.int64)
    b_len = np.array([p[1] for p in segs], dtype=np.int64)
    a_off = np.concatenate([[0], np.cumsum(a_len)[:-1]]).astype(np.int64)
    b_off = (a_off[-1] + a_len[-1] + np.concatenate([[0], np.cumsum(b_len)[:-1]])).astype(np.int64)
    rows = (rng.standard_normal(int(b_off[-1] + b_len[-1])) * np.exp(-np.arange(int(b_off[-1] + b_len[-1])) % 997 / 200.0)).astype(np.float32)
    d_rows = ctx.malloc(rows.nbytes)
    ctx.h2d(d_rows, rows)
    arg, val = ctx.xcorr_argmax_device(d_rows, a_off, a_len, b_off, b_len)
    ctx.free(d_rows)
    ctx.free(d_src)
    a64 = [rows[o:o + n].astype(np.float64) for o, n in zip(a_off, a_len)]
    b64 = [rows[o:o + n].astype(np.float64) for o, n in zip(b_off, b_len)]
    arg_h, val_h = ctx.xcorr_argmax(a64, b64)
    assert np.array_equal(arg, arg_h) and np.array_equal(val, val_h)
    for k, (x, y) in enumerate(zip(a64, b64)):
        corr = np.correlate(x, y, mode="full")
        assert int(arg[k]) == int(np.argmax(corr)), k
        assert abs(val[k] - corr.max()) <= 1e-12 * max(1.0, abs(corr.max())), k


def test_resident_slice_random_layouts_against_the_oracle():
    """Ten seeded random cases (1 - 4 speakers incl. FL in random order, alignments on, a decay dict in half of them,
    different decay times, levels and noise floors) through the resident sequence and through the ORACLE composition of the
    reference's stages: the discrete decisions agree (crop_tails length exactly; peaks, lags and knees through it) and the
    samples agree to the fp32 tolerance of the path."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice, _fir_taps
    from oracle import estimator as oest
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    oe = oest.Estimator(min_duration=1.0, fs=fs)
    rng = np.random.default_rng(2024)
    others = ["FR", "FC", "BL", "BR", "SL", "SR", "WL", "TFL", "TFR"]
    checked = 0
    for case in range(10):
        spk = ["FL"] + [others[i] for i in rng.permutation(len(others))[:int(rng.integers(0, 4))]]
        rng.shuffle(spk)
        decay = {sp: float(rng.choice([0.3, 0.6, 2.0])) for sp in spk if rng.integers(0, 2)} if case % 2 else None
        decay = decay or None
        fr = synth_frames(e, spk, int(rng.integers(1 << 30)), rt60=float(rng.uniform(0.12, 0.4)), level=float(rng.uniform(0.1, 0.7)),
                          noise_db=float(rng.uniform(-100, -70)))
        layout = Layout(e, [(fr.shape[0], 2, spk)])
        firs = synth_firs(layout.tasks, _fir_taps(fs), int(rng.integers(1 << 30)))
        rs = ResidentSlice(e, layout, max_measurements=1)
        rs.set_firs(firs)
        rs.set_alignment(True)
        rs.set_decay(decay)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                got = rs.run([[fr]])
            except TypeError:                              # no decay time defined: the reference raises too (core/decay.py:367)
                rs.close()
                continue
        _, res = rs.slice.results()
        tail_ind, g, o_irs = oracle_measurement(oe, [(fr, spk)], firs, fs, decay=decay, align=True)
        assert list(got[0][0].irs) == list(o_irs), (case, spk)
        assert len(got[0][0].irs[spk[0]]["left"].peek()) == tail_ind + _fir_taps(fs) - 1, (case, spk)
        assert got[0][1] == pytest.approx(g, abs=1e-5), (case, spk)
        for sp in o_irs:
            for sd in o_irs[sp]:
                assert rel(got[0][0].irs[sp][sd].peek(), o_irs[sp][sd]) <= 2 * TIME_TOL, (case, sp, sd)
        checked += 1
        rs.close()
    assert checked >= 8


def test_slice_refuses_bad_arguments_loudly():
    """The C entry points of the slice say what is wrong instead of computing something: no FIRs yet, a decay target that is
    not a positive time, more measurements than the slice was made for, an output pitch below keep_cap + taps - 1, a pack
    call that does not match the last execute, an onset leader out of range."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, ResidentSlice, _fir_taps
    from impulse_hip import _native
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR"]
    fr = synth_frames(e, spk, 77)
    layout = Layout(e, [(fr.shape[0], 2, spk)])
    rs = ResidentSlice(e, layout, max_measurements=2)
    ctx = rs.ctx
    sl = rs.slice
    d_rec = ctx.malloc(2 * layout.samples * 4)
    ctx.h2d(d_rec, layout.pack([fr]))
    ctx.h2d(d_rec + layout.samples * 4, layout.pack([fr]))
    d_out = ctx.malloc(2 * sl.rows * rs.out_pitch * 4)
    with pytest.raises(_native.NativeError, match="no FIRs"):
        sl.execute_device(d_rec, layout.samples, 1, d_out, rs.out_pitch)
    rs.set_firs(synth_firs(layout.tasks, _fir_taps(fs), 1))
    with pytest.raises(_native.NativeError, match="capacity"):
        sl.execute_device(d_rec, layout.samples, 3, d_out, rs.out_pitch)
    with pytest.raises(_native.NativeError, match="out_pitch"):
        sl.execute_device(d_rec, layout.samples, 1, d_out, sl.out_len_max - 1)
    with pytest.raises(_native.NativeError, match="positive time"):
        sl.set_decay([0.3, -1.0, np.nan, np.nan])
    with pytest.raises(ValueError):
        sl.set_decay([0.3])
    with pytest.raises(_native.NativeError, match="out of range"):
        sl.set_alignment([(0, 1)], [5, -1], 0, 1440)
    with pytest.raises(_native.NativeError, match="reference pair"):
        sl.set_alignment([(0, 1)], [-1, -1], 2, 1440)
    sl.execute_device(d_rec, layout.samples, 2, d_out, rs.out_pitch)
    d_pk = ctx.malloc(2 * sl.rows * sl.out_len_max * 8)
    with pytest.raises(_native.NativeError, match="last call"):
        sl.pack_f64(d_out, rs.out_pitch, 1, d_pk, sl.rows * sl.out_len_max)
    with pytest.raises(_native.NativeError, match="too small"):
        sl.pack_f64(d_out, rs.out_pitch, 2, d_pk, sl.rows * sl.out_len_max - 1)
    sl.pack_f64(d_out, rs.out_pitch, 2, d_pk, sl.rows * sl.out_len_max)
    rows, meas = sl.results()
    assert len(meas) == 2 and np.array_equal(meas["out_len"][0:1], meas["out_len"][1:2])       # the same recording twice
    for p in (d_rec, d_out, d_pk):
        ctx.free(p)
    rs.close()
    with pytest.raises(ValueError, match="PCM"):
        Layout(e, [(fr.shape[0], 2, spk)], dtype=np.float32)
    with pytest.raises(ValueError, match="twice"):
        Layout(e, [(fr.shape[0], 2, spk), (fr.shape[0], 2, ["FL", "FC"])])


def test_decay_times_of_device_rows_have_the_bits_of_their_float64_copies():
    from impulse_hip import _native
    from impulse_hip.decay import decay_params, decay_times
    ctx = _native.default_context()
    fs = 48000
    rng = np.random.default_rng(77)
    rows = []
    for k in range(4):
        n = 30000 + 5000 * k
        t = np.arange(n) / fs
        x = rng.standard_normal(n) * 10 ** (-3.0 * t / (0.15 + 0.05 * k)) + rng.standard_normal(n) * 1e-4
        x[:200] = 0
        x[200] = 2.0
        rows.append(x.astype(np.float32))
    lens = np.array([len(r) for r in rows], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    flat = np.concatenate(rows)
    d = ctx.malloc(flat.nbytes)
    ctx.h2d(d, flat)
    par = [decay_params(r.astype(np.float64), fs) for r in rows]
    got = ctx.decay_times_device(d, offs, lens, [p[0] for p in par], [p[1] for p in par], [p[2] for p in par], [p[3] for p in par], fs)
    ctx.free(d)
    for k, r in enumerate(rows):
        want = decay_times(r.astype(np.float64), fs, *par[k])
        assert any(w is not None for w in want)
        for a, b in zip(got[k], want):
            assert (np.isnan(a) and b is None) or a == b, (k, got[k], want)


def test_run_slice_jobs_workers_overlap_and_order():
    """A job of five measurements through run_slice_jobs (two worker threads, a context and a one-measurement slice each):
    results in job order, each identical to the staged path of its measurement, responses on the host as float64."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, run_slice_jobs
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR"]
    meas = [[synth_frames(e, spk, 500 + m, rt60=0.18 + 0.02 * m)] for m in range(5)]
    layout = Layout(e, [(meas[0][0].shape[0], 2, spk)])
    from impulse_hip.resident_slice import _fir_taps
    firs = synth_firs(layout.tasks, _fir_taps(fs), 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = run_slice_jobs(e, layout, meas, firs, workers=2)
    assert len(got) == 5
    for m in range(5):
        assert got[m][0].irs["FL"]["left"]._data is not None and got[m][0].irs["FL"]["left"].data.dtype == np.float64
        assert_same_as_staged(got[m], staged_measurement(e, [(meas[m][0], spk)], firs))


def test_slice_pipeline_order_regrowth_recycling_and_errors():
    """The three-stage runner (upload | compute | download, a thread and a stream each).  A job of six measurements, one of
    them with responses longer than the slice was first sized for (the slice and the float64 hand-over ring are re-made
    in mid-job): results in job order, float64 on the host, each identical to the staged path.  A second job after the
    first one's results were dropped pins no new memory (the blocks are recycled); a job whose third recording has the
    wrong shape raises and leaves the runner usable; responses left on the device (to_host=False) are the same samples."""
    import gc
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, SlicePipeline, _fir_taps
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR"]
    rt = [0.18, 0.2, 0.22, 0.75, 0.24, 0.19]
    meas = [[synth_frames(e, spk, 700 + m, rt60=rt[m], noise_db=-100.0 if rt[m] > 0.5 else -85.0)] for m in range(6)]
    layout = Layout(e, [(meas[0][0].shape[0], 2, spk)])
    firs = synth_firs(layout.tasks, _fir_taps(fs), 5)
    runner = SlicePipeline(e, layout, keep_cap=4000)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = runner.run(meas, firs)
            want = [staged_measurement(e, [(meas[m][0], spk)], firs) for m in range(6)]
            assert len(got) == 6
            for m in range(6):
                ir = got[m][0].irs["FL"]["left"]
                assert ir._data is not None and ir.data.dtype == np.float64 and ir.data.flags["C_CONTIGUOUS"]
                assert_same_as_staged(got[m], want[m])
            lens = {len(got[m][0].irs["FL"]["left"].data) for m in range(6)}
            assert len(lens) > 1 and runner.rs.stats["regrown"] >= 1, (lens, runner.rs.stats)
            pinned = runner.pool.allocations
            assert pinned >= 1
            keep = got[3][0].irs["FR"]["right"].data           # a view outliving the other results keeps ITS block only
            snapshot = keep.copy()
            del got, ir
            gc.collect()
            again = runner.run(meas, firs)
            assert runner.pool.allocations <= pinned + 1, (runner.pool.allocations, pinned)
            assert np.array_equal(keep, snapshot)
            for m in range(6):
                assert_same_as_staged(again[m], want[m])
            bad = [list(x) for x in meas]
            bad[2] = [meas[2][0][:-1]]
            with pytest.raises(ValueError):
                runner.run(bad, firs)
            on_dev = runner.run(meas[:3], firs, to_host=False)
            for m in range(3):
                assert on_dev[m][0].irs["FL"]["left"]._data is None
                assert_same_as_staged(on_dev[m], want[m])
            # two callers at once: each gets its own job's results, complete and in order
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=2) as tp:
                fa, fb = tp.submit(runner.run, meas[:4], firs), tp.submit(runner.run, meas[2:], firs)
                ra, rb = fa.result(), fb.result()
            for m in range(4):
                assert_same_as_staged(ra[m], want[m])
                assert_same_as_staged(rb[m], want[2 + m])
    finally:
        runner.close()


def test_slice_fleet_folded_on_one_device():
    """SliceFleet: a SlicePipeline per entry of the device list, the job's measurements cut into contiguous blocks.  Folded
    here (two pipelines on device 0, as IMPULSE_HIP_DEVICES=0,0 would give): seven measurements come back in job order, each
    identical to the staged path, with FIRs that a design left on the device (handed over in place on their own device;
    the host-taps route for another device is the same call with a device mismatch, covered by _firs_for below)."""
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.resident_slice import Layout, SliceFleet, _firs_for, _fir_taps
    from impulse_hip import _native
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR"]
    meas = [[synth_frames(e, spk, 1500 + m, rt60=0.18 + 0.01 * m)] for m in range(7)]
    layout = Layout(e, [(meas[0][0].shape[0], 2, spk)])
    firs = synth_firs(layout.tasks, _fir_taps(fs), 17)
    fleet = SliceFleet(e, layout, devices=[0, 0])
    try:
        assert len(fleet.pipes) == 2
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = fleet.run(meas, firs, align=True)
            assert len(got) == 7
            for m in range(7):
                assert_same_as_staged(got[m], staged_measurement(e, [(meas[m][0], spk)], firs, align=True))
            assert fleet.pipes[0].rs.stats["measurements"] == 4 and fleet.pipes[1].rs.stats["measurements"] == 3
            one = fleet.run(meas[:1], firs)                       # fewer measurements than devices: one pipeline takes them
            assert_same_as_staged(one[0], staged_measurement(e, [(meas[0][0], spk)], firs))
    finally:
        fleet.close()
    # FIRs designed on one device, a slice on another: host taps, the same numbers

    class OtherDevice:
        device = 1
    ctx = _native.default_context()
    mat = np.stack([firs[t] for t in layout.tasks])
    d = ctx.malloc(mat.nbytes)
    ctx.h2d(d, mat)
    batch = _native.DeviceFirs(ctx, d, mat.shape[0], mat.shape[1])
    same = _firs_for(batch, layout, ctx)
    assert same is batch
    moved = _firs_for(batch, layout, OtherDevice())
    assert isinstance(moved, np.ndarray) and np.array_equal(moved, mat)
    moved = _firs_for({t: r for t, r in zip(layout.tasks, batch.rows())}, layout, OtherDevice())
    assert all(np.array_equal(moved[t], firs[t]) for t in layout.tasks)
    batch.close()


def test_job_from_wav_files_on_disk(tmp_path):
    """WavMeasurements: a job whose measurements are PCM WAV files (one binaural file per measurement here); the runner's
    upload stage reads measurement i + 1 while measurement i computes.  Results = the staged class path opening the same
    files (HRIR.open_recording), PCM_16 and PCM_32; a float file is refused with the reason."""
    from impulse_hip.audio_io import write_wav, write_wav_frames
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.pipeline_slice import run_slice
    from impulse_hip.resident_slice import SlicePipeline, WavMeasurements, _fir_taps
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR", "FC"]
    for bits in (32, 16):
        paths = []
        for m in range(4):
            fr = synth_frames(e, spk, 1700 + m + bits, rt60=0.2 + 0.02 * m)
            path = str(tmp_path / f"m{bits}_{m}.wav")
            write_wav_frames(path, fs, fr if bits == 32 else (fr >> 16).astype(np.int16), bits)
            paths.append([path])
        job = WavMeasurements(paths, fs=fs)
        assert len(job) == 4 and len(job[1:]) == 3
        layout = job.layout(e, [spk])
        assert layout.dtype == (np.int32 if bits == 32 else np.int16)
        firs = synth_firs(layout.tasks, _fir_taps(fs), 23)
        runner = SlicePipeline(e, layout)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = runner.run(job, firs, align=True)
                for m in range(4):
                    want = run_slice(e, [(paths[m][0], spk)], firs=firs, align=True)
                    assert_same_as_staged(got[m], want)
        finally:
            runner.close()
    # measurement directories as the reference lays them out: <speaker list>.wav files, speakers from the names
    dirs = []
    for m in range(2):
        d = tmp_path / f"measurement{m}"
        d.mkdir()
        write_wav_frames(str(d / "FL,FR.wav"), fs, synth_frames(e, ["FL", "FR"], 1900 + m), 32)
        write_wav_frames(str(d / "FC.wav"), fs, synth_frames(e, ["FC"], 1950 + m), 32)
        (d / "notes.txt").write_text("not a recording")
        dirs.append(str(d))
    job, spk_per_file = WavMeasurements.from_dirs(dirs, fs=fs)
    assert sorted(map(tuple, spk_per_file)) == [("FC",), ("FL", "FR")] and len(job) == 2
    layout = job.layout(e, spk_per_file)
    firs = synth_firs(layout.tasks, _fir_taps(fs), 29)
    runner = SlicePipeline(e, layout)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = runner.run(job, firs, align=True)
            for m in range(2):
                want = run_slice(e, [(p, sp) for p, sp in zip(job.files[m], spk_per_file)], firs=firs, align=True)
                assert_same_as_staged(got[m], want)
    finally:
        runner.close()
    # the whole flow for the directories: FIRs designed once from the (default, flat) target, one call
    from impulse_hip.pipeline_slice import run_measurement_dirs
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        whole = run_measurement_dirs(e, dirs, decay={"FC": 0.4})
        for m in range(2):
            want = run_slice(e, [(p, sp) for p, sp in zip(job.files[m], spk_per_file)], decay={"FC": 0.4}, align=True)
            assert_same_as_staged(whole[m], want)
            assert whole[m][0].irs["FC"]["left"].data.dtype == np.float64
            # ... down to the files the reference's flow ends with (core/pipeline.py:865-876): hrir.wav and hesuvi.wav
            from impulse_hip.constants import HESUVI_TRACK_ORDER
            for name, order in (("hrir", None), ("hesuvi", HESUVI_TRACK_ORDER)):
                a, b = str(tmp_path / f"{name}_{m}_resident.wav"), str(tmp_path / f"{name}_{m}_staged.wav")
                whole[m][0].write_wav(a, track_order=order)
                want[0].write_wav(b, track_order=order)
                assert open(a, "rb").read() == open(b, "rb").read(), (name, m)
    (tmp_path / "measurement1" / "SL,SR.wav").write_bytes(b"")
    with pytest.raises(ValueError, match="differ"):
        WavMeasurements.from_dirs(dirs)
    odd = str(tmp_path / "float.wav")
    write_wav(odd, fs, np.zeros((2, 1000)), bit_depth=24)
    with pytest.raises(ValueError, match="PCM"):
        WavMeasurements([[odd]])[0]


def test_firs_left_on_the_device_are_the_same_firs():
    """process_equalization_batch(on_device=True) leaves the minimum-phase FIRs on the device (core/pipeline.py:690-691 hands
    every FIR straight to ImpulseResponse.equalize: they never need to visit the host): the rows are the host version's bits,
    ConvPlan.set_filters_device forms the same spectra as set_filters, HRIR.equalize_channels and the resident slice take
    the batch where it is - and the slice's results do not change."""
    from impulse_hip import ConvPlan, _native
    from impulse_hip.frequency_response import FrequencyResponse
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    from impulse_hip.parallel_workers import process_equalization_batch
    from impulse_hip.pipeline_slice import run_slice
    from impulse_hip.resident_slice import Layout, ResidentSlice
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR", "FC"]
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    target = FrequencyResponse(name="target", frequency=common.copy(), raw=0)
    rng = np.random.default_rng(8)
    room = {sp: {sd: FrequencyResponse("r", frequency=common.copy(), raw=0, error=np.cumsum(rng.standard_normal(len(common))) * 0.2)
                 for sd in ("left", "right")} for sp in spk}
    tasks = [(sp, sd) for sp in spk for sd in ("left", "right")]
    host = process_equalization_batch(tasks, room, None, None, None, None, target, common, fs)
    dev = process_equalization_batch(tasks, room, None, None, None, None, target, common, fs, on_device=True)
    assert all(isinstance(f, _native.DeviceFir) for _, _, f in dev)
    for (sp, sd, fh), (sp2, sd2, fd) in zip(host, dev):
        assert (sp, sd) == (sp2, sd2) and len(fd) == len(fh) == 9600
        assert np.array_equal(np.asarray(fd), fh)
    batch = dev[0][2].batch
    # the spectra of a plan: from the device batch == from the host matrix
    ctx = batch.ctx
    x = rng.standard_normal((6, 30000)).astype(np.float32)
    p_host = ConvPlan(ctx, np.stack([f for _, _, f in host]), 30000, "full", ws_channels=6)
    p_dev = ConvPlan(ctx, None, 30000, "full", ws_channels=6, empty_M=9600, n_filters=6)
    p_dev.set_filters_device(batch.ready().ptr, 9600)
    assert np.array_equal(p_host.execute(x), p_dev.execute(x))
    p_host.close()
    p_dev.close()
    # the staged path and the resident slice with the device batch == with the host FIRs
    frames = synth_frames(e, spk, 31337)
    layout = Layout(e, [(frames.shape[0], 2, spk)])
    f_host = {(sp, sd): f for sp, sd, f in host}
    f_dev = {(sp, sd): f for sp, sd, f in dev}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = run_slice(e, [((fs, frames), spk)], firs=f_host)
        got = run_slice(e, [((fs, frames), spk)], firs=f_dev)
        assert_same_as_staged(got, want)
        rs = ResidentSlice(e, layout, max_measurements=1)
        rs.set_firs(f_dev)
        assert isinstance(rs.firs, _native.DeviceFirs)
        assert_same_as_staged(rs.run([[frames]])[0], want)
        rs.close()


def test_single_process_multi_device_fan_out_folded(monkeypatch):
    """IMPULSE_HIP_DEVICES = "0,0,0": three contexts folded onto the one device of this box - what a process on a multi-GPU
    node does with "0,1,2".  estimate_batch shards its rows (pairs kept together) and open_recording_frames the columns of
    the recording over the contexts, one host thread each, every context's plan holding a COPY of the root's inverse-sweep
    spectrum (imp_plan_copy_spectrum); the deconvolved rows are gathered on the root device.  Results: bit-identical to
    the single-context run, for the one-channel-per-transform plan (estimate_batch) and the pair plan (recording ingest)."""
    from impulse_hip import _native
    from impulse_hip.hrir import HRIR
    from impulse_hip.impulse_response_estimator import ImpulseResponseEstimator
    fs = 48000
    e = ImpulseResponseEstimator(min_duration=1.0, fs=fs)
    spk = ["FL", "FR", "FC", "BL", "BR"]
    frames = synth_frames(e, spk, 2718)
    L = len(e) + 2 * fs
    rows = np.ascontiguousarray(frames[2 * fs: 2 * fs + 3 * L].T.reshape(2, 3, L).transpose(1, 0, 2).reshape(6, L)).astype(np.float32)
    monkeypatch.delenv("IMPULSE_HIP_DEVICES", raising=False)
    one = e.estimate_batch(rows, dtype=np.float32)
    h1 = HRIR(e)
    h1.open_recording_frames(fs, frames, spk)
    want = {(sp, sd): h1.irs[sp][sd].peek() for sp in spk for sd in ("left", "right")}
    monkeypatch.setenv("IMPULSE_HIP_DEVICES", "0,0,0")
    ctxs = _native.device_contexts()
    assert len(ctxs) == 3 and len({id(c) for c in ctxs}) == 3 and ctxs[0] is _native.root_context()
    many = e.estimate_batch(rows, dtype=np.float32)
    assert np.array_equal(one, many)
    h3 = HRIR(e)
    h3.open_recording_frames(fs, frames, spk)
    assert list(h3.irs) == spk
    for key, ref in want.items():
        got = h3.irs[key[0]][key[1]]
        assert got._row is not None and got._row.block.ctx is ctxs[0]          # gathered on the root device
        assert np.array_equal(got.peek(), ref), key
    # the later stages run on the root device as before
    h3.crop_heads()
    h1.crop_heads()
    h3.crop_tails()
    h1.crop_tails()
    for sp in spk:
        assert np.array_equal(h3.irs[sp]["left"].peek(), h1.irs[sp]["left"].peek())
    # every context holds a plan of its own for this length, with the root's spectrum
    plans = [p for k, p in e._plans.items() if k[0] == L and p._h]
    assert len({id(p.ctx) for p in plans}) >= 3
