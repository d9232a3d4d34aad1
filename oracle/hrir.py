"""Oracle: HRIR container operations (reference core/hrir.py) on plain dicts of float64 arrays.

irs = {speaker: {"left": ndarray, "right": ndarray}}
"""
import numpy as np

from . import decay
from .impulse_response import magnitude_response, peak_index
from .scipy_restated import hann, next_fast_len_real


def split_recording(recording, speakers, n_sweep, fs, side=None, silence_length=2.0):
    """Column/track geometry of core/hrir.py:146-219, 302-355 for recordings that are long enough
    (the short-recording fallbacks :221-299 are not restated).  recording: [tracks, samples].
    Returns [(speaker, side, column_view)] in the reference's processing order."""
    recording = np.asarray(recording)
    sil = int(silence_length * fs)
    k = 2 if side is None else 1
    n_cols = round(len(speakers) / (recording.shape[0] // k))
    rec = recording[:, sil:]
    col = sil + n_sweep
    cols = []
    for i in range(n_cols):
        a, b = i * col, min((i + 1) * col, rec.shape[1])
        if b > a and (b - a) >= n_sweep:
            cols.append(rec[:, a:b])
    out = []
    i = 0
    while i < rec.shape[0]:
        for j, c in enumerate(cols):
            n = int(i // 2 * len(cols) + j)
            if n >= len(speakers):
                continue
            sp = speakers[n]
            if side is None:
                if i + 1 < rec.shape[0]:
                    out.append((sp, "left", c[i, :]))
                    out.append((sp, "right", c[i + 1, :]))
            else:
                out.append((sp, side, c[i, :]))
        i += k
    return out


def crop_heads(irs, fs, head_ms=1):
    """core/hrir.py:548-612 with SPEAKER_DELAYS == 0 (core/constants.py:59-61). Returns new dict."""
    out = {}
    head = int(head_ms * fs / 1000)
    for sp, pair in irs.items():
        left, right = pair["left"], pair["right"]
        pl, pr = peak_index(left), peak_index(right)
        first = pl if pl < pr else pr            # ties take the right-ear branch; same crop index
        i0 = max(0, first - head)
        left, right = left[i0:].copy(), right[i0:].copy()
        if len(left) >= head and len(right) >= head:
            w = hann(head * 2)[:head]
            left[:head] *= w
            right[:head] *= w
        out[sp] = {"left": left, "right": right}
    return out


def crop_tails(irs, fs, n_sweep, n_octaves):
    """core/hrir.py:614-653. Returns (tail_ind, new dict)."""
    tails, lengths = [], []
    for pair in irs.values():
        for d in pair.values():
            tails.append(decay.decay_params(d, fs)[1])
            lengths.append(len(d))
    sec_per_oct = n_sweep / fs / n_octaves
    fo = 2 * int(fs * sec_per_oct * (1 / 24))
    w = hann(fo)[fo // 2:]
    tail_ind = min(np.min(lengths), next_fast_len_real(max(tails)))
    out = {}
    for sp, pair in irs.items():
        out[sp] = {}
        for sd, d in pair.items():
            d = d[:tail_ind].copy()
            d *= np.concatenate([np.ones(len(d) - len(w)), w])
            out[sp][sd] = d
    return int(tail_ind), out


def normalization_gain_db(irs, fs, peak_target=-0.1, avg_target=None):
    """core/hrir.py:457-521: gain from the summed-ear magnitude responses."""
    def summed(side):
        arrs = [p[side] for p in irs.values() if p[side].size > 0]
        n = max(len(a) for a in arrs)
        return np.sum(np.vstack([np.pad(a, (0, n - len(a))) for a in arrs]), axis=0)

    fl, ml = magnitude_response(summed("left"), fs)
    fr, mr = magnitude_response(summed("right"), fs)
    if peak_target is not None and avg_target is None:
        return np.max(np.vstack([ml, mr])) * -1 + peak_target
    if peak_target is None and avg_target is not None:
        band = np.concatenate([ml[np.logical_and(fl > 80, fl < 6000)], mr[np.logical_and(fr > 80, fr < 6000)]])
        return np.mean(band) * -1 + avg_target
    raise ValueError('One and only one of the parameters "peak_target" and "avg_target" must be given!')


def ipsilateral_lag(a, b, segment_len):
    """core/hrir.py:930-937 / :944-949 (HRIR.align_ipsilateral_all): lag of the cross-correlation peak of
    the first `segment_len` samples.  scipy.signal.correlate(x, y, "full")[k] = sum_l x[l + k - (len(y)-1)] y[l]
    (SciPy picks FFT or direct by size; both agree to rounding, np.correlate is the direct sum)."""
    x, y = np.asarray(a, dtype=np.float64)[:segment_len], np.asarray(b, dtype=np.float64)[:segment_len]
    corr = np.correlate(x, y, mode="full")
    lags = np.arange(-len(x) + 1, len(x))
    return int(lags[np.argmax(corr)])


def align_ipsilateral_all(irs, fs, speaker_pairs, segment_ms=30):
    """core/hrir.py:921-958 on a dict {speaker: {side: fp64 array}}; returns the shifted copies.
    `shift` follows core/impulse_response.py:92-108 (length-preserving)."""
    def shift(d, s):
        n = len(d)
        if s > 0:
            return np.concatenate((np.zeros(s), d))[:n]
        if s < 0:
            t = d[-s:]
            return np.pad(t, (0, n - len(t))) if len(t) < n else t
        return d
    out = {sp: {sd: np.array(x, dtype=np.float64) for sd, x in pair.items()} for sp, pair in irs.items()}
    seg = int(fs * segment_ms / 1000)
    for one, two in speaker_pairs:
        if one not in out or two not in out:
            continue
        if one == two:
            lag = ipsilateral_lag(out[one]["left"], out[one]["right"], seg)
            if lag > 0:
                out[one]["right"] = shift(out[one]["right"], lag)
            elif lag < 0:
                out[one]["left"] = shift(out[one]["left"], -lag)
            continue
        lag = ipsilateral_lag(out[one]["left"], out[two]["right"], seg)
        if lag > 0:
            for sd in ("left", "right"):
                out[two][sd] = shift(out[two][sd], lag)
        elif lag < 0:
            for sd in ("left", "right"):
                out[one][sd] = shift(out[one][sd], -lag)
    return out


def align_onset_groups_peak_leftref(irs, groups=None):
    """core/hrir.py:960-1001 on a dict {speaker: {side: fp64 array}}; returns the shifted copies: every group is shifted
    by -(peak_index of its first speaker's left ear - peak_index of FL's left ear); a group whose first speaker is absent
    is left alone; a missing FL raises."""
    from .impulse_response import peak_index

    def shift(d, s):                                       # core/impulse_response.py:92-108
        n = len(d)
        if s > 0:
            return np.concatenate((np.zeros(s), d))[:n]
        if s < 0:
            t = d[-s:]
            return np.pad(t, (0, n - len(t))) if len(t) < n else t
        return d
    if groups is None:
        groups = [("FL", "FR"), ("SL", "SR"), ("BL", "BR"), ("WL", "WR"), ("TFL", "TFR"), ("TSL", "TSR"), ("TBL", "TBR"), ("FC",)]
    out = {sp: {sd: np.array(x, dtype=np.float64) for sd, x in pair.items()} for sp, pair in irs.items()}

    def lead(group):
        sp = group[0]
        return peak_index(out[sp]["left"]) if sp in out and "left" in out[sp] else None
    ref = lead(("FL", "FR"))
    if ref is None:
        raise RuntimeError("Cannot find FL left channel reference for onset alignment.")
    for group in groups:
        if group == ("FL", "FR"):
            continue
        pk = lead(group)
        if pk is None:
            continue
        for sp in group:
            if sp in out:
                for sd in ("left", "right"):
                    out[sp][sd] = shift(out[sp][sd], -(pk - ref))
    return out


# core/constants.py:95-104
HESUVI_TRACK_ORDER = ['FL-left', 'FL-right', 'SL-left', 'SL-right', 'BL-left', 'BL-right', 'FC-left', 'FR-right',
                      'FR-left', 'SR-right', 'SR-left', 'BR-right', 'BR-left', 'FC-right', 'WL-left', 'WL-right',
                      'WR-left', 'WR-right', 'TFL-left', 'TFL-right', 'TFR-left', 'TFR-right', 'TSL-left',
                      'TSL-right', 'TSR-left', 'TSR-right', 'TBL-left', 'TBL-right', 'TBR-left', 'TBR-right']
HEXADECAGONAL_TRACK_ORDER = ['FL-left', 'FL-right', 'FR-left', 'FR-right', 'FC-left', 'FC-right', 'LFE-left',
                             'LFE-right', 'BL-left', 'BL-right', 'BR-left', 'BR-right', 'SL-left', 'SL-right',
                             'SR-left', 'SR-right', 'WL-left', 'WL-right', 'WR-left', 'WR-right', 'TFL-left',
                             'TFL-right', 'TFR-left', 'TFR-right', 'TSL-left', 'TSL-right', 'TSR-left',
                             'TSR-right', 'TBL-left', 'TBL-right', 'TBR-left', 'TBR-right']


def write_wav_frames(irs, track_order=None):
    """core/hrir.py:426-455 + core/audio_io.py:82-97: the float matrix handed to soundfile, [frames, tracks]: one
    column per name in track_order ('<speaker>-<side>'), zeros (length of the FIRST response) for absent channels;
    the [tracks, frames] stack is transposed when it has more columns than rows."""
    if track_order is None:
        track_order = HEXADECAGONAL_TRACK_ORDER
    by_name = {f"{sp}-{sd}": np.asarray(d) for sp, pair in irs.items() for sd, d in pair.items()}
    if not by_name:
        raise ValueError("No impulse responses available for WAV output.")
    n = len(next(iter(by_name.values())))
    data = np.vstack([by_name.get(ch, np.zeros(n)) for ch in track_order])
    if data.ndim > 1 and data.shape[1] > data.shape[0]:
        data = data.T
    return data


def pcm_quantise(frames, bit_depth):
    """libsndfile's float -> PCM conversion as soundfile.write(subtype='PCM_16'|'PCM_24'|'PCM_32') performs it
    (core/audio_io.py:82-97 -> soundfile; python-soundfile enables libsndfile's clipping on every file it opens, which
    selects the *_clip_array conversions): q32 = clip(lrint(x * 2^31), -2^31, 2^31 - 1), and a narrower subtype keeps
    the top bits, q = q32 >> (32 - bits) (an arithmetic shift: floor).
    PCM_32 is PINNED by reference-held data: the four sweep WAVs under /root/reference/data were written by
    core/impulse_response_estimator.py:306-322 with this call, peak at 0.99999999996 of full scale, and are reproduced
    by this rule (tests/golden/sweep_wavs.npz; lrint(x * (2^31 - 1)) reproduces 38 % of their samples).  libsndfile is a
    third-party dependency of the reference (via `soundfile`, unpinned) and no 16- / 24-bit file ships: for those widths
    this restates its published clip path (src/pcm.c d2s_clip_array / d2let_clip_array): PARITY UNPINNED."""
    q32 = np.clip(np.rint(np.asarray(frames, dtype=np.float64) * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
    return q32 >> (32 - bit_depth)


def equalize_all(irs, fir):
    """core/hrir.py:858-888: row 0 of `fir` filters every left response, row 1 every right one ('full'
    convolution); a single row / 1-D filter serves both sides."""
    from .scipy_restated import fft_convolve
    fir = np.asarray(fir, dtype=np.float64)
    if fir.ndim == 1 or fir.shape[0] == 1:
        fir = np.tile(fir, (2, 1))
    return {sp: {sd: fft_convolve(np.asarray(d, dtype=np.float64), fir[0] if sd == "left" else fir[1], "full")
                 for sd, d in pair.items()} for sp, pair in irs.items()}


def reflection_levels(irs, fs, direct_sound_duration_ms=2, early_ref_start_ms=20, early_ref_end_ms=50,
                      late_ref_start_ms=50, late_ref_end_ms=150, epsilon=1e-12):
    """core/hrir.py:1003-1090 HRIR.calculate_reflection_levels: for every response, RMS of [peak, peak + direct) and of the
    early / late windows after the peak (all clipped at the end of the data), early_db / late_db = 20 log10(rms / rms_direct
    + epsilon) with rms_direct floored at epsilon and an empty window counting as zero.
    irs: {speaker: {side: 1-D array}} -> {speaker: {side: {"early_db", "late_db"}}}"""
    out = {}
    for sp, pair in irs.items():
        out[sp] = {}
        for sd, data in pair.items():
            data = np.asarray(data, dtype=np.float64)
            pk = peak_index(data)
            n = len(data)

            def at(ms):
                return pk + int(ms * fs / 1000)

            direct = data[pk:min(at(direct_sound_duration_ms), n)]
            early = data[min(at(early_ref_start_ms), n):min(at(early_ref_end_ms), n)]
            late = data[min(at(late_ref_start_ms), n):min(at(late_ref_end_ms), n)]
            rms_direct = np.sqrt(np.mean(direct ** 2)) if len(direct) > 0 else epsilon
            rms_early = np.sqrt(np.mean(early ** 2)) if len(early) > 0 else 0
            rms_late = np.sqrt(np.mean(late ** 2)) if len(late) > 0 else 0
            rms_direct = rms_direct if rms_direct > epsilon else epsilon
            out[sp][sd] = {"early_db": 20 * np.log10(rms_early / rms_direct + epsilon),
                           "late_db": 20 * np.log10(rms_late / rms_direct + epsilon)}
    return out
