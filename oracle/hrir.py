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
