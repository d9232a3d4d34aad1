"""Decay analysis (Lundeby knee, Schroeder decay times) and decay-window adjustment.

Surface of reference core/decay.py:12-403.  The Lundeby knee search is scalar control flow over at
most ~200 window levels; the reductions it asks for - max|x|, squaring, np.mean over windows and
ranges of <= 2 s of samples - run on the device in fp64 with NumPy's summation order (K7), the
first-peak search (K3) and the decay window (K8) too; decay_times (Schroeder integral, moving average,
line fits) is one device kernel per batch.
"""
import numpy as np

from . import _native

EPSILON = 1e-20


def _peak_index(data, start=0, end=None, peak_height=0.12589):
    """First local extremum within -18 dB of the maximum (device kernel K3)."""
    n = len(data)
    if n == 0:
        return 0
    if end is None:
        end = n
    seg = np.asarray(data)[start:end]
    if len(seg) == 0:
        return start
    idx, _ = _native.default_context().peak_index([seg], peak_height)
    return int(idx[0]) + start


class _Grid:
    """t = np.linspace(0, n / fs, n) of the analysis segment, without materialising it: NumPy forms
    t[i] = i * step with step = (n / fs) / (n - 1) and pins the last sample to n / fs."""

    def __init__(self, n, fs):
        self.n = n
        self.stop = n / fs
        self.step = self.stop / (n - 1) if n > 1 else 0.0

    def at(self, i):
        if self.n == 1:
            return 0.0
        return self.stop if i == self.n - 1 else i * self.step

    def nearest(self, time):
        """int(np.argmin(np.abs(t - time))): the exact float comparison on the few candidates around time/step."""
        if self.n <= 1 or self.step == 0.0:
            return 0
        guess = int(min(max(time / self.step, 0.0), self.n - 1))
        best, best_d = None, None
        for i in range(max(guess - 2, 0), min(guess + 3, self.n)):
            d = abs(self.at(i) - time)
            if best is None or d < best_d:          # first minimum wins, like argmin
                best, best_d = i, d
        return best


def _fit_line(x, y):
    """Least-squares slope/intercept exactly as scipy.stats.linregress forms them: slope = cov(x, y) / var(x) from
    np.cov(x, y, bias=1).  The knee search calls this ~20 times per response, so np.cov's own primitive sequence (row
    means, centring in place, dot(X, X.T.conj()), scaling by the reciprocal count) is issued directly: the same
    operations on the same layout - bit-identical, a quarter of the call overhead."""
    n = len(x)
    X = np.empty((2, n), dtype=np.float64)
    X[0] = x
    X[1] = y
    avg = X.mean(axis=1)
    xm, ym = avg[0], avg[1]
    X -= avg[:, None]
    c = np.dot(X, X.T.conj())
    c *= np.true_divide(1, n)
    slope = c[0, 1] / c[0, 0]
    return slope, ym - slope * xm


def _first_le(values, level):
    hit = np.flatnonzero(values <= level)
    return int(hit[0]) if hit.size else None


def _db(mean):
    return 10 * np.log10(np.maximum(mean, EPSILON))


def _ranges(a, b):
    return np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64)


def _lundeby(n_sq, fs):
    """The Lundeby knee search of core/decay.py:103-253 as a coroutine over the squared, peak-normalised
    segment of n_sq samples: it yields the ranges it needs as a pair of int64 arrays (a, b) and is sent
    np.mean(sq[a:b]) for each (device K7, NumPy's summation order); everything else is scalar control flow on a few
    hundred window levels.
    Returns (knee offset in samples, noise floor dB, window size)."""
    grid = _Grid(n_sq, fs)
    wd = 0.03
    n = int(n_sq / fs / wd) if fs > 0 else 0
    if n == 0:
        (whole,) = yield _ranges([0], [n_sq])
        return n_sq, float(_db(whole)), max(1, n_sq)
    w0 = max(int(n_sq / n), 1)
    tail_from = int(n_sq * 0.9)
    starts = np.arange(n + 1, dtype=np.int64) * w0               # n windows of w0 samples, then the noise tail
    ends = starts + w0
    starts[n], ends[n] = (tail_from, n_sq) if tail_from < n_sq else (0, n_sq)
    means = yield starts, ends
    levels, floor = _db(np.asarray(means[:n])), float(_db(means[n]))
    t_win = np.arange(n) * wd + wd / 2

    close = np.flatnonzero(levels <= floor + 10.0)
    stop = int(close[0]) if close.size and close[0] > 0 else len(levels)
    if stop < 2:
        if len(levels) < 2:
            return n_sq, floor, w0
        stop = len(levels)
    slope, icpt = _fit_line(t_win[:stop], levels[:stop])
    if np.isnan(slope) or abs(slope) < EPSILON:
        return n_sq, floor, w0
    t_first, t_last = grid.at(0), grid.at(n_sq - 1)
    knee_time = np.clip((floor - icpt) / slope, t_first, t_last)

    # re-window: three windows per 10 dB of decay
    per10 = abs(slope) * 3
    wd = t_last / 3.0 if per10 < EPSILON else 10 / per10
    n = int(n_sq / fs / wd) if (fs > 0 and wd > EPSILON) else 1
    n = max(n, 1)
    w = max(int(n_sq / n), 1)
    starts = np.arange(n, dtype=np.int64) * w
    means = yield starts, starts + w
    levels = _db(np.asarray(means))
    t_win = np.arange(n) * wd + wd / 2

    after = np.flatnonzero(t_win >= knee_time)
    if after.size:
        k_idx = int(after[0])
    else:
        k_idx = len(t_win) - 1
        knee_time = t_win[-1]
    k_level = levels[k_idx]

    total = t_last
    for _ in range(5):
        i0 = _first_le(levels, k_level - 5)
        if i0 is None:
            break
        t0 = max(t_win[i0], 0.1 * total)
        if t0 > t_win[-1]:
            break
        a, b = grid.nearest(t0), grid.nearest(min(t0 + knee_time, total))
        if a >= b:
            break
        (m,) = yield _ranges([a], [b])
        floor = float(_db(m))
        hi = _first_le(levels, floor + 8)
        lo = _first_le(levels, floor + 28)
        if hi is None or lo is None:
            break
        hi, lo = hi - 1, max(lo - 1, 0)
        if hi <= lo + 1:
            break
        s2, i2 = _fit_line(t_win[lo:hi], levels[lo:hi])
        if np.isnan(s2) or abs(s2) < EPSILON:
            break
        t_new = np.clip((floor - i2) / s2, t_win[0], t_win[-1])
        after = np.flatnonzero(t_win >= t_new)
        new_idx = int(after[0]) if after.size else len(t_win) - 1
        same = new_idx == k_idx
        k_idx = new_idx
        knee_time = t_win[k_idx]
        if same:
            break
        k_level = levels[k_idx]

    return grid.nearest(knee_time), floor, w


def _knee_searches(ctx, lengths, peaks, segset_for, fs):
    """The lock-step Lundeby searches shared by the host-array and the device-row entry points.  lengths / peaks per
    response; segset_for(starts, seg_lens) builds the K7 segment set of the analysis spans."""
    results = [None] * len(lengths)
    live = [k for k, n in enumerate(lengths) if n >= 10]
    for k, n in enumerate(lengths):
        if n < 10:
            results[k] = (0, n, -200.0, n if n > 0 else 1)
    if not live:
        return results
    starts, seg_lens, seg_peak = [], [], {}
    for k in live:
        n, pk = lengths[k], int(peaks[k])
        end = min(pk + int(2 * fs), n)
        if pk >= end:
            pk = min(max(pk, 0), n - 1)
            starts.append(pk)
            seg_lens.append(1)
        else:
            starts.append(pk)
            seg_lens.append(end - pk)
        seg_peak[k] = pk
    segset = segset_for(live, starts, seg_lens)
    try:
        runs = {}
        pending = {}
        for j, k in enumerate(live):
            gen = _lundeby(seg_lens[j], fs)
            runs[k] = (j, gen)
            pending[k] = next(gen)                         # every search asks at least one question
        while pending:
            order = list(pending)
            seg = np.concatenate([np.full(len(pending[k][0]), runs[k][0], dtype=np.int64) for k in order])
            means = segset.range_means_arrays(seg, np.concatenate([pending[k][0] for k in order]),
                                              np.concatenate([pending[k][1] for k in order]))
            pos = 0
            nxt = {}
            for k in order:
                cnt = len(pending[k][0])
                try:
                    nxt[k] = runs[k][1].send(means[pos:pos + cnt])
                except StopIteration as done:
                    knee_off, floor, w = done.value
                    results[k] = (seg_peak[k], seg_peak[k] + int(knee_off), floor, w)
                pos += cnt
            pending = nxt
    finally:
        segset.close()
    return results


def decay_params_batch(datas, fs):
    """decay_params for many responses at once: one upload of all analysis segments, then the knee
    searches advance in lock step so that each round of np.mean queries is ONE device call (K7)."""
    datas = [np.asarray(d, dtype=np.float64) for d in datas]
    ctx = _native.default_context()
    lengths = [len(d) for d in datas]
    live = [k for k, n in enumerate(lengths) if n >= 10]
    peaks = np.zeros(len(datas), dtype=np.int64)
    if live:
        peaks[live] = ctx.peak_index([datas[k] for k in live])[0]

    def segset_for(idx, starts, seg_lens):
        return _native.SegSet(ctx, [datas[k][a:a + n] for k, a, n in zip(idx, starts, seg_lens)])

    return _knee_searches(ctx, lengths, peaks, segset_for, fs)


def decay_params_rows(rows, fs):
    """decay_params_batch for device-resident responses (device_rows.Row): the peak search and the analysis spans are
    read where the rows are; only peak indices and window levels come back."""
    from .device_rows import span
    ctx = _native.default_context()
    base, offs, lens = span(rows)
    lengths = [int(n) for n in lens]
    peaks, _ = ctx.peak_index_device(base, offs, lens)

    def segset_for(idx, starts, seg_lens):
        # (the row maxima stay on the device: the set is ready in stream order, no round trip of its own)
        return _native.SegSet.from_device(ctx, base, [offs[k] + a for k, a in zip(idx, starts)], seg_lens, want_max=False)

    return _knee_searches(ctx, lengths, peaks, segset_for, fs)


def knee_indices_rows(rows, fs, stats=None):
    """decay_params(...)[1] of device-resident responses, for crop_tails: peak search and the whole Lundeby search run
    on the device in one sequence of launches (K7c, csrc/decay_kernels.hip.h).  np.log10 and linregress's BLAS dot
    product cannot be reproduced bit for bit on the device, so each decision of the device search carries a guard band;
    rows with a decision inside its band (flags != 0) are searched again here by the host flow above - the knees
    returned are the host search's integers either way.  stats (dict) receives the number of rows the host decided."""
    from .device_rows import span
    ctx = _native.default_context()
    base, offs, lens = span(rows)
    if not (fs > 0):
        return [p[1] for p in decay_params_rows(rows, fs)]
    peaks, knees, _, _, flags = ctx.decay_knees_device(base, offs, lens, fs)
    redo = [k for k in range(len(rows)) if flags[k]]
    if stats is not None:
        stats["host_rows"] = stats.get("host_rows", 0) + len(redo)
        stats["rows"] = stats.get("rows", 0) + len(rows)
    if redo:
        def segset_for(idx, starts, seg_lens):
            return _native.SegSet.from_device(ctx, base, [offs[redo[k]] + a for k, a in zip(idx, starts)], seg_lens,
                                              want_max=False)

        host = _knee_searches(ctx, [int(lens[k]) for k in redo], [int(peaks[k]) for k in redo], segset_for, fs)
        for k, res in zip(redo, host):
            knees[k] = res[1]
    return [int(v) for v in knees]


def decay_params(data, fs):
    """(peak_index, knee_point_index, noise_floor_dB, window_size) by the Lundeby method."""
    return decay_params_batch([data], fs)[0]


def decay_times(data, fs, peak_ind=None, knee_point_ind=None, noise_floor=None, window_size=None):
    """EDT, RT20, RT30, RT60 from the Schroeder backward integral (None where the dynamic range
    above the noise floor is insufficient).  Integral, moving average and line fits run on the device."""
    ir = np.asarray(data, dtype=np.float64)
    if peak_ind is None or knee_point_ind is None or noise_floor is None:
        peak_ind, knee_point_ind, noise_floor, window_size = decay_params(ir, fs)
    vals = _native.default_context().decay_times([ir], [peak_ind], [knee_point_ind], [noise_floor], [window_size], fs)[0]
    return tuple(None if np.isnan(v) else float(v) for v in vals)


def decay_adjustment_params(data, fs, target):
    """(window_start, half_window, knee_point_index, window_level) or None when the measured
    decay is already faster than ``target`` seconds per 60 dB."""
    peak, knee, _, _ = decay_params(data, fs)
    return _adjustment(peak, knee, decay_times(data, fs), fs, target)


def _adjustment(peak, knee, times, fs, target):
    """decay_adjustment_params from decay_params' and decay_times' results (core/decay.py:359-380)"""
    measured = None
    for rt, span in zip(times, (-10, -20, -30, -60)):
        if not rt:
            break
        measured = span / rt                     # dB/s from the longest defined decay time
    wanted = -60 / target
    if wanted > measured:                        # TypeError when nothing is defined, as the reference
        return None
    knee_s = knee / fs
    start = peak + 2 * (fs // 1000)
    return start, knee - start, knee, wanted * knee_s - measured * knee_s


def adjust_decay_rows(rows, fs, targets, stats=None):
    """process_decay_worker (core/parallel_workers.py:24-39) for responses that are on the device (device_rows.Row), in
    place and without bringing a sample to the host: decay_params by the device knee search (K3 + K7c; rows with a
    decision inside a guard band are searched again by the host flow, as in knee_indices_rows), decay_times from the
    rows where they are (K7b), the window by K8 in place.  targets: one RT60 in seconds per row.  Raises what the
    reference raises (TypeError when no decay time is defined, ValueError when the window does not tile the response)."""
    from .device_rows import span
    if not rows:
        return
    ctx = _native.default_context()
    base, offs, lens = span(rows)
    peaks, knees, floors, wins, flags = ctx.decay_knees_device(base, offs, lens, fs)
    redo = [k for k in range(len(rows)) if flags[k]]
    if stats is not None:
        stats["host_rows"] = stats.get("host_rows", 0) + len(redo)
        stats["rows"] = stats.get("rows", 0) + len(rows)
    if redo:
        def segset_for(idx, starts, seg_lens):
            return _native.SegSet.from_device(ctx, base, [offs[redo[k]] + a for k, a in zip(idx, starts)], seg_lens, want_max=False)

        host = _knee_searches(ctx, [int(lens[k]) for k in redo], [int(peaks[k]) for k in redo], segset_for, fs)
        for k, (pk, kn, fl, w) in zip(redo, host):
            peaks[k], knees[k], floors[k], wins[k] = pk, kn, fl, w
    times = ctx.decay_times_device(base, offs, lens, peaks, knees, floors, wins, fs)
    todo, params = [], []
    for k, row in enumerate(rows):
        vals = tuple(None if np.isnan(v) else float(v) for v in times[k])
        p = _adjustment(int(peaks[k]), int(knees[k]), vals, fs, targets[k])
        if p is None:
            continue
        start, half, knee, level = p
        if start + half != knee or knee > row.n or start < 0 or half < 0:
            raise ValueError("operands could not be broadcast together: decay window does not tile the data")
        todo.append(k)
        params.append(dict(gain=1.0, decay_start=start, decay_half=half, decay_knee=knee, decay_level_db=level))
    if todo:
        ctx.apply_window_device(base, offs[todo], base, offs[todo], lens[todo], params)
        for k in todo:
            rows[k].block.touch()


def apply_decay_window(data, params):
    """In-place ``data *= 10**(-level*(window-1)/20)`` with window = ones | falling Hann | zeros
    (device kernel K8)."""
    if params is None:
        return data
    start, half, knee, level = params
    if start + half != knee or knee > len(data) or start < 0 or half < 0:
        raise ValueError("operands could not be broadcast together: decay window does not tile the data")
    out = _native.default_context().apply_window(
        [data], [dict(gain=1.0, decay_start=start, decay_half=half, decay_knee=knee, decay_level_db=level)])[0]
    data[:] = out
    return data
