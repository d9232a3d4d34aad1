"""HRIR: the {speaker: {side: ImpulseResponse}} container (class surface of reference
core/hrir.py:366-1090) with recording ingest batched onto the device.

Where the reference deconvolves column after column in a Python loop (core/hrir.py:307-355), the
ingest here collects every (track, column) slice of a recording and hands them to the GPU as one
batch (ImpulseResponseEstimator.estimate_batch).
"""
import warnings

import numpy as np

from . import _native
from .audio_io import magnitude_response, read_wav, read_wav_pcm, write_wav
from .constants import (HEXADECAGONAL_TRACK_ORDER, IPSILATERAL_PAIRS, SPEAKER_DELAYS, SPEAKER_NAMES,
                        speaker_side, track_name)
from .impulse_response import ImpulseResponse

def _unit_impulse(n):
    x = np.zeros(int(n))
    x[0] = 1.0
    return x


def channel_balance_groups():
    """core/hrir.py:38-40: one group per ipsilateral pair (a centre speaker stands alone)."""
    return [[one] if one == two else [one, two] for one, two in IPSILATERAL_PAIRS]


def get_center_value(fr, frequency_range):
    """core/hrir.py:43-73: minus the shift FrequencyResponse.center would apply, without mutating `fr`
    (band mean of `raw` on the curve's own grid, or the log-frequency linear interpolation at a point)."""
    from .frequency_response import log_interp
    if isinstance(frequency_range, (list, np.ndarray)) and len(frequency_range) > 1:
        band = np.logical_and(fr.frequency >= frequency_range[0], fr.frequency <= frequency_range[1])
        return -np.mean(fr.raw[band])
    if isinstance(frequency_range, (list, np.ndarray)):
        frequency_range = frequency_range[0]
    return -float(log_interp(fr.frequency, fr.raw, np.array([float(frequency_range)]))[0])


def _hann(M):
    if M <= 0:
        return np.zeros(0)
    if M == 1:
        return np.ones(1)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(M) / (M - 1))


def next_fast_len(n):
    """Smallest 2^a 3^b 5^c >= n (what scipy.fftpack.next_fast_len returns)."""
    n = int(n)
    if n <= 6:
        return n
    best = 1 << (n - 1).bit_length()
    p5 = 1
    while p5 < best:
        p35 = p5
        while p35 < best:
            q = p35
            while q < n:
                q *= 2
            best = min(best, q)
            p35 *= 3
        p5 *= 5
    return best


def _column_slices(n_samples, n_columns, column_size, n_sweep):
    cols = []
    for i in range(n_columns):
        a, b = i * column_size, min((i + 1) * column_size, n_samples)
        if b > a and (b - a) >= n_sweep:
            cols.append((a, b))
    return cols


def split_recording(recording, speakers, n_sweep, fs, side=None, silence_length=2.0):
    """Geometry of a sweep-sequence recording (reference core/hrir.py:146-355).

    recording: [tracks, samples] (any strides).  Returns (trimmed_recording, jobs) where jobs is a
    list of (speaker, side, track_index, start, stop) into the trimmed recording."""
    if silence_length * fs != int(silence_length * fs):
        raise ValueError("Silence length must produce full samples with given sampling rate.")
    lead = int(silence_length * fs)
    per_speaker = 2 if side is None else 1
    n_tracks = recording.shape[0]
    n_columns = round(len(speakers) / (n_tracks // per_speaker))
    rec = recording[:, lead:]
    avail = rec.shape[1]
    column = lead + n_sweep
    if column > avail:
        if n_columns <= 1:
            column, n_columns = avail, 1
        else:
            column = avail // n_columns
    cols = _column_slices(avail, n_columns, column, n_sweep)
    if not cols:
        # short recording: (1) give up part of the lead silence, (2) accept >= 80 % of a sweep
        if lead > 0:
            spare = avail - n_sweep
            if spare >= int(0.5 * fs):
                cut = max(int(0.5 * fs), spare)
                trial = rec[:, cut:]
                cols = _column_slices(trial.shape[1], n_columns, n_sweep, n_sweep)
                if cols:
                    rec = trial
        if not cols and avail > n_sweep * 0.8:
            if n_columns == 1:
                cols = [(0, avail)]
            else:
                step = avail // n_columns
                cols = [(i * step, min((i + 1) * step, avail)) for i in range(n_columns)
                        if min((i + 1) * step, avail) > i * step]
        if not cols:
            raise ValueError(
                "No valid columns could be extracted even with fallback methods.\n"
                f"Recording length ({avail} samples, {avail / fs:.2f}s) is too short for the required "
                f"estimator length ({n_sweep} samples, {n_sweep / fs:.2f}s).")
    jobs = []
    track = 0
    while track < rec.shape[0]:
        for j, (a, b) in enumerate(cols):
            n = int(track // 2 * len(cols) + j)
            if n >= len(speakers):
                continue
            sp = speakers[n]
            if sp not in SPEAKER_NAMES:
                continue
            if side is None:
                if track + 1 < rec.shape[0]:
                    jobs.append((sp, "left", track, a, b))
                    jobs.append((sp, "right", track + 1, a, b))
            else:
                jobs.append((sp, side, track, a, b))
        track += per_speaker
    return rec, jobs


def ingest_recording(estimator, expected_fs, fs, recording, speakers, side=None, silence_length=2.0):
    """{speaker: {side: ImpulseResponse}} for one recording array [tracks, samples]."""
    if fs != expected_fs:
        raise ValueError("Sampling rate of recording must match sampling rate of test signal.")
    if recording.ndim == 1:
        recording = recording[None, :]
    rec, jobs = split_recording(recording, speakers, len(estimator), fs, side, silence_length)
    irs = {}
    by_len = {}
    for job in jobs:
        by_len.setdefault(job[4] - job[3], []).append(job)
    for length, group in by_len.items():
        cols = [rec[tr, a:b] for (_, _, tr, a, b) in group]
        if hasattr(estimator, "estimate_batch"):
            est = estimator.estimate_batch(np.stack(cols))
        else:                                            # duck-typed estimator: per channel
            est = [estimator.estimate(c) for c in cols]
        for (sp, sd, _, _, _), col, y in zip(group, cols, est):
            irs.setdefault(sp, {})[sd] = ImpulseResponse(np.asarray(y), fs, col)
    # keep the reference's insertion order (speaker order of the jobs)
    ordered = {}
    for sp, sd, *_ in jobs:
        ordered.setdefault(sp, {})[sd] = irs[sp][sd]
    return ordered


def ingest_pcm_frames(estimator, expected_fs, fs, frames, speakers, side=None, silence_length=2.0):
    """Same result as ingest_recording, from the WAV's own interleaved PCM frames [n_frames, tracks]
    (int16/int32): the block goes to the GPU untouched; scaling to [-1, 1) and de-interleaving happen in
    the column-pass loader (imp_conv_execute_device_pcm)."""
    if fs != expected_fs:
        raise ValueError("Sampling rate of recording must match sampling rate of test signal.")
    n_frames, tracks = frames.shape
    shape_only = np.broadcast_to(np.zeros(1), (tracks, n_frames))        # geometry needs shapes only
    rec, jobs = split_recording(shape_only, speakers, len(estimator), fs, side, silence_length)
    origin = n_frames - rec.shape[1]                                     # every trim is from the front
    scale = 1.0 / float(2 ** (8 * frames.dtype.itemsize - 1))
    by_len = {}
    for job in jobs:
        by_len.setdefault(job[4] - job[3], []).append(job)
    irs = {}
    from .device_rows import DeviceBlock, Row
    for length, group in by_len.items():
        starts = sorted({origin + a for (_, _, _, a, _) in group})
        # a binaural recording (two tracks = the two ears of every column) takes the pair-mode plan
        plan = estimator._plan(length, paired=(side is None and tracks == 2))
        # the deconvolved columns stay on the device as rows of one block (device_rows.py); `recording`, the raw
        # column the reference keeps beside every response (core/hrir.py:336-341), is cut from the PCM block on demand
        pitch = (plan.out_len + 63) // 64 * 64
        block = DeviceBlock(plan.ctx, len(starts) * tracks * pitch)
        ctxs = _native.device_contexts() if getattr(_native._thread_ctx, "ctx", None) is None else None
        if ctxs is not None and len(ctxs) > 1 and len(starts) > 1 and plan.ctx is ctxs[0]:
            _ingest_columns_sharded(estimator, ctxs, frames, starts, length, plan, block, pitch, side is None and tracks == 2)
        else:
            plan.execute_pcm_columns_device(frames, starts, block.ptr, pitch)
        for sp, sd, tr, a, b in group:
            row = Row(block, (starts.index(origin + a) * tracks + tr) * pitch, plan.out_len)
            column = (lambda a=a, b=b, tr=tr: frames[origin + a: origin + b, tr].astype(np.float64) * scale)
            irs.setdefault(sp, {})[sd] = ImpulseResponse.on_device(row, fs, column)
    ordered = {}
    for sp, sd, *_ in jobs:
        ordered.setdefault(sp, {})[sd] = irs[sp][sd]
    return ordered


def _ingest_columns_sharded(estimator, ctxs, frames, starts, length, plan, block, pitch, paired):
    """The columns of a recording over the devices of IMPULSE_HIP_DEVICES: every device uploads the frames of ITS columns
    (a contiguous stretch of the PCM block), deconvolves them with its own copy of the inverse-sweep spectrum, and the rows
    are gathered (peer copies) into the root device's block, where the later stages run - same rows, same bits."""
    from .sharding import device_shards, run_sharded
    tracks = frames.shape[1]
    root = ctxs[0]

    def work(ctx, lo, hi):
        cols = starts[lo:hi]
        first, last = cols[0], cols[-1] + length
        sub = frames[first:last]
        rel = [c - first for c in cols]
        if ctx is root:
            plan.execute_pcm_columns_device(sub, rel, block.ptr + lo * tracks * pitch * 4, pitch)
            return None
        p = estimator._plan(length, paired=paired)
        nbytes = len(cols) * tracks * pitch * 4
        tmp = ctx.malloc(nbytes)
        p.execute_pcm_columns_device(sub, rel, tmp, pitch)
        return (ctx, tmp, lo, nbytes)

    shards = device_shards(len(starts), len(ctxs), keep_pairs=False)       # the unit is a column (both ears of a speaker)
    for res in run_sharded(ctxs, shards, work):
        if res is not None:
            ctx, tmp, lo, nbytes = res
            root.copy_from(block.ptr + lo * tracks * pitch * 4, ctx, tmp, nbytes)
            root.synchronize()                               # the copy has landed: the source block may go
            ctx.free(tmp)


def _rows_matrix(rows):
    """np.stack(rows) - or, when the rows already ARE the consecutive rows of one C-contiguous matrix (the FIR batch of
    process_equalization_batch), that matrix without the copy."""
    first = rows[0]
    base = first.base
    if (base is not None and isinstance(base, np.ndarray) and base.ndim == 2 and base.flags.c_contiguous
            and base.dtype == first.dtype and base.shape[1] == first.shape[0] and len(rows) <= base.shape[0]
            and all(r.base is base and r.flags.c_contiguous for r in rows)):
        step = base.strides[0]
        p0 = first.ctypes.data
        i0, rem = divmod(p0 - base.ctypes.data, step)
        if rem == 0 and 0 <= i0 and i0 + len(rows) <= base.shape[0] and \
                all(r.ctypes.data == p0 + i * step for i, r in enumerate(rows)):
            return base[i0:i0 + len(rows)]
    return np.stack(rows)


class HRIR(object):
    def __init__(self, estimator):
        self.estimator = estimator
        self.fs = self.estimator.fs
        self.irs = dict()

    def copy(self):
        other = HRIR(self.estimator)
        other.irs = {sp: {"left": pair["left"].copy(), "right": pair["right"].copy()}
                     for sp, pair in self.irs.items()}
        return other

    def subset(self, speakers, copy_irs=False):
        other = HRIR(self.estimator)
        other.irs = {sp: {sd: (ir.copy() if copy_irs else ir) for sd, ir in self.irs[sp].items()}
                     for sp in speakers if sp in self.irs}
        return other

    def _require_matching_fs(self, what):
        if self.fs != self.estimator.fs:
            raise ValueError(f"Refusing to {what} because HRIR's sampling rate doesn't match impulse "
                             "response estimator's sampling rate.")

    # ---- ingest --------------------------------------------------------------------------
    def open_recording(self, file_path, speakers, side=None, silence_length=2.0, debug=False):
        """Split a combined sweep recording into speaker-ear impulse responses (batched on GPU)."""
        self._require_matching_fs("open recording")
        pcm = read_wav_pcm(file_path) if hasattr(self.estimator, "_plan") else None
        if pcm is not None:
            got = ingest_pcm_frames(self.estimator, self.fs, pcm[0], pcm[1], speakers, side, silence_length)
            for sp, sides in got.items():
                self.irs.setdefault(sp, {}).update(sides)
            return
        fs, recording = read_wav(file_path, expand=True)                   # 24-bit / float files, duck-typed estimators
        self.open_recording_data(fs, recording, speakers, side=side, silence_length=silence_length)

    def open_recording_frames(self, fs, frames, speakers, side=None, silence_length=2.0):
        """open_recording for a recording that is already in memory as interleaved PCM frames [n_frames, tracks]
        (int16 / int32, WAV wire order - what a capture buffer or a WAV data chunk holds): the block goes to the GPU
        as it is and the deconvolved responses stay there (see device_rows.py)."""
        self._require_matching_fs("open recording")
        frames = np.asarray(frames)
        if frames.ndim != 2 or frames.dtype not in (np.int16, np.int32):
            raise ValueError("frames must be int16 / int32 [n_frames, tracks]")
        got = ingest_pcm_frames(self.estimator, self.fs, fs, frames, speakers, side, silence_length)
        for sp, sides in got.items():
            self.irs.setdefault(sp, {}).update(sides)

    def open_recording_data(self, fs, recording, speakers, side=None, silence_length=2.0):
        self._require_matching_fs("open recording")
        got = ingest_recording(self.estimator, self.fs, fs, np.asarray(recording), speakers, side, silence_length)
        for sp, sides in got.items():
            self.irs.setdefault(sp, {}).update(sides)

    def write_wav(self, file_path, track_order=None, bit_depth=32):
        if track_order is None:
            track_order = HEXADECAGONAL_TRACK_ORDER
        named = [(track_name(sp, sd), ir) for sp, pair in self.irs.items() for sd, ir in pair.items()]
        dev = self._device_rows([ir for _, ir in named]) if named else None
        if dev is not None and len({r.n for r in dev}) == 1 and bit_depth in (16, 24, 32):
            # responses on the device: ordering, silence for absent channels and the PCM conversion happen there and
            # the WAV data chunk comes back as it will be written (imp_rows_to_pcm_device)
            from .audio_io import write_wav_frames
            from .device_rows import span
            names = [nm for nm, _ in named]
            n = dev[0].n
            if n <= len(track_order):                            # the reference's transpose rule: fewer frames than tracks
                dev = None
            else:
                base, offs, lens = span(dev)
                frames = _native.default_context().rows_to_pcm_device(
                    base, offs, lens, [names.index(ch) if ch in names else -1 for ch in track_order], n, bit_depth)
                write_wav_frames(file_path, self.fs, frames, bit_depth)
                return
        by_name = {track_name(sp, sd): ir.data for sp, pair in self.irs.items() for sd, ir in pair.items()}
        if not by_name:
            raise ValueError("No impulse responses available for WAV output.")
        n = len(next(iter(by_name.values())))
        write_wav(file_path, self.fs, np.vstack([by_name.get(ch, np.zeros(n)) for ch in track_order]),
                  bit_depth=bit_depth)

    # ---- level ---------------------------------------------------------------------------
    def normalize(self, peak_target=-0.1, avg_target=None):
        """Scale all channels so the summed-ear magnitude peak (or 80-6000 Hz mean) hits the target."""
        def summed(side):
            arrs = [pair[side].data for pair in self.irs.values() if pair[side].data.size > 0]
            if not arrs:
                raise ValueError("No valid impulse response data found for normalization. "
                                 "All channels appear to be empty.")
            n = max(len(a) for a in arrs)
            return np.sum(np.vstack([np.pad(a, (0, n - len(a)), "constant") for a in arrs]), axis=0)

        sides = [(sd, pair[sd]) for pair in self.irs.values() for sd in ("left", "right")]
        dev = self._device_rows([ir for _, ir in sides]) if sides and all(len(ir) > 0 for _, ir in sides) else None
        if dev is not None:
            from .device_rows import span
            ctx = _native.default_context()
            base, offs, lens = span(dev)
            n = int(max(lens))
            groups = [0 if sd == "left" else 1 for sd, _ in sides]
            if peak_target is not None and avg_target is None:
                # np.max of the stacked spectra is all this mode reads: the two maxima are reduced where the spectra are
                m_l, m_r = (np.array([v]) for v in ctx.magnitude_db_sum_peak_device(base, offs, lens, groups, 2, n))
                f_l = f_r = None
            else:
                m_l, m_r = ctx.magnitude_db_sum_device(base, offs, lens, groups, 2, n)
                f_l = f_r = np.arange(int(np.ceil(n / 2))) * (self.fs / n)
        else:
            f_l, m_l = magnitude_response(summed("left"), self.fs)
            f_r, m_r = magnitude_response(summed("right"), self.fs)
        if peak_target is not None and avg_target is None:
            gain = np.max(np.vstack([m_l, m_r])) * -1 + peak_target
        elif peak_target is None and avg_target is not None:
            mid = np.concatenate([m_l[np.logical_and(f_l > 80, f_l < 6000)],
                                  m_r[np.logical_and(f_r > 80, f_r < 6000)]])
            gain = np.mean(mid) * -1 + avg_target
        else:
            raise ValueError('One and only one of the parameters "peak_target" and "avg_target" must be given!')
        g = 10 ** (gain / 20)
        if dev is not None:
            ctx.apply_window_device(base, offs, base, offs, lens, [dict(gain=g)] * len(dev))
            for r in dev:
                r.block.touch()
            return gain
        for pair in self.irs.values():
            for ir in pair.values():
                ir.data *= g
        return gain

    def calculate_reflection_levels(self, direct_sound_duration_ms=2, early_ref_start_ms=20, early_ref_end_ms=50,
                                    late_ref_start_ms=50, late_ref_end_ms=150, epsilon=1e-12):
        """Early and late reflection levels relative to the direct sound, per channel (core/hrir.py:1003-1090):
        {speaker: {side: {"early_db", "late_db"}}}.  One batched peak search (K3) and one set of window means of the squared
        responses (K7) for all channels, from host arrays or from device rows; the few scalars per channel on the host."""
        items = self._all_irs()
        out = {sp: {} for sp, _, _ in items}
        if not items:
            return out
        ctx = _native.default_context()
        dev = self._device_rows([ir for _, _, ir in items])
        if dev is not None:
            from .device_rows import span
            base, offs, lens = span(dev)
            peaks, _ = ctx.peak_index_device(base, offs, lens)
            seg = _native.SegSet.from_device(ctx, base, offs, lens)
            lengths = [int(v) for v in lens]
        else:
            rows = [np.asarray(ir.data, dtype=np.float64) for _, _, ir in items]
            lengths = [len(r) for r in rows]
            live = [k for k, n in enumerate(lengths) if n > 0]
            peaks = np.zeros(len(rows), dtype=np.int64)
            if live:
                peaks[live] = ctx.peak_index([rows[k] for k in live])[0]
            seg = _native.SegSet(ctx, rows)
        try:
            windows = ((0.0, direct_sound_duration_ms), (early_ref_start_ms, early_ref_end_ms), (late_ref_start_ms, late_ref_end_ms))
            q_seg, q_a, q_b = [], [], []
            for k, n in enumerate(lengths):
                pk = int(peaks[k])
                for w, (t0, t1) in enumerate(windows):
                    a = pk if w == 0 else min(pk + int(t0 * self.fs / 1000), n)
                    b = min(pk + int(t1 * self.fs / 1000), n)
                    q_seg.append(k)
                    q_a.append(min(a, n))
                    q_b.append(max(b, min(a, n)))
            means = seg.range_means_arrays(q_seg, q_a, q_b).reshape(-1, 3)           # NaN where a window is empty
            tops = np.asarray(seg.maxabs, dtype=np.float64)
        finally:
            seg.close()
        for k, (sp, sd, _) in enumerate(items):
            # the segments hold (x / max|x|)^2 (x^2 when max|x| < 1e-20): back to mean(x^2)
            scale = tops[k] ** 2 if tops[k] >= 1e-20 else 1.0
            rms = [np.sqrt(m * scale) if not np.isnan(m) else None for m in means[k]]
            rms_direct = rms[0] if rms[0] is not None else epsilon
            rms_direct = rms_direct if rms_direct > epsilon else epsilon
            rms_early = rms[1] if rms[1] is not None else 0
            rms_late = rms[2] if rms[2] is not None else 0
            out[sp][sd] = {"early_db": 20 * np.log10(rms_early / rms_direct + epsilon),
                           "late_db": 20 * np.log10(rms_late / rms_direct + epsilon)}
        return out

    # ---- cropping ------------------------------------------------------------------------
    def _all_irs(self):
        return [(sp, sd, ir) for sp, pair in self.irs.items() for sd, ir in pair.items()]

    def to_host(self):
        """Bring every device-resident response to the host (one transfer per device block); returns self."""
        for pair in self.irs.values():
            for ir in pair.values():
                ir.data                                         # noqa: B018 - the property does the transfer
        return self

    def _device_rows(self, irs):
        """the device rows of `irs` if EVERY one of them still lives on the default context's device, else None"""
        rows = [getattr(ir, "_row", None) if getattr(ir, "_data", 0) is None else None for ir in irs]
        if not rows or any(r is None for r in rows):
            return None
        ctx = _native.default_context()
        return rows if all(r.block.ctx is ctx for r in rows) else None

    def crop_heads(self, head_ms=1):
        """Crop leading silence of every pair at the earlier ear's first peak minus ``head_ms``
        (interaural delay preserved) and fade the head in.  Peak search is one batched launch."""
        self._require_matching_fs("crop heads")
        pairs = list(self.irs.items())
        head = int(head_ms * self.fs / 1000)
        dev = self._device_rows([pair[sd] for _, pair in pairs for sd in ("left", "right")])
        if dev is not None:
            from .device_rows import span
            ctx = _native.default_context()
            base, offs, lens = span(dev)
            peaks, _ = ctx.peak_index_device(base, offs, lens)
        else:
            flat = [pair[sd].data for _, pair in pairs for sd in ("left", "right")]
            peaks, _ = _native.default_context().peak_index(flat) if flat else (np.zeros(0, np.int64), None)
        rows, owners = [], []
        for i, (sp, pair) in enumerate(pairs):
            p_left, p_right = int(peaks[2 * i]), int(peaks[2 * i + 1])
            delay = int(np.round(SPEAKER_DELAYS[sp] * self.fs)) + head
            if p_left < p_right:
                wrong, first = "right", p_left
            else:
                wrong, first = "left", p_right
            if speaker_side(sp) == wrong:
                early = "left" if wrong == "right" else "right"
                itd_ms = abs(p_left - p_right) / self.fs * 1000
                warnings.warn(
                    f"Warning: {sp} measurement has lower delay to {early} ear than to {wrong} ear. "
                    f"{sp} should be at the {wrong} side of the head so the sound should arrive first in the "
                    f"{wrong} ear. This is usually a problem with the measurement process or the speaker order "
                    f"given is not correct. Detected delay difference is {itd_ms:.4f} milliseconds.")
            at = max(0, first - delay)
            if dev is not None:                              # a crop is a change of (offset, length)
                for sd in ("left", "right"):
                    r = pair[sd]._row
                    cut = min(at, r.n)
                    r.off, r.n = r.off + cut, r.n - cut
                if pair["left"]._row.n >= head and pair["right"]._row.n >= head:
                    rows.extend(pair[sd]._row for sd in ("left", "right"))
                continue
            pair["left"].data = pair["left"].data[at:]
            pair["right"].data = pair["right"].data[at:]
            if len(pair["left"].data) >= head and len(pair["right"].data) >= head:
                for sd in ("left", "right"):
                    rows.append(pair[sd].data[:head])
                    owners.append(pair[sd])
        if dev is not None:
            if rows and head > 0:
                base, offs, _ = span(rows)
                ctx.apply_window_device(base, offs, base, offs, [head] * len(rows), [dict(fade_in=head)] * len(rows))
                for r in rows:
                    r.block.touch()
            return
        if rows and head > 0:
            faded = _native.default_context().apply_window(rows, [dict(fade_in=head)] * len(rows))
            for ir, seg in zip(owners, faded):
                if not ir.data.flags.writeable or ir.data.base is not None:
                    ir.data = ir.data.copy()
                ir.data[:head] = seg

    def crop_tails(self):
        """Truncate all channels at an FFT-friendly length past the latest Lundeby knee and fade out."""
        self._require_matching_fs("crop tails")
        items = self._all_irs()
        if not items:
            return 0
        from .decay import decay_params_batch
        dev = self._device_rows([ir for _, _, ir in items])
        if dev is not None:
            from .decay import knee_indices_rows
            from .device_rows import DeviceBlock, Row, span
            ctx = _native.default_context()
            knees = knee_indices_rows(dev, self.fs)
            per_octave = len(self.estimator) / self.estimator.fs / self.estimator.n_octaves
            fade = 2 * int(self.fs * per_octave * (1 / 24)) // 2
            keep = int(min(min(r.n for r in dev), next_fast_len(max(knees))))
            if fade > keep:
                raise ValueError("operands could not be broadcast together: fade-out longer than the response")
            # truncate + fade out into rows of a common pitch: the layout the FIR stage (K5) reads
            pitch = (keep + 63) // 64 * 64
            block = DeviceBlock(ctx, len(dev) * pitch)
            base, offs, _ = span(dev)
            ctx.apply_window_device(base, offs, block.ptr, [i * pitch for i in range(len(dev))], [keep] * len(dev),
                                    [dict(fade_out=fade)] * len(dev))
            for i, (_, _, ir) in enumerate(items):
                ir._row = Row(block, i * pitch, keep)
            return keep
        lengths = [len(ir.data) for _, _, ir in items]
        try:                                                # all knee searches in lock step on the device (K7)
            knees = [p[1] for p in decay_params_batch([ir.data for _, _, ir in items], self.fs)]
        except (_native.NativeError, _native.NativeUnavailable):
            raise
        except Exception:                                   # noqa: BLE001 - the reference tolerates analysis failures
            knees = []
            for _, _, ir in items:
                try:
                    knees.append(ir.decay_params()[1])
                except (_native.NativeError, _native.NativeUnavailable):
                    raise
                except Exception:                           # noqa: BLE001
                    knees.append(len(ir.data))
        per_octave = len(self.estimator) / self.estimator.fs / self.estimator.n_octaves
        fade = 2 * int(self.fs * per_octave * (1 / 24)) // 2
        keep = min(np.min(lengths), next_fast_len(max(knees)))
        for _, _, ir in items:
            ir.data = np.array(ir.data[:keep], dtype=np.float64)
            if fade > len(ir.data):
                raise ValueError("operands could not be broadcast together: fade-out longer than the response")
        out = _native.default_context().apply_window([ir.data for _, _, ir in items],
                                                     [dict(fade_out=fade)] * len(items))
        for (_, _, ir), y in zip(items, out):
            ir.data[:] = y
        return keep

    # ---- filtering -----------------------------------------------------------------------
    def equalize(self, fir):
        """Apply one FIR to all left and one to all right responses (rows 0 / 1 of ``fir``)."""
        if isinstance(fir, list):
            if isinstance(fir[0], ImpulseResponse):
                fir = np.vstack([fir[0].data, fir[1].data]) if len(fir) > 1 else fir[0].data.copy()
            else:
                fir = np.vstack(fir) if isinstance(fir[0], np.ndarray) else np.array(fir)
        fir = np.asarray(fir)
        if fir.ndim == 1 or fir.shape[0] == 1:
            fir = np.tile(fir, (2, 1))
        # one batched device convolution for all channels instead of the reference's per-channel loop
        self.equalize_channels({(sp, sd): (fir[0] if sd == "left" else fir[1])
                                for sp, pair in self.irs.items() for sd in pair})

    # ---- channel balance (core/hrir.py:655-799) ---------------------------------------------
    def channel_balance_firs(self, left_fr, right_fr, method):
        """Two FIRs (left, right) that bring the ears of a speaker group to a common response (core/hrir.py:655-764).
        The two curves travel as one [2, n] matrix: smoothing, gain-limited inversion and FIR design run on the device
        (K12 -> K6) for both ears at once."""
        from .frequency_response import equalization_firs, minimum_phase_impulse_response, smooth_curves
        fs = self.fs

        def band_mean(raws, lo, hi):                            # rows of raws on the curves' own grid
            band = np.logical_and(left_fr.frequency >= lo, left_fr.frequency <= hi)
            return np.mean(raws[:, band], axis=1)

        def unit_gain_pair(gain):
            n = int(round(fs * 0.1))
            return [_unit_impulse(n), _unit_impulse(n) * gain]

        grid = left_fr.frequency
        raws = np.stack([left_fr.raw, right_fr.raw])
        if method == "mids":
            ml, mr = band_mean(raws, 100, 3000)
            return unit_gain_pair(10 ** ((ml - mr) / 20))
        if method == "trend":
            trend = smooth_curves(grid, raws[0] - raws[1], 2, 1 / 3, 20000, int(round(fs / 2)))
            right_fr.equalization = trend
            fir = minimum_phase_impulse_response(grid, trend, fs, f_res=10, normalize=False)   # the method's default f_res
            return [_unit_impulse(len(fir)), fir]
        if method in ("left", "right"):
            ref, subj = (left_fr, right_fr) if method == "left" else (right_fr, left_fr)
            ref.smoothed = smooth_curves(grid, ref.raw, 1 / 3, 1 / 3, 20000, int(round(fs / 2)))
            gain = ref.center([100, 10000])                     # shifts ref.raw and ref.smoothed
            subj.raw = subj.raw + gain
            subj.target = ref.smoothed
            subj.error = subj.raw - subj.target
            eq, fir = equalization_firs(grid, subj.error, fs, smoothen_first=True, max_gain=15, treble_f_lower=20000,
                                        treble_f_upper=fs / 2, normalize=False)
            subj.equalization = eq
            return [_unit_impulse(len(fir)), fir] if method == "left" else [fir, _unit_impulse(len(fir))]
        if method in ("avg", "min"):
            raws = raws - np.mean(band_mean(raws, 100, 10000))   # (get_center_value(left) + get_center_value(right)) / 2
            target = np.mean(raws, axis=0) if method == "avg" else np.min(raws, axis=0)
            errors = smooth_curves(grid, raws - target, 1 / 3, 1 / 3, 20000, 23999)
            eqs, firs = equalization_firs(grid, errors, fs, smoothen_first=False, max_gain=15, treble_f_lower=2000,
                                          treble_f_upper=fs / 2, normalize=False)
            for fr, raw, eq in zip((left_fr, right_fr), raws, eqs):
                fr.raw, fr.target, fr.error, fr.equalization = raw, target, raw - target, eq
            return [firs[0], firs[1]]
        try:
            gain = 10 ** (float(method) / 20)
        except ValueError:
            raise ValueError(f'"{method}" is not valid value for channel balance method.')
        return unit_gain_pair(gain)

    def correct_channel_balance(self, method):
        """core/hrir.py:766-799: per speaker group, equalize the ears to the same response."""
        eqir = HRIR(self.estimator)
        for speakers in channel_balance_groups():
            if any(sp not in self.irs for sp in speakers):
                continue                                   # balancing needs the whole group
            left = np.mean(np.vstack([self.irs[sp]["left"].data for sp in speakers]), axis=0)
            right = np.mean(np.vstack([self.irs[sp]["right"].data for sp in speakers]), axis=0)
            firs = self.channel_balance_firs(ImpulseResponse(left, self.fs).frequency_response(),
                                             ImpulseResponse(right, self.fs).frequency_response(), method)
            for sp in speakers:
                self.irs[sp]["left"].equalize(firs[0])
                self.irs[sp]["right"].equalize(firs[1])
        return eqir

    def equalize_channels(self, firs):
        """`firs`: {(speaker, side): taps}.  Same result as calling ir.equalize(taps) on each channel
        (core/pipeline.py:690-691 does exactly that loop), sent to the device as one batch."""
        from .impulse_response import _k5_plans, fir_convolve_full_batch
        keys = [k for k in firs if k[0] in self.irs and k[1] in self.irs[k[0]]]
        dev = self._device_rows([self.irs[sp][sd] for sp, sd in keys])
        # FIRs the design left on the device (process_equalization_batch(on_device=True)): the rows of ONE batch, in order,
        # on this device - they go to K5 where they are
        raw = [firs[k] for k in keys]
        batch = None
        if dev is not None and raw and all(isinstance(t, _native.DeviceFir) for t in raw):
            b0 = raw[0].batch
            if (all(t.batch is b0 for t in raw) and [t.index for t in raw] == list(range(b0.B))
                    and b0.ctx.device == _native.default_context().device):
                batch = b0
        taps = raw if batch is not None else [np.asarray(t, dtype=np.float64) for t in raw]
        if dev is not None and len({len(t) for t in taps}) == 1 and len(taps[0]) > 0 and dev[0].n > 0:
            from .device_rows import DeviceBlock, Row, uniform
            pitch = uniform(dev)
            if pitch is not None:                            # device rows in, device rows out
                n, k = dev[0].n, len(taps[0])
                out_pitch = (n + k - 1 + 63) // 64 * 64
                block = DeviceBlock(_native.default_context(), len(dev) * out_pitch)
                out_len = _k5_plans.run_device(dev[0].ptr, len(dev), pitch, n, batch if batch is not None else _rows_matrix(taps),
                                               block.ptr, out_pitch)
                for i, (sp, sd) in enumerate(keys):
                    self.irs[sp][sd]._row = Row(block, i * out_pitch, out_len)
                return
        ys = fir_convolve_full_batch([self.irs[sp][sd].data for sp, sd in keys], [np.asarray(firs[k], dtype=np.float64) for k in keys])
        for (sp, sd), y in zip(keys, ys):
            self.irs[sp][sd].data = y

    def resample(self, fs):
        raise NotImplementedError("resample depends on nnresample (no oracle here, parity unpinned)")

    def correct_microphone_deviation(self, correction_strength=0.7, anchor="auto", plot_analysis=False, plot_dir=None):
        """core/hrir.py:801-857 hands the HRIR to core.microphone_deviation_correction, which is outside the hot-path
        scope: a caller that sets the flag (core/pipeline.py:631) gets a clear refusal."""
        raise NotImplementedError(
            "correct_microphone_deviation is outside the device path's scope (core.microphone_deviation_correction is "
            "not part of the accelerated hot path); run the pipeline without microphone-deviation correction")

    # ---- alignment (small host-side correlations; 'next' tier of the scope table) ----------
    def align_ipsilateral_all(self, speaker_pairs=None, segment_ms=30):
        """core/hrir.py:921-958.  The lag searches (scipy.signal.correlate + argmax per pair) run as one
        batch on the device (K10, imp_xcorr_argmax); pairs that touch a speaker an earlier pair of the
        batch may have shifted are searched after that shift, as in the reference's loop."""
        pairs = list(IPSILATERAL_PAIRS) if speaker_pairs is None else list(speaker_pairs)
        seg = int(self.fs * segment_ms / 1000)
        pairs = [(one, two) for one, two in pairs if one in self.irs and two in self.irs]
        ctx = _native.default_context()
        i = 0
        while i < len(pairs):
            batch, touched = [], set()
            while i < len(pairs) and not ({pairs[i][0], pairs[i][1]} & touched):
                batch.append(pairs[i])
                touched |= {pairs[i][0], pairs[i][1]}
                i += 1
            ears_a = [self.irs[one]["left"] for one, _ in batch]
            ears_b = [self.irs[two]["right"] for _, two in batch]
            resident = all(ir._data is None and ir._row is not None for ir in ears_a + ears_b)
            if resident:
                # the segments where the rows are: nothing comes to the host but the lags
                from .device_rows import span
                base, offs, lens = span([ir._row for ir in ears_a + ears_b])
                nb = len(batch)
                a_len = [min(seg, int(n)) for n in lens[:nb]]
                arg, _ = ctx.xcorr_argmax_device(base, offs[:nb], a_len, offs[nb:], [min(seg, int(n)) for n in lens[nb:]])
            else:
                a = [ir.data[:seg] for ir in ears_a]
                arg, _ = ctx.xcorr_argmax(a, [ir.data[:seg] for ir in ears_b])
                a_len = [len(x) for x in a]
            todo = []
            for (one, two), k, la in zip(batch, arg, a_len):
                lags = np.arange(-la + 1, la)                  # the reference indexes this with the argmax
                lag = int(lags[int(k)])
                if one == two:
                    if lag > 0:
                        todo.append((self.irs[one]["right"], lag))
                    elif lag < 0:
                        todo.append((self.irs[one]["left"], -lag))
                    continue
                target, amount = (two, lag) if lag > 0 else (one, -lag)
                if lag != 0:
                    todo.extend((self.irs[target][sd], amount) for sd in ("left", "right"))
            self._shift_all(todo)

    @staticmethod
    def _shift_all(todo):
        """[(ImpulseResponse, samples)]: device-resident responses of the list are shifted with one launch"""
        from .device_rows import shift_rows
        dev = [(ir, n) for ir, n in todo if ir._data is None and ir._row is not None and n != 0]
        if dev:
            for (ir, _), row in zip(dev, shift_rows([ir._row for ir, _ in dev], [n for _, n in dev])):
                ir._row = row
        for ir, n in todo:
            if not (ir._data is None and ir._row is not None):
                ir.shift(n)

    def align_onset_groups_peak_leftref(self, groups=None):
        if groups is None:
            groups = [("FL", "FR"), ("SL", "SR"), ("BL", "BR"), ("WL", "WR"), ("TFL", "TFR"),
                      ("TSL", "TSR"), ("TBL", "TBR"), ("FC",)]

        leaders = [g[0] for g in [("FL", "FR")] + [g for g in groups if g != ("FL", "FR")]
                   if g[0] in self.irs and "left" in self.irs[g[0]]]
        peaks = {}
        on_dev = [sp for sp in leaders if self.irs[sp]["left"]._data is None and self.irs[sp]["left"]._row is not None]
        if on_dev:                                             # the leaders' peaks in one device call (K3)
            from .device_rows import span
            base, offs, lens = span([self.irs[sp]["left"]._row for sp in on_dev])
            idx, _ = _native.default_context().peak_index_device(base, offs, lens)
            peaks = {sp: int(k) for sp, k in zip(on_dev, idx)}

        def lead_peak(group):
            sp = group[0]
            if sp not in self.irs or "left" not in self.irs[sp]:
                return None
            return peaks[sp] if sp in peaks else self.irs[sp]["left"].peak_index()

        ref = lead_peak(("FL", "FR"))
        if ref is None:
            raise RuntimeError("Cannot find FL left channel reference for onset alignment.")
        todo = []
        for group in groups:
            if group == ("FL", "FR"):
                continue
            pk = lead_peak(group)
            if pk is None:
                continue
            for sp in group:
                if sp in self.irs:
                    todo.extend((self.irs[sp][sd], -(pk - ref)) for sd in ("left", "right"))
        self._shift_all(todo)
