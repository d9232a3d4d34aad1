"""The hot-path slice of the reference's pipeline for MANY measurements per call, device resident.

`pipeline_slice.run_slice` walks one measurement through the reference's stages (core/pipeline.py:565-573, 585-601,
647-692, 725-735) with the class surface between them: every stage brings the scalars it decides with - peaks, knees,
spectrum maxima - back to the host.  A job of many measurements with one layout (the same files, speakers and
equalisation curves: a listener measured again, a room measured at several seats) does not need that: here the stage
sequence of M measurements is ONE stream-ordered sequence of launches (imp_slice, include/impulse_hip.h), decisions
are taken where the data is, and the scalars come back once at the end.

What the device cannot promise to decide exactly as the host flow would - a Lundeby search with a decision inside its
guard band, a gain on an fp32 rounding boundary, a crop_tails length above the capacity the slice was sized for - is
flagged, never guessed: those measurements (rare) go through `run_slice`, the staged path, so every result is the staged
path's result bit for bit.
"""
import time
import warnings

import numpy as np

from . import _native
from .constants import IPSILATERAL_PAIRS, SPEAKER_DELAYS, speaker_side
from .device_rows import DeviceBlock, Row
from .hrir import HRIR, next_fast_len, split_recording
from .impulse_response import ImpulseResponse


class Layout:
    """The files of one measurement: [(n_frames, tracks, speakers[, silence_length])], every file interleaved PCM frames of
    one sample type with two tracks per speaker column (left ear, right ear: core/hrir.py:326-341)."""

    def __init__(self, estimator, files, dtype=np.int32):
        self.estimator = estimator
        self.fs = estimator.fs
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.int16), np.dtype(np.int32)):
            raise ValueError("recordings are int16 / int32 PCM frames (other sample types: HRIR.open_recording)")
        self.files = []
        self.tasks = []                    # (speaker, side) in the order HRIR.irs lists them
        pair_offsets, speakers, columns = [], [], []
        base = 0
        tracks0 = None
        for spec in files:
            n_frames, tracks, names = int(spec[0]), int(spec[1]), list(spec[2])
            silence = spec[3] if len(spec) > 3 else 2.0
            if tracks0 is None:
                tracks0 = tracks
            if tracks != tracks0 or tracks % 2:
                raise ValueError("the resident slice takes recordings of one even track count")
            shape_only = np.broadcast_to(np.zeros(1), (tracks, n_frames))
            rec, jobs = split_recording(shape_only, names, len(estimator), self.fs, None, silence)
            origin = n_frames - rec.shape[1]
            for k in range(0, len(jobs), 2):
                (sp, sd, tr, a, b), (sp2, sd2, tr2, a2, b2) = jobs[k], jobs[k + 1]
                if (sp2, sd, sd2, tr2, a2, b2) != (sp, "left", "right", tr + 1, a, b):
                    raise ValueError("recording geometry is not left / right pairs")
                if sp in speakers:
                    raise ValueError(f"speaker {sp} appears twice in the measurement")
                speakers.append(sp)
                columns.append((len(self.files), origin + a, b - a, tr))
                pair_offsets.append(base + (origin + a) * tracks + tr)
            self.files.append((n_frames, tracks, names, silence, base))
            base += n_frames * tracks
        if not speakers:
            raise ValueError("no speaker column in the layout")
        if len({c[2] for c in columns}) != 1:
            raise ValueError("the resident slice takes columns of one length")
        self.tracks = tracks0
        self.samples = base                                # samples per measurement, all files
        self.speakers = speakers
        self.columns = columns
        self.column_len = columns[0][2]
        self.pair_offsets = np.array(pair_offsets, dtype=np.int64)
        self.tasks = [(sp, sd) for sp in speakers for sd in ("left", "right")]

    def pack(self, recordings):
        """the files of one measurement ([frames[n_frames, tracks], ...]) as one block of samples"""
        if len(recordings) != len(self.files):
            raise ValueError("one frame block per file of the layout")
        out = np.empty(self.samples, dtype=self.dtype)
        for fr, (n_frames, tracks, _, _, base) in zip(recordings, self.files):
            fr = np.asarray(fr)
            if fr.shape != (n_frames, tracks) or fr.dtype != self.dtype:
                raise ValueError(f"expected frames {(n_frames, tracks)} of {self.dtype}, got {fr.shape} of {fr.dtype}")
            out[base:base + n_frames * tracks] = fr.reshape(-1)
        return out


ONSET_GROUPS = [("FL", "FR"), ("SL", "SR"), ("BL", "BR"), ("WL", "WR"), ("TFL", "TFR"), ("TSL", "TSR"), ("TBL", "TBR"), ("FC",)]


def alignment_tables(speakers):
    """The two alignments of core/hrir.py:921-1001 as tables over the ear pairs of a layout (pair q = speakers[q]):
    (ipsilateral pairs [(q1, q2)] of IPSILATERAL_PAIRS whose speakers are both present, leader_of_pair[q] = the pair whose
    left ear gives the onset group of q its peak - or -1: the reference group FL/FR, speakers in no group, groups whose
    FIRST speaker is absent (the reference skips those) -, FL's pair).  Raises as the reference does without FL."""
    idx = {sp: q for q, sp in enumerate(speakers)}
    if "FL" not in idx:
        raise RuntimeError("Cannot find FL left channel reference for onset alignment.")
    ipsi = [(idx[a], idx[b]) for a, b in IPSILATERAL_PAIRS if a in idx and b in idx]
    leader = [-1] * len(speakers)
    for group in ONSET_GROUPS:
        if group == ("FL", "FR") or group[0] not in idx:
            continue
        for sp in group:
            if sp in idx:
                leader[idx[sp]] = idx[group[0]]
    return ipsi, leader, idx["FL"]


class WavMeasurements:
    """The measurements of a job as WAV files on disk: measurement i = files[i] = [path of file 0, path of file 1, ...] in the
    order of the layout's files.  A sequence for the job runners (len, [i]): measurement i's PCM blocks are read
    (`audio_io.read_wav_pcm`: the file's own int16 / int32 frames, no conversion) when a runner's upload stage asks for
    them - the read of measurement i + 1 overlaps the compute and the download of measurement i - and nothing is kept."""

    def __init__(self, files, fs=None):
        self.files = [list(f) if isinstance(f, (list, tuple)) else [f] for f in files]
        self.fs = fs

    @classmethod
    def from_dirs(cls, dir_paths, fs=None):
        """(measurements, speakers_per_file) for measurement DIRECTORIES as the reference lays them out
        (core/pipeline_stages.py:504-522 open_binaural_measurements): every file named `<speaker list>.wav` (FL,FR.wav,
        FC.wav, ...) is one recording of the measurement, its speakers taken from the name, in the directory's listing
        order.  All directories must hold the same file names (one layout per job)."""
        import os
        import re
        from .room_correction import SPEAKER_LIST_PATTERN
        pattern = re.compile(rf"^{SPEAKER_LIST_PATTERN}\.wav$")
        names = None
        files = []
        for d in dir_paths:
            found = [f for f in os.listdir(d) if pattern.match(f)]
            if not found:
                raise ValueError("No HRIR recordings found in the directory.")
            if names is None:
                names = found
            elif sorted(found) != sorted(names):
                raise ValueError(f"{d}: recordings {sorted(found)} differ from the job's layout {sorted(names)}")
            files.append([os.path.join(d, f) for f in names])
        speakers = [re.search(SPEAKER_LIST_PATTERN, f)[0].split(",") for f in names]
        return cls(files, fs), speakers

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return WavMeasurements(self.files[i], self.fs)
        from .audio_io import read_wav_pcm
        out = []
        for path in self.files[i]:
            got = read_wav_pcm(path)
            if got is None:
                raise ValueError(f"{path}: the resident slice takes 16 / 32-bit PCM files (others: HRIR.open_recording)")
            fs, frames = got
            if self.fs is not None and fs != self.fs:
                raise ValueError(f"{path}: sampling rate {fs}, expected {self.fs}")
            out.append(frames)
        return out

    def layout(self, estimator, speakers_per_file, silence_length=2.0):
        """the Layout of these measurements, from the first one's files: speakers_per_file[k] = the speakers recorded in
        file k (core/hrir.py:307-355's `speakers` argument)"""
        first = self[0]
        if len(first) != len(speakers_per_file):
            raise ValueError("one speaker list per file of a measurement")
        files = [(fr.shape[0], fr.shape[1], list(sp), silence_length) for fr, sp in zip(first, speakers_per_file)]
        return Layout(estimator, files, dtype=first[0].dtype)


class ResidentSlice:
    """ingest -> crop_heads -> crop_tails -> equalize -> normalize for up to `max_measurements` measurements of one
    layout per call.  FIRs are per job (set_firs); results are HRIR objects whose responses are device rows."""

    def __init__(self, estimator, layout, max_measurements=8, head_ms=1, peak_target=-0.1, taps=None, keep_cap=None,
                 paired=True):
        self.estimator, self.layout = estimator, layout
        fs = estimator.fs
        self.fs = fs
        self.head_ms, self.peak_target = head_ms, peak_target
        self.head = int(head_ms * fs / 1000)
        per_octave = len(estimator) / estimator.fs / estimator.n_octaves
        self.fade = 2 * int(fs * per_octave * (1 / 24)) // 2
        self.taps = int(taps) if taps is not None else _fir_taps(fs)
        self.plan = estimator._plan(layout.column_len, paired=paired and layout.tracks == 2)
        self.ctx = self.plan.ctx
        self.max_measurements = int(max_measurements)
        self.delays = [int(np.round(SPEAKER_DELAYS[sp] * fs)) + self.head for sp in layout.speakers]
        self.bits = {np.dtype(np.int16): 16, np.dtype(np.int32): 32}[layout.dtype]
        self.firs = None
        self.decay = None
        self.align = False
        self.slice = None
        self.stats = dict(measurements=0, staged=0, regrown=0)
        self._make(self._cap_for(int(1.1 * fs)) if keep_cap is None else int(keep_cap))

    def _cap_for(self, keep):
        """the largest crop_tails length whose normalisation transform (2^k >= 2 (keep + taps - 1) - 1 points) is the one
        `keep` needs: 55 937 samples (1.17 s) at 48 kHz / 9 600 taps, 111 873 at 96 kHz / 19 200"""
        want = int(keep) + self.taps
        size = 1 << max(2 * want - 2, 1).bit_length()
        return (size + 1) // 2 - self.taps + 1

    def _make(self, keep_cap):
        if self.slice is not None:
            self.slice.close()
        self.keep_cap = int(min(keep_cap, self.plan.out_len))
        self.slice = _native.Slice(self.plan, self.layout.pair_offsets, self.delays, self.layout.tracks, self.bits, self.head,
                                   self.fade, self.taps, self.keep_cap, self.fs, peak_target=self.peak_target,
                                   max_measurements=self.max_measurements)
        self.out_pitch = (self.slice.out_len_max + 63) // 64 * 64
        if self.firs is not None:
            self.set_firs(self.firs)
        if self.decay is not None:
            self.set_decay(self.decay)
        if self.align:
            self.set_alignment(True)

    def set_alignment(self, on=True):
        """the alignments `_stage_crop_and_align` runs between crop_heads and crop_tails (core/pipeline.py:593-597):
        align_ipsilateral_all over IPSILATERAL_PAIRS with 30 ms segments, then align_onset_groups_peak_leftref - lag searches,
        leader peaks and the shifted rows all on the device"""
        self.align = bool(on)
        if not on:
            self.slice.set_alignment(None, None, 0, 0)
            return
        ipsi, leader, ref = alignment_tables(self.layout.speakers)
        self.slice.set_alignment(ipsi, leader, ref, int(self.fs * 30 / 1000))

    def set_decay(self, decay):
        """the optional stage between equalize and normalize (core/pipeline.py:694-716): decay = None (off), a target RT60
        in seconds for every speaker, or {speaker: seconds} for the speakers to adjust"""
        self.decay = decay
        if decay is None:
            self.slice.set_decay(None)
            return
        if isinstance(decay, dict):
            per_row = [float(decay[sp]) if sp in decay else np.nan for sp, _ in self.layout.tasks]
        else:
            per_row = [float(decay)] * len(self.layout.tasks)
        self.slice.set_decay(per_row)

    def grow_for(self, rows):
        """After a call that flagged IMP_SLICE_KEEP_CAP: size the slice for the crop_tails lengths those rows ask for (the
        knees came back with the scalars), so that the call can be repeated.  The capacity only ever grows."""
        R = self.slice.rows
        need = 0
        for m in range(len(rows) // R):
            r = rows[m * R:(m + 1) * R]
            need = max(need, min(int(r["len"].min()), next_fast_len(int(r["knee"].max()))))
        if need <= self.keep_cap:
            return False
        self.stats["regrown"] += 1
        self._make(self._cap_for(need))
        return True

    def set_firs(self, firs):
        """firs: {(speaker, side): taps}, a [2 * speakers, taps] matrix in layout.tasks order, or the _native.DeviceFirs
        batch a design left on this device (process_equalization_batch(on_device=True): rows in layout.tasks order) - that
        one is taken where it is: no upload, no wait on the slice's stream beyond the design's completion"""
        if isinstance(firs, dict):
            rows = [firs[t] for t in self.layout.tasks]
            if rows and all(isinstance(r, _native.DeviceFir) for r in rows) and all(r.batch is rows[0].batch for r in rows) \
                    and [r.index for r in rows] == list(range(rows[0].batch.B)):
                firs = rows[0].batch
            else:
                firs = np.stack([np.asarray(r, dtype=np.float64) for r in rows])
        if isinstance(firs, _native.DeviceFirs):
            if (firs.B, firs.taps) != (self.slice.rows, self.taps) or firs.ctx.device != self.ctx.device:
                raise ValueError(f"device FIRs must be [{self.slice.rows}, {self.taps}] on device {self.ctx.device}")
            self.firs = firs
            if firs.ctx is not self.ctx:
                firs.ready()                               # designed on another stream: wait for it (once)
            self.slice.set_firs_device(firs.ptr, firs.taps)     # same context: stream order is enough
            return
        self.firs = np.ascontiguousarray(firs, dtype=np.float64)
        self.slice.set_firs(self.firs)

    def firs_by_task(self):
        """{(speaker, side): taps} on the host (what the staged path takes)"""
        host = self.firs.host() if isinstance(self.firs, _native.DeviceFirs) else self.firs
        return {t: host[i] for i, t in enumerate(self.layout.tasks)}

    def execute_device(self, d_rec, M, d_out=None):
        """M measurements already on the device (layout.samples apart, first at d_rec).  Asynchronous; returns the output
        block (rows [M * rows][out_pitch] fp32).  collect() brings the scalars back and builds the HRIRs."""
        block = None
        if d_out is None:
            block = DeviceBlock(self.ctx, M * self.slice.rows * self.out_pitch)
            d_out = block.ptr
        self.slice.execute_device(d_rec, self.layout.samples, M, d_out, self.out_pitch)
        return block

    def collect(self, block, recordings=None, staged=None):
        """[(HRIR, gain dB)] of the last call.  Measurements the device flagged are replaced by staged(m) (the staged
        path of measurement m); without `staged` a flagged measurement raises."""
        rows, meas = self.slice.results()
        R = self.slice.rows
        out = []
        for m in range(len(meas)):
            self.stats["measurements"] += 1
            fl = int(meas["flags"][m])
            if fl & _native.SLICE_REDO:
                self.stats["staged"] += 1
                if staged is None:
                    raise _native.NativeError(-3, f"measurement {m}: the device left a decision to the staged path (flags {fl})")
                out.append(staged(m))
                continue
            self._warn_sides(rows[m * R:(m + 1) * R])
            hrir = HRIR(self.estimator)
            n = int(meas["out_len"][m])
            for q, sp in enumerate(self.layout.speakers):
                pair = {}
                for s, sd in enumerate(("left", "right")):
                    b = m * R + 2 * q + s
                    pair[sd] = ImpulseResponse.on_device(Row(block, b * self.out_pitch, n), self.fs,
                                                         self._column(recordings, m, q, s))
                hrir.irs[sp] = pair
            out.append((hrir, float(meas["gain_db"][m])))
        return out

    def collect_host(self, host_rows, rows, meas, recordings=None, m=0):
        """(HRIR, gain dB) of measurement m of the last call from its float64 rows on the host ([R, out_len], the packed
        layout imp_slice_pack_f64 leaves): the responses are views of host_rows, nothing is copied"""
        R = self.slice.rows
        self.stats["measurements"] += 1
        self._warn_sides(rows[m * R:(m + 1) * R])
        hrir = HRIR(self.estimator)
        for q, sp in enumerate(self.layout.speakers):
            hrir.irs[sp] = {sd: ImpulseResponse(host_rows[2 * q + s], self.fs, self._column(recordings, m, q, s))
                            for s, sd in enumerate(("left", "right"))}
        return hrir, float(meas["gain_db"][m])

    def _column(self, recordings, m, q, s):
        if recordings is None:
            return None
        f, start, length, tr = self.layout.columns[q]
        scale = 1.0 / float(2 ** (8 * self.layout.dtype.itemsize - 1))
        return lambda: np.asarray(recordings[m][f])[start:start + length, tr + s].astype(np.float64) * scale

    def _warn_sides(self, rows):
        """crop_heads' warning for a speaker whose sound reaches the far ear first (core/hrir.py:569-596)"""
        for q, sp in enumerate(self.layout.speakers):
            p_left, p_right = int(rows["peak"][2 * q]), int(rows["peak"][2 * q + 1])
            wrong = "right" if p_left < p_right else "left"
            if speaker_side(sp) == wrong:
                early = "left" if wrong == "right" else "right"
                itd_ms = abs(p_left - p_right) / self.fs * 1000
                warnings.warn(
                    f"Warning: {sp} measurement has lower delay to {early} ear than to {wrong} ear. "
                    f"{sp} should be at the {wrong} side of the head so the sound should arrive first in the "
                    f"{wrong} ear. This is usually a problem with the measurement process or the speaker order "
                    f"given is not correct. Detected delay difference is {itd_ms:.4f} milliseconds.")

    def run(self, measurements):
        """measurements: [[frames of file 0, frames of file 1, ...], ...] in host memory.  Returns [(HRIR, gain dB)]:
        uploads, runs the resident sequence in calls of up to max_measurements, replaces flagged measurements by the
        staged path."""
        from .pipeline_slice import run_slice
        if self.firs is None:
            raise ValueError("set_firs first")
        results = []
        firs = None
        for m0 in range(0, len(measurements), self.max_measurements):
            batch = measurements[m0:m0 + self.max_measurements]
            M = len(batch)
            d_rec = self.ctx.malloc(M * self.layout.samples * self.layout.dtype.itemsize)
            try:
                for m, recs in enumerate(batch):
                    self.ctx.h2d(d_rec + m * self.layout.samples * self.layout.dtype.itemsize, self.layout.pack(recs))
                block = self.execute_device(d_rec, M)
                rows, meas = self.slice.results()
                if np.any(meas["flags"] & _native.SLICE_KEEP_CAP) and self.grow_for(rows):
                    block = self.execute_device(d_rec, M)      # once more, with room for the longest response
            finally:
                self.ctx.free(d_rec)                       # ordered on the stream

            def staged(m, batch=batch):
                jobs = [((self.fs, np.asarray(fr)), spec[2], None) for fr, spec in zip(batch[m], self.layout.files)]
                return run_slice(self.estimator, jobs, head_ms=self.head_ms, peak_target=self.peak_target, firs=self.firs_by_task(),
                                 decay=self.decay, align=self.align)

            results.extend(self.collect(block, batch, staged))
        return results

    def upload(self, d_rec, recordings, ctx=None):
        """the files of one measurement to their places in a device block (no host-side packing: every file goes up as it
        is); ctx: the context whose stream carries the copies (default: the slice's)"""
        ctx = ctx or self.ctx
        item = self.layout.dtype.itemsize
        for fr, (n_frames, tracks, _, _, base) in zip(recordings, self.layout.files):
            fr = np.asarray(fr)
            if fr.shape != (n_frames, tracks) or fr.dtype != self.layout.dtype:
                raise ValueError(f"expected frames {(n_frames, tracks)} of {self.layout.dtype}, got {fr.shape} of {fr.dtype}")
            ctx.h2d(d_rec + base * item, fr)

    def close(self):
        self.slice.close()


class SliceRunner:
    """Jobs of many measurements of one layout, end to end from host memory.

    `workers` lanes, each a host thread with a context (stream) of its own and a one-measurement resident slice, all kept
    for the runner's life: while one measurement's recording crosses PCIe, another's stage sequence runs and a third's
    responses come back - the upload of measurement i + 1 overlaps the compute of measurement i."""

    def __init__(self, estimator, layout, workers=3, head_ms=1, peak_target=-0.1, pinned_mb=None):
        import queue
        import threading
        self.estimator, self.layout = estimator, layout
        self.head_ms, self.peak_target = head_ms, peak_target
        self.lanes = []
        self._lock = threading.Lock()
        self._run_lock = threading.Lock()
        self.pool = _native.PinnedPool(pinned_mb)
        for k in range(max(1, int(workers))):
            ln = dict(todo=queue.Queue(), done=queue.Queue(), ready=threading.Event(), error=None, times={})
            ln["thread"] = threading.Thread(target=self._worker, args=(ln,), name=f"impulse-slice-{k}", daemon=True)
            ln["thread"].start()
            self.lanes.append(ln)
        for ln in self.lanes:
            ln["ready"].wait()
            if ln["error"] is not None:
                raise ln["error"]

    def _worker(self, ln):
        """a lane's thread: makes its context and slice, then serves jobs until it is told to stop"""
        from .pipeline_slice import run_slice
        est, layout = self.estimator, self.layout
        try:
            ctx = _native.Context(_native.default_device())
            with _native.using_context(ctx):
                rs = ResidentSlice(est, layout, max_measurements=1, head_ms=self.head_ms, peak_target=self.peak_target)
            d_rec = ctx.malloc(layout.samples * layout.dtype.itemsize)
            ln.update(ctx=ctx, rs=rs, d_rec=d_rec)
        except BaseException as exc:                       # noqa: BLE001 - reported to the constructor
            ln["error"] = exc
            ln["ready"].set()
            return
        ln["ready"].set()
        with _native.using_context(ctx):
            while True:
                job = ln["todo"].get()
                if job is None:
                    break
                try:
                    rs.set_firs(job["firs"])
                    rs.set_decay(job["decay"])
                    rs.set_alignment(job["align"])
                    while True:
                        with self._lock:
                            i = job["next"]
                            job["next"] += 1
                        if i >= len(job["measurements"]):
                            break
                        recs = job["measurements"][i]
                        t0 = time.perf_counter()
                        rs.upload(d_rec, recs)
                        t1 = time.perf_counter()
                        host = job["to_host"]
                        block = self._lane_execute(ln, host)
                        t2 = time.perf_counter()
                        rows, meas = rs.slice.results()
                        t3 = time.perf_counter()
                        if np.any(meas["flags"] & _native.SLICE_KEEP_CAP) and rs.grow_for(rows):
                            block = self._lane_execute(ln, host)
                            rows, meas = rs.slice.results()

                        def staged(m, recs=recs):
                            jobs = [((est.fs, np.asarray(fr)), spec[2], None) for fr, spec in zip(recs, layout.files)]
                            return run_slice(est, jobs, head_ms=self.head_ms, peak_target=self.peak_target, firs=job["firs"],
                                             decay=job["decay"], align=job["align"])

                        if not host:
                            res = rs.collect(block, [recs], staged)[0]
                            t4 = t5 = time.perf_counter()
                        elif int(meas["flags"][0]) & _native.SLICE_REDO:
                            rs.stats["measurements"] += 1
                            rs.stats["staged"] += 1
                            res = staged(0)
                            t4 = time.perf_counter()
                            res[0].to_host()
                            t5 = time.perf_counter()
                        else:
                            # the packed float64 rows over the link into recycled page-locked memory: one linear copy
                            R, n = rs.slice.rows, int(meas["out_len"][0])
                            blk = self.pool.take(ctx, R * n)
                            flat = np.asarray(blk)[:R * n] if blk is not None else np.empty(R * n)
                            ctx.d2h(flat, ln["d_packed"])
                            t4 = time.perf_counter()
                            res = rs.collect_host(flat.reshape(R, n), rows, meas, [recs])
                            t5 = time.perf_counter()
                        job["out"][i] = res
                        for k, dt in (("upload", t1 - t0), ("launch", t2 - t1), ("wait", t3 - t2), ("to_host", t4 - t3),
                                      ("collect", t5 - t4), ("measurements", 1)):
                            ln["times"][k] = ln["times"].get(k, 0.0) + dt
                    ln["done"].put(None)
                except BaseException as exc:               # noqa: BLE001 - re-raised in the caller's thread
                    ln["done"].put(exc)
            ctx.free(d_rec)
            for k in ("d_out", "d_packed"):
                if ln.get(k):
                    ctx.free(ln[k])
            rs.close()
            est._forget_context(ctx)
        ctx.close()

    def _lane_execute(self, ln, host):
        """one measurement through the lane's slice.  host: the rows stay in a block the lane keeps and are packed as float64
        for the copy-out (returns None); otherwise a device block of their own is returned, as ResidentSlice.execute_device"""
        rs, ctx = ln["rs"], ln["ctx"]
        if not host:
            return rs.execute_device(ln["d_rec"], 1)
        R, cap = rs.slice.rows, rs.slice.out_len_max
        if ln.get("cap") != cap:                           # first measurement, or the slice was re-made for longer responses
            for k in ("d_out", "d_packed"):
                if ln.get(k):
                    ctx.free(ln[k])
            ln["d_out"] = ctx.malloc(R * rs.out_pitch * 4)
            ln["d_packed"] = ctx.malloc(R * cap * 8)
            ln["cap"] = cap
        rs.execute_device(ln["d_rec"], 1, ln["d_out"])
        rs.slice.pack_f64(ln["d_out"], rs.out_pitch, 1, ln["d_packed"], R * cap)
        return None

    def run(self, measurements, firs, to_host=True, decay=None, align=False):
        """[(HRIR, gain dB)] in the order of `measurements` ([[frames of file 0, ...], ...]).  firs: {(speaker, side):
        taps}, designed once per job (the curves belong to the job, core/pipeline.py:668-688).  to_host: True = the
        responses as float64 host arrays (as the reference's classes hold them), converted inside the workers;
        False = left on the device.  decay: as ResidentSlice.set_decay; align: as ResidentSlice.set_alignment."""
        job = dict(measurements=measurements, firs=firs, to_host=to_host, decay=decay, align=align, next=0, out=[None] * len(measurements))
        lanes = self.lanes[:max(1, min(len(self.lanes), len(measurements)))]
        with self._run_lock:                               # one job at a time: the lanes' completion markers carry no job identity
            for ln in lanes:
                ln["todo"].put(job)
            errors = [e for e in (ln["done"].get() for ln in lanes) if e is not None]
        if errors:
            raise errors[0]
        return job["out"]

    def times(self, reset=True):
        """seconds the lanes spent per stage since the last reset, summed over lanes: {upload, launch, wait, collect,
        to_host, measurements}"""
        tot = {}
        for ln in self.lanes:
            for k, v in ln["times"].items():
                tot[k] = tot.get(k, 0.0) + v
            if reset:
                ln["times"] = {}
        return tot

    def close(self):
        for ln in self.lanes:
            ln["todo"].put(None)
        for ln in self.lanes:
            ln["thread"].join()
        self.lanes = []
        self.pool.close()


class SlicePipeline:
    """Jobs of many measurements of one layout, end to end from host memory, as a three-stage pipeline.

    upload | compute | download: three host threads with a stream each and rings of device buffers between them.  The
    upload thread does nothing but push recordings across the link (one synchronous copy after the other: the link is the
    slowest stage and never waits for a launch or a readback), the compute thread runs imp_slice one measurement per call
    and packs the rows as float64 on the device, the download thread brings them into recycled page-locked memory
    (_native.PinnedPool) with one linear copy per measurement and builds the HRIR objects.  Hand-overs are host queues
    after synchronous copies / imp_slice_results, so no cross-stream event is needed.  Same surface as SliceRunner."""

    def __init__(self, estimator, layout, depth=3, head_ms=1, peak_target=-0.1, pinned_mb=None, keep_cap=None, device=None):
        import queue
        import threading
        self.estimator, self.layout = estimator, layout
        self.device = _native.default_device() if device is None else int(device)
        self.head_ms, self.peak_target, self.keep_cap = head_ms, peak_target, keep_cap
        self.depth = max(2, int(depth))
        self.pool = _native.PinnedPool(pinned_mb)
        self.free_rec, self.uploaded, self.free_packed, self.packed = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
        self.jobs = [queue.Queue() for _ in range(2)]       # compute, upload; the download stage takes its jobs from the items
        self._submit = threading.Lock()
        self._times = {}
        self._tlock = threading.Lock()
        self.state = dict(error=None, ready=threading.Event())
        self.ctxs = [None, None, None]
        self.threads = [threading.Thread(target=fn, name=f"impulse-slice-{name}", daemon=True)
                        for fn, name in ((self._compute, "compute"), (self._upload, "upload"), (self._download, "download"))]
        self.threads[0].start()                            # makes the slice and the rings, then the others start
        self.state["ready"].wait()
        if self.state["error"] is not None:
            raise self.state["error"]
        for t in self.threads[1:]:
            t.start()

    def frame_buffers(self):
        """One measurement's recording buffers in page-locked memory: [frames[n_frames, tracks] of the layout's sample
        type, one per file], for a reader to fill (a WAV data chunk read straight into them): their upload needs no
        staging copy on the way to the link.  They go back to the runner's pool when the caller drops them."""
        out = []
        for n_frames, tracks, _, _, _ in self.layout.files:
            nbytes = n_frames * tracks * self.layout.dtype.itemsize
            blk = self.pool.take(self.ctxs[0], -(-nbytes // 8))
            if blk is None:
                out.append(np.empty((n_frames, tracks), dtype=self.layout.dtype))
            else:
                out.append(np.asarray(blk).view(self.layout.dtype)[:n_frames * tracks].reshape(n_frames, tracks))
        return out

    # ---- bookkeeping
    def _add(self, **kv):
        with self._tlock:
            for k, v in kv.items():
                self._times[k] = self._times.get(k, 0.0) + v

    def times(self, reset=True):
        """seconds per stage since the last reset: {upload, upload_stall, launch, wait, compute_stall, to_host, collect,
        download_stall, measurements}; *_stall = the stage waiting for its neighbour"""
        with self._tlock:
            out = dict(self._times)
            if reset:
                self._times = {}
        return out

    def _fail(self, job, exc):
        if job["error"] is None:
            job["error"] = exc

    # ---- stage 1: the link, upward
    def _upload(self):
        ctx = self.ctxs[1] = _native.Context(self.device)
        with _native.using_context(ctx):
            while True:
                job = self.jobs[1].get()
                if job is None:
                    break
                for i, recs in enumerate(job["measurements"]):
                    t0 = time.perf_counter()
                    k = self.free_rec.get()
                    t1 = time.perf_counter()
                    ok = job["error"] is None
                    if ok:
                        try:
                            self.rs.upload(self.d_rec[k], recs, ctx)
                        except BaseException as exc:       # noqa: BLE001 - re-raised by run()
                            self._fail(job, exc)
                            ok = False
                    self._add(upload_stall=t1 - t0, upload=time.perf_counter() - t1)
                    self.uploaded.put((i, k, recs, ok))
        ctx.close()

    # ---- stage 2: the stage sequence
    def _rings(self, held=None):
        """the fp32 row block and the float64 hand-over ring for the slice's present capacity.  held: the ring index the
        compute stage holds when the slice was re-made for longer responses - the others are collected first (the download
        stage hands them back as it finishes), so nothing is freed under a copy"""
        rs, ctx = self.rs, self.ctxs[0]
        R, cap = rs.slice.rows, rs.slice.out_len_max
        others = [] if held is None else [self.free_packed.get() for _ in range(self.depth - 1)]
        try:
            for ptr in [getattr(self, "d_out", 0)] + list(getattr(self, "d_packed", [])):
                if ptr:
                    ctx.free(ptr)
            self.d_out, self.d_packed, self.cap = 0, [], None
            self.d_out = ctx.malloc(R * rs.out_pitch * 4)
            self.d_packed = [ctx.malloc(R * cap * 8) for _ in range(self.depth)]
            ctx.synchronize()
            self.cap = cap
        finally:                                           # (an allocation that fails must not strand the ring's indices)
            for j in (range(self.depth) if held is None else others):
                self.free_packed.put(j)

    def _compute(self):
        from .pipeline_slice import run_slice
        est, layout = self.estimator, self.layout
        try:
            ctx = self.ctxs[0] = _native.Context(self.device)
            with _native.using_context(ctx):
                rs = self.rs = ResidentSlice(est, layout, max_measurements=1, head_ms=self.head_ms, peak_target=self.peak_target,
                                             keep_cap=self.keep_cap)
                self.d_rec = [ctx.malloc(layout.samples * layout.dtype.itemsize) for _ in range(self.depth)]
                for k in range(self.depth):
                    self.free_rec.put(k)
                self._rings()
        except BaseException as exc:                       # noqa: BLE001 - reported to the constructor
            self.state["error"] = exc
            self.state["ready"].set()
            return
        self.state["ready"].set()
        with _native.using_context(ctx):
            while True:
                job = self.jobs[0].get()
                if job is None:
                    break
                host = job["to_host"]
                try:
                    rs.set_firs(_firs_for(job["firs"], layout, ctx))
                    rs.set_decay(job["decay"])
                    rs.set_alignment(job["align"])
                except BaseException as exc:               # noqa: BLE001
                    self._fail(job, exc)
                for _ in range(len(job["measurements"])):
                    t0 = time.perf_counter()
                    i, k, recs, ok = self.uploaded.get()
                    t1 = time.perf_counter()
                    j = self.free_packed.get() if host else None
                    t2 = t3 = t4 = time.perf_counter()
                    out = None
                    try:
                        if not ok or job["error"] is not None:
                            raise _Skip()

                        def once():
                            if not host:
                                return rs.execute_device(self.d_rec[k], 1)
                            if self.cap != rs.slice.out_len_max:      # grown by a job that left its rows on the device, or a failed re-make
                                self._rings(held=j)
                            rs.execute_device(self.d_rec[k], 1, self.d_out)
                            rs.slice.pack_f64(self.d_out, rs.out_pitch, 1, self.d_packed[j], rs.slice.rows * self.cap)
                            return None

                        block = once()
                        t3 = time.perf_counter()
                        rows, meas = rs.slice.results()
                        t4 = time.perf_counter()
                        if np.any(meas["flags"] & _native.SLICE_KEEP_CAP) and rs.grow_for(rows):
                            if j is not None:
                                self._rings(held=j)
                            block = once()
                            rows, meas = rs.slice.results()
                        if int(meas["flags"][0]) & _native.SLICE_REDO:
                            rs.stats["measurements"] += 1
                            rs.stats["staged"] += 1
                            jobs = [((est.fs, np.asarray(fr)), spec[2], None) for fr, spec in zip(recs, layout.files)]
                            res = run_slice(est, jobs, head_ms=self.head_ms, peak_target=self.peak_target, firs=job["firs"],
                                            decay=job["decay"], align=job["align"])
                            if host:
                                res[0].to_host()
                            out = ("result", res)
                        elif host:
                            out = ("packed", j, rows, meas, recs)
                            j = None
                        else:
                            out = ("result", rs.collect(block, [recs])[0])
                    except _Skip:
                        pass
                    except BaseException as exc:           # noqa: BLE001
                        self._fail(job, exc)
                    self.free_rec.put(k)
                    if j is not None:
                        self.free_packed.put(j)
                    self._add(compute_stall=(t1 - t0) + (t2 - t1), launch=t3 - t2, wait=t4 - t3, measurements=1)
                    self.packed.put((job, i, out))
            for ptr in self.d_rec + [getattr(self, "d_out", 0)] + list(getattr(self, "d_packed", [])):
                if ptr:
                    ctx.free(ptr)
            rs.close()
            est._forget_context(ctx)
        ctx.close()

    # ---- stage 3: the link, downward
    def _download(self):
        ctx = self.ctxs[2] = _native.Context(self.device)
        with _native.using_context(ctx):
            while True:
                t0 = time.perf_counter()
                item = self.packed.get()
                if item is None:
                    break
                job, i, out = item
                t1 = t2 = time.perf_counter()
                try:
                    if out is None:
                        pass
                    elif out[0] == "result":
                        job["out"][i] = out[1]
                    else:
                        _, j, rows, meas, recs = out
                        try:
                            R, n = self.rs.slice.rows, int(meas["out_len"][0])
                            blk = self.pool.take(ctx, R * n)
                            flat = np.asarray(blk)[:R * n] if blk is not None else np.empty(R * n)
                            ctx.d2h(flat, self.d_packed[j])
                        finally:
                            self.free_packed.put(j)
                        t2 = time.perf_counter()
                        job["out"][i] = self.rs.collect_host(flat.reshape(R, n), rows, meas, [recs])
                except BaseException as exc:               # noqa: BLE001
                    self._fail(job, exc)
                self._add(download_stall=t1 - t0, to_host=t2 - t1, collect=time.perf_counter() - t2)
                job["left"] -= 1
                if job["left"] == 0:
                    job["done"].set()
        ctx.close()

    def run(self, measurements, firs, to_host=True, decay=None, align=False):
        """[(HRIR, gain dB)] in the order of `measurements` ([[frames of file 0, ...], ...]); firs, to_host, decay and align
        as SliceRunner.run"""
        if not len(measurements):
            return []
        import threading
        job = dict(measurements=measurements, firs=firs, to_host=to_host, decay=decay, align=align, out=[None] * len(measurements),
                   left=len(measurements), error=None, done=threading.Event())
        with self._submit:                                 # jobs of concurrent callers enter both stage queues in one order
            for q in self.jobs:
                q.put(job)
        job["done"].wait()
        if job["error"] is not None:
            raise job["error"]
        return job["out"]

    def close(self):
        if not self.threads:
            return
        for q in self.jobs:
            q.put(None)
        self.threads[0].join()
        self.threads[1].join()
        self.packed.put(None)
        self.threads[2].join()
        self.threads = []
        self.pool.close()


def _firs_for(firs, layout, ctx):
    """a job's FIRs for a slice on ctx's device: FIRs a design left on ANOTHER device come as host taps (one readback per job)"""
    if isinstance(firs, _native.DeviceFirs):
        return firs if firs.ctx.device == ctx.device else firs.host()
    if isinstance(firs, dict):
        rows = [firs[t] for t in layout.tasks]
        if any(isinstance(r, _native.DeviceFir) and r.batch.ctx.device != ctx.device for r in rows):
            return {t: np.asarray(r) for t, r in zip(layout.tasks, rows)}
    return firs


class SliceFleet:
    """Jobs of many measurements over SEVERAL devices from one process: a SlicePipeline per entry of IMPULSE_HIP_DEVICES (its
    own three streams, its own link), the measurements of a job cut into contiguous blocks - one per device, as
    sharding.shard_channels cuts channels - and run concurrently; results in job order.  With one device this is one
    SlicePipeline.  The deconvolution spectrum is prepared once (first device) and peer-copied by the estimator's plan
    cache; FIRs a design left on the first device reach the others as host taps, once per job."""

    def __init__(self, estimator, layout, devices=None, **kw):
        self.devices = list(_native.device_list() if devices is None else devices)
        self.pipes = []
        try:
            for d in self.devices:
                self.pipes.append(SlicePipeline(estimator, layout, device=d, **kw))
        except BaseException:
            self.close()
            raise

    def run(self, measurements, firs, to_host=True, decay=None, align=False):
        from concurrent.futures import ThreadPoolExecutor
        from .sharding import shard_channels
        n = len(measurements)
        blocks = [(k,) + shard_channels(n, len(self.pipes), k, keep_pairs=False) for k in range(len(self.pipes))]
        blocks = [(k, lo, hi) for k, lo, hi in blocks if hi > lo]
        if len(blocks) <= 1:
            return self.pipes[0].run(measurements, firs, to_host=to_host, decay=decay, align=align) if n else []

        def part(item):
            k, lo, hi = item
            return self.pipes[k].run(measurements[lo:hi], firs, to_host=to_host, decay=decay, align=align)

        with ThreadPoolExecutor(max_workers=len(blocks), thread_name_prefix="impulse-fleet") as pool:
            parts = list(pool.map(part, blocks))
        return [r for p in parts for r in p]

    def times(self, reset=True):
        out = {}
        for p in self.pipes:
            for k, v in p.times(reset).items():
                out[k] = out.get(k, 0.0) + v
        return out

    def close(self):
        for p in self.pipes:
            p.close()
        self.pipes = []


class _Skip(Exception):
    """a measurement of a job that has already failed: passed through the stages untouched"""


def run_slice_jobs(estimator, layout, measurements, firs, workers=None, head_ms=1, peak_target=-0.1, decay=None, align=False):
    """one job through a runner made for it (responses on the host): a three-stage SlicePipeline per device of
    IMPULSE_HIP_DEVICES (SliceFleet), or with `workers` that many SliceRunner lanes; callers with several jobs keep a runner"""
    if workers is None:
        runner = SliceFleet(estimator, layout, head_ms=head_ms, peak_target=peak_target)
    else:
        runner = SliceRunner(estimator, layout, workers=workers, head_ms=head_ms, peak_target=peak_target)
    try:
        return runner.run(measurements, firs, to_host=True, decay=decay, align=align)
    finally:
        runner.close()


def _fir_taps(fs):
    """taps of minimum_phase_impulse_response(fs, f_res=5): next_fast_len(round(fs // 2 / (f_res / 2)))
    (autoeq/frequency_response.py:637-681 as process_equalization_worker calls it)"""
    from .hrir import next_fast_len
    return next_fast_len(int(round(fs // 2 / (5 / 2))))
