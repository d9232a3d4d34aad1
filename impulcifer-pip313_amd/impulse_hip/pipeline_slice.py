"""The hot-path slice of the reference's BRIR pipeline, as one function.

Counterpart of the stages `_stage_open_measurements` (core/pipeline.py:565-573),
`_stage_crop_and_align` (:585-601, cropping part), `_stage_equalize` (:647-692), `_stage_decay`
(:694-716) and `_stage_normalize` (:725-735).  Orchestration, file discovery, plotting and report
writing of the full pipeline are out of scope; this harness exists so the stage sequence can be
parity-checked end to end and benchmarked.
"""
import numpy as np

import threading
from concurrent.futures import ThreadPoolExecutor

from . import _native
from .constants import IPSILATERAL_PAIRS
from .frequency_response import FrequencyResponse
from .hrir import HRIR
from .decay import adjust_decay_rows
from .parallel_workers import process_decay_worker, process_equalization_batch

_designer = None
_designer_lock = threading.Lock()


def _design_early(tasks, *args):
    """process_equalization_batch(tasks, ...) on a worker thread and a context of its own (a second stream).  The FIRs
    depend on the room / headphone / user curves and the target only - not on the recording - so their design (K12 + K6,
    0.46 ms for a 7.1 layout) runs while the recording crosses PCIe and is deconvolved.  Returns a Future."""
    global _designer
    with _designer_lock:
        if _designer is None:
            _designer = ThreadPoolExecutor(max_workers=1, thread_name_prefix="impulse-eq")

    def work():
        with _native.using_context(_native.aux_context()):
            # the FIRs stay on the device: equalize_channels takes them there (they reach the host only if read)
            return process_equalization_batch(tasks, *args, on_device=True)

    return _designer.submit(work)


def _expected_tasks(recordings):
    """(speaker, side) of every response the recordings will bring, in the order HRIR.irs lists them"""
    seen, out = set(), []
    for rec in recordings:
        side = rec[2] if len(rec) > 2 else None
        for sp in rec[1]:
            for sd in (("left", "right") if side is None else (side,)):
                if (sp, sd) not in seen:
                    seen.add((sp, sd))
                    out.append((sp, sd))
    return out


def run_slice(estimator, recordings, room_frs=None, target=None, head_ms=1, decay=None, peak_target=-0.1,
              hp_left=None, hp_right=None, eq_left=None, eq_right=None, stages=None, firs=None, align=False):
    """recordings: list of (path_or_(fs, array), speakers[, side]) measurement files.
    Returns the HRIR after: ingest (batched GPU deconvolution) -> crop_heads -> crop_tails ->
    per-channel minimum-phase FIR (batched GPU design) + equalize -> optional decay adjustment ->
    normalize.  ``stages`` (dict) receives copies of the intermediate channel data when given.
    The FIR design is a function of the curves and the target alone, so it is started first, on a worker thread with a
    context (stream) of its own, and collected where the reference's stage order needs it: same FIRs, same stage
    results, the design's 0.46 ms hidden behind the upload of the recording.
    ``firs`` ({(speaker, side): taps}): FIRs already designed for the job (the curves are per job, not per measurement);
    no design runs then.
    ``align``: run the two alignments `_stage_crop_and_align` has between crop_heads and crop_tails (core/pipeline.py:593-597:
    align_ipsilateral_all over IPSILATERAL_PAIRS with 30 ms segments, align_onset_groups_peak_leftref); responses that are
    on the device stay there."""
    hrir = HRIR(estimator)
    fs = estimator.fs
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    if target is None:
        target = FrequencyResponse(name="target", frequency=common.copy(), raw=0)
    eq_args = (room_frs, hp_left, hp_right, eq_left, eq_right, target, common, fs)
    planned = _expected_tasks(recordings)
    design = _design_early(planned, *eq_args) if (planned and firs is None) else None
    for rec in recordings:
        src, speakers = rec[0], rec[1]
        side = rec[2] if len(rec) > 2 else None
        if isinstance(src, str):
            hrir.open_recording(src, speakers, side=side)
        elif np.asarray(src[1]).dtype in (np.int16, np.int32):
            hrir.open_recording_frames(src[0], src[1], speakers, side=side)     # PCM frames [n_frames, tracks]
        else:
            hrir.open_recording_data(src[0], src[1], speakers, side=side)

    def snap(name):
        if stages is not None:
            # peek(): a copy that leaves device-resident responses where they are
            stages[name] = {(sp, sd): ir.peek() for sp, pair in hrir.irs.items() for sd, ir in pair.items()}

    snap("ingest")
    hrir.crop_heads(head_ms=head_ms)
    snap("crop_heads")
    if align:
        hrir.align_ipsilateral_all(speaker_pairs=list(IPSILATERAL_PAIRS), segment_ms=30)
        hrir.align_onset_groups_peak_leftref()
        snap("align")
    hrir.crop_tails()
    snap("crop_tails")

    tasks = [(sp, sd) for sp, pair in hrir.irs.items() for sd in pair]
    if firs is None:
        firs = {(sp, sd): fir for sp, sd, fir in design.result()} if design is not None else {}
    if not set(tasks) <= set(firs):                          # (a recording that brought other channels than announced)
        firs = {(sp, sd): fir for sp, sd, fir in process_equalization_batch(tasks, *eq_args)}
    hrir.equalize_channels({t: firs[t] for t in tasks})
    snap("equalize")

    if decay is not None:
        # core/pipeline.py:702-716: the speakers named in `decay` (a number: every speaker)
        todo = [(sp, sd, ir, decay[sp] if isinstance(decay, dict) else decay) for sp, pair in hrir.irs.items()
                if not isinstance(decay, dict) or sp in decay for sd, ir in pair.items()]
        if todo and all(ir._data is None and ir._row is not None for _, _, ir, _ in todo):
            adjust_decay_rows([ir._row for _, _, ir, _ in todo], fs, [tg for _, _, _, tg in todo])      # where the rows are
        else:
            for sp, sd, ir, tg in todo:
                _, _, ir.data = process_decay_worker((sp, sd, ir.data, fs, tg))
        snap("decay")

    gain = hrir.normalize(peak_target=peak_target)
    snap("normalize")
    return hrir, gain


def run_measurement_dirs(estimator, dir_paths, room_frs=None, target=None, head_ms=1, decay=None, peak_target=-0.1,
                         hp_left=None, hp_right=None, eq_left=None, eq_right=None, align=True):
    """The same stage sequence for MANY measurement directories of one layout (a listener measured again, a room measured at
    several seats), each laid out as `open_binaural_measurements` reads it (core/pipeline_stages.py:504-522: `<speaker
    list>.wav` files): the equalisation FIRs are designed once (the curves belong to the job), the recordings are read,
    uploaded, run through the device-resident sequence (imp_slice) and brought back as overlapping stages - one pipeline
    per device of IMPULSE_HIP_DEVICES.  Returns [(HRIR, gain dB)] in the order of dir_paths; every result is what
    run_slice gives for that directory with the same arguments (align defaults to True here: the reference's flow runs
    the two alignments between crop_heads and crop_tails)."""
    from .resident_slice import WavMeasurements, run_slice_jobs
    job, speakers = WavMeasurements.from_dirs(dir_paths, fs=estimator.fs)
    layout = job.layout(estimator, speakers)
    fs = estimator.fs
    common = FrequencyResponse.generate_frequencies(f_min=10, f_max=fs / 2, f_step=1.01)
    if target is None:
        target = FrequencyResponse(name="target", frequency=common.copy(), raw=0)
    firs = {(sp, sd): fir for sp, sd, fir in process_equalization_batch(layout.tasks, room_frs, hp_left, hp_right, eq_left, eq_right,
                                                                        target, common, fs, on_device=True)}
    return run_slice_jobs(estimator, layout, job, firs, head_ms=head_ms, peak_target=peak_target, decay=decay, align=align)
