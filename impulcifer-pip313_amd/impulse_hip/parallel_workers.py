"""Per-channel worker functions (surface of reference core/parallel_workers.py:9-39): tuples of
plain arrays in, tuples out, so they remain picklable.  The array work runs on the device."""


def process_plot_worker(args):
    """(speaker, side, ir_data, test_signal, fs) -> (speaker, side, convolve(test_signal, ir_data, 'full'))."""
    speaker, side, ir_data, test_signal, fs = args
    from .impulse_response import fir_convolve_full
    return (speaker, side, fir_convolve_full(test_signal, ir_data))


def process_decay_worker(args):
    """(speaker, side, ir_data, fs, target) -> (speaker, side, decay-adjusted copy of ir_data)."""
    speaker, side, ir_data, fs, target = args
    from .decay import apply_decay_window, decay_adjustment_params
    adjusted = ir_data.copy()
    apply_decay_window(adjusted, decay_adjustment_params(adjusted, fs, target))
    return (speaker, side, adjusted)


_EQUALIZATION_CONTEXT = None


def init_equalization_worker(room_frs, hp_left, hp_right, eq_left, eq_right, target, common_freq, estimator_fs):
    """Install the shared equalisation inputs once per worker (reference :45-66)."""
    global _EQUALIZATION_CONTEXT
    _EQUALIZATION_CONTEXT = (room_frs, hp_left, hp_right, eq_left, eq_right, target, common_freq, estimator_fs)


def equalization_errors(tasks, room_frs, hp_left, hp_right, eq_left, eq_right, target, common_freq):
    """[B, n] error matrix of the channels in ``tasks``: room + headphone + user EQ - target on the common grid
    (reference :98-122), one row per (speaker, side)."""
    import numpy as np
    from .frequency_response import FrequencyResponse
    n = len(common_freq)
    err = np.zeros((len(tasks), n))
    for row, (speaker, side) in enumerate(tasks):
        if room_frs is not None and speaker in room_frs and side in room_frs[speaker]:
            err[row] += room_frs[speaker][side].error
        hp = hp_left if side == 'left' else hp_right
        if hp is not None:
            err[row] += hp.error
        eq = eq_left if side == 'left' else eq_right
        if eq is not None and isinstance(eq, FrequencyResponse):
            err[row] += eq.error
    err -= target.raw
    return err


def equalization_curve(speaker, side, room_frs, hp_left, hp_right, eq_left, eq_right, target, common_freq,
                       estimator_fs):
    """FrequencyResponse whose ``equalization`` is the curve the FIR must realise for one speaker-ear
    channel: error = room + headphone + user EQ - target, smoothed heavy/light, inverted with a
    40 dB gain limit (6 dB above 10 kHz) - reference :98-126."""
    from .frequency_response import FrequencyResponse
    fr = FrequencyResponse(name=f'{speaker}-{side} eq', frequency=common_freq.copy(), raw=0,
                           error=equalization_errors([(speaker, side)], room_frs, hp_left, hp_right, eq_left, eq_right,
                                                     target, common_freq)[0])
    fr.smoothen_heavy_light()
    fr.equalize(max_gain=40, treble_f_lower=10000, treble_f_upper=estimator_fs / 2)
    return fr


def process_equalization_worker(args):
    """(speaker, side[, context...]) -> (speaker, side, minimum-phase FIR): curve conditioning and FIR design both on
    the GPU, one launch chain."""
    if len(args) == 2:
        if _EQUALIZATION_CONTEXT is None:
            raise RuntimeError("Equalization worker context was not initialized.")
        speaker, side = args
        ctx = _EQUALIZATION_CONTEXT
    else:
        speaker, side, ctx = args[0], args[1], tuple(args[2:])
    return process_equalization_batch([(speaker, side)], *ctx)[0]


def process_equalization_batch(tasks, room_frs, hp_left, hp_right, eq_left, eq_right, target, common_freq,
                               estimator_fs, on_device=False):
    """All (speaker, side) FIRs of a measurement in ONE device launch chain - what replaces the reference's process
    pool over channels (core/pipeline.py:668-688): the error matrix goes up, the FIRs come down; smoothing, gain-limited
    inversion, FIR design grid and the minimum-phase design never leave the device.  on_device: neither do the FIRs - the
    third element of every result is then a _native.DeviceFir (row of one device batch) that HRIR.equalize_channels and
    the resident slice consume where it is; np.asarray(fir) brings it to the host."""
    from .frequency_response import equalization_firs
    tasks = list(tasks)
    if not tasks:
        return []
    errors = equalization_errors(tasks, room_frs, hp_left, hp_right, eq_left, eq_right, target, common_freq)
    _, firs = equalization_firs(common_freq, errors, estimator_fs, smoothen_first=True, max_gain=40,
                                treble_f_lower=10000, treble_f_upper=estimator_fs / 2, f_res=5, normalize=False,
                                on_device=on_device)
    if on_device:
        firs = firs.rows()
    return [(sp, sd, fir) for (sp, sd), fir in zip(tasks, firs)]
