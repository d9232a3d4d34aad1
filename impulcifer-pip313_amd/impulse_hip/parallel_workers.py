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
