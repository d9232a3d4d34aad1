"""impulse_hip - MI355X-native impulse-response engine behind Impulcifer's DSP class surfaces.

Host side of libimpulse_hip.so (hand-written HIP for gfx950, C ABI in include/impulse_hip.h).
"""
from ._native import (Context, ConvPlan, NativeError, NativeUnavailable, default_context,  # noqa: F401
                      load_library, library_path)

__version__ = "0.1.0"
