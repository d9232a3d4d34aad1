"""ctypes binding of libimpulse_hip.so (C ABI: include/impulse_hip.h).

The HIP library is the product path: there is NO CPU fallback.  If the shared object cannot be
loaded, or no gfx950 device is present, every compute entry point raises ``NativeUnavailable``.
"""
import atexit
import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.normpath(os.path.join(_HERE, "..", "csrc", "libimpulse_hip.so"))

IMP_MODE_SAME = 0
IMP_MODE_FULL = 1
IMP_ERR_UNSUPPORTED = -3


class NativeUnavailable(RuntimeError):
    """libimpulse_hip.so is missing or no MI355X (gfx950) device is usable."""


class NativeError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"impulse_hip error {code}: {message}")
        self.code = code


class WindowParams(C.Structure):
    _fields_ = [
        ("gain", C.c_float),
        ("fade_in", C.c_int64),
        ("fade_out", C.c_int64),
        ("decay_start", C.c_int64),
        ("decay_half", C.c_int64),
        ("decay_knee", C.c_int64),
        ("decay_level_db", C.c_float),
    ]


class SliceGeometry(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_int64),
        ("elem_stride", C.c_int64),
        ("bits", C.c_int),
        ("pair_offset", C.POINTER(C.c_int64)),
        ("delay", C.POINTER(C.c_int64)),
        ("head", C.c_int64),
        ("fade_out", C.c_int64),
        ("taps", C.c_int64),
        ("keep_cap", C.c_int64),
        ("fs", C.c_double),
        ("peak_height", C.c_double),
        ("peak_target_db", C.c_double),
        ("gain_guard_rel", C.c_double),
    ]


class SliceRowResult(C.Structure):
    _fields_ = [("peak", C.c_int64), ("cut", C.c_int64), ("len", C.c_int64), ("knee", C.c_int64),
                ("knee_flags", C.c_int32), ("knee_why", C.c_int32), ("decay_peak", C.c_int64), ("decay_knee", C.c_int64),
                ("decay_slope", C.c_double), ("decay_level_db", C.c_double), ("decay_state", C.c_int32), ("decay_flags", C.c_int32),
                ("shift_ipsilateral", C.c_int64), ("shift_onset", C.c_int64)]


class SliceResult(C.Structure):
    _fields_ = [("keep", C.c_int64), ("out_len", C.c_int64), ("peak_db", C.c_double * 2), ("gain_db", C.c_double),
                ("gain", C.c_float), ("flags", C.c_int32)]


(SLICE_KNEE_GUARD, SLICE_KNEE_RANGE, SLICE_KEEP_CAP, SLICE_FADE, SLICE_GAIN_GUARD, SLICE_GAIN_NONFINITE, SLICE_SHORT, SLICE_DECAY_GUARD,
 SLICE_ALIGN_GUARD) = (1, 2, 4, 8, 16, 32, 64, 128, 256)
SLICE_REDO = (SLICE_KNEE_GUARD | SLICE_KNEE_RANGE | SLICE_KEEP_CAP | SLICE_FADE | SLICE_GAIN_GUARD | SLICE_GAIN_NONFINITE
              | SLICE_DECAY_GUARD | SLICE_ALIGN_GUARD)

_vp = C.c_void_p
_i64 = C.c_int64
_pf = C.POINTER(C.c_float)
_pd = C.POINTER(C.c_double)
_pi64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); mirrors include/impulse_hip.h one to one
SIGNATURES = {
    "imp_version": (C.c_char_p, []),
    "imp_last_error": (C.c_char_p, []),
    "imp_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "imp_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "imp_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "imp_ctx_synchronize": (C.c_int, [_vp]),
    "imp_ctx_destroy": (None, [_vp]),
    "imp_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "imp_free": (C.c_int, [_vp, _vp]),
    "imp_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "imp_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "imp_memcpy_d2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "imp_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    "imp_conv_plan_create": (C.c_int, [_vp, _pd, _i64, _i64, _i64, _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "imp_conv_plan_create_empty": (C.c_int, [_vp, _i64, _i64, _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "imp_conv_plan_create_paired": (C.c_int, [_vp, _pd, _i64, _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "imp_conv_plan_create_empty_paired": (C.c_int, [_vp, _i64, _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "imp_plan_is_paired": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "imp_conv_plan_create_ex": (C.c_int, [_vp, _pd, _i64, _i64, _i64, _i64, C.c_int, _i64, C.c_int, C.POINTER(_vp)]),
    "imp_plan_kind": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "imp_conv_execute_device_pairs": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _i64, _vp, _i64]),
    "imp_plan_destroy": (None, [_vp]),
    "imp_plan_info": (C.c_int, [_vp, _pi64, _pi64, _pi64, _pi64]),
    "imp_plan_spectrum": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "imp_plan_copy_spectrum": (C.c_int, [_vp, _vp]),
    "imp_memcpy_peer": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t]),
    "imp_conv_execute": (C.c_int, [_vp, _pf, _i64, _i64, _pf, _i64]),
    "imp_conv_execute_interleaved": (C.c_int, [_vp, _pf, _i64, _pf, _i64]),
    "imp_conv_execute_device": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64]),
    "imp_conv_execute_device_pcm": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _vp, _i64]),
    "imp_plan_set_overlap": (C.c_int, [_vp, C.c_int]),
    "imp_chain_create": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, C.c_double, C.POINTER(_vp)]),
    "imp_chain_execute_device": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "imp_chain_destroy": (None, [_vp]),
    "imp_slice_create": (C.c_int, [_vp, C.POINTER(SliceGeometry), _i64, C.POINTER(_vp)]),
    "imp_slice_destroy": (None, [_vp]),
    "imp_slice_info": (C.c_int, [_vp, _pi64, _pi64, _pi64, _pi64]),
    "imp_slice_set_firs": (C.c_int, [_vp, _pd, _i64]),
    "imp_slice_execute_device": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64]),
    "imp_slice_results": (C.c_int, [_vp, C.POINTER(SliceRowResult), C.POINTER(SliceResult)]),
    "imp_slice_set_decay": (C.c_int, [_vp, _pd]),
    "imp_slice_set_alignment": (C.c_int, [_vp, _i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _i64]),
    "imp_xcorr_argmax_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _pi64, _pi64, _i64, _pi64, _pd]),
    "imp_shift_rows_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _pi64, _i64, _vp, _pi64]),
    "imp_slice_pack_f64": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64]),
    "imp_host_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "imp_host_free": (C.c_int, [_vp]),
    "imp_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "imp_comm_create": (C.c_int, [_vp, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(_vp)]),
    "imp_comm_destroy": (None, [_vp]),
    "imp_comm_probe": (C.c_int, []),
    "imp_comm_nranks": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "imp_comm_broadcast": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    "imp_plan_broadcast_spectrum": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(C.c_size_t)]),
    "imp_plan_set_filters": (C.c_int, [_vp, _pd, _i64]),
    "imp_plan_set_filters_device": (C.c_int, [_vp, _vp, _i64]),
    "imp_curves_equalization_fir_device": (C.c_int, [_vp, _pd, _i64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                                     C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, _pd, C.POINTER(_vp), _pi64]),
    "imp_slice_set_firs_device": (C.c_int, [_vp, _vp, _i64]),
    "imp_plan_set_timing": (C.c_int, [_vp, C.c_int]),
    "imp_plan_get_timing": (C.c_int, [_vp, _pd, _pi64, C.c_int]),
    "imp_debug_plan_geometry": (C.c_int, [_i64, _i64, C.c_int, _pi64, _pi64, _pi64]),
    "imp_debug_plan_geometry_fused": (C.c_int, [_i64, _i64, C.c_int, _pi64, _pi64, _pi64, _pi64]),
    "imp_debug_plan_geometry_paired": (C.c_int, [_i64, _i64, C.c_int, _pi64, _pi64, _pi64, _pi64]),
    "imp_debug_host_spectrum": (C.c_int, [_pd, _i64, C.c_int, _pf]),
    "imp_plan_debug_run_stage": (C.c_int, [_vp, _pf, _i64, _i64, C.c_int, _pf]),
    "imp_segset_create": (C.c_int, [_vp, _pd, _pi64, _pi64, _i64, C.POINTER(_vp), _pd]),
    "imp_segset_range_means": (C.c_int, [_vp, _pi64, _pi64, _pi64, _i64, _pd]),
    "imp_segset_destroy": (None, [_vp]),
    "imp_decay_times": (C.c_int, [_vp, _pd, _pi64, _pi64, _i64, _pi64, _pi64, _pd, _pi64, C.c_double, _pd]),
    "imp_decay_times_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _i64, _pi64, _pi64, _pd, _pi64, C.c_double, _pd]),
    "imp_sosfilt": (C.c_int, [_vp, _pd, _i64, _pd, _pi64, _pi64, _i64, _pd]),
    "imp_xcorr_argmax": (C.c_int, [_vp, _pd, _pi64, _pi64, _pd, _pi64, _pi64, _i64, _pi64, _pd]),
    "imp_minphase_fir": (C.c_int, [_vp, _pd, _i64, _i64, C.c_double, _pd]),
    "imp_curves_create": (C.c_int, [_vp, _pd, _i64, C.POINTER(_vp)]),
    "imp_curves_destroy": (None, [_vp]),
    "imp_curves_window_size": (C.c_int, [_vp, C.c_double, C.POINTER(C.c_int)]),
    "imp_curves_smooth": (C.c_int, [_vp, _pd, _i64, C.c_double, C.c_double, C.c_double, C.c_double, _pd]),
    "imp_curves_equalization": (C.c_int, [_vp, _pd, _i64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int, _pd, _pd, C.POINTER(C.c_int)]),
    "imp_curves_fir_taps": (C.c_int, [_vp, C.c_double, C.c_double, _pi64]),
    "imp_curves_fir": (C.c_int, [_vp, _pd, _i64, C.c_double, C.c_double, C.c_int, _pd, _pd]),
    "imp_curves_equalization_fir": (C.c_int, [_vp, _pd, _i64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                              C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, _pd, _pd]),
    "imp_magnitude_db": (C.c_int, [_vp, _pd, _i64, _i64, _pd]),
    "imp_debug_minphase_stage": (C.c_int, [_vp, _pd, _i64, _i64, C.c_double, C.c_int, _pd]),
    "imp_debug_fft64": (C.c_int, [_vp, _pd, _i64, _i64, C.c_int, _pd, C.POINTER(C.c_int)]),
    "imp_peak_index": (C.c_int, [_vp, _pf, _pi64, _pi64, _i64, C.c_double, _pi64, _pf]),
    "imp_peak_index_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _i64, C.c_double, _pi64, _pf]),
    "imp_apply_window": (C.c_int, [_vp, _pf, _pi64, _pi64, _i64, C.POINTER(WindowParams)]),
    "imp_apply_window_device": (C.c_int, [_vp, _vp, _pi64, _vp, _pi64, _pi64, _i64, C.POINTER(WindowParams)]),
    "imp_segset_create_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _i64, C.POINTER(_vp), _pd]),
    "imp_decay_knees_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _i64, C.c_double, C.c_double, _pi64, _pi64, _pd, _pi64,
                                         C.POINTER(C.c_int32)]),
    "imp_rows_to_pcm_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _i64, _pi64, _i64, _i64, C.c_int, _vp]),
    "imp_magnitude_db_sum_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _pi64, _i64, _i64, _i64, _pd]),
    "imp_magnitude_db_sum_peak_device": (C.c_int, [_vp, _vp, _pi64, _pi64, _pi64, _i64, _i64, _i64, _pd]),
}

_lib = None
_lib_lock = threading.Lock()
_live_contexts = weakref.WeakSet()


@atexit.register
def _shutdown():
    """Release every context (and the plans made on it) while the interpreter and the HIP runtime
    are both still fully alive.  Objects kept alive past this point (e.g. by a traceback) would
    otherwise reach the HIP runtime's own exit-time teardown with live streams/buffers, which aborts
    the process (std::bad_variant_access inside libamdhip64 on ROCm 7.2)."""
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


def library_path():
    return os.environ.get("IMPULSE_HIP_LIB", _DEFAULT_LIB)


def load_library():
    """Load the shared object and declare every prototype. Raises NativeUnavailable if missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise NativeUnavailable(
                f"{path} not found - build it with `python impulcifer-pip313_amd/build.py` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        try:
            lib = C.CDLL(path)
        except OSError as exc:
            raise NativeUnavailable(f"cannot load {path}: {exc}") from exc
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _check(rc):
    if rc != 0:
        msg = load_library().imp_last_error()
        raise NativeError(rc, msg.decode("utf-8", "replace") if msg else "")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr_f(a):
    return a.ctypes.data_as(_pf)


def _ptr_i64(a):
    return a.ctypes.data_as(_pi64)


class Context:
    """One GPU + one stream (imp_ctx)."""

    def __init__(self, device=0):
        lib = load_library()
        n = C.c_int(0)
        rc = lib.imp_device_count(C.byref(n))
        if rc != 0 or n.value <= 0:
            raise NativeUnavailable("no HIP device visible: the impulse_hip product path needs an MI355X (gfx950)")
        h = _vp()
        rc = lib.imp_ctx_create(int(device), C.byref(h))
        if rc != 0:
            msg = lib.imp_last_error().decode("utf-8", "replace")
            raise NativeUnavailable(f"imp_ctx_create({device}) failed: {msg}")
        self._lib = lib
        self._h = h
        self.device = int(device)
        self._plans = weakref.WeakSet()       # plans hold a raw pointer to this context: they go first
        _live_contexts.add(self)

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr):
        _check(self._lib.imp_ctx_set_stream(self._h, _vp(int(stream_ptr))))

    def synchronize(self):
        _check(self._lib.imp_ctx_synchronize(self._h))

    def malloc(self, nbytes):
        p = _vp()
        _check(self._lib.imp_malloc(self._h, int(nbytes), C.byref(p)))
        return p.value or 0

    def free(self, dptr):
        _check(self._lib.imp_free(self._h, _vp(int(dptr))))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(self._lib.imp_memcpy_h2d(self._h, _vp(int(dptr)), arr.ctypes.data_as(_vp), arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        _check(self._lib.imp_memcpy_d2h(self._h, arr.ctypes.data_as(_vp), _vp(int(dptr)), arr.nbytes))

    def d2d(self, dst, src, nbytes):
        _check(self._lib.imp_memcpy_d2d(self._h, _vp(int(dst)), _vp(int(src)), int(nbytes)))

    def copy_from(self, dst, src_ctx, src, nbytes):
        """nbytes from `src` on src_ctx's device to `dst` on this context's device (peer copy; waits for src_ctx's stream,
        asynchronous on this context's)"""
        _check(self._lib.imp_memcpy_peer(self._h, _vp(int(dst)), src_ctx.handle, _vp(int(src)), int(nbytes)))

    def memset(self, dptr, value, nbytes):
        _check(self._lib.imp_memset(self._h, _vp(int(dptr)), int(value), int(nbytes)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # chains (and other composites) hold raw pointers to plans: they go before the plans they are made of
            members = list(getattr(self, "_plans", ()))
            for plan in sorted(members, key=lambda m: 0 if getattr(m, "_composite", False) else 1):
                plan.close()
            self._lib.imp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ragged helpers ------------------------------------------------------------------
    @staticmethod
    def _pack_rows(rows):
        lens = np.array([len(r) for r in rows], dtype=np.int64)
        offs = np.zeros(len(rows), dtype=np.int64)
        if len(rows):
            offs[1:] = np.cumsum(lens)[:-1]
        flat = np.zeros(int(lens.sum()) if len(rows) else 0, dtype=np.float32)
        for r, o, n in zip(rows, offs, lens):
            flat[o:o + n] = r
        return flat, offs, lens

    def peak_index(self, rows, peak_height=0.12589):
        """Batched ImpulseResponse.peak_index over a list of 1-D arrays. Returns (idx[int64], maxabs[f32])."""
        rows = [np.asarray(r) for r in rows]
        B = len(rows)
        idx = np.zeros(B, dtype=np.int64)
        mx = np.zeros(B, dtype=np.float32)
        if B == 0:
            return idx, mx
        flat, offs, lens = self._pack_rows(rows)
        if flat.size == 0:
            return idx, mx
        _check(self._lib.imp_peak_index(self._h, _ptr_f(flat), _ptr_i64(offs), _ptr_i64(lens), B,
                                        float(peak_height), _ptr_i64(idx), _ptr_f(mx)))
        return idx, mx

    # ---- device-resident rows (fp32 at dptr + off[b], len[b] samples) ---------------------------
    @staticmethod
    def _window_array(params, B):
        arr = (WindowParams * B)()
        for i, p in enumerate(params):
            arr[i] = WindowParams(float(p.get("gain", 1.0)), int(p.get("fade_in", 0)), int(p.get("fade_out", 0)),
                                  int(p.get("decay_start", 0)), int(p.get("decay_half", -1)),
                                  int(p.get("decay_knee", 0)), float(p.get("decay_level_db", 0.0)))
        return arr

    def peak_index_device(self, dptr, offs, lens, peak_height=0.12589):
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        B = len(offs)
        idx = np.zeros(B, dtype=np.int64)
        mx = np.zeros(B, dtype=np.float32)
        if B:
            _check(self._lib.imp_peak_index_device(self._h, _vp(int(dptr)), _ptr_i64(offs), _ptr_i64(lens), B,
                                                   float(peak_height), _ptr_i64(idx), _ptr_f(mx)))
        return idx, mx

    def decay_knees_device(self, dptr, offs, lens, fs, peak_height=0.12589):
        """Peak + Lundeby knee search of device rows in one stream-ordered sequence: (peak, knee, floor dB, window,
        flags); flags[b] != 0 = the device left row b to the host search (include/impulse_hip.h)."""
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        B = len(offs)
        peak, knee, win = (np.zeros(B, dtype=np.int64) for _ in range(3))
        floor = np.zeros(B, dtype=np.float64)
        flags = np.zeros(B, dtype=np.int32)
        if B:
            _check(self._lib.imp_decay_knees_device(self._h, _vp(int(dptr)), _ptr_i64(offs), _ptr_i64(lens), B, float(fs),
                                                    float(peak_height), _ptr_i64(peak), _ptr_i64(knee), floor.ctypes.data_as(_pd),
                                                    _ptr_i64(win), flags.ctypes.data_as(C.POINTER(C.c_int32))))
        return peak, knee, floor, win, flags

    def apply_window_device(self, d_src, src_off, d_dst, dst_off, lens, params):
        src_off = np.ascontiguousarray(src_off, dtype=np.int64)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        B = len(lens)
        if B:
            _check(self._lib.imp_apply_window_device(self._h, _vp(int(d_src)), _ptr_i64(src_off), _vp(int(d_dst)),
                                                     _ptr_i64(dst_off), _ptr_i64(lens), B, self._window_array(params, B)))

    def rows_to_pcm_device(self, dptr, offs, lens, row_of_track, n_frames, bits):
        """Interleaved PCM frames [n_frames, n_tracks] (int16 / int32) of device rows: track t = row row_of_track[t]
        or silence (-1); libsndfile's float -> PCM conversion."""
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        rot = np.ascontiguousarray(row_of_track, dtype=np.int64)
        out = np.empty((int(n_frames), len(rot)), dtype=np.int16 if bits == 16 else np.int32)
        _check(self._lib.imp_rows_to_pcm_device(self._h, _vp(int(dptr)), _ptr_i64(offs), _ptr_i64(lens), len(offs),
                                                _ptr_i64(rot), len(rot), int(n_frames), int(bits),
                                                out.ctypes.data_as(_vp)))
        return out

    def magnitude_db_sum_device(self, dptr, offs, lens, groups, n_groups, n):
        """[n_groups, ceil(n/2)] dB spectra of the per-group sums of device rows (HRIR.normalize)."""
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        groups = np.ascontiguousarray(groups, dtype=np.int64)
        out = np.empty((int(n_groups), (int(n) + 1) // 2), dtype=np.float64)
        _check(self._lib.imp_magnitude_db_sum_device(self._h, _vp(int(dptr)), _ptr_i64(offs), _ptr_i64(lens),
                                                     _ptr_i64(groups), len(offs), int(n_groups), int(n),
                                                     out.ctypes.data_as(_pd)))
        return out

    def magnitude_db_sum_peak_device(self, dptr, offs, lens, groups, n_groups, n):
        """np.max of each of magnitude_db_sum_device's spectra, reduced on the device: [n_groups]."""
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        groups = np.ascontiguousarray(groups, dtype=np.int64)
        out = np.empty(int(n_groups), dtype=np.float64)
        _check(self._lib.imp_magnitude_db_sum_peak_device(self._h, _vp(int(dptr)), _ptr_i64(offs), _ptr_i64(lens),
                                                          _ptr_i64(groups), len(offs), int(n_groups), int(n),
                                                          out.ctypes.data_as(_pd)))
        return out

    def decay_times(self, rows, peaks, knees, noise_floors, windows, fs):
        """Batched core/decay.py decay_times on the device: [B, 4] = EDT, RT20, RT30, RT60 (NaN = undefined)."""
        rows = [np.ascontiguousarray(r, dtype=np.float64).ravel() for r in rows]
        B = len(rows)
        out = np.full((B, 4), np.nan)
        if B == 0:
            return out
        lens = np.array([len(r) for r in rows], dtype=np.int64)
        offs = np.zeros(B, dtype=np.int64)
        offs[1:] = np.cumsum(lens)[:-1]
        flat = np.concatenate(rows) if lens.sum() else np.zeros(1)
        pk = np.ascontiguousarray(peaks, dtype=np.int64)
        kn = np.ascontiguousarray(knees, dtype=np.int64)
        nf = np.ascontiguousarray(noise_floors, dtype=np.float64)
        ws = np.ascontiguousarray(windows, dtype=np.int64)
        _check(self._lib.imp_decay_times(self._h, flat.ctypes.data_as(_pd), _ptr_i64(offs), _ptr_i64(lens), B,
                                         _ptr_i64(pk), _ptr_i64(kn), nf.ctypes.data_as(_pd), _ptr_i64(ws),
                                         float(fs), out.ctypes.data_as(_pd)))
        return out

    def xcorr_argmax_device(self, dptr, a_off, a_len, b_off, b_len):
        """argmax of the full cross-correlation of segment pairs of fp32 device rows (K10): index into correlate(a, b, 'full')"""
        a_off, a_len, b_off, b_len = (np.ascontiguousarray(v, dtype=np.int64) for v in (a_off, a_len, b_off, b_len))
        B = len(a_off)
        arg = np.zeros(B, dtype=np.int64)
        val = np.zeros(B, dtype=np.float64)
        if B:
            _check(self._lib.imp_xcorr_argmax_device(self._h, _vp(int(dptr)), _ptr_i64(a_off), _ptr_i64(a_len), _ptr_i64(b_off), _ptr_i64(b_len),
                                                     B, _ptr_i64(arg), val.ctypes.data_as(_pd)))
        return arg, val

    def shift_rows_device(self, d_src, src_off, lens, shifts, d_dst, dst_off):
        """ImpulseResponse.shift of fp32 device rows into other rows (asynchronous)"""
        src_off, lens, shifts, dst_off = (np.ascontiguousarray(v, dtype=np.int64) for v in (src_off, lens, shifts, dst_off))
        if len(lens):
            _check(self._lib.imp_shift_rows_device(self._h, _vp(int(d_src)), _ptr_i64(src_off), _ptr_i64(lens), _ptr_i64(shifts), len(lens),
                                                   _vp(int(d_dst)), _ptr_i64(dst_off)))

    def decay_times_device(self, dptr, offs, lens, peaks, knees, noise_floors, windows, fs):
        """decay_times for fp32 rows that are on the device: [B, 4] = EDT, RT20, RT30, RT60 (NaN = undefined), the bits the
        rows' float64 copies give through decay_times()"""
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        B = len(offs)
        out = np.full((B, 4), np.nan)
        if B:
            pk = np.ascontiguousarray(peaks, dtype=np.int64)
            kn = np.ascontiguousarray(knees, dtype=np.int64)
            nf = np.ascontiguousarray(noise_floors, dtype=np.float64)
            ws = np.ascontiguousarray(windows, dtype=np.int64)
            _check(self._lib.imp_decay_times_device(self._h, _vp(int(dptr)), _ptr_i64(offs), _ptr_i64(lens), B, _ptr_i64(pk), _ptr_i64(kn),
                                                    nf.ctypes.data_as(_pd), _ptr_i64(ws), float(fs), out.ctypes.data_as(_pd)))
        return out

    def sosfilt(self, sos, rows):
        """scipy.signal.sosfilt(sos, row) for every row (fp64 on the device, bit-identical). Returns a list."""
        sos = np.ascontiguousarray(sos, dtype=np.float64).reshape(-1, 6)
        rows = [np.ascontiguousarray(r, dtype=np.float64).ravel() for r in rows]
        B = len(rows)
        if B == 0:
            return []
        lens = np.array([len(r) for r in rows], dtype=np.int64)
        offs = np.zeros(B, dtype=np.int64)
        offs[1:] = np.cumsum(lens)[:-1]
        flat = np.concatenate(rows) if lens.sum() else np.zeros(1)
        out = np.zeros_like(flat)
        _check(self._lib.imp_sosfilt(self._h, sos.ctypes.data_as(_pd), len(sos), flat.ctypes.data_as(_pd), _ptr_i64(offs),
                                     _ptr_i64(lens), B, out.ctypes.data_as(_pd)))
        return [out[o:o + n].copy() for o, n in zip(offs, lens)]

    def xcorr_argmax(self, a_rows, b_rows):
        """np.argmax(scipy.signal.correlate(a, b, "full")) for every pair (fp64 on the device).
        Returns (argmax[int64], peak value[f64])."""
        B = len(a_rows)
        if B != len(b_rows):
            raise ValueError("xcorr_argmax: the two lists differ in length")
        arg = np.zeros(B, dtype=np.int64)
        val = np.zeros(B, dtype=np.float64)
        if B == 0:
            return arg, val

        def pack(rows):
            rows = [np.ascontiguousarray(r, dtype=np.float64).ravel() for r in rows]
            lens = np.array([len(r) for r in rows], dtype=np.int64)
            offs = np.zeros(B, dtype=np.int64)
            offs[1:] = np.cumsum(lens)[:-1]
            return (np.concatenate(rows) if lens.sum() else np.zeros(1)), offs, lens

        fa, oa, la = pack(a_rows)
        fb, ob, lb = pack(b_rows)
        _check(self._lib.imp_xcorr_argmax(self._h, fa.ctypes.data_as(_pd), _ptr_i64(oa), _ptr_i64(la),
                                          fb.ctypes.data_as(_pd), _ptr_i64(ob), _ptr_i64(lb), B,
                                          _ptr_i64(arg), val.ctypes.data_as(_pd)))
        return arg, val

    def minphase_fir(self, gain, fs):
        """Batched firwin2 + homomorphic minimum_phase (fp64 on the device): gain [B, n] linear gains on
        linspace(0, fs//2, n) -> FIR taps [B, n]."""
        g = np.ascontiguousarray(gain, dtype=np.float64)
        one = g.ndim == 1
        if one:
            g = g[None, :]
        out = np.empty_like(g)
        _check(self._lib.imp_minphase_fir(self._h, g.ctypes.data_as(_pd), g.shape[0], g.shape[1], float(fs),
                                          out.ctypes.data_as(_pd)))
        return out[0] if one else out

    def magnitude_db(self, x):
        """20 log10 |rfft(x)| on the first ceil(n/2) bins, batched over rows: [B, n] -> [B, ceil(n/2)] float64."""
        a = np.ascontiguousarray(x, dtype=np.float64)
        one = a.ndim == 1
        if one:
            a = a[None, :]
        out = np.empty((a.shape[0], (a.shape[1] + 1) // 2), dtype=np.float64)
        if out.size:
            _check(self._lib.imp_magnitude_db(self._h, a.ctypes.data_as(_pd), a.shape[0], a.shape[1],
                                              out.ctypes.data_as(_pd)))
        return out[0] if one else out

    def fft64(self, x, inverse=False):
        """the library's batched fp64 complex transform on host data [B, N] (test hook): (result, used the LDS tile form)"""
        a = np.ascontiguousarray(x, dtype=np.complex128)
        one = a.ndim == 1
        if one:
            a = a[None, :]
        out = np.empty_like(a)
        tiles = C.c_int(0)
        _check(self._lib.imp_debug_fft64(self._h, a.ctypes.data_as(_pd), a.shape[0], a.shape[1], 1 if inverse else -1,
                                         out.ctypes.data_as(_pd), C.byref(tiles)))
        return (out[0] if one else out), bool(tiles.value)

    def minphase_debug_stage(self, gain, fs, stage):
        g = np.ascontiguousarray(gain, dtype=np.float64)
        if g.ndim == 1:
            g = g[None, :]
        out = np.empty((g.shape[0], 2 * g.shape[1]), dtype=np.float64)
        _check(self._lib.imp_debug_minphase_stage(self._h, g.ctypes.data_as(_pd), g.shape[0], g.shape[1], float(fs),
                                                  int(stage), out.ctypes.data_as(_pd)))
        return out

    def apply_window(self, rows, params):
        """In-place-style windowing of a list of rows; returns new float32 arrays."""
        rows = [np.asarray(r) for r in rows]
        B = len(rows)
        if B == 0:
            return []
        flat, offs, lens = self._pack_rows(rows)
        arr = (WindowParams * B)()
        for i, p in enumerate(params):
            arr[i] = WindowParams(float(p.get("gain", 1.0)), int(p.get("fade_in", 0)), int(p.get("fade_out", 0)),
                                  int(p.get("decay_start", 0)), int(p.get("decay_half", -1)),
                                  int(p.get("decay_knee", 0)), float(p.get("decay_level_db", 0.0)))
        _check(self._lib.imp_apply_window(self._h, _ptr_f(flat), _ptr_i64(offs), _ptr_i64(lens), B, arr))
        return [flat[o:o + n].copy() for o, n in zip(offs, lens)]


class SegSet:
    """K7: fp64 analysis segments kept on the device as e = (x / max|x|)^2; range_means() answers
    np.mean(e[seg][a:b]) queries with NumPy's summation order (bit-identical levels)."""

    def __init__(self, ctx, rows):
        self.ctx = ctx
        self._lib = ctx._lib
        rows = [np.ascontiguousarray(r, dtype=np.float64).ravel() for r in rows]
        self.lens = np.array([len(r) for r in rows], dtype=np.int64)
        B = len(rows)
        offs = np.zeros(max(B, 1), dtype=np.int64)[:B]
        if B:
            offs[1:] = np.cumsum(self.lens)[:-1]
        flat = np.concatenate(rows) if B and self.lens.sum() else np.zeros(1)
        self.maxabs = np.zeros(max(B, 1), dtype=np.float64)[:B]
        h = _vp()
        _check(self._lib.imp_segset_create(ctx.handle, flat.ctypes.data_as(_pd), _ptr_i64(offs), _ptr_i64(self.lens), B,
                                           C.byref(h), self.maxabs.ctypes.data_as(_pd)))
        self._h = h
        ctx._plans.add(self)

    @classmethod
    def from_device(cls, ctx, dptr, offs, lens, want_max=True):
        """Segments cut from fp32 device rows (dptr + offs[b], lens[b] samples), converted on the device.  want_max=False:
        the row maxima stay on the device and the call returns without waiting for it."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self._lib = ctx._lib
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        self.lens = np.ascontiguousarray(lens, dtype=np.int64)
        B = len(offs)
        self.maxabs = np.zeros(max(B, 1), dtype=np.float64)[:B]
        h = _vp()
        _check(self._lib.imp_segset_create_device(ctx.handle, _vp(int(dptr)), _ptr_i64(offs), _ptr_i64(self.lens), B,
                                                  C.byref(h), self.maxabs.ctypes.data_as(_pd) if want_max else None))
        if not want_max:
            self.maxabs = None
        self._h = h
        ctx._plans.add(self)
        return self

    def range_means(self, queries):
        """queries: iterable of (segment, a, b) with 0 <= a <= b <= len(segment).  Returns float64 means."""
        q = np.asarray(list(queries), dtype=np.int64).reshape(-1, 3)
        return self.range_means_arrays(q[:, 0], q[:, 1], q[:, 2])

    def range_means_arrays(self, seg, a, b):
        """np.mean(e[seg[i]][a[i]:b[i]]) for every i (three equally long integer arrays)."""
        seg, a, b = (np.ascontiguousarray(v, dtype=np.int64) for v in (seg, a, b))
        out = np.zeros(len(seg), dtype=np.float64)
        if len(seg):
            _check(self._lib.imp_segset_range_means(self._h, _ptr_i64(seg), _ptr_i64(a), _ptr_i64(b), len(seg),
                                                    out.ctypes.data_as(_pd)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                self._lib.imp_segset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class FirChain:
    """K1 -> K3 -> K4 -> K5 in stream order, no host round trip (imp_chain): recordings on the device in, equalised
    cropped responses on the device out."""
    _composite = True

    def __init__(self, deconv_plan, fir_plan, B, head, fade_in, fade_out, peak_height=0.12589):
        self.ctx = deconv_plan.ctx
        self._lib = self.ctx._lib
        self._plans = (deconv_plan, fir_plan)            # keep them alive
        h = _vp()
        _check(self._lib.imp_chain_create(deconv_plan.handle, fir_plan.handle, int(B), int(head), int(fade_in),
                                          int(fade_out), float(peak_height), C.byref(h)))
        self._h = h
        # the chain holds raw pointers into BOTH contexts: whichever closes first must close the chain before its plans
        self.ctx._plans.add(self)
        fir_plan.ctx._plans.add(self)

    def execute_device(self, d_x, chan_stride_in, d_out, chan_stride_out, d_peaks=0, elem_stride_in=1):
        if not getattr(self, "_h", None):
            raise NativeError(-1, "the chain is closed (one of its contexts was closed)")
        _check(self._lib.imp_chain_execute_device(self._h, _vp(int(d_x)), int(chan_stride_in), int(elem_stride_in),
                                                  _vp(int(d_out)), int(chan_stride_out), _vp(int(d_peaks)) if d_peaks else None))

    def close(self):
        if getattr(self, "_h", None):
            # imp_chain_destroy locks both contexts and drains both streams: both must still be alive
            if all(getattr(p.ctx, "_h", None) for p in self._plans):
                self._lib.imp_chain_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class Slice:
    """imp_slice: ingest -> crop_heads -> crop_tails -> equalize -> normalize for M measurements per call, device resident,
    no host readback between the stages (include/impulse_hip.h).  Rows of flagged measurements (results()[1].flags &
    SLICE_REDO) are not valid: the caller runs those through the staged path."""
    _composite = True

    def __init__(self, deconv_plan, pair_offsets, delays, elem_stride, bits, head, fade_out, taps, keep_cap, fs,
                 peak_target=-0.1, max_measurements=1, peak_height=0.12589, gain_guard_rel=0.0):
        self.ctx = deconv_plan.ctx
        self._lib = self.ctx._lib
        self._plan = deconv_plan                          # keep it alive
        po = np.ascontiguousarray(pair_offsets, dtype=np.int64)
        dl = np.ascontiguousarray(delays, dtype=np.int64)
        if po.shape != dl.shape or po.ndim != 1 or len(po) < 1:
            raise ValueError("pair_offsets and delays: one entry per ear pair")
        g = SliceGeometry(len(po), int(elem_stride), int(bits), _ptr_i64(po), _ptr_i64(dl), int(head), int(fade_out), int(taps),
                          int(keep_cap), float(fs), float(peak_height), float(peak_target), float(gain_guard_rel))
        h = _vp()
        _check(self._lib.imp_slice_create(deconv_plan.handle, C.byref(g), int(max_measurements), C.byref(h)))
        self._h = h
        self.ctx._plans.add(self)
        r, m, o, f = _i64(), _i64(), _i64(), _i64()
        _check(self._lib.imp_slice_info(h, C.byref(r), C.byref(m), C.byref(o), C.byref(f)))
        self.rows, self.max_measurements, self.out_len_max, self.norm_fft_len = r.value, m.value, o.value, f.value
        self.taps, self.keep_cap = int(taps), int(keep_cap)
        self.last_M = 0

    def set_firs(self, firs):
        f = np.ascontiguousarray(firs, dtype=np.float64)
        if f.shape != (self.rows, self.taps):
            raise ValueError(f"FIRs must be [{self.rows}, {self.taps}], got {f.shape}")
        _check(self._lib.imp_slice_set_firs(self._h, f.ctypes.data_as(_pd), self.taps))

    def set_firs_device(self, d_firs, ld=None):
        """FIRs already on the device (fp64 [rows][ld]): no upload, no wait"""
        _check(self._lib.imp_slice_set_firs_device(self._h, _vp(int(d_firs)), int(ld or self.taps)))

    def execute_device(self, d_rec, rec_stride, M, d_out, out_pitch):
        _check(self._lib.imp_slice_execute_device(self._h, _vp(int(d_rec)), int(rec_stride), int(M), _vp(int(d_out)),
                                                  int(out_pitch)))
        self.last_M = int(M)

    def results(self):
        """(rows, measurements) of the last call as structured NumPy arrays; waits for the call."""
        M = self.last_M
        rows = (SliceRowResult * max(M * self.rows, 1))()
        meas = (SliceResult * max(M, 1))()
        _check(self._lib.imp_slice_results(self._h, rows, meas))
        r = np.ctypeslib.as_array(rows)[:M * self.rows].copy() if M else np.zeros(0)
        m = np.ctypeslib.as_array(meas)[:M].copy() if M else np.zeros(0)
        return r, m

    def set_alignment(self, ipsi_pairs, leader_of_pair, ref_pair, segment):
        """ipsi_pairs: [(first ear pair, second ear pair)]; leader_of_pair: per ear pair the pair whose left ear leads its
        onset group (-1: no onset shift); ref_pair: FL's ear pair; segment: samples of the lag search.  None switches the
        stage off."""
        p32 = C.POINTER(C.c_int32)
        if ipsi_pairs is None:
            _check(self._lib.imp_slice_set_alignment(self._h, 0, None, None, None, 0, 0))
            return
        a = np.ascontiguousarray([p[0] for p in ipsi_pairs], dtype=np.int32)
        b = np.ascontiguousarray([p[1] for p in ipsi_pairs], dtype=np.int32)
        ld = np.ascontiguousarray(leader_of_pair, dtype=np.int32)
        if ld.shape != (self.rows // 2,):
            raise ValueError(f"one onset leader per ear pair ({self.rows // 2}), got {ld.shape}")
        _check(self._lib.imp_slice_set_alignment(self._h, len(a), a.ctypes.data_as(p32), b.ctypes.data_as(p32), ld.ctypes.data_as(p32),
                                                 int(ref_pair), int(segment)))

    def set_decay(self, targets):
        """target RT60 in seconds per row of a measurement (NaN: leave the row alone); None switches the stage off"""
        if targets is None:
            _check(self._lib.imp_slice_set_decay(self._h, None))
            return
        t = np.ascontiguousarray(targets, dtype=np.float64)
        if t.shape != (self.rows,):
            raise ValueError(f"one decay target per row of a measurement ({self.rows}), got {t.shape}")
        _check(self._lib.imp_slice_set_decay(self._h, t.ctypes.data_as(_pd)))

    def pack_f64(self, d_out, out_pitch, M, d_packed, meas_stride):
        """the last call's rows as float64, every measurement packed as a [rows][out_len] array (asynchronous)"""
        _check(self._lib.imp_slice_pack_f64(self._h, _vp(int(d_out)), int(out_pitch), int(M), _vp(int(d_packed)), int(meas_stride)))

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                self._lib.imp_slice_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class PinnedBlock:
    """float64 page-locked host memory from a PinnedPool.  np.asarray(block) is a view whose base is the block: when the last
    view of it is gone the memory goes back to the pool (mapped and pinned as it is) for the next job."""

    def __init__(self, pool, ptr, doubles):
        self._pool, self.ptr, self.doubles = pool, int(ptr), int(doubles)
        self.__array_interface__ = {"shape": (self.doubles,), "typestr": "<f8", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            self._pool._give(self.ptr, self.doubles)
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class PinnedPool:
    """Page-locked float64 result blocks, recycled: a job's responses are written by the link straight into them
    (imp_memcpy_d2h, no staging copy, no first-touch page faults), and come back here when the caller has dropped the
    arrays.  Pinning is the expensive part (about 3 ms per 8 MB block, measured; a first-touch of ordinary memory is 0.6 ms),
    so it only pays when blocks come back: `limit_mb` bounds what the pool pins at any time (IMPULSE_HIP_PINNED_MB, default
    1024 - a caller that keeps every result pays for at most that much pinning, once); take() returns None beyond it and the
    caller uses ordinary memory."""
    GRAIN = 1 << 18                                        # doubles: blocks come in multiples of 2 MiB

    def __init__(self, limit_mb=None):
        self._lib = load_library()
        self.limit = int(float(os.environ.get("IMPULSE_HIP_PINNED_MB", 1024) if limit_mb is None else limit_mb) * (1 << 20))
        self._lock = threading.Lock()
        self._free = {}                                    # doubles -> [ptr]
        self.pinned = 0                                    # bytes pinned now (in use + free)
        self.allocations = 0
        self._closed = False

    def take(self, ctx, doubles):
        """a block of at least `doubles` float64 (pinned through ctx's device when a new one is needed), or None"""
        size = -(-int(doubles) // self.GRAIN) * self.GRAIN
        with self._lock:
            ptrs = self._free.get(size)
            if ptrs:
                return PinnedBlock(self, ptrs.pop(), size)
            if self._closed or self.pinned + 8 * size > self.limit:
                return None
            self.pinned += 8 * size
            self.allocations += 1
        p = _vp()
        try:
            _check(self._lib.imp_host_alloc(ctx.handle, 8 * size, C.byref(p)))
        except BaseException:
            with self._lock:
                self.pinned -= 8 * size
            raise
        return PinnedBlock(self, p.value, size)

    def _give(self, ptr, size):
        with self._lock:
            if not self._closed:
                self._free.setdefault(size, []).append(ptr)
                return
            self.pinned -= 8 * size
        self._lib.imp_host_free(_vp(ptr))

    def close(self):
        """frees the blocks that are back; blocks still held by arrays are freed when those arrays go"""
        with self._lock:
            self._closed = True
            ptrs = [(p, size) for size, lst in self._free.items() for p in lst]
            self._free = {}
            self.pinned -= sum(8 * size for _, size in ptrs)
        for p, _ in ptrs:
            self._lib.imp_host_free(_vp(p))


def comm_probe():
    """True if librccl loads with every entry point the library uses (no communicator is made)."""
    return load_library().imp_comm_probe() == 0


def comm_unique_id():
    """128 bytes that identify a new RCCL communicator (rank 0 makes them and shares them with the other ranks)."""
    buf = (C.c_ubyte * 128)()
    _check(load_library().imp_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """RCCL communicator of one rank (imp_comm): collective construction, broadcast of device buffers."""

    def __init__(self, ctx, unique_id, rank, nranks):
        if len(unique_id) != 128:
            raise ValueError("the unique id is 128 bytes")
        self.ctx, self._lib = ctx, ctx._lib
        self.rank, self.nranks = int(rank), int(nranks)
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        h = _vp()
        _check(self._lib.imp_comm_create(ctx.handle, buf, self.rank, self.nranks, C.byref(h)))
        self._h = h
        ctx._plans.add(self)

    def nranks_seen(self):
        """ranks of the communicator as RCCL itself counts them"""
        n = C.c_int(0)
        _check(self._lib.imp_comm_nranks(self._h, C.byref(n)))
        return int(n.value)

    def broadcast(self, dptr, nbytes, root=0):
        _check(self._lib.imp_comm_broadcast(self._h, _vp(int(dptr)), int(nbytes), int(root)))

    def broadcast_plan_spectrum(self, plan, root=0):
        n = C.c_size_t(0)
        _check(self._lib.imp_plan_broadcast_spectrum(plan.handle, self._h, int(root), C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                self._lib.imp_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class Curves:
    """K12: curve conditioning on one frequency grid (imp_curves): batched [B, n] fp64 dB curves in, out."""

    def __init__(self, ctx, frequency):
        self.ctx = ctx
        self._lib = ctx._lib
        f = np.ascontiguousarray(frequency, dtype=np.float64)
        self.n = len(f)
        h = _vp()
        _check(self._lib.imp_curves_create(ctx.handle, f.ctypes.data_as(_pd), self.n, C.byref(h)))
        self._h = h
        ctx._plans.add(self)

    def _rows(self, x):
        a = np.ascontiguousarray(x, dtype=np.float64)
        one = a.ndim == 1
        if one:
            a = a[None, :]
        if a.ndim != 2 or a.shape[1] != self.n:
            raise ValueError(f"curves must be [B, {self.n}], got {a.shape}")
        return a, one

    def window_size(self, octaves):
        w = C.c_int(0)
        _check(self._lib.imp_curves_window_size(self._h, float(octaves), C.byref(w)))
        return w.value

    def smooth(self, x, window_oct, treble_window_oct, treble_f_lower, treble_f_upper):
        a, one = self._rows(x)
        y = np.empty_like(a)
        _check(self._lib.imp_curves_smooth(self._h, a.ctypes.data_as(_pd), a.shape[0], float(window_oct),
                                           float(treble_window_oct), float(treble_f_lower), float(treble_f_upper),
                                           y.ctypes.data_as(_pd)))
        return y[0] if one else y

    def equalization(self, error, smoothen_first, max_gain, treble_f_lower, treble_f_upper, treble_max_gain, treble_gain_k,
                     smoothen_kinks=True):
        """(error_smoothed, equalization, spline_used[B]) of autoeq's smoothen_heavy_light (optional) + equalize"""
        a, one = self._rows(error)
        es, eq = np.empty_like(a), np.empty_like(a)
        used = np.zeros(a.shape[0], dtype=np.int32)
        _check(self._lib.imp_curves_equalization(self._h, a.ctypes.data_as(_pd), a.shape[0], 1 if smoothen_first else 0,
                                                 float(max_gain), float(treble_f_lower), float(treble_f_upper),
                                                 float(treble_max_gain), float(treble_gain_k), 1 if smoothen_kinks else 0,
                                                 es.ctypes.data_as(_pd), eq.ctypes.data_as(_pd),
                                                 used.ctypes.data_as(C.POINTER(C.c_int))))
        return (es[0], eq[0], used) if one else (es, eq, used)

    def fir_taps(self, fs, f_res):
        n = _i64()
        _check(self._lib.imp_curves_fir_taps(self._h, float(fs), float(f_res), C.byref(n)))
        return n.value

    def fir(self, equalization, fs, f_res, normalize, want_gain=False):
        a, one = self._rows(equalization)
        taps = self.fir_taps(fs, f_res)
        fir = np.empty((a.shape[0], taps), dtype=np.float64)
        gain = np.empty_like(fir) if want_gain else None
        _check(self._lib.imp_curves_fir(self._h, a.ctypes.data_as(_pd), a.shape[0], float(fs), float(f_res),
                                        1 if normalize else 0, gain.ctypes.data_as(_pd) if want_gain else None,
                                        fir.ctypes.data_as(_pd)))
        if want_gain:
            return (fir[0], gain[0]) if one else (fir, gain)
        return fir[0] if one else fir

    def equalization_fir(self, error, smoothen_first, max_gain, treble_f_lower, treble_f_upper, treble_max_gain,
                         treble_gain_k, fs, f_res, normalize, smoothen_kinks=True):
        """error curves -> (equalization [B, n], FIR [B, taps]) in one device chain"""
        a, one = self._rows(error)
        taps = self.fir_taps(fs, f_res)
        eq = np.empty_like(a)
        fir = np.empty((a.shape[0], taps), dtype=np.float64)
        _check(self._lib.imp_curves_equalization_fir(self._h, a.ctypes.data_as(_pd), a.shape[0], 1 if smoothen_first else 0,
                                                     float(max_gain), float(treble_f_lower), float(treble_f_upper),
                                                     float(treble_max_gain), float(treble_gain_k),
                                                     1 if smoothen_kinks else 0, float(fs), float(f_res),
                                                     1 if normalize else 0, eq.ctypes.data_as(_pd), fir.ctypes.data_as(_pd)))
        return (eq[0], fir[0]) if one else (eq, fir)

    def equalization_fir_device(self, error, smoothen_first, max_gain, treble_f_lower, treble_f_upper, treble_max_gain,
                                treble_gain_k, fs, f_res, normalize, smoothen_kinks=True, want_equalization=False):
        """error curves -> FIRs LEFT ON THE DEVICE: (equalization [B, n] or None, DeviceFirs)"""
        a, one = self._rows(error)
        eq = np.empty_like(a) if want_equalization else None
        d = _vp()
        taps = _i64()
        _check(self._lib.imp_curves_equalization_fir_device(
            self._h, a.ctypes.data_as(_pd), a.shape[0], 1 if smoothen_first else 0, float(max_gain), float(treble_f_lower),
            float(treble_f_upper), float(treble_max_gain), float(treble_gain_k), 1 if smoothen_kinks else 0, float(fs), float(f_res),
            1 if normalize else 0, eq.ctypes.data_as(_pd) if want_equalization else None, C.byref(d), C.byref(taps)))
        return eq, DeviceFirs(self.ctx, d.value, a.shape[0], taps.value)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                self._lib.imp_curves_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class DeviceFirs:
    """A batch of FIRs [B, taps] (fp64) that the design left on the device.  Consumers on the same device take the
    device pointer (ConvPlan.set_filters_device, Slice.set_firs_device); rows() hands out DeviceFir objects that turn
    into host arrays only when somebody reads them (np.asarray(row) / row.host())."""

    def __init__(self, ctx, dptr, B, taps):
        self.ctx, self.ptr, self.B, self.taps = ctx, int(dptr), int(B), int(taps)
        self._host = None
        self._ready = False

    def ready(self):
        """wait (once) until the design's stream has produced the taps: consumers on OTHER streams call this first"""
        if not self._ready:
            self.ctx.synchronize()
            self._ready = True
        return self

    def host(self):
        if self._host is None:
            out = np.empty((self.B, self.taps), dtype=np.float64)
            self.ctx.synchronize()
            self.ctx.d2h(out, self.ptr)
            self._host, self._ready = out, True
        return self._host

    def rows(self):
        return [DeviceFir(self, i) for i in range(self.B)]

    def close(self):
        if self.ptr and getattr(self.ctx, "_h", None):
            self.ctx.free(self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class DeviceFir:
    """row i of a DeviceFirs batch: an array-like that becomes a host array when read"""
    __slots__ = ("batch", "index")

    def __init__(self, batch, index):
        self.batch, self.index = batch, index

    def host(self):
        return self.batch.host()[self.index]

    def __array__(self, dtype=None, copy=None):
        a = self.host()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return self.batch.taps

    @property
    def shape(self):
        return (self.batch.taps,)


class ConvPlan:
    """Batched FFT convolution plan (imp_plan): scipy.signal.convolve(x, h, mode) for fixed (h, L)."""

    def __init__(self, ctx, filt, L, mode="same", ws_channels=0, empty_M=None, n_filters=None, paired=False, fused=True):
        """paired: False = one channel per transform; True = pair mode (two channels per complex transform, one shared
        filter; NativeError if the lengths need more than 256 rows); "auto" = pair mode where it is available.
        fused: False keeps the three-launch transform for filters short enough for the fused overlap-save kernel."""
        self._lib = ctx._lib
        self.ctx = ctx
        self.L = int(L)
        self.mode = {"same": IMP_MODE_SAME, "full": IMP_MODE_FULL}[mode]
        h = _vp()
        if filt is None:
            self.M = int(empty_M)
            self.n_filters = int(n_filters or 1)
            f = None
        else:
            f = np.ascontiguousarray(filt, dtype=np.float64)
            if f.ndim == 1:
                f = f[None, :]
            self.n_filters, self.M = int(f.shape[0]), int(f.shape[1])
        if paired and self.n_filters == 1 and os.environ.get("IMPULSE_HIP_NO_PAIRS") != "1":
            if f is None:
                rc = self._lib.imp_conv_plan_create_empty_paired(ctx.handle, self.M, self.L, self.mode, int(ws_channels),
                                                                 C.byref(h))
            else:
                rc = self._lib.imp_conv_plan_create_paired(ctx.handle, f.ctypes.data_as(_pd), self.M, self.L, self.mode,
                                                           int(ws_channels), C.byref(h))
            if rc == IMP_ERR_UNSUPPORTED and paired == "auto":
                h = _vp()                                     # too long for pair mode: one channel per transform
            else:
                _check(rc)
        elif paired is True:
            raise ValueError("pair mode needs one shared filter")
        if not h:
            _check(self._lib.imp_conv_plan_create_ex(ctx.handle, None if f is None else f.ctypes.data_as(_pd), self.M,
                                                     self.n_filters, self.M, self.L, self.mode, int(ws_channels),
                                                     0 if fused else 2, C.byref(h)))
        self._h = h
        kind = C.c_int(0)
        _check(self._lib.imp_plan_kind(h, C.byref(kind)))
        self.paired, self.fused = kind.value == 1, kind.value == 2
        ctx._plans.add(self)
        nfft, out_len, wsc, n1 = _i64(), _i64(), _i64(), _i64()
        _check(self._lib.imp_plan_info(self._h, C.byref(nfft), C.byref(out_len), C.byref(wsc), C.byref(n1)))
        self.nfft, self.out_len, self.ws_channels, self.n1 = nfft.value, out_len.value, wsc.value, n1.value

    @property
    def handle(self):
        return self._h

    def copy_spectrum_from(self, src_plan):
        """the prepared spectrum of a plan of the same geometry on another context / device (peer copy)"""
        _check(self._lib.imp_plan_copy_spectrum(self._h, src_plan.handle))

    def spectrum_buffer(self):
        p, n = _vp(), C.c_size_t()
        _check(self._lib.imp_plan_spectrum(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def execute(self, x):
        """x: [B, L] (or [L]) float -> [B, out_len] float32."""
        x = _f32(x)
        one = x.ndim == 1
        if one:
            x = x[None, :]
        if x.shape[1] != self.L:
            raise ValueError(f"plan was made for L={self.L}, got rows of {x.shape[1]}")
        B = x.shape[0]
        y = np.empty((B, self.out_len), dtype=np.float32)
        _check(self._lib.imp_conv_execute(self._h, _ptr_f(x), B, x.shape[1], _ptr_f(y), y.shape[1]))
        return y[0] if one else y

    def execute_interleaved(self, frames):
        """frames: [L, C] float (WAV wire order) -> [C, out_len] float32."""
        frames = _f32(frames)
        if frames.ndim != 2 or frames.shape[0] != self.L:
            raise ValueError(f"frames must be [L={self.L}, C]")
        Cn = frames.shape[1]
        y = np.empty((Cn, self.out_len), dtype=np.float32)
        _check(self._lib.imp_conv_execute_interleaved(self._h, _ptr_f(frames), Cn, _ptr_f(y), y.shape[1]))
        return y

    def execute_device(self, d_x, B, chan_stride_in, d_y, chan_stride_out, elem_stride_in=1):
        _check(self._lib.imp_conv_execute_device(self._h, _vp(int(d_x)), int(B), int(chan_stride_in),
                                                 int(elem_stride_in), _vp(int(d_y)), int(chan_stride_out)))

    def execute_device_pcm(self, d_pcm, bits, B, chan_stride_in, elem_stride_in, d_y, chan_stride_out):
        _check(self._lib.imp_conv_execute_device_pcm(self._h, _vp(int(d_pcm)), int(bits), int(B), int(chan_stride_in),
                                                     int(elem_stride_in), _vp(int(d_y)), int(chan_stride_out)))

    def execute_device_pairs(self, d_x, bits, n_pairs, pair_stride, right_offset, elem_stride, d_y, chan_stride_out):
        """pair-mode plans: pair q = samples d_x[q*pair_stride + i*elem_stride] (left) and [... + right_offset] (right);
        bits 0 = float32, 16 / 32 = PCM; outputs rows 2q, 2q + 1 of d_y"""
        _check(self._lib.imp_conv_execute_device_pairs(self._h, _vp(int(d_x)), int(bits), int(n_pairs), int(pair_stride),
                                                       int(right_offset), int(elem_stride), _vp(int(d_y)),
                                                       int(chan_stride_out)))

    def execute_pcm_columns(self, frames, column_starts):
        """frames: interleaved PCM [n_frames, tracks] int16/int32 (WAV wire order).  Deconvolves, for every
        column start s, the L frames frames[s:s+L] of every track: returns float32 [len(column_starts), tracks,
        out_len].  The block is uploaded once, untouched; scaling and de-interleaving happen in the loader."""
        frames = np.ascontiguousarray(frames)
        if frames.dtype not in (np.int16, np.int32) or frames.ndim != 2:
            raise ValueError("frames must be int16/int32 [n_frames, tracks]")
        bits = 16 if frames.dtype == np.int16 else 32
        n_frames, tracks = frames.shape
        ctx = self.ctx
        pitch = (self.out_len + 1) & ~1
        d_in = ctx.malloc(frames.nbytes)
        d_out = ctx.malloc(max(1, len(column_starts) * tracks) * pitch * 4)
        try:
            ctx.h2d(d_in, frames)
            self._launch_pcm_columns(d_in, frames, column_starts, d_out, pitch)
            ctx.synchronize()
            out = np.empty((len(column_starts), tracks, pitch), dtype=np.float32)
            ctx.d2h(out, d_out)
        finally:
            ctx.free(d_in)
            ctx.free(d_out)
        return out[:, :, :self.out_len]

    def execute_pcm_columns_device(self, frames, column_starts, d_out, pitch):
        """As execute_pcm_columns, but the result stays on the device: column j, track t -> d_out + 4 * (j * tracks
        + t) * pitch (pitch >= out_len floats).  The PCM block is uploaded once and freed after the launches."""
        frames = np.ascontiguousarray(frames)
        if frames.dtype not in (np.int16, np.int32) or frames.ndim != 2:
            raise ValueError("frames must be int16/int32 [n_frames, tracks]")
        bits = 16 if frames.dtype == np.int16 else 32
        n_frames, tracks = frames.shape
        ctx = self.ctx
        d_in = ctx.malloc(frames.nbytes)
        try:
            ctx.h2d(d_in, frames)
            self._launch_pcm_columns(d_in, frames, column_starts, d_out, pitch)
        finally:
            ctx.free(d_in)                                  # ordered on the context's stream: no wait

    def _launch_pcm_columns(self, d_in, frames, column_starts, d_out, pitch):
        """Launch groups for the columns of an uploaded PCM block: column j, track t -> row j * tracks + t of d_out.
        Equally spaced columns (sweep_sequence lays them out so) take ONE launch group per track whose channels are the
        columns - 2 groups of 8 channels for a 7.1 recording instead of 8 groups of 2."""
        n_frames, tracks = frames.shape
        bits = 16 if frames.dtype == np.int16 else 32
        starts = [int(v) for v in column_starts]
        if any(s0 < 0 or s0 + self.L > n_frames for s0 in starts):
            raise ValueError("column outside the recording")
        if self.paired and tracks == 2:
            # a binaural recording: the two ears of a column are ONE complex signal, a stereo frame is one load (core/hrir.py:
            # 326-341: track 0 = left ear, track 1 = right ear); columns in runs of equal spacing, one launch group per run
            j = 0
            while j < len(starts):
                k = j + 1
                step = starts[k] - starts[j] if k < len(starts) else 0
                while step > 0 and k < len(starts) and starts[k] - starts[k - 1] == step:
                    k += 1
                if step <= 0:
                    k = j + 1
                self.execute_device_pairs(d_in + starts[j] * tracks * frames.itemsize, bits, k - j, step * tracks, 1, tracks,
                                          d_out + 2 * j * pitch * 4, pitch)
                j = k
            return
        step = starts[1] - starts[0] if len(starts) > 1 else 0
        uniform = len(starts) > 1 and step > 0 and all(b - a == step for a, b in zip(starts, starts[1:]))
        if uniform:
            for t in range(tracks):
                self.execute_device_pcm(d_in + (starts[0] * tracks + t) * frames.itemsize, bits, len(starts),
                                        step * tracks, tracks, d_out + t * pitch * 4, tracks * pitch)
        else:
            for j, s0 in enumerate(starts):
                self.execute_device_pcm(d_in + s0 * tracks * frames.itemsize, bits, tracks, 1, tracks,
                                        d_out + j * tracks * pitch * 4, pitch)

    def set_overlap(self, lanes):
        """lanes > 1: successive launch groups of execute_device overlap on that many streams (inputs must
        be ready before each call; outputs are complete after ctx.synchronize())."""
        _check(self._lib.imp_plan_set_overlap(self._h, int(lanes)))

    def set_filters(self, filt):
        """New filter(s) of the same shape for this plan (recomputes the spectra on the device, in place)."""
        f = np.ascontiguousarray(filt, dtype=np.float64)
        if f.ndim == 1:
            f = f[None, :]
        if f.shape != (self.n_filters, self.M):
            raise ValueError(f"plan holds {self.n_filters} filter(s) of {self.M} taps, got {f.shape}")
        _check(self._lib.imp_plan_set_filters(self._h, f.ctypes.data_as(_pd), self.M))

    def set_filters_device(self, d_filt, ld=None):
        """New filters of the same shape from fp64 device memory [n_filters][ld]: no upload, no wait (stream order)."""
        _check(self._lib.imp_plan_set_filters_device(self._h, _vp(int(d_filt)), int(ld or self.M)))

    def set_timing(self, every_n):
        """0/False = off; n = bracket the three passes of every n-th launch group with HIP events."""
        _check(self._lib.imp_plan_set_timing(self._h, int(every_n)))

    def get_timing(self, reset=True):
        ms = (C.c_double * 3)()
        n = _i64()
        _check(self._lib.imp_plan_get_timing(self._h, ms, C.byref(n), 1 if reset else 0))
        return [ms[0], ms[1], ms[2]], n.value

    def debug_stage(self, x, stage):
        x = _f32(x)
        if x.ndim == 1:
            x = x[None, :]
        B = x.shape[0]
        out = np.empty((B, self.n1, 4096), dtype=np.complex64)
        _check(self._lib.imp_plan_debug_run_stage(self._h, _ptr_f(x), B, x.shape[1], int(stage),
                                                  out.ctypes.data_as(_pf)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if getattr(self.ctx, "_h", None):          # the context closes its plans before it goes
                self._lib.imp_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_geometry(M, L, mode="same"):
    """(nfft, out_start, out_len) the library picks for a plan; needs no GPU."""
    lib = load_library()
    a, b, c = _i64(), _i64(), _i64()
    _check(lib.imp_debug_plan_geometry(int(M), int(L), {"same": IMP_MODE_SAME, "full": IMP_MODE_FULL}[mode],
                                       C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def plan_geometry_fused(M, L, mode="same"):
    """(history, valid, first_block, blocks) of a fused FIR plan, or None beyond 24 577 taps; needs no GPU."""
    lib = load_library()
    a, b, c, d = _i64(), _i64(), _i64(), _i64()
    rc = lib.imp_debug_plan_geometry_fused(int(M), int(L), {"same": IMP_MODE_SAME, "full": IMP_MODE_FULL}[mode],
                                           C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    if rc == IMP_ERR_UNSUPPORTED:
        return None
    _check(rc)
    return a.value, b.value, c.value, d.value


def plan_geometry_paired(M, L, mode="same"):
    """(nfft, out_start, out_len, n1_rows) of a pair-mode plan, or None where pair mode is not available; needs no GPU."""
    lib = load_library()
    a, b, c, d = _i64(), _i64(), _i64(), _i64()
    rc = lib.imp_debug_plan_geometry_paired(int(M), int(L), {"same": IMP_MODE_SAME, "full": IMP_MODE_FULL}[mode],
                                            C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    if rc == IMP_ERR_UNSUPPORTED:
        return None
    _check(rc)
    return a.value, b.value, c.value, d.value


def host_spectrum(filt, n1_rows):
    """alpha/beta planes [n1_rows, 4096, 4] float32 exactly as a plan uploads them; needs no GPU."""
    lib = load_library()
    f = np.ascontiguousarray(filt, dtype=np.float64)
    out = np.empty((int(n1_rows), 4096, 4), dtype=np.float32)
    _check(lib.imp_debug_host_spectrum(f.ctypes.data_as(_pd), len(f), int(n1_rows), out.ctypes.data_as(_pf)))
    return out


_default_ctx = None
_aux_ctx = None
_default_lock = threading.Lock()
_thread_ctx = threading.local()


def default_device():
    """the device index the package's default context lives on: the first entry of IMPULSE_HIP_DEVICES, else
    IMPULSE_HIP_DEVICE, else 0"""
    spec = os.environ.get("IMPULSE_HIP_DEVICES", "").strip()
    if spec:
        return int(spec.replace(";", ",").split(",")[0])
    return int(os.environ.get("IMPULSE_HIP_DEVICE", "0"))


def device_list():
    """IMPULSE_HIP_DEVICES = "0,1,2,...": the devices the classes shard channel pairs over from this ONE process (one
    context and one host thread per entry; an index may repeat: two contexts folded on one device).  Unset: the default
    device alone."""
    spec = os.environ.get("IMPULSE_HIP_DEVICES", "").strip()
    if not spec:
        return [default_device()]
    return [int(v) for v in spec.replace(";", ",").split(",") if v.strip() != ""]


_device_ctxs = None
_device_spec = None


def root_context():
    """the process-wide default context, whatever the calling thread has installed with using_context"""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None or not _default_ctx._h:
            _default_ctx = Context(default_device())
        return _default_ctx


def device_contexts():
    """[root context, one more context per further entry of IMPULSE_HIP_DEVICES]: what estimate_batch and the recording
    ingest shard over.  Made once per value of the variable."""
    global _device_ctxs, _device_spec
    devs = device_list()
    root = root_context()
    with _default_lock:
        if _device_ctxs is None or _device_spec != tuple(devs) or any(not c._h for c in _device_ctxs):
            _device_ctxs = [root] + [Context(d) for d in devs[1:]]
            _device_spec = tuple(devs)
        return list(_device_ctxs)


def default_context():
    """Process-wide context on IMPULSE_HIP_DEVICE (default 0) - or the one `using_context` installed for this thread."""
    global _default_ctx
    override = getattr(_thread_ctx, "ctx", None)
    if override is not None:
        return override
    return root_context()


def aux_context():
    """A second context (its own stream) on the default context's device, for work that does not depend on what the
    default context is doing: pipeline_slice designs the equalisation FIRs there while the recording uploads."""
    global _aux_ctx
    with _default_lock:
        if _aux_ctx is None or not _aux_ctx._h:
            _aux_ctx = Context(default_device())
        return _aux_ctx


class using_context:
    """with using_context(ctx): ... - the classes of this package, called from THIS thread inside the block, take `ctx`
    wherever they would take default_context()."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.prev = getattr(_thread_ctx, "ctx", None)
        _thread_ctx.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _thread_ctx.ctx = self.prev
        return False
