"""Responses that stay on the GPU between the stages of a measurement.

The reference keeps every impulse response as a float64 NumPy array and each stage reads and rewrites it
(core/hrir.py:457-653, 858-888).  Here the deconvolved responses of a recording are left on the device as fp32 rows of
one allocation; a head or tail crop is a change of (offset, length), the Hann fades, the FIR filtering and the
normalisation gain are device kernels on those rows, and only what the host decides with - peak indices, knee-search
window levels, one spectrum per ear - crosses the bus.  `ImpulseResponse.data` remains the public, writable array of
the class surface: reading it brings the row to the host and from then on the host array is the truth (the device row
is dropped), so code that mutates `.data` in place behaves exactly as with the reference.
"""
import numpy as np


class DeviceBlock:
    """One device allocation of fp32 samples that rows are cut from."""

    def __init__(self, ctx, n_floats):
        self.ctx = ctx
        self.n = int(n_floats)
        self.ptr = ctx.malloc(max(self.n, 1) * 4)
        self._host = None

    def touch(self):
        """the device content changed: a cached host copy is stale"""
        self._host = None

    def host(self):
        """the whole block on the host (one transfer, cached until the block changes)"""
        if self._host is None:
            out = np.empty(self.n, dtype=np.float32)
            if self.n:
                self.ctx.synchronize()
                self.ctx.d2h(out, self.ptr)
            self._host = out
        return self._host

    def close(self):
        if self.ptr and getattr(self.ctx, "_h", None):
            self.ctx.free(self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 - interpreter shutdown
            pass


class Row:
    """n samples at block.ptr + 4 * off"""
    __slots__ = ("block", "off", "n")

    def __init__(self, block, off, n):
        self.block, self.off, self.n = block, int(off), int(n)

    @property
    def ptr(self):
        return self.block.ptr + 4 * self.off

    def to_host(self):
        return self.block.host()[self.off:self.off + self.n].astype(np.float64)


def span(rows):
    """(base pointer, offsets in floats, lengths) addressing rows of possibly different blocks of one context"""
    base = min(r.block.ptr for r in rows)
    offs = np.array([(r.block.ptr - base) // 4 + r.off for r in rows], dtype=np.int64)
    lens = np.array([r.n for r in rows], dtype=np.int64)
    return base, offs, lens


def uniform(rows):
    """pitch if the rows are equally long, in one block and equally spaced in order; else None"""
    if not rows or any(r.block is not rows[0].block or r.n != rows[0].n for r in rows):
        return None
    if len(rows) == 1:
        return max(rows[0].n, 1)
    pitch = rows[1].off - rows[0].off
    if pitch < rows[0].n or any(rows[i].off - rows[0].off != i * pitch for i in range(len(rows))):
        return None
    return pitch


def shift_rows(rows, shifts):
    """ImpulseResponse.shift (core/impulse_response.py:92-108) for device rows: rows[k] delayed by shifts[k] > 0 or advanced
    by -shifts[k], length kept.  One new block and one launch for the batch; returns the new rows (the old ones untouched)."""
    if not rows:
        return []
    ctx = rows[0].block.ctx
    base, offs, lens = span(rows)
    pitch = [(int(n) + 63) // 64 * 64 for n in lens]
    dst_off = np.concatenate([[0], np.cumsum(pitch)[:-1]]).astype(np.int64)
    block = DeviceBlock(ctx, int(sum(pitch)))
    ctx.shift_rows_device(base, offs, lens, np.asarray(shifts, dtype=np.int64), block.ptr, dst_off)
    return [Row(block, int(o), int(n)) for o, n in zip(dst_off, lens)]
