"""WAV I/O and small DSP primitives (surface of reference core/audio_io.py).

WAV files are read/written with a self-contained RIFF codec (PCM 16/24/32, IEEE float 32/64,
WAVE_FORMAT_EXTENSIBLE) so the package has no libsndfile dependency; PCM samples map to
[-1, 1) as int / 2^(bits-1), the same convention soundfile uses (reference
core/audio_truehd.py:153-185 transposes to [tracks, samples]).
"""
import os
import struct

import numpy as np


def _parse_riff(blob, file_path):
    if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{file_path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    data = None
    while pos + 8 <= len(blob):
        cid, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        body = memoryview(blob)[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            tag, nch, fs, _, align, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:            # extensible: real tag in the GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, nch, fs, bits, align)
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{file_path}: missing fmt/data chunk")
    return fmt, data


def read_wav_pcm(file_path):
    """(fs, frames[n_frames, tracks]) with the file's own int16/int32 samples in wire order, or None when
    the file is not 16/32-bit PCM.  This is what the device loader consumes directly."""
    with open(file_path, "rb") as fh:
        blob = fh.read()
    (tag, nch, fs, bits, _), data = _parse_riff(blob, file_path)
    if tag != 1 or bits not in (16, 32):
        return None
    nframes = len(data) // ((bits // 8) * nch)
    frames = np.frombuffer(data, dtype="<i2" if bits == 16 else "<i4", count=nframes * nch).reshape(nframes, nch)
    return int(fs), frames


def read_wav(file_path, expand=False):
    """Returns (fs, data[tracks, samples] float64); 1-D for mono unless ``expand``."""
    with open(file_path, "rb") as fh:
        blob = fh.read()
    if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{file_path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    data = None
    while pos + 8 <= len(blob):
        cid, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            tag, nch, fs, _, align, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:            # extensible: real tag in the GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, nch, fs, bits, align)
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{file_path}: missing fmt/data chunk")
    tag, nch, fs, bits, align = fmt
    nbytes = bits // 8
    nframes = len(data) // (nbytes * nch)
    data = data[: nframes * nbytes * nch]
    if tag == 1:
        if bits == 16:
            x = np.frombuffer(data, dtype="<i2").astype(np.float64) / 2.0 ** 15
        elif bits == 32:
            x = np.frombuffer(data, dtype="<i4").astype(np.float64) / 2.0 ** 31
        elif bits == 24:
            b = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
            v = np.where(v & 0x800000, v - (1 << 24), v)
            x = v.astype(np.float64) / 2.0 ** 23
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8").astype(np.float64)
    else:
        raise ValueError(f"unsupported WAV format tag {tag}")
    x = x.reshape(nframes, nch).T                      # [tracks, samples] view, like the reference
    if nch == 1 and not expand:
        x = x[0]
    return int(fs), x


def pcm_quantise(frames, bit_depth):
    """float -> PCM integers the way the reference's writer does it.  The reference writes through
    soundfile.write (core/audio_io.py:82-97); python-soundfile switches libsndfile's clipping on for every
    file it opens, which selects libsndfile's ``*_clip_array`` conversions: the sample is scaled to the 32-bit
    range, ``q32 = clip(lrint(x * 2^31), -2^31, 2^31 - 1)`` (round half to even, saturating), and narrower
    subtypes keep its top bits, ``q = q32 >> (32 - bits)``.

    PCM_32 is PINNED by the four sweep WAVs the reference ships under data/ (written by its own
    core/impulse_response_estimator.py:306-322; the sweep peaks at 0.99999999996, so they show both the 2^31
    scale and the saturation at +2147483647): tests/golden/sweep_wavs.npz.  No 16- or 24-bit file ships, so for
    those two widths this restates libsndfile's published clip path (src/pcm.c, d2s_clip_array /
    d2let_clip_array): parity of PCM_16 / PCM_24 is unpinned."""
    if bit_depth not in (16, 24, 32):
        raise ValueError('Invalid bit depth. Accepted values are 16, 24 and 32.')
    q32 = np.clip(np.rint(np.asarray(frames, dtype=np.float64) * 2.0 ** 31), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
    return q32 >> (32 - bit_depth)


def write_wav_frames(file_path, fs, frames, bit_depth):
    """PCM WAV from already quantised interleaved frames [n_frames, n_tracks] (int16 for 16 bit, integers in int32 for
    24 / 32 bit) - the block a device-side conversion returns (imp_rows_to_pcm_device)."""
    if bit_depth not in (16, 24, 32):
        raise ValueError('Invalid bit depth. Accepted values are 16, 24 and 32.')
    d = os.path.dirname(file_path)
    if d:
        os.makedirs(d, exist_ok=True)
    q = np.ascontiguousarray(frames)
    nframes, nch = q.shape
    if bit_depth == 16:
        raw = q.astype("<i2").tobytes()
    elif bit_depth == 32:
        raw = q.astype("<i4").tobytes()
    else:
        u = (q.astype(np.int64) & 0xFFFFFF).astype(np.uint32).reshape(-1)
        raw = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
    nbytes = bit_depth // 8
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(raw), b"WAVE", b"fmt ", 16, 1, nch, int(fs),
                      int(fs) * nch * nbytes, nch * nbytes, bit_depth, b"data", len(raw))
    with open(file_path, "wb") as fh:
        fh.write(hdr + raw)


def write_wav(file_path, fs, data, bit_depth=32):
    """PCM writer; rows are tracks (reference core/audio_io.py:82-97)."""
    if bit_depth not in (16, 24, 32):
        raise ValueError('Invalid bit depth. Accepted values are 16, 24 and 32.')
    d = os.path.dirname(file_path)
    if d:
        os.makedirs(d, exist_ok=True)
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 1:
        data = data[None, :]
    elif data.shape[1] <= data.shape[0]:
        data = data.T                                   # frames were on rows already
    nch, nframes = data.shape
    frames = np.ascontiguousarray(data.T)
    q = pcm_quantise(frames, bit_depth)
    if bit_depth == 16:
        raw = q.astype("<i2").tobytes()
    elif bit_depth == 32:
        raw = q.astype("<i4").tobytes()
    else:
        u = (q & 0xFFFFFF).astype(np.uint32).reshape(-1)
        raw = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
    nbytes = bit_depth // 8
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(raw), b"WAVE", b"fmt ", 16, 1, nch, int(fs),
                      int(fs) * nch * nbytes, nch * nbytes, bit_depth, b"data", len(raw))
    with open(file_path, "wb") as fh:
        fh.write(hdr + raw)


def magnitude_response(x, fs):
    """20 log10 |rfft(x)| on the first ceil(n/2) bins (reference core/audio_io.py:100-113), computed
    on the GPU in fp64 for any length (kernel K2: Bluestein on a Stockham FFT).  Exact zeros give -inf,
    as in the reference (no epsilon)."""
    from . import _native
    n = len(x)
    half = int(np.ceil(n / 2))
    f = np.arange(half) * (fs / n) if n else np.zeros(0)
    if n == 0:
        return f, np.zeros(0)
    return f, _native.default_context().magnitude_db(np.asarray(x, dtype=np.float64))


def magnitude_responses(rows, fs):
    """Batched form for equally long rows: [B, n] -> (f, dB[B, ceil(n/2)]) in one launch chain."""
    from . import _native
    rows = np.asarray(rows, dtype=np.float64)
    n = rows.shape[1]
    half = int(np.ceil(n / 2))
    return np.arange(half) * (fs / n), _native.default_context().magnitude_db(rows)


def running_mean(x, N):
    c = np.cumsum(np.insert(x, 0, 0))
    return (c[N:] - c[:-N]) / float(N)


def to_db(x):
    return 20 * np.log10(np.abs(x) + 1e-10)


def db_to_gain(x):
    return 10 ** (x / 20)
