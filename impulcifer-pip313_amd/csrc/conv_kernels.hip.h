// Batched FFT linear convolution for gfx950 (MI355X): three HBM passes per transform pair.
//
// Replaces scipy.signal.convolve(x, h, 'same'|'full') as called by the reference at
//   core/impulse_response_estimator.py:151  (estimate: recording (*) inverse_filter, 'same')
//   core/impulse_response.py:119,135        (equalize / convolve: 'full')
//
// One real channel x[L] is packed even/odd into z[n] = x[2n] + i x[2n+1] (Nc = nfft/2 complex
// points, Nc = N1*N2, N2 = 4096, N1 = F*R2 in {16,24,32,40,48,64,72,80,96,128,144,160,192,256}) and transformed with a
// four-step FFT:
//   pass A  cols_kernel<fwd>   : length-N1 column FFTs (stride N2), x w_Nc^(n2 k1)     -> ws[k1][n2]
//   pass B  rows_kernel        : length-4096 row FFT -> real-FFT unpack * H * repack
//                                (W = alpha Z[k] + beta conj Z[Nc-k]) -> row IFFT, in place
//   pass C  cols_kernel<inv>   : x conj w_Nc^(n2 k1), column IFFTs, unpack + crop     -> out
// The spectrum never leaves the transposed [k1][k2] order, so there is no transpose pass.
// No MFMA: the work is butterflies (VALU) + LDS exchanges; the bound is the bytes moved across the L2 boundary.
#pragma once
#include "fft_regs.hip.h"

namespace imp {

constexpr int kN2 = 4096;        // row length (complex points)
constexpr int kLogN2 = 12;
constexpr int kRowPad = 272;     // LDS row pitch (float2) of the 16x256 exchange planes
#ifndef IMP_AB_PREFETCH
#define IMP_AB_PREFETCH 8
#endif
constexpr int kAbPrefetch = IMP_AB_PREFETCH;   // alpha/beta bins fetched before the partner exchange (register budget)

// ---------------------------------------------------------------------------------------------
// Twiddle tables (device, fp32 rounded from fp64 on the host).  Every table is laid out in the
// order the lanes read it, so a wave's read is one contiguous segment (no gathers):
//   full[k1*4096 + n2] = exp(-2 pi i (k1 n2 mod Nc) / Nc)      four-step twiddle, tiled like ws[]
//   hi[m]              = exp(-2 pi i 1024 m / Nc), m < Nc/1024  (w_N1^j = hi[4 j], wave-uniform reads)
//   t1[a*256 + t]      = w4096^(t a)                            row pass, stage after the 1st FFT16
//   t2[q*16 + l]       = w4096^(16 l q)   (= w256^(l q))        row pass, stage after the 2nd FFT16
//   t4[j*256 + t]      = w4096^((16 j + (t&15)) (t>>4))         row pass, inverse 2nd twiddle
// ---------------------------------------------------------------------------------------------
struct Twiddles {
  const cf* __restrict__ full;
  const cf* __restrict__ hi;
  const cf* __restrict__ t1;
  const cf* __restrict__ t2;
  const cf* __restrict__ t4;
};

// Blocks b and b+8 share an XCD (round-robin dispatch; a speed assumption only).  Work items are
// (tile, channel) with the SAME tile on the same XCD for every channel, channels fastest, so the
// tile's shared tables (twiddles, alpha/beta) are fetched once per XCD and then hit its L2.
// Requires tiles % 8 == 0 (always true here: 64/128 column tiles, N1/2 in {8..128} row pairs).
__device__ __forceinline__ void xcd_work_item(int nchan, int& tile, int& chan) {
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int tl = slot / nchan;
  chan = slot - tl * nchan;
  tile = tl * 8 + xcd;
}

// Raw buffer (SRSRC) addressing.  Every global access of the three passes is a buffer instruction:
// one 32-bit VGPR offset per thread + a scalar offset per element instead of 64-bit VGPR address
// pairs (rows: 138 VGPRs -> spills; columns: two thirds of the VALU work was address arithmetic
// and bounds branches), and the hardware range check does the zero fill and the crop:
//   * a load whose offset is >= num_records returns 0 without touching memory,
//   * an out-of-range store is dropped,
//   * both per dword, so a dwordx2 straddling the end keeps its valid half,
//   * voffset and soffset both take part; a voffset that is itself out of range (e.g. negative) is
//     out of range whatever soffset adds (tools/probes/buffer_oob_probe.hip, measured on gfx950).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Cache policy of the once-touched input / output streams: nt (aux = 2).  Measured with inputs streamed from HBM
// (1 GiB of batches in rotation): C2 +1.4 %, C3 +4 %, C5 +3 % over the default policy.  (When a benchmark re-reads
// one cache-resident batch every step nt costs 6 % instead - it gives up exactly that residency.)
#ifndef IMP_STREAM_AUX
#define IMP_STREAM_AUX 2
#endif
constexpr int kStreamAux = IMP_STREAM_AUX;
// Cache policy of the workspace traffic (written by one kernel, read once by the next): experiments only, default 0
#ifndef IMP_AUX_ROWS_LD
#define IMP_AUX_ROWS_LD 0
#endif
#ifndef IMP_AUX_ROWS_ST
#define IMP_AUX_ROWS_ST 0
#endif
#ifndef IMP_AUX_COLS_LD
#define IMP_AUX_COLS_LD 0
#endif
#ifndef IMP_AUX_COLS_ST
#define IMP_AUX_COLS_ST 0
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// AUX = cache policy bits of the instruction: 0 default, 1 sc0, 2 nt, 16 sc1 (agent scope: bypasses this CU's L1 and is
// served by the XCD's L2), 18 nt sc1
template <int AUX = 0>
__device__ __forceinline__ cf bload_cf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
  return make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
}
__device__ __forceinline__ float bload_f(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <int AUX = 0>
__device__ __forceinline__ float4 bload_f4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
  return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}
template <int AUX = 0>
__device__ __forceinline__ void bstore_cf(cf v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  u32x2 x;
  x.x = __float_as_uint(v.x);
  x.y = __float_as_uint(v.y);
  __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, AUX);
}

// ---------------------------------------------------------------------------------------------
// Loaders / storers for the column passes.  A column thread owns 16 complex points
//   e_j = e0 + j * step      (complex index n1*4096 + n2; e0 per thread, step wave-uniform)
// `column<STEP>(b, e0, v)` fetches them for channel b (b is wave-uniform).
// ---------------------------------------------------------------------------------------------

// Real fp32 channel, planar [B][ld] (elem_stride 1) or interleaved frames [L][C] (elem_stride = C,
// the channel offset folded into the base): z = (x[2n], x[2n+1]), zero beyond `len`.
struct LoadRealPacked {
  const float* __restrict__ base;   // channel 0, sample 0
  long long chan_stride;            // elements between channels
  long long elem_stride;            // elements between consecutive samples of a channel
  long long len;                    // valid samples per channel
  // the block [first, first + maxlen) of every channel (overlap-add over long inputs)
  __host__ __device__ LoadRealPacked shifted(long long first, long long maxlen) const {
    const long long rest = len - first;
    return LoadRealPacked{base + first * elem_stride, chan_stride, elem_stride, rest < maxlen ? rest : maxlen};
  }
  // fir_block_kernel's view of a channel: (x[s], x[s + 1]) for EVEN s, zero outside [0, len) - s may be negative (the
  // whole byte offset goes through the VGPR, so a negative one is out of range)
  struct Row {
    __amdgpu_buffer_rsrc_t r;
    unsigned es;                      // bytes between samples
    __device__ __forceinline__ cf finish(cf v, int) const { return v; }
    __device__ __forceinline__ cf pair_at(int s) const {               // |s| < 2^29: fused plans stop there
      if (es == 4u) return bload_cf<kStreamAux>(r, (unsigned)s * 4u, 0u);
      return make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)s * es, 0u, kStreamAux)),
                         __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)(s + 1) * es, 0u, kStreamAux)));
    }
  };
  __device__ __forceinline__ Row open(int b) const {
    const unsigned span = len > 0 ? ((unsigned)(len - 1) * (unsigned)elem_stride + 1u) * 4u : 0u;
    return Row{make_rsrc(base + (long long)b * chan_stride, span), (unsigned)elem_stride * 4u};
  }
  template <int STEP, int F>
  __device__ __forceinline__ void column(int b, unsigned e0, cf (&v)[F]) const {
    const float* p = base + (long long)b * chan_stride;
    if (elem_stride == 1) {
      const __amdgpu_buffer_rsrc_t r = make_rsrc(p, (unsigned)len * 4u);
#pragma unroll
      for (int j = 0; j < F; ++j) {
        const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, e0 * 8u, (unsigned)(j * STEP) * 8u, kStreamAux);
        v[j] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
      }
    } else {
      const unsigned es = (unsigned)elem_stride * 4u;                          // bytes between samples
      const __amdgpu_buffer_rsrc_t r = make_rsrc(p, ((unsigned)(len - 1) * (unsigned)elem_stride + 1u) * 4u);
      const unsigned vo = 2u * e0 * es;
#pragma unroll
      for (int j = 0; j < F; ++j) {
        const unsigned so = 2u * (unsigned)(j * STEP) * es;
        v[j].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, kStreamAux));
        v[j].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so + es, kStreamAux));
      }
    }
  }
};

// Raw PCM frames as they sit in a WAV file: sample i of channel c at base[i*elem_stride + c*chan_stride]
// (interleaved: elem_stride = tracks, chan_stride = 1), int32 or int16, scaled to [-1, 1) by 2^-(bits-1)
// exactly like soundfile/libsndfile and the reference's reader do (core/audio_truehd.py:153-185).
// int -> float is exact for 16-bit and rounds 32-bit PCM to fp32 (the device dtype).
template <class Sample>
struct LoadPcmPacked {
  const Sample* __restrict__ base;
  long long chan_stride;
  long long elem_stride;
  long long len;
  float scale;
  __host__ __device__ LoadPcmPacked shifted(long long first, long long maxlen) const {
    const long long rest = len - first;
    return LoadPcmPacked{base + first * elem_stride, chan_stride, elem_stride, rest < maxlen ? rest : maxlen, scale};
  }
  __device__ __forceinline__ float sample(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) const {
    if constexpr (sizeof(Sample) == 2) return (float)(short)__builtin_amdgcn_raw_buffer_load_b16(r, vo, so, kStreamAux) * scale;
    else return (float)(int)__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, kStreamAux) * scale;
  }
  struct Row {                        // see LoadRealPacked::Row
    __amdgpu_buffer_rsrc_t r;
    unsigned es;
    float scale;
    __device__ __forceinline__ float one(unsigned off) const {
      if constexpr (sizeof(Sample) == 2) return (float)(short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0u, kStreamAux) * scale;
      else return (float)(int)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0u, kStreamAux) * scale;
    }
    __device__ __forceinline__ cf finish(cf v, int) const { return v; }
    __device__ __forceinline__ cf pair_at(int s) const {
      return make_float2(one((unsigned)s * es), one((unsigned)(s + 1) * es));
    }
  };
  __device__ __forceinline__ Row open(int b) const {
    const unsigned span = len > 0 ? ((unsigned)(len - 1) * (unsigned)elem_stride + 1u) * (unsigned)sizeof(Sample) : 0u;
    return Row{make_rsrc(base + (long long)b * chan_stride, span), (unsigned)elem_stride * (unsigned)sizeof(Sample), scale};
  }
  template <int STEP, int F>
  __device__ __forceinline__ void column(int b, unsigned e0, cf (&v)[F]) const {
    const Sample* p = base + (long long)b * chan_stride;
    const unsigned es = (unsigned)elem_stride * (unsigned)sizeof(Sample);
    const __amdgpu_buffer_rsrc_t r =
        make_rsrc(p, ((unsigned)(len - 1) * (unsigned)elem_stride + 1u) * (unsigned)sizeof(Sample));
    const unsigned vo = 2u * e0 * es;
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const unsigned so = 2u * (unsigned)(j * STEP) * es;
      v[j].x = sample(r, vo, so);
      v[j].y = sample(r, vo, so + es);
    }
  }
};

struct LoadWorkspace {
  const cf* __restrict__ ws;   // [B][N1][4096]
  int n1_total;
  template <int STEP, int F>
  __device__ __forceinline__ void column(int b, unsigned e0, cf (&v)[F]) const {
    const __amdgpu_buffer_rsrc_t r = make_rsrc(ws + (long long)b * n1_total * kN2, (unsigned)n1_total * kN2 * 8u);
#pragma unroll
    for (int j = 0; j < F; ++j) v[j] = bload_cf<IMP_AUX_COLS_LD>(r, e0 * 8u, (unsigned)(j * STEP) * 8u);
  }
};

// Storers: bind(b) gives the channel's buffer; put(r, b, e, step_elems, v) writes complex point e + step_elems;
// begin(tid, threads) / end(b, tile, tiles, n1, tid, threads) bracket a workgroup's puts (used by StoreRealCropMax)
// (e per thread, step_elems a compile-time multiple of 4096).
struct StoreWorkspace {
  cf* __restrict__ ws;
  int n1_total;
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bind(int b) const {
    return make_rsrc(ws + (long long)b * n1_total * kN2, (unsigned)n1_total * kN2 * 8u);
  }
  __device__ __forceinline__ void begin(int, int) const {}
  __device__ __forceinline__ void end(int, int, int, int, int, int) const {}
  __device__ __forceinline__ void put(__amdgpu_buffer_rsrc_t r, int /*b*/, unsigned e, unsigned step_elems, cf v) const {
    bstore_cf<IMP_AUX_COLS_ST>(v, r, e * 8u, step_elems * 8u);
  }
};

// Unpack y[2n] = re, y[2n+1] = im and keep the window [start, start+len) of the linear
// convolution ('same': start = (M-1)/2, len = L; 'full': start = 0, len = L+M-1).  The crop is the
// buffer's range check: sample 2n - start as an unsigned byte offset is out of range on both sides.
struct StoreRealCrop {
  float* __restrict__ base;
  long long chan_stride;
  long long start;
  long long len;
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bind(int b) const {
    return make_rsrc(base + (long long)b * chan_stride, (unsigned)len * 4u);
  }
  __device__ __forceinline__ void begin(int, int) const {}
  __device__ __forceinline__ void end(int, int, int, int, int, int) const {}
  __device__ __forceinline__ void put(__amdgpu_buffer_rsrc_t r, int /*b*/, unsigned e, unsigned step_elems, cf v) const {
    // the whole offset goes through the VGPR: a negative voffset stays out of range whatever soffset adds
    const unsigned off = (2u * (e + step_elems) - (unsigned)start) * 4u;
    if (start & 1) {
      // odd start: one point straddles y[-1] | y[0]; two dword stores let the range check split it.
      // The second offset is made opaque so that the two are not merged back into one dwordx2,
      // which is dropped whole when its voffset is negative (measured).
      unsigned off_im = off + 4u;
      asm volatile("" : "+v"(off_im));
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), r, off, 0u, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), r, off_im, 0u, 0);
    } else {
      u32x2 x;
      x.x = __float_as_uint(v.x);
      x.y = __float_as_uint(v.y);
      __builtin_amdgcn_raw_buffer_store_b64(x, r, off, 0u, kStreamAux);
    }
  }
};

// Overlap-add: the piece's full convolution is ADDED into the window [start, start+len) of the long result
// (start may be negative: the piece then begins inside the window).  Pieces of one output run in stream
// order, so the read-modify-write needs no atomics; the range check clips as in StoreRealCrop.
struct StoreRealCropAdd {
  float* __restrict__ base;
  long long chan_stride;
  long long start;
  long long len;
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bind(int b) const {
    return make_rsrc(base + (long long)b * chan_stride, (unsigned)len * 4u);
  }
  __device__ __forceinline__ void begin(int, int) const {}
  __device__ __forceinline__ void end(int, int, int, int, int, int) const {}
  __device__ __forceinline__ void put(__amdgpu_buffer_rsrc_t r, int /*b*/, unsigned e, unsigned step_elems, cf v) const {
    const unsigned off = (2u * (e + step_elems) - (unsigned)start) * 4u;
    unsigned off_im = off + 4u;
    asm volatile("" : "+v"(off_im));            // two dword accesses: the straddling point splits at either edge
    const float re = bload_f(r, off, 0u), im = bload_f(r, off_im, 0u);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(re + v.x), r, off, 0u, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(im + v.y), r, off_im, 0u, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// Pair mode: TWO real channels travel as ONE complex signal z[n] = x_L[n] + i x_R[n].  The filter is real, so
// (x_L + i x_R) (*) h = y_L + i y_R: no even/odd packing, no real-FFT unpack - the row pass is a pointwise product
// with H - and a stereo frame of the recording (core/hrir.py:326-341: track i = left ear, i + 1 = right ear) is one
// 8-byte load.  Channel pair p = channels (2p, 2p + 1) of the launch group; an odd last channel pairs with silence.
// A column thread's point e is SAMPLE e of both channels (not samples 2e, 2e + 1 of one).
// ---------------------------------------------------------------------------------------------
// swap with the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float lane_swap1(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
}

template <class Sample> struct PairSample;
template <> struct PairSample<float> {
  static __device__ __forceinline__ float get(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, kStreamAux));
  }
  static __device__ __forceinline__ cf get2(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float) {
    const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, kStreamAux);
    return make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
  }
};
template <> struct PairSample<int> {
  static __device__ __forceinline__ float get(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float scale) {
    return (float)(int)__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, kStreamAux) * scale;
  }
  static __device__ __forceinline__ cf get2(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float scale) {
    const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, kStreamAux);
    return make_float2((float)(int)x.x * scale, (float)(int)x.y * scale);
  }
};
template <> struct PairSample<short> {
  static __device__ __forceinline__ float get(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float scale) {
    return (float)(short)__builtin_amdgcn_raw_buffer_load_b16(r, vo, so, kStreamAux) * scale;
  }
  static __device__ __forceinline__ cf get2(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float scale) {
    const unsigned x = __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, kStreamAux);      // one stereo frame of 16-bit PCM
    return make_float2((float)(short)(x & 0xFFFFu) * scale, (float)(short)(x >> 16) * scale);
  }
};

// Sample = float (scale unused), int (PCM_32, scale 2^-31) or short (PCM_16, scale 2^-15).  Sample i of pair q:
//   left  at base[q * pair_stride + i * elem_stride],   right at base[q * pair_stride + right_off + i * elem_stride]
// planar rows [B][pitch]: pair_stride = 2 pitch, right_off = pitch, elem_stride = 1; the columns of a binaural WAV
// (frames [n][2], column q starting at frame s_q = s_0 + q step): pair_stride = 2 step, right_off = 1, elem_stride = 2 -
// a stereo frame is then ONE load.  An odd channel count leaves the last pair without a right channel (silence).
template <class Sample>
struct LoadPair {
  const Sample* __restrict__ base;
  long long pair_stride;
  long long right_off;
  long long elem_stride;
  long long len;
  int nchan;                        // channels in the launch group (pairs = (nchan + 1) / 2)
  float scale;
  __host__ __device__ LoadPair shifted(long long first, long long maxlen) const {
    const long long rest = len - first;
    return LoadPair{base + first * elem_stride, pair_stride, right_off, elem_stride, rest < maxlen ? rest : maxlen, nchan, scale};
  }
  template <int STEP, int F>
  __device__ __forceinline__ void column(int p, unsigned e0, cf (&v)[F]) const {
    const Sample* pl = base + (long long)p * pair_stride;
    const bool has_r = 2 * p + 1 < nchan;                                   // wave-uniform
    const unsigned es = (unsigned)elem_stride * (unsigned)sizeof(Sample);   // bytes between samples
    const unsigned span = ((unsigned)(len - 1) * (unsigned)elem_stride + 1u) * (unsigned)sizeof(Sample);
    const unsigned vo = e0 * es;
    // whole frames only where they are naturally aligned (odd track counts / odd first tracks take the two-load path)
    const bool frames = right_off == 1 && has_r && ((unsigned long long)pl % (2 * sizeof(Sample))) == 0 &&
                        es % (2u * (unsigned)sizeof(Sample)) == 0;
    if (frames) {
      const __amdgpu_buffer_rsrc_t r = make_rsrc(pl, len > 0 ? span + (unsigned)sizeof(Sample) : 0u);
#pragma unroll
      for (int j = 0; j < F; ++j) v[j] = PairSample<Sample>::get2(r, vo, (unsigned)(j * STEP) * es, scale);
    } else if (elem_stride == 1 && has_r && right_off > 0 && ((unsigned long long)pl % (2 * sizeof(Sample))) == 0 &&
               (right_off & 1) == 0 && (unsigned long long)(right_off + len) * sizeof(Sample) < 0xFFFFFFFFull) {
      // planar rows: lanes are consecutive samples, so the even lane of a lane pair fetches x_L[n], x_L[n + 1] and the
      // odd one x_R[n - 1], x_R[n] as ONE aligned 8-byte load each (both rows through one buffer), then they swap a word
      const bool odd = (threadIdx.x & 1) != 0;                             // == e0 & 1: tiles start at even columns
      const __amdgpu_buffer_rsrc_t r = make_rsrc(pl, (unsigned)(right_off + len) * (unsigned)sizeof(Sample));
      const unsigned n0 = e0 & ~1u;
      const unsigned vo2 = (n0 + (odd ? (unsigned)right_off : 0u)) * (unsigned)sizeof(Sample);
      const unsigned ulen = (unsigned)len;
#pragma unroll
      for (int j = 0; j < F; ++j) {
        cf x = PairSample<Sample>::get2(r, vo2, (unsigned)(j * STEP) * (unsigned)sizeof(Sample), scale);
        const unsigned i0 = n0 + (unsigned)(j * STEP);
        x.x = i0 < ulen ? x.x : 0.f;                                       // the left row's padding is the right row's head
        x.y = i0 + 1u < ulen ? x.y : 0.f;
        const float got = lane_swap1(odd ? x.x : x.y);                     // even lanes receive x_R[n], odd lanes x_L[n]
        v[j] = odd ? make_float2(got, x.y) : make_float2(x.x, got);
      }
    } else {
      const __amdgpu_buffer_rsrc_t rl = make_rsrc(pl, len > 0 ? span : 0u);
      const __amdgpu_buffer_rsrc_t rr = make_rsrc(pl + right_off, (has_r && len > 0) ? span : 0u);
#pragma unroll
      for (int j = 0; j < F; ++j) {
        v[j].x = PairSample<Sample>::get(rl, vo, (unsigned)(j * STEP) * es, scale);
        v[j].y = PairSample<Sample>::get(rr, vo, (unsigned)(j * STEP) * es, scale);     // absent channel: empty range -> 0
      }
    }
  }
};

// y_L[n - start] = re, y_R[n - start] = im, window [start, start + len) of the linear convolution.  Lanes are consecutive
// samples, so two neighbouring lanes hold (re, im) of samples n, n + 1 (n even): they swap one word each and the even lane
// writes y_L[n], y_L[n + 1], the odd lane y_R[n], y_R[n + 1] - ONE 8-byte store per lane and point, 256 contiguous bytes per
// channel and wave instruction (two 4-byte stores per lane: pass C of C2 34 us instead of 26 us for the mono plan).  Both
// channels go through one buffer (base = the left row, the right row chan_stride further), so the crop is an explicit
// range test here, not the hardware's; the samples at the window's two edges take 4-byte stores.
struct StorePairCrop {
  float* __restrict__ base;
  long long chan_stride;
  long long start;
  long long len;
  int nchan;
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bind(int p) const {
    const bool has_r = 2 * p + 1 < nchan;
    return make_rsrc(base + (long long)(2 * p) * chan_stride, (unsigned)((has_r ? chan_stride : 0) + len) * 4u);
  }
  __device__ __forceinline__ void begin(int, int) const {}
  __device__ __forceinline__ void end(int, int, int, int, int, int) const {}
  __device__ __forceinline__ void put(__amdgpu_buffer_rsrc_t r, int p, unsigned e, unsigned step_elems, cf v) const {
    const unsigned n = e + step_elems;
#ifdef IMP_EXPERIMENT_FRAMES_OUT
    // timing experiment only (wrong layout for the callers): the pair's two rows written as one block of stereo frames
    if (n - (unsigned)start < (unsigned)len) bstore_cf<kStreamAux>(v, r, (n - (unsigned)start) * 8u, 0u);
    return;
#endif
    const bool odd = (threadIdx.x & 1) != 0;                   // == n & 1: tiles start at even columns
    const float got = lane_swap1(odd ? v.x : v.y);             // even lanes receive re[n + 1], odd lanes im[n - 1]
    const float w0 = odd ? got : v.x, w1 = odd ? v.y : got;    // the lane's channel at samples n0, n0 + 1
    const unsigned i0 = (n & ~1u) - (unsigned)start;           // below the window: wraps out of range
    const unsigned ulen = (unsigned)len;
    const bool ok0 = i0 < ulen, ok1 = i0 + 1u < ulen;
    const bool chan_ok = !odd || 2 * p + 1 < nchan;
    const unsigned off = i0 * 4u + (odd ? (unsigned)chan_stride * 4u : 0u);
    if (ok0 && ok1 && chan_ok) {
      u32x2 x;
      x.x = __float_as_uint(w0);
      x.y = __float_as_uint(w1);
      __builtin_amdgcn_raw_buffer_store_b64(x, r, off, 0u, kStreamAux);
    } else if (chan_ok) {
      if (ok0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w0), r, off, 0u, 0);
      if (ok1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w1), r, off + 4u, 0u, 0);
    }
  }
};

// overlap-add of 'full' pieces in pair mode (StoreRealCropAdd)
struct StorePairCropAdd {
  float* __restrict__ base;
  long long chan_stride;
  long long start;
  long long len;
  int nchan;
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bind(int p) const {
    return make_rsrc(base + (long long)(2 * p) * chan_stride, (unsigned)len * 4u);
  }
  __device__ __forceinline__ void begin(int, int) const {}
  __device__ __forceinline__ void end(int, int, int, int, int, int) const {}
  __device__ __forceinline__ void put(__amdgpu_buffer_rsrc_t r, int p, unsigned e, unsigned step_elems, cf v) const {
    const __amdgpu_buffer_rsrc_t rr =
        make_rsrc(base + (long long)(2 * p + 1) * chan_stride, (2 * p + 1 < nchan) ? (unsigned)len * 4u : 0u);
    const unsigned off = (e + step_elems - (unsigned)start) * 4u;
    const float yl = bload_f(r, off, 0u), yr = bload_f(rr, off, 0u);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yl + v.x), r, off, 0u, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yr + v.y), rr, off, 0u, 0);
  }
};

// wave-uniform twiddle (SGPR pair): one VOP3P may read one scalar pair, so the multiply needs no copy
template <int DIR>
__device__ __forceinline__ cf ctw_uniform(cf a, cf w) {
  v2f t, r;
  if constexpr (DIR < 0) {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(to_v(a)), "s"(to_v(w)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=v"(r) : "v"(to_v(a)), "s"(to_v(w)), "v"(t));
  } else {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(to_v(a)), "s"(to_v(w)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]"
        : "=v"(r) : "v"(to_v(a)), "s"(to_v(w)), "v"(t));
  }
  return to_c(r);
}

// ---------------------------------------------------------------------------------------------
// Column pass.  Tile = TC columns x N1 rows, one thread = 16 rows of one column.
//   thread (g, c): g = tid / TC in [0, R2), c = tid % TC; rows i = g + R2*j, j = 0..15
//   FFT16 over j -> a ; x w_N1^(g a) ; LDS exchange ; FFT_R2 over g -> b ; output row = a + 16 b
// DIR < 0: forward, four-step twiddle applied to the outputs (pass A).
// DIR > 0: inverse, conj four-step twiddle applied to the inputs (pass C).
// LDS plane index = a*T + tid with the column as the lane index: conflict-free both ways.
// ---------------------------------------------------------------------------------------------
template <int R2>
struct ColsCfg {
  static constexpr int TC = (R2 >= 16) ? 32 : 64;
  static constexpr int T = TC * R2;
  static constexpr int G = 16 / R2;            // groups of R2 per thread in the 2nd stage
  static constexpr size_t lds_bytes = (R2 > 1) ? sizeof(cf) * 16 * T : 0;
};

// first stage shared by both column kernels: fetch F rows (i = g + R2 j), (inverse: conj four-step twiddle,)
// F-point FFT over j -> a, x w_N1^(g a)
template <int F, int R2, int TC, int DIR, class Load>
__device__ __forceinline__ void cols_first_stage(const Load& ld, const Twiddles& tw, __amdgpu_buffer_rsrc_t r_full,
                                                 int b, int g, unsigned e0, cf (&v)[F]) {
  ld.template column<R2 * kN2, F>(b, e0, v);
  if constexpr (DIR > 0) {
#pragma unroll
    for (int j = 0; j < F; ++j) v[j] = cmulc(v[j], bload_cf(r_full, e0 * 8u, (unsigned)(j * R2 * kN2) * 8u));
  }
  fft_first<DIR, F>(v);   // index a
  if constexpr (R2 > 1) {
    // w_N1^(g a) = hi[4 g a]  (Nc/1024 = 4 N1 entries); g is wave-uniform for 64-column tiles
    if constexpr (TC == 64) {
      const int gu = __builtin_amdgcn_readfirstlane(g);
#pragma unroll
      for (int a = 1; a < F; ++a) v[a] = ctw_uniform<DIR>(v[a], tw.hi[4 * gu * a]);
    } else {
#pragma unroll
      for (int a = 1; a < F; ++a) v[a] = ctw<DIR>(v[a], tw.hi[4 * g * a]);
    }
  }
}

template <int R2, int DIR, class Load, class Store>
__global__ __launch_bounds__(ColsCfg<R2>::T) void cols_kernel(Load ld, Store st, Twiddles tw, int nchan, int n1_total) {
  using Cfg = ColsCfg<R2>;
  constexpr int TC = Cfg::TC, T = Cfg::T, G = Cfg::G;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cf* buf = reinterpret_cast<cf*>(smem_raw);

  const int tid = threadIdx.x;
  const int c = tid % TC;
  const int g = tid / TC;
  int b, tile;
  xcd_work_item(nchan, tile, b);
  const unsigned n2 = (unsigned)(tile * TC + c);
  const __amdgpu_buffer_rsrc_t r_full = make_rsrc(tw.full, (unsigned)n1_total * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_out = st.bind(b);
  st.begin(tid, T);

  constexpr int GG = (R2 > 1) ? G : 16, KB = (R2 > 1) ? R2 : 1;
  // pass A: four-step twiddles of this thread's outputs, fetched with the inputs (see cols_mixed_kernel)
  cf twd[GG][KB];
  if constexpr (DIR < 0) {
#pragma unroll
    for (int i = 0; i < GG; ++i)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        twd[i][kb] = bload_cf(r_full, ((unsigned)((R2 > 1 ? g * G : 0) + i) * kN2 + n2) * 8u, (unsigned)(kb * 16 * kN2) * 8u);
  }
  cf v[16];
  cols_first_stage<16, R2, TC, DIR>(ld, tw, r_full, b, g, (unsigned)g * kN2 + n2, v);

  if constexpr (R2 > 1) {
#pragma unroll
    for (int a = 0; a < 16; ++a) buf[a * T + tid] = v[a];
    __syncthreads();
    // thread (q = g, c) now owns a in {q*G .. q*G+G-1}, all g'
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int gp = 0; gp < R2; ++gp) v[i * R2 + gp] = buf[(g * G + i) * T + gp * TC + c];
    fft_groups<DIR, R2>(v);
  }
#pragma unroll
  for (int i = 0; i < GG; ++i) {
    const unsigned e = (unsigned)((R2 > 1 ? g * G : 0) + i) * kN2 + n2;      // row a = g*G + i (R2 = 1: a = i)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      cf z = v[i * KB + kb];
      if constexpr (DIR < 0) z = cmul(z, twd[i][kb]);
      st.put(r_out, b, e, (unsigned)(kb * 16 * kN2), z);
    }
  }
  st.end(b, tile, kN2 / TC, n1_total, tid, T);
}

// ---------------------------------------------------------------------------------------------
// Column pass for N1 = F*R2 with an odd factor: F = 16 rows per thread and R2 in {3, 5, 6, 9, 10, 12}, or
// F = 8 and R2 in {3, 5, 9} (N1 = 24, 40, 72).  Same first stage as cols_kernel (F-point FFT over j); the
// second stage is F DFTs of R2 points per column, dealt to the R2 threads of the column as
// ka = g, g + R2, ... (< F), so a thread holds up to G = ceil(F/R2) butterflies.  These sizes exist
// because the circular length only has to cover L + ceil((M-1)/2) in 'same' mode: 72 rows of 4096
// complex points (589 824 samples) instead of 128 for the 7.1 / 6.15 s case, and every row not
// transformed is 32 KiB less workspace traffic in each of the four trips.
// ---------------------------------------------------------------------------------------------
template <int F, int R2>
struct MixCfg {
  // 64 columns = 512-byte row segments; F = 16 with R2 = 12 would need 96 KiB of LDS (one workgroup per CU),
  // so it takes 32-column tiles (48 KiB, three per CU): C5 +5 %.  Narrower tiles everywhere were slower
  // (256-byte row segments: C2 327 k -> 300 k IR/s, C3 -6 %).
  static constexpr int TC = (F * R2 >= 192) ? 32 : 64;
  static constexpr int T = TC * R2;
  static constexpr int G = (F + R2 - 1) / R2;
  static constexpr size_t lds_bytes = sizeof(cf) * F * T;
  // the 12-wave workgroups of 11 x 12 (pair mode's 132 rows) fit a CU twice only below 81 VGPRs, the 9-wave ones of
  // 16 x 18 (288 rows) below 97: ask for the waves per SIMD that two workgroups need (HIP's second launch-bounds figure is
  // waves per execution unit)
  static constexpr int min_waves = (T >= 512 && lds_bytes <= 80 * 1024) ? (2 * (T / 64) + 3) / 4 : 1;
};

template <int F, int R2, int DIR, class Load, class Store>
__global__ __launch_bounds__((MixCfg<F, R2>::T), (MixCfg<F, R2>::min_waves)) void cols_mixed_kernel(Load ld, Store st, Twiddles tw, int nchan,
                                                                       int n1_total) {
  using Cfg = MixCfg<F, R2>;
  constexpr int TC = Cfg::TC, T = Cfg::T, G = Cfg::G;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cf* buf = reinterpret_cast<cf*>(smem_raw);

  const int tid = threadIdx.x;
  const int c = tid % TC;
  const int g = tid / TC;
  int b, tile;
  xcd_work_item(nchan, tile, b);
  const unsigned n2 = (unsigned)(tile * TC + c);
  const __amdgpu_buffer_rsrc_t r_full = make_rsrc(tw.full, (unsigned)n1_total * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_out = st.bind(b);
  st.begin(tid, T);

  // pass A: the four-step twiddles of this thread's outputs are fetched with the inputs, not after the
  // exchange where their latency would sit in front of the stores
  // (the 12-wave workgroups of 11 x 12 have to stay below 81 VGPRs - MixCfg::min_waves - and fetch them between the
  // exchange's writes and its barrier instead, when the first stage's registers are free again)
  constexpr bool kLateTwiddles = Cfg::min_waves > 1;
  cf twd[G][R2];
  auto fetch_twiddles = [&]() {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int ka = g + R2 * i;
      if (ka < F) {
#pragma unroll
        for (int kb = 0; kb < R2; ++kb)
          twd[i][kb] = bload_cf(r_full, ((unsigned)ka * kN2 + n2) * 8u, (unsigned)(kb * F * kN2) * 8u);
      }
    }
  };
  if constexpr (DIR < 0 && !kLateTwiddles) fetch_twiddles();
  cf v[F];
  cols_first_stage<F, R2, TC, DIR>(ld, tw, r_full, b, g, (unsigned)g * kN2 + n2, v);
#pragma unroll
  for (int a = 0; a < F; ++a) buf[a * T + tid] = v[a];
  if constexpr (DIR < 0 && kLateTwiddles) fetch_twiddles();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int ka = g + R2 * i;
    if (ka < F) {
      cf y[R2];
#pragma unroll
      for (int gp = 0; gp < R2; ++gp) y[gp] = buf[ka * T + gp * TC + c];
      fft_small<DIR, R2>(y);
      const unsigned e = (unsigned)ka * kN2 + n2;
#pragma unroll
      for (int kb = 0; kb < R2; ++kb) {                    // output row k1 = ka + F kb
        cf z = y[kb];
        if constexpr (DIR < 0) z = cmul(z, twd[i][kb]);
        st.put(r_out, b, e, (unsigned)(kb * F * kN2), z);
      }
    }
  }
  st.end(b, tile, kN2 / TC, n1_total, tid, T);
}

// ---------------------------------------------------------------------------------------------
// Column pass for the short transforms N1 = F in {4, 8} (32 768 / 65 536 real points): K5's everyday size - a 0.7 s
// response (*) a 9 600-tap equalisation FIR is 43 k samples, a third of the smallest 16-row plan.  One thread owns
// the F rows of one column; no second stage, no LDS.  256 columns per workgroup.
// ---------------------------------------------------------------------------------------------
template <int F, int DIR, class Load, class Store>
__global__ __launch_bounds__(256) void cols_small_kernel(Load ld, Store st, Twiddles tw, int nchan, int n1_total) {
  static_assert(F == 4 || F == 8, "short column passes hold 4 or 8 rows per thread");
  int b, tile;
  xcd_work_item(nchan, tile, b);
  const unsigned n2 = (unsigned)(tile * 256 + threadIdx.x);
  const __amdgpu_buffer_rsrc_t r_full = make_rsrc(tw.full, (unsigned)n1_total * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_out = st.bind(b);
  st.begin((int)threadIdx.x, 256);
  cf twd[F];
#pragma unroll
  for (int k = 0; k < F; ++k) twd[k] = bload_cf(r_full, n2 * 8u, (unsigned)(k * kN2) * 8u);
  cf v[F];
  ld.template column<kN2, F>(b, n2, v);
  if constexpr (DIR > 0) {
#pragma unroll
    for (int k = 0; k < F; ++k) v[k] = cmulc(v[k], twd[k]);
  }
  if constexpr (F == 8) fft8<DIR>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  else bfly4<DIR>(v[0], v[1], v[2], v[3]);
#pragma unroll
  for (int k = 0; k < F; ++k) {
    cf z = v[k];
    if constexpr (DIR < 0) z = cmul(z, twd[k]);
    st.put(r_out, b, n2, (unsigned)(k * kN2), z);
  }
  st.end(b, tile, kN2 / 256, n1_total, (int)threadIdx.x, 256);
}

// ---------------------------------------------------------------------------------------------
// Row pass (pass B).  One workgroup = 512 threads = two rows (k1, N1-k1), or the two
// self-paired rows (0, N1/2) when pair == 0.  Everything between the load and the store of a
// row happens in registers + 68 KiB of LDS:
//   fwd FFT4096 = [FFT16, x w4096^(t a), X1, FFT16, x w256^(t2 b), X2, FFT16]
//   middle      = W[k] = alpha[k] Z[k] + beta[k] conj(Z[Nc-k])   (partner read through LDS)
//   inv FFT4096 = mirror image with conjugate twiddles
// Register layout after the forward FFT: thread u = 16*ka + kb1 holds k2 = ka + 16 kb1 + 256 kb2
// in v[kb2]; the plan stores alpha/beta in exactly that order: ab[k1][kb2*256 + u].
// ---------------------------------------------------------------------------------------------
struct RowsArgs {
  cf* __restrict__ ws;             // [B][N1][4096], in place
  const float4* __restrict__ ab;   // [HB][N1][4096] (alpha.x, alpha.y, beta.x, beta.y)
  long long ab_chan_stride;        // 0: one spectrum shared by all channels; N1*4096: per channel
  int n1_total;                    // N1 (even; need not be a power of two)
  int npairs;                      // N1/2: pair 0 = rows (0, N1/2), pair p = rows (p, N1-p)
  int nchan;                       // channels in this launch group
};

// W = alpha z + beta conj(zp) in four packed instructions (alpha = ab.xy, beta = ab.zw)
__device__ __forceinline__ cf filter_bin(float4 ab, cf z, cf zp) {
  const v2f al = {ab.x, ab.y}, be = {ab.z, ab.w};
  v2f t = to_v(cmul(z, to_c(al))), w;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(to_v(zp)), "v"(be), "v"(t));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
      : "=v"(w) : "v"(to_v(zp)), "v"(be), "v"(t));
  return to_c(w);
}

// Orders this wave's LDS traffic (all earlier DS ops retired, nothing moved across by the
// compiler) without stalling the other waves of the workgroup.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef IMP_PHASE_TRACE
// Diagnostic build only (tools/phase_trace.py): per-workgroup phase timestamps of wave 0.
__device__ unsigned long long g_phase_trace[8192 * 16];
#define IMP_MARK(i)                                                                       \
  do {                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_phase_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define IMP_MARK_MEM(i)                                   \
  do {                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
    IMP_MARK(i);                                          \
  } while (0)
#define IMP_MARK_WALL(i)                                                                  \
  do {                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_phase_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define IMP_MARK(i)
#define IMP_MARK_MEM(i)
#define IMP_MARK_WALL(i)
#endif

// The row pass between its loads and its stores: v[j] = row[t + 256 j] in, the filtered, inverse-transformed row out, in
// the same registers.  `half` (which row of the pair), `pair` and k1 are wave-uniform; lds = the pair's two 34 KiB planes;
// r_ab = the alpha/beta row of (channel, k1).  Used by rows_pair (rows in the workspace) and by fir_block_kernel (rows that
// never leave the CU).
template <int AB_AUX, int AB_PREFETCH = kAbPrefetch>
__device__ __forceinline__ void rows_core(cf (&v)[16], cf* lds, const int half, const int pair, const int k1,
                                          const __amdgpu_buffer_rsrc_t r_ab, const Twiddles& tw, const int t) {
  cf* buf = lds + half * (16 * kRowPad);
  const __amdgpu_buffer_rsrc_t r_t1 = make_rsrc(tw.t1, 16 * 256 * sizeof(cf));
  const __amdgpu_buffer_rsrc_t r_t2 = make_rsrc(tw.t2, 16 * 16 * sizeof(cf));
  const __amdgpu_buffer_rsrc_t r_t4 = make_rsrc(tw.t4, 16 * 256 * sizeof(cf));
  const int hi4 = t >> 4;   // "ka" of the (ka, x) thread naming
  const int lo4 = t & 15;
  const unsigned vo8 = (unsigned)t * 8u;       // byte offset of element t in a [..][256] cf table / the row
  const unsigned vo16 = (unsigned)t * 16u;     // same for float4
  const unsigned vl8 = (unsigned)lo4 * 8u;

  // Every table read below is issued one phase ahead of its use, into whichever of the two register
  // arrays is idle at that point, so its latency hides behind the butterflies / the LDS exchange
  // (phase trace: the four twiddle fetches were ~1 us of exposed latency each).
  cf u[16];
#pragma unroll
  for (int a = 1; a < 16; ++a) u[a] = bload_cf(r_t1, vo8, a * 256 * 8);      // stage-1 twiddles
  IMP_MARK_MEM(1);
  // ---- forward FFT4096 ----
  fft16<-1>(v);                                            // over j -> a
#pragma unroll
  for (int a = 1; a < 16; ++a) v[a] = cmul(v[a], u[a]);
#pragma unroll
  for (int a = 0; a < 16; ++a) buf[a * kRowPad + t] = v[a];
#pragma unroll
  for (int q = 1; q < 16; ++q) v[q] = bload_cf(r_t2, vl8, q * 16 * 8);       // stage-2 twiddles
  __syncthreads();
  IMP_MARK(2);
  // thread (ka = hi4, t2 = lo4) gathers j2 = 0..15 (t = 16 j2 + t2)
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) u[j2] = buf[hi4 * kRowPad + 16 * j2 + lo4];
  fft16<-1>(u);                                            // over j2 -> kb1
#pragma unroll
  for (int q = 1; q < 16; ++q) u[q] = cmul(u[q], v[q]);
  // The X2 exchange stays inside one 16-lane group (same ka): both the plane-row it overwrites
  // (read just above by the same 16 lanes) and the values it reads back are private to that group,
  // which lives in one wave.  LDS executes a wave's DS ops in order, so a wave-level scheduling
  // barrier replaces the workgroup barrier here (4 of the 9 barriers of this kernel).
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < 16; ++q) buf[hi4 * kRowPad + lo4 * 17 + q] = u[q];
  wave_lds_fence();
  // thread (ka = hi4, kb1 = lo4) gathers t2 = 0..15
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) v[t2] = buf[hi4 * kRowPad + t2 * 17 + lo4];
  float4 ab_pre[AB_PREFETCH];                              // first half of this thread's alpha/beta, a phase early
#pragma unroll
  for (int q = 0; q < AB_PREFETCH; ++q) ab_pre[q] = bload_f4<AB_AUX>(r_ab, vo16, q * 256 * 16);
  fft16<-1>(v);                                            // over t2 -> kb2
  __syncthreads();
  IMP_MARK(3);

  // ---- partner exchange: plane [kb2][17 ka + kb1] (pitch 17 keeps the mirrored read conflict-free) ----
#pragma unroll
  for (int q = 0; q < 16; ++q) buf[q * kRowPad + t + hi4] = v[q];
  __syncthreads();

  // partner bins conj Z[Nc-k] (u[] is free here).  With k = k1 + N1 k2, k2 = ka + 16 kb1 + 256 kb2:
  //   k1 != 0: Nc - k = (N1 - k1) + N1 (4095 - k2)  -> the other row of the pair, column 4095 - k2,
  //            i.e. (15 - ka, 15 - kb1, 15 - kb2): one per-thread base + compile-time plane offsets
  //   k1 == 0: Nc - k = N1 (4096 - k2) mod Nc       -> row 0 itself, column (4096 - k2) mod 4096
  // In the self-paired workgroup (rows 0 and N1/2) the partner row is the thread's own row.
  const int phalf = (pair == 0) ? half : 1 - half;
  const cf* pbuf = lds + phalf * (16 * kRowPad);
  if (k1 != 0) {                                           // wave-uniform
    const cf* pm = pbuf + 17 * (15 - hi4) + (15 - lo4);
#pragma unroll
    for (int q = 0; q < 16; ++q) u[q] = pm[(15 - q) * kRowPad];
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const unsigned k2 = (unsigned)hi4 + 16u * (unsigned)lo4 + 256u * (unsigned)q;
      const unsigned pk2 = (4096u - k2) & 4095u;
      u[q] = pbuf[(pk2 >> 8) * kRowPad + 17 * (pk2 & 15u) + ((pk2 >> 4) & 15u)];
    }
  }
  IMP_MARK(4);
  // bin 0 of the packed transform carries DC and Nyquist: ab.x = H[0]/Nc, ab.z = H[Nc]/Nc
  const bool dc_lane = (k1 == 0) && (t == 0);
  cf w_dc = make_float2(0.f, 0.f);
  if (dc_lane) {
    const float4 ab = bload_f4<AB_AUX>(r_ab, 0u, 0u);
    const float x0 = v[0].x + v[0].y, xn = v[0].x - v[0].y;
    w_dc = make_float2(0.5f * (x0 * ab.x + xn * ab.z), 0.5f * (x0 * ab.x - xn * ab.z));
  }
  // W = alpha Z + beta conj(Z[Nc-k]), in place
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    v[q] = filter_bin(q < AB_PREFETCH ? ab_pre[q] : bload_f4<AB_AUX>(r_ab, vo16, q * 256 * 16), v[q], u[q]);
  }
  if (dc_lane) v[0] = w_dc;
  IMP_MARK_MEM(5);
  __syncthreads();
  IMP_MARK(6);

  // ---- inverse FFT4096 (mirror) ----
#pragma unroll
  for (int q = 1; q < 16; ++q) u[q] = bload_cf(r_t2, vl8, q * 16 * 8);       // u[] is free after the multiply
  fft16<+1>(v);                                            // over kb2 -> t2
#pragma unroll
  for (int q = 1; q < 16; ++q) v[q] = cmulc(v[q], u[q]);
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) buf[hi4 * kRowPad + t2 * 17 + lo4] = v[t2];
  wave_lds_fence();                                        // intra-group exchange, see above
  // thread (ka = hi4, t2 = lo4) gathers kb1 = 0..15
#pragma unroll
  for (int q = 0; q < 16; ++q) u[q] = buf[hi4 * kRowPad + lo4 * 17 + q];
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) v[j2] = bload_cf(r_t4, vo8, j2 * 256 * 8);  // v[] is free: inverse 2nd twiddle
  fft16<+1>(u);                                            // over kb1 -> j2
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) u[j2] = cmulc(u[j2], v[j2]);
  wave_lds_fence();                                        // X1' rows are written by the group that read X2'
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) buf[hi4 * kRowPad + 16 * j2 + lo4] = u[j2];
  __syncthreads();
  IMP_MARK(7);
#pragma unroll
  for (int a = 0; a < 16; ++a) v[a] = buf[a * kRowPad + t];
  fft16<+1>(v);                                            // over ka -> j
  IMP_MARK(8);
}

// One row pair: workspace slot ws_b, spectrum of channel `chan`; 512 threads, 68 KiB of LDS at `lds`.  WS_AUX / AB_AUX: cache policy of
// the workspace loads and of the alpha/beta loads (the launches use the defaults).
template <int WS_AUX, int AB_AUX>
__device__ __forceinline__ void rows_pair(const RowsArgs& args, const Twiddles& tw, int ws_b, int chan, int pair, cf* lds,
                                          const int tid) {
  IMP_MARK_WALL(12);
  IMP_MARK(0);
  // a wave never straddles the two rows: everything derived from `half` is wave-uniform (SGPRs)
  const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int t = tid & 255;
  const int N1 = args.n1_total;
  const int rowA = pair;
  const int rowB = (pair == 0) ? (N1 >> 1) : (N1 - pair);
  const int k1 = half ? rowB : rowA;
  const __amdgpu_buffer_rsrc_t r_row = make_rsrc(args.ws + ((long long)ws_b * N1 + k1) * kN2, kN2 * sizeof(cf));
  const __amdgpu_buffer_rsrc_t r_ab =
      make_rsrc(args.ab + (long long)chan * args.ab_chan_stride + (long long)k1 * kN2, kN2 * sizeof(float4));
  const unsigned vo8 = (unsigned)t * 8u;
  cf v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = bload_cf<WS_AUX>(r_row, vo8, j * 256 * 8);
  rows_core<AB_AUX>(v, lds, half, pair, k1, r_ab, tw, t);
#pragma unroll
  for (int j = 0; j < 16; ++j) bstore_cf<IMP_AUX_ROWS_ST>(v[j], r_row, vo8, j * 256 * 8);
  IMP_MARK_MEM(9);
  IMP_MARK_WALL(13);
#ifdef IMP_PHASE_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_phase_trace[blockIdx.x * 16 + 14] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
#endif
}

__global__ __launch_bounds__(512, 4) void rows_kernel(RowsArgs args, Twiddles tw) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int b, pair;
  xcd_work_item(args.nchan, pair, b);
  if (pair >= args.npairs) return;              // the grid is padded to a multiple of 8 pairs (whole workgroup exits)
  rows_pair<IMP_AUX_ROWS_LD, 0>(args, tw, b, b, pair, reinterpret_cast<cf*>(smem_raw), (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Row pass of pair mode.  One workgroup = 256 threads = ONE row k1 of a channel pair's transform:
//   fwd FFT4096 -> W[k] = H[k] Z[k] (H = spectrum of the real filter / Nc, in this kernel's register order) -> inv FFT4096
// No partner row, no alpha/beta: 8 bytes of spectrum per bin instead of 16, half the LDS (34 KiB: four workgroups per
// CU instead of two) and two workgroup barriers instead of five - between the X1 exchanges every LDS word a 16-lane
// group touches is private to it.
// ---------------------------------------------------------------------------------------------
struct RowsPairArgs {
  cf* __restrict__ ws;             // [pairs][N1][4096], in place
  const cf* __restrict__ hs;       // [N1][4096]: H[k1 + N1 k2] / Nc at [k1][kb2*256 + u], k2 = (u>>4) + 16 (u&15) + 256 kb2
  int n1_total;
  int npairs;                      // channel pairs in this launch group
};

__global__ __launch_bounds__(256, 4) void rows_single_kernel(RowsPairArgs args, Twiddles tw) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cf* buf = reinterpret_cast<cf*>(smem_raw);
  int b, k1;
  xcd_work_item(args.npairs, k1, b);
  const int N1 = args.n1_total;
  if (k1 >= N1) return;                         // the grid is padded to a multiple of 8 rows (whole workgroup exits)
  const int t = threadIdx.x;

  const __amdgpu_buffer_rsrc_t r_row = make_rsrc(args.ws + ((long long)b * N1 + k1) * kN2, kN2 * sizeof(cf));
  const __amdgpu_buffer_rsrc_t r_h = make_rsrc(args.hs + (long long)k1 * kN2, kN2 * sizeof(cf));
  const __amdgpu_buffer_rsrc_t r_t1 = make_rsrc(tw.t1, 16 * 256 * sizeof(cf));
  const __amdgpu_buffer_rsrc_t r_t2 = make_rsrc(tw.t2, 16 * 16 * sizeof(cf));
  const __amdgpu_buffer_rsrc_t r_t4 = make_rsrc(tw.t4, 16 * 256 * sizeof(cf));

  const int hi4 = t >> 4, lo4 = t & 15;
  const unsigned vo8 = (unsigned)t * 8u, vl8 = (unsigned)lo4 * 8u;

  cf v[16], u[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = bload_cf<IMP_AUX_ROWS_LD>(r_row, vo8, j * 256 * 8);
#pragma unroll
  for (int a = 1; a < 16; ++a) u[a] = bload_cf(r_t1, vo8, a * 256 * 8);      // stage-1 twiddles

  // ---- forward FFT4096 (as rows_pair) ----
  fft16<-1>(v);
#pragma unroll
  for (int a = 1; a < 16; ++a) v[a] = cmul(v[a], u[a]);
#pragma unroll
  for (int a = 0; a < 16; ++a) buf[a * kRowPad + t] = v[a];
#pragma unroll
  for (int q = 1; q < 16; ++q) v[q] = bload_cf(r_t2, vl8, q * 16 * 8);       // stage-2 twiddles
  __syncthreads();
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) u[j2] = buf[hi4 * kRowPad + 16 * j2 + lo4];
  fft16<-1>(u);
#pragma unroll
  for (int q = 1; q < 16; ++q) u[q] = cmul(u[q], v[q]);
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < 16; ++q) buf[hi4 * kRowPad + lo4 * 17 + q] = u[q];
  wave_lds_fence();
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) v[t2] = buf[hi4 * kRowPad + t2 * 17 + lo4];
#pragma unroll
  for (int q = 0; q < 16; ++q) u[q] = bload_cf(r_h, vo8, q * 256 * 8);       // this thread's 16 bins of H, a phase early
  fft16<-1>(v);                                            // over t2 -> kb2: thread (ka, kb1) holds k2 = ka + 16 kb1 + 256 kb2

  // ---- W = H Z ----
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], u[q]);

  // ---- inverse FFT4096 (mirror) ----
#pragma unroll
  for (int q = 1; q < 16; ++q) u[q] = bload_cf(r_t2, vl8, q * 16 * 8);
  fft16<+1>(v);                                            // over kb2 -> t2
#pragma unroll
  for (int q = 1; q < 16; ++q) v[q] = cmulc(v[q], u[q]);
  wave_lds_fence();                                        // the group's plane row was last read by these same lanes
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) buf[hi4 * kRowPad + t2 * 17 + lo4] = v[t2];
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < 16; ++q) u[q] = buf[hi4 * kRowPad + lo4 * 17 + q];
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) v[j2] = bload_cf(r_t4, vo8, j2 * 256 * 8);  // inverse 2nd twiddle
  fft16<+1>(u);                                            // over kb1 -> j2
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) u[j2] = cmulc(u[j2], v[j2]);
  wave_lds_fence();
#pragma unroll
  for (int j2 = 0; j2 < 16; ++j2) buf[hi4 * kRowPad + 16 * j2 + lo4] = u[j2];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 16; ++a) v[a] = buf[a * kRowPad + t];
  fft16<+1>(v);                                            // over ka -> j
#pragma unroll
  for (int j = 0; j < 16; ++j) bstore_cf<IMP_AUX_ROWS_ST>(v[j], r_row, vo8, j * 256 * 8);
}

// ---------------------------------------------------------------------------------------------
// K5 as ONE launch: overlap-save blocks that never leave the CU (core/impulse_response.py:110-119 equalize / :126-135
// convolve, core/parallel_workers.py:9-21: x[n] (*) fir[K], K <= 24 577).  A workgroup of 1024 threads owns one
// (channel, block): 32 768 input samples starting kp = K - 1 (rounded up to even) before the block's first output, as a
// four-row four-step transform held in registers and LDS:
//   load      four columns per thread (eight loads in flight at a time), radix-4 over the four rows, x the four-step twiddle,
//             through LDS [4][4096] to the threads that own the rows (a thread reading all four rows of its 16 row-pass
//             columns instead was 2.5x slower: sixteen dependent round trips to memory)
//   rows      rows_core on the row pairs (0, 2) and (1, 3): forward FFT4096, W = alpha Z + beta conj Z[Nc - k] with the
//             channel's own alpha/beta planes (those of a 4-row plan), inverse FFT4096
//   store     rows -> LDS [4][4096] -> inverse radix-4 per column -> the 32 768 - kp valid samples of the block
// Against the three-launch short plan (8 rows of workspace, three kernels of 128 - 512 small workgroups): no workspace
// traffic at all, one launch, and the spectrum planes are half as long.
// ---------------------------------------------------------------------------------------------
struct FirBlockArgs {
  const float4* __restrict__ ab;   // [n_filters][4][4096], register order of rows_core
  long long ab_chan_stride;        // 0: one filter shared by all channels; 4 * 4096: per channel
  float* __restrict__ out;
  long long out_stride;            // samples between output rows
  long long out_start, out_len;    // window of the linear convolution that is kept ('full': 0, L + M - 1)
  int kp;                          // taps - 1 rounded up to even: samples of history a block starts with
  int valid;                       // 32768 - kp output samples per block
  int blocks;                      // blocks per channel in this launch
  int first_block;                 // the launch's block 0 is block first_block of the convolution ('same': the window starts late)
  int nchan;
};

constexpr int kFirBlockPoints = 4 * kN2;                       // complex points of a block
constexpr size_t kFirBlockLds = sizeof(cf) * 4 * 16 * kRowPad;

// a loader whose row lengths live in device memory (slice_kernels.hip.h LoadRowsDeviceLen): the kernel then takes the
// kept window's length and the filter of a row from the loader, and the blocks past a row's end exit at once
template <class L, class = void> struct LoadHasDeviceLen { static constexpr bool value = false; };
template <class L> struct LoadHasDeviceLen<L, decltype((void)L::kDeviceLen)> { static constexpr bool value = true; };

template <class Load>
__global__ __launch_bounds__(1024, 4) void fir_block_kernel(Load ld, FirBlockArgs a, Twiddles tw) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cf* lds = reinterpret_cast<cf*>(smem_raw);
  // a channel's blocks share its alpha/beta planes: keep them on one XCD (blocks b and b + 8 share an XCD)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chan = (slot / a.blocks) * 8 + xcd, blk = a.first_block + slot % a.blocks;
  if (chan >= a.nchan) return;                                 // the grid is padded to eight channels (whole workgroup exits)
  long long out_len = a.out_len;
  int filt = chan;
  if constexpr (LoadHasDeviceLen<Load>::value) {
    out_len = ld.out_len(chan);
    filt = ld.filter_of(chan);
    if ((long long)blk * a.valid - a.out_start >= out_len) return;   // the grid covers the longest row the plan allows
  }
  const int tid = threadIdx.x;
  const int pairq = __builtin_amdgcn_readfirstlane(tid >> 9);  // 0: rows (0, 2), 1: rows (1, 3)
  const int half = __builtin_amdgcn_readfirstlane((tid >> 8) & 1);
  const int t = tid & 255;
  const int k1 = pairq + 2 * half;
  const int s0 = blk * a.valid - a.kp;                         // even: first input sample of the block (|s0| < 2^29)
  const __amdgpu_buffer_rsrc_t r_full = make_rsrc(tw.full, 4u * kN2 * 8u);
  const auto row = ld.open(chan);

  // columns n2 = tid + 1024 c: sixteen 8-byte loads and twelve twiddles in flight per thread, radix-4 over the rows,
  // x the four-step twiddle, then through LDS [4][4096] to the threads that own the rows
  cf v[16];
#pragma unroll
  for (int h = 0; h < 2; ++h) {                                // two columns at a time: eight loads + six twiddles in flight
    cf x[2][4], w[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n2 = tid + 1024 * (2 * h + c);
#pragma unroll
      for (int n1 = 0; n1 < 4; ++n1) x[c][n1] = row.pair_at(s0 + 2 * (n1 * kN2 + n2));
#pragma unroll
      for (int k = 1; k < 4; ++k) w[c][k - 1] = bload_cf(r_full, (unsigned)(k * kN2 + n2) * 8u, 0u);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n2 = tid + 1024 * (2 * h + c);
#pragma unroll
      for (int n1 = 0; n1 < 4; ++n1) x[c][n1] = row.finish(x[c][n1], s0 + 2 * (n1 * kN2 + n2));   // (the chain's fades)
      bfly4<-1>(x[c][0], x[c][1], x[c][2], x[c][3]);
      lds[n2] = x[c][0];
#pragma unroll
      for (int k = 1; k < 4; ++k) lds[k * kN2 + n2] = cmul(x[c][k], w[c][k - 1]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = lds[k1 * kN2 + t + 256 * j];
  __syncthreads();                                             // rows_core reuses the same LDS
  const __amdgpu_buffer_rsrc_t r_ab =
      make_rsrc(a.ab + (long long)filt * a.ab_chan_stride + (long long)k1 * kN2, kN2 * sizeof(float4));
  // (16 waves per CU leave 128 VGPRs: four alpha/beta bins prefetched instead of the row pass's eight)
  rows_core<0, 4>(v, lds + pairq * (2 * 16 * kRowPad), half, pairq, k1, r_ab, tw, t);

  __syncthreads();                                             // every plane is free
#pragma unroll
  for (int j = 0; j < 16; ++j) lds[k1 * kN2 + t + 256 * j] = v[j];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out + (long long)chan * a.out_stride, (unsigned)out_len * 4u);
  const int o0 = blk * a.valid - a.kp - (int)a.out_start;      // output index of the block's position 0
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n2 = tid + 1024 * c;
    cf w0 = lds[n2], w1 = lds[kN2 + n2], w2 = lds[2 * kN2 + n2], w3 = lds[3 * kN2 + n2];
    w1 = cmulc(w1, bload_cf(r_full, (unsigned)(kN2 + n2) * 8u, 0u));
    w2 = cmulc(w2, bload_cf(r_full, (unsigned)(2 * kN2 + n2) * 8u, 0u));
    w3 = cmulc(w3, bload_cf(r_full, (unsigned)(3 * kN2 + n2) * 8u, 0u));
    bfly4<+1>(w0, w1, w2, w3);                                 // samples 2 (n1 4096 + n2), + 1 of the block, n1 = 0..3
    const cf y[4] = {w0, w1, w2, w3};
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
      const int pos = 2 * (n1 * kN2 + n2);
      if (pos < a.kp) continue;                                // the block's history: wrapped-around garbage
      // the window's two edges fall to the range check (an offset below the window wraps out of range); an odd window
      // start splits the pair into two 4-byte stores, like StoreRealCrop
      const unsigned off = (unsigned)(o0 + pos) * 4u;
      if (a.out_start & 1) {
        unsigned off_im = off + 4u;
        asm volatile("" : "+v"(off_im));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[n1].x), r_out, off, 0u, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[n1].y), r_out, off_im, 0u, 0);
      } else {
        bstore_cf<kStreamAux>(y[n1], r_out, off, 0u);
      }
    }
  }
}

}  // namespace imp
