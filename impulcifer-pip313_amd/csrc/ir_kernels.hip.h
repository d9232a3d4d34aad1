// Small batched kernels on ragged sets of impulse responses (rows at x + off[b], length len[b]).
//   K3  first significant peak   core/impulse_response.py:32-70  (twin core/decay.py:12-41)
//   K4  gain + Hann fades        core/hrir.py:530-544, :591-612, :642-651
//   K8  decay window             core/decay.py:383-403
// All of them are HBM-bound streaming passes (1 read, or 1 read + 1 write); no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_kernels.hip.h"

namespace imp {

struct RowPeak {
  unsigned int maxabs_bits;         // bits of max|x| (non-negative floats order like unsigned ints)
  unsigned int pad;
  unsigned long long first_peak;    // smallest index passing the find_peaks rule, ~0 if none
  unsigned long long first_max;     // smallest index with |x| == max|x|; looked for only when there is no peak
};

// ---------------------------------------------------------------------------------------------
// K3 in two steps.  Step 1 leaves one number per CHUNK of kPeakChunk samples: the bits of max|x| over the chunk.  It is
// either row_chunk_max_kernel below (one read of the rows) or, when the rows come out of K1, the epilogue of pass C
// (StoreRealCropMax further down: a chunk is then one four-step row, 8 192 samples of the un-cropped result, `shift` is
// the crop start, and the maxima arrive per column tile).  Step 2 (row_first_peak_chunked_kernel, one workgroup per row)
// reduces the chunk maxima to the row maximum and then reads only the chunks that CAN hold the answer: the find_peaks
// height test is monotone in |x|, so the first peak starts in the first chunk whose maximum passes it.
// Chunk c of row b covers the samples i with (i + shift) / kPeakChunk == c.
// ---------------------------------------------------------------------------------------------
constexpr int kPeakChunk = 8192;
#ifdef IMP_PEAK_DIAG
#define IMP_PEAK_T(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) peak_ts[k] = clock64(); } while (0)
#else
#define IMP_PEAK_T(k) do { } while (0)
#endif
constexpr int kPeakThreads = 1024;
constexpr int kMaxPlanRows = 256;      // four-step rows (N1) a chained plan may have

__device__ __forceinline__ int64_t peak_chunks(int64_t n, int64_t shift) {
  return n > 0 ? (n + shift + kPeakChunk - 1) / kPeakChunk : 0;
}

// block-wide reductions over NW waves; the result is returned to every thread
template <int NW>
__device__ __forceinline__ float block_max_f32(float m, float* wm /*[NW], 16-byte aligned*/) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
  __syncthreads();                                                        // wm may still be read from an earlier call
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  float r = wm[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) r = fmaxf(r, wm[k]);
  return r;
}

template <int NW>
__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, unsigned long long* wm /*[NW]*/) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const unsigned long long o = __shfl_xor(v, s, 64);
    v = o < v ? o : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = wm[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) r = wm[k] < r ? wm[k] : r;
  return r;
}

// grid = (max chunks per row, B), 256 threads
__global__ __launch_bounds__(256) void row_chunk_max_kernel(const float* __restrict__ x, const int64_t* __restrict__ off,
                                                            const int64_t* __restrict__ len, int64_t shift,
                                                            unsigned* __restrict__ chunk_max, int64_t pitch) {
  const int b = blockIdx.y;
  const int64_t n = len[b], c = blockIdx.x;
  if (c >= peak_chunks(n, shift)) return;
  const int64_t lo = c * kPeakChunk - shift > 0 ? c * kPeakChunk - shift : 0;
  const int64_t hi = (c + 1) * kPeakChunk - shift < n ? (c + 1) * kPeakChunk - shift : n;
  const float* row = x + off[b];
  float m = 0.f;
  int64_t i = lo + threadIdx.x;
  for (; i + 7 * 256 < hi; i += 8 * 256) {                    // eight loads in flight per thread
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = row[i + k * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v[k]));
  }
  for (; i < hi; i += 256) m = fmaxf(m, fabsf(row[i]));
  __shared__ __attribute__((aligned(16))) float wm[4];
  m = block_max_f32<4>(m, wm);
  // NaN-free inputs assumed (as the reference: np.max would propagate NaN and find_peaks none)
  if (threadIdx.x == 0) chunk_max[(int64_t)b * pitch + c] = __float_as_uint(m);
}

// SciPy _local_maxima_1d on +x and -x, height filter x[peak]/max >= h (inclusive), min index.
//  - sample i starts a candidate iff v[i-1] < v[i]
//  - plateau: run of equal samples i..j; it is a peak iff j+1 < n and v[j+1] < v[j];
//    reported index = (i + j) / 2; first and last samples are never peaks
// Plateaus are disjoint and ordered, so the candidate that STARTS first also reports the smallest index: the search can
// stop at the first chunk that yields one.  A chunk is staged in LDS (with one sample either side) by loads that are all
// in flight at once; only a plateau that runs past the chunk goes back to memory.
// grid = B, kPeakThreads threads.
// tiles > 0: step 1 was pass C - the chunk maxima are max over the tiles of tile_max[b][tile][r] (r < pitch) and
// chunk_max is not read.
// peaks_out (optional, device): ImpulseResponse.peak_index of every row.
__global__ __launch_bounds__(kPeakThreads) void row_first_peak_chunked_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len, int64_t shift,
    const unsigned* __restrict__ tile_max, int tiles, const unsigned* __restrict__ chunk_max, int64_t pitch,
    RowPeak* __restrict__ res, double height, long long* __restrict__ peaks_out) {
  const int b = blockIdx.x, tid = threadIdx.x;
#ifdef IMP_PEAK_DIAG
  long long peak_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  IMP_PEAK_T(0);
  const int64_t n = len[b];
  const float* row = x + off[b];
  const int64_t nch = peak_chunks(n, shift);
  constexpr int NW = kPeakThreads / 64;
  __shared__ __attribute__((aligned(16))) float wmf[NW];
  __shared__ __attribute__((aligned(16))) unsigned long long wmu[NW];
  __shared__ float sx[kPeakChunk + 2];
  // the chunk maxima are consulted several times: keep them in LDS (rows of up to 16 M samples; longer ones stay in memory)
  constexpr int kLdsChunks = 2048;
  __shared__ unsigned srows[kLdsChunks];
  const unsigned* cm = chunk_max + (int64_t)b * pitch;
  if (tiles > 0) {
    // lanes along the rows (coalesced), waves along the tiles; nch <= pitch <= kMaxPlanRows here
    for (int r = tid; r < kMaxPlanRows; r += kPeakThreads) srows[r] = 0u;
    __syncthreads();
    const unsigned* tm = tile_max + (int64_t)b * tiles * pitch;
    const int lane = tid & 63, w = tid >> 6;
    for (int64_t r = lane; r < nch; r += 64) {
      unsigned m = 0u;
#pragma unroll 4
      for (int t = w; t < tiles; t += kPeakThreads / 64) {
        const unsigned v = tm[(int64_t)t * pitch + r];
        m = v > m ? v : m;
      }
      if (m) atomicMax(&srows[r], m);
    }
    __syncthreads();
    cm = srows;
  } else if (nch <= kLdsChunks) {
    for (int64_t c = tid; c < nch; c += kPeakThreads) srows[c] = cm[c];
    __syncthreads();
    cm = srows;
  }
  IMP_PEAK_T(1);
  float mf = 0.f;
  for (int64_t c = tid; c < nch; c += kPeakThreads) mf = fmaxf(mf, __uint_as_float(cm[c]));
  const float maxabs = block_max_f32<NW>(mf, wmf);
  RowPeak out;
  out.maxabs_bits = __float_as_uint(maxabs);
  out.pad = 0u;
  out.first_peak = ~0ull;
  out.first_max = ~0ull;
  if (maxabs >= 1e-20f) {
    // The reference tests data / max >= height on float64 data.  Division by a positive number is monotone, so on fp32
    // samples that is |x| >= thr with thr the smallest float passing the fp64 test: found once (by one thread: an fp64
    // division costs a wave ~400 cycles), no division per sample.
    __shared__ float s_thr;
    if (tid == 0) {
      const double dmax = (double)maxabs;
      float t = (float)(height * dmax);
      if (!(t >= 0.f)) t = 0.f;
      for (int k = 0; k < 4 && t > 0.f && (double)__uint_as_float(__float_as_uint(t) - 1u) / dmax >= height; ++k)
        t = __uint_as_float(__float_as_uint(t) - 1u);
      for (int k = 0; k < 4 && !((double)t / dmax >= height); ++k) t = __uint_as_float(__float_as_uint(t) + 1u);
      s_thr = t;
    }
    __syncthreads();
    const float thr = s_thr;
    IMP_PEAK_T(2);
    unsigned long long c_q = ~0ull;                             // first chunk that can hold a peak
    for (int64_t c = tid; c < nch; c += kPeakThreads)
      if (__uint_as_float(cm[c]) >= thr && (unsigned long long)c < c_q) c_q = (unsigned long long)c;
    c_q = block_min_u64<NW>(c_q, wmu);
    constexpr int U = (kPeakChunk + kPeakThreads - 1) / kPeakThreads;      // samples per thread and chunk
    int64_t lo = 0, hi = 0;
    auto stage = [&](int64_t c) {                              // sx[k] = row[lo - 1 + k], zero outside the row
      lo = c * kPeakChunk - shift > 0 ? c * kPeakChunk - shift : 0;
      hi = (c + 1) * kPeakChunk - shift < n ? (c + 1) * kPeakChunk - shift : n;
      const int64_t cnt = hi - lo + 2;
      float t[U + 1];
#pragma unroll
      for (int u = 0; u <= U; ++u) {                           // all loads in flight before the first LDS write
        const int64_t k = tid + (int64_t)u * kPeakThreads, i = lo - 1 + k;
        t[u] = (k < cnt && i >= 0 && i < n) ? row[i] : 0.f;
      }
      __syncthreads();                                         // the previous chunk is no longer read
#pragma unroll
      for (int u = 0; u <= U; ++u) {
        const int64_t k = tid + (int64_t)u * kPeakThreads;
        if (k < cnt) sx[k] = t[u];
      }
      __syncthreads();
    };
    IMP_PEAK_T(3);
    for (int64_t c = c_q == ~0ull ? nch : (int64_t)c_q; c < nch; ++c) {          // every bound here is block-uniform
      if (!(__uint_as_float(cm[c]) >= thr)) continue;
      stage(c);
      IMP_PEAK_T(4);
      float xs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = lo + tid + (int64_t)u * kPeakThreads;
        xs[u] = i < hi ? sx[i - lo + 1] : 0.f;
      }
      unsigned long long best = ~0ull;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = lo + tid + (int64_t)u * kPeakThreads;
        const float xi = xs[u];
        if (best != ~0ull || !(fabsf(xi) >= thr) || i >= hi || i == 0 || i >= n - 1) continue;   // the rare test first
        const float xm = sx[i - lo];
        // polarity: +x peak needs xm < xi ; -x peak needs xm > xi
        const bool up = xm < xi, dn = xm > xi;
        if (!up && !dn) continue;
        if (!((up ? xi : -xi) >= thr)) continue;
        int64_t j = i;
        while (j + 1 < n && (j + 1 <= hi ? sx[j + 1 - lo + 1] : row[j + 1]) == xi) ++j;
        if (j + 1 >= n) continue;
        const float xn = j + 1 <= hi ? sx[j + 1 - lo + 1] : row[j + 1];
        // this thread's later samples can only report later peaks: `best` set ends its search
        if (up ? (xn < xi) : (xn > xi)) best = (unsigned long long)((i + j) / 2);
      }
      best = block_min_u64<NW>(best, wmu);
      if (best != ~0ull) {
        out.first_peak = best;
        break;
      }
    }
    IMP_PEAK_T(5);
    if (out.first_peak == ~0ull) {
      // no peak anywhere (a monotone row, say): the answer is the first sample at the maximum (argmax fallback)
      unsigned long long c_max = ~0ull;
      for (int64_t c = tid; c < nch; c += kPeakThreads)
        if (__uint_as_float(cm[c]) == maxabs && (unsigned long long)c < c_max) c_max = (unsigned long long)c;
      c_max = block_min_u64<NW>(c_max, wmu);
      stage((int64_t)c_max);
      unsigned long long best = ~0ull;
      for (int64_t i = lo + tid; i < hi; i += kPeakThreads)
        if (fabsf(sx[i - lo + 1]) == maxabs) {
          best = (unsigned long long)i;
          break;
        }
      out.first_max = block_min_u64<NW>(best, wmu);
    }
  }
  IMP_PEAK_T(6);
#ifdef IMP_PEAK_DIAG
  if (tid == 0 && b == 0)
    printf("peak phases (clk): entry->fold %lld, max+thr %lld, cq %lld, stage %lld, search %lld, fallback %lld\n", peak_ts[1] - peak_ts[0],
           peak_ts[2] - peak_ts[1], peak_ts[3] - peak_ts[2], peak_ts[4] - peak_ts[3], peak_ts[5] - peak_ts[4], peak_ts[6] - peak_ts[5]);
#endif
  if (tid == 0) {
    res[b] = out;
    if (peaks_out) {
      long long pk;
      if (n == 0 || !(maxabs >= 1e-20f)) pk = 0;                                   // EPSILON rule, impulse_response.py:56-58
      else pk = (long long)(out.first_peak != ~0ull ? out.first_peak : out.first_max);   // argmax fallback, :66-67
      peaks_out[b] = pk;
    }
  }
}

__device__ __forceinline__ double hann_sym_fwd(int64_t i, int64_t N) {     // scipy hann(N, sym=True)[i]
  if (N <= 1) return 1.0;
  return 0.5 - 0.5 * cospi(2.0 * (double)i / (double)(N - 1));
}

// ---------------------------------------------------------------------------------------------
// K1 -> K3 -> K4 -> K5 without leaving the device (imp_chain): two functors that put K3's first step and K4 into the
// column passes either side of them.
// ---------------------------------------------------------------------------------------------

// Pass C store of K1 that also leaves the maxima K3 starts from: max|y| over the KEPT samples of every four-step row
// within the workgroup's column tile (point p = e + step lies in row p >> 12; its samples are 2p - start and
// 2p + 1 - start).  A wave holds 64 consecutive columns of one row, or 2 x 32 columns of two rows (ColsCfg with TC = 32):
// the maximum is taken over each half wave with DPP moves, lanes 31 / 63 fold it into a per-row LDS word, and end()
// writes the workgroup's n1 words as one coalesced store to tile_max[b][tile][row].  Plain stores, every word written by
// exactly one workgroup: nothing to zero, no global atomics (one atomicMax per half wave and put made pass C 3x slower).
__device__ __forceinline__ unsigned* crop_max_lds() {
  __shared__ unsigned rows[kMaxPlanRows];
  return rows;
}

struct StoreRealCropMax {
  StoreRealCrop crop;
  unsigned* __restrict__ tile_max;   // [B][tiles][n1]
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bind(int b) const { return crop.bind(b); }
  __device__ __forceinline__ void begin(int tid, int threads) const {
    unsigned* rows = crop_max_lds();
    for (int r = tid; r < kMaxPlanRows; r += threads) rows[r] = 0u;
    __syncthreads();
  }
  __device__ __forceinline__ void put(__amdgpu_buffer_rsrc_t r, int b, unsigned e, unsigned step_elems, cf v) const {
    crop.put(r, b, e, step_elems, v);
    const unsigned p = e + step_elems;
    const unsigned i_re = 2u * p - (unsigned)crop.start;          // wraps out of range below the window
    float m = i_re < (unsigned)crop.len ? fabsf(v.x) : 0.f;
    if (i_re + 1u < (unsigned)crop.len) m = fmaxf(m, fabsf(v.y));
    int mi = __float_as_int(m);
    // max over each row of 16 lanes (quad swaps, half mirror, mirror), then row 0 -> 1 and row 2 -> 3 (row_bcast:15)
    mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0xB1, 0xF, 0xF, true))));
    mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x4E, 0xF, 0xF, true))));
    mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x141, 0xF, 0xF, true))));
    mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x140, 0xF, 0xF, true))));
    mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x142, 0xA, 0xF, true))));
    if ((threadIdx.x & 31) == 31 && mi != 0) atomicMax(crop_max_lds() + (p >> 12), (unsigned)mi);
  }
  __device__ __forceinline__ void end(int b, int tile, int tiles, int n1, int tid, int threads) const {
    __syncthreads();
    const unsigned* rows = crop_max_lds();
    unsigned* out = tile_max + ((long long)b * tiles + tile) * n1;
    for (int r = tid; r < n1; r += threads) out[r] = rows[r];
  }
};

// The same in pair mode: sample n of BOTH channels sits in point p = n, a four-step row holds 4 096 samples of each channel,
// and the maxima are kept per channel and per 8 192-sample chunk (two four-step rows): tile_max[channel][tile][n1 / 2].
__device__ __forceinline__ unsigned* crop_max_lds_r() {
  __shared__ unsigned rows_r[kMaxPlanRows];
  return rows_r;
}

__device__ __forceinline__ int half_wave_max(float m) {      // max over each 32-lane half, valid in lanes 31 / 63
  int mi = __float_as_int(m);
  mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0xB1, 0xF, 0xF, true))));
  mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x4E, 0xF, 0xF, true))));
  mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x141, 0xF, 0xF, true))));
  mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x140, 0xF, 0xF, true))));
  mi = __float_as_int(fmaxf(__int_as_float(mi), __int_as_float(__builtin_amdgcn_update_dpp(0, mi, 0x142, 0xA, 0xF, true))));
  return mi;
}

struct StorePairCropMax {
  StorePairCrop crop;
  unsigned* __restrict__ tile_max;   // [channels][tiles][n1 / 2]
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t bind(int p) const { return crop.bind(p); }
  __device__ __forceinline__ void begin(int tid, int threads) const {
    unsigned *rl = crop_max_lds(), *rr = crop_max_lds_r();
    for (int r = tid; r < kMaxPlanRows; r += threads) rl[r] = rr[r] = 0u;
    __syncthreads();
  }
  __device__ __forceinline__ void put(__amdgpu_buffer_rsrc_t r, int p, unsigned e, unsigned step_elems, cf v) const {
    crop.put(r, p, e, step_elems, v);
    const unsigned n = e + step_elems;
    const bool kept = n - (unsigned)crop.start < (unsigned)crop.len;      // wraps out of range below the window
    const int ml = half_wave_max(kept ? fabsf(v.x) : 0.f), mr = half_wave_max(kept ? fabsf(v.y) : 0.f);
    if ((threadIdx.x & 31) == 31) {
      if (ml != 0) atomicMax(crop_max_lds() + (n >> 13), (unsigned)ml);
      if (mr != 0) atomicMax(crop_max_lds_r() + (n >> 13), (unsigned)mr);
    }
  }
  __device__ __forceinline__ void end(int p, int tile, int tiles, int n1, int tid, int threads) const {
    __syncthreads();
    const int chunks = n1 >> 1;
    const unsigned *rl = crop_max_lds(), *rr = crop_max_lds_r();
    unsigned* out = tile_max + ((long long)(2 * p) * tiles + tile) * chunks;
    for (int r = tid; r < chunks; r += threads) out[r] = rl[r];
    if (2 * p + 1 < crop.nchan) {
      out += (long long)tiles * chunks;
      for (int r = tid; r < chunks; r += threads) out[r] = rr[r];
    }
  }
};

// Pass A load of K5 that IS K4: channel b is read from start = clamp(peak - head, 0, row_len - n) of its row, n samples
// long, with a Hann fade-in of `fade_in` and fade-out of `fade_out` samples (core/impulse_response.py:82-90 crop_head +
// the fades of core/hrir.py:591-612, :642-651 at a fixed length).  peak = ImpulseResponse.peak_index as K3 left it in
// res[b].  Samples past the row's end read as zero (the buffer's range check), as does the transform's padding.
struct LoadCropAtPeak {
  const float* __restrict__ base;   // row 0, sample 0
  long long chan_stride;            // samples between rows
  long long row_len;                // samples per row
  const RowPeak* __restrict__ res;
  long long n, head, fade_in, fade_out;
  // hann(2 fade_in)[:fade_in] followed by hann(2 fade_out)[fade_out:], fp64, made once per chain (fade_table_kernel): the
  // loaders multiply by table entries instead of evaluating a cosine per sample (the library cospi keeps a stack frame,
  // and a kernel with scratch pays for it at every launch)
  const double* __restrict__ win;
  __host__ __device__ LoadCropAtPeak shifted(long long, long long) const { return *this; }   // overlap-add plans are refused
  static __device__ __forceinline__ float shaped(float x, long long i, long long n, long long fade_in, long long fade_out,
                                                 const double* __restrict__ win) {
    double g = 1.0;
    if (i < fade_in) g *= win[i];
    if (fade_out > 0 && i >= n - fade_out && i < n) g *= win[fade_in + (i - (n - fade_out))];
    return (float)((double)x * g);
  }
  __device__ __forceinline__ long long crop_start(int b) const {
    const RowPeak rp = res[b];
    long long pk;
    if (row_len == 0 || !(__uint_as_float(rp.maxabs_bits) >= 1e-20f)) pk = 0;
    else pk = (long long)(rp.first_peak != ~0ull ? rp.first_peak : rp.first_max);
    long long start = pk - head;
    if (start > row_len - n) start = row_len - n;
    if (start < 0) start = 0;
    return start;
  }
  struct Row {                        // fir_block_kernel's view of a channel (see LoadRealPacked::Row)
    __amdgpu_buffer_rsrc_t r;
    int n, fade_in, fade_out;
    const double* __restrict__ win;
    __device__ __forceinline__ cf pair_at(int s) const { return bload_cf<0>(r, (unsigned)s * 4u, 0u); }
    // the window, applied once the loads have landed (kept out of the load loop: fewer registers live across it)
    __device__ __forceinline__ cf finish(cf v, int s) const {
      if ((unsigned)s < (unsigned)n && (s < fade_in || s + 1 >= n - fade_out)) {      // only the two ends pay for the window
        v.x = shaped(v.x, s, n, fade_in, fade_out, win);
        v.y = shaped(v.y, s + 1, n, fade_in, fade_out, win);
      }
      return v;
    }
  };
  __device__ __forceinline__ Row open(int b) const {
    const long long start = crop_start(b), avail = row_len - start;
    return Row{make_rsrc(base + (long long)b * chan_stride + start, (unsigned)(avail < n ? avail : n) * 4u), (int)n, (int)fade_in,
               (int)fade_out, win};
  }
  template <int STEP, int F>
  __device__ __forceinline__ void column(int b, unsigned e0, cf (&v)[F]) const {
    const long long start = crop_start(b), avail = row_len - start;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(base + (long long)b * chan_stride + start, (unsigned)(avail < n ? avail : n) * 4u);
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, e0 * 8u, (unsigned)(j * STEP) * 8u, 0);
      v[j] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
      const long long i = 2ll * ((long long)e0 + (long long)j * STEP);
      if (i < n && (i < fade_in || i + 1 >= n - fade_out)) {     // only the two ends pay for the window
        v[j].x = shaped(v[j].x, i, n, fade_in, fade_out, win);
        v[j].y = shaped(v[j].y, i + 1, n, fade_in, fade_out, win);
      }
    }
  }
};

__global__ __launch_bounds__(256) void fade_table_kernel(double* __restrict__ win, long long fade_in, long long fade_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < fade_in) win[i] = hann_sym_fwd(i, 2 * fade_in);
  else if (i < fade_in + fade_out) win[i] = hann_sym_fwd(fade_out + (i - fade_in), 2 * fade_out);
}

// layout-compatible with imp_window_params (include/impulse_hip.h)
struct WindowParams {
  float gain;
  int64_t fade_in;
  int64_t fade_out;
  int64_t decay_start;
  int64_t decay_half;
  int64_t decay_knee;
  float decay_level_db;
};

// scipy.signal.windows.hann(N, sym=True)[i] = 0.5 - 0.5 cos(2 pi i / (N - 1))
__device__ __forceinline__ double hann_sym(int64_t i, int64_t N) {
  if (N <= 1) return 1.0;
  return 0.5 - 0.5 * cospi(2.0 * (double)i / (double)(N - 1));
}

__global__ __launch_bounds__(256) void apply_window_kernel(float* __restrict__ x,
                                                           const int64_t* __restrict__ off,
                                                           const int64_t* __restrict__ len,
                                                           const WindowParams* __restrict__ par) {
  const int b = blockIdx.y;
  const int64_t n = len[b];
  float* row = x + off[b];
  const WindowParams p = par[b];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double g = (double)p.gain;
    if (i < p.fade_in) g *= hann_sym(i, 2 * p.fade_in);                        // hann(2F)[:F]
    if (p.fade_out > 0 && i >= n - p.fade_out)
      g *= hann_sym(p.fade_out + (i - (n - p.fade_out)), 2 * p.fade_out);      // hann(2F)[F:]
    if (p.decay_half >= 0) {
      // window = concat(ones(start), hann(2h)[h:], zeros(n - knee)) - 1 ; x *= 10^(-level*window/20)
      double w;
      if (i < p.decay_start) w = 1.0;
      else if (i < p.decay_knee) w = hann_sym(p.decay_half + (i - p.decay_start), 2 * p.decay_half);
      else w = 0.0;
      g *= pow(10.0, (w - 1.0) * (-(double)p.decay_level_db) / 20.0);
    }
    row[i] = (float)((double)row[i] * g);
  }
}

// the same window, read from one set of rows and written to another (device-resident responses: head/tail crops are an
// offset into the source row, the result is compacted into rows of a common pitch for the next stage)
__global__ __launch_bounds__(256) void apply_window_copy_kernel(const float* __restrict__ src, const int64_t* __restrict__ src_off,
                                                                float* __restrict__ dst, const int64_t* __restrict__ dst_off,
                                                                const int64_t* __restrict__ len,
                                                                const WindowParams* __restrict__ par) {
  const int b = blockIdx.y;
  const int64_t n = len[b];
  const float* in = src + src_off[b];
  float* out = dst + dst_off[b];
  const WindowParams p = par[b];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double g = (double)p.gain;
    if (i < p.fade_in) g *= hann_sym(i, 2 * p.fade_in);
    if (p.fade_out > 0 && i >= n - p.fade_out) g *= hann_sym(p.fade_out + (i - (n - p.fade_out)), 2 * p.fade_out);
    if (p.decay_half >= 0) {
      double w;
      if (i < p.decay_start) w = 1.0;
      else if (i < p.decay_knee) w = hann_sym(p.decay_half + (i - p.decay_start), 2 * p.decay_half);
      else w = 0.0;
      g *= pow(10.0, (w - 1.0) * (-(double)p.decay_level_db) / 20.0);
    }
    out[i] = (float)((double)in[i] * g);
  }
}

// HRIR.write_wav on device rows (core/hrir.py:426-455 -> core/audio_io.py:82-97 -> soundfile, which turns libsndfile's
// clipping on): out[i][t] = clip(lrint(row_t[i] * 2^31), -2^31, 2^31 - 1) >> (32 - bits), zeros for tracks without a row;
// interleaved frames, the WAV wire order.  PCM_32 is pinned by the sweep WAVs the reference ships (tests/golden/
// sweep_wavs.npz); see audio_io.pcm_quantise.  row_of_track[t] = row index or -1.  Samples are stored as int32 (bits 24/32)
// or int16 (bits 16).
__global__ __launch_bounds__(256) void rows_to_pcm_kernel(const float* __restrict__ src, const int64_t* __restrict__ off,
                                                          const int64_t* __restrict__ len,
                                                          const int64_t* __restrict__ row_of_track, int n_tracks,
                                                          int64_t n_frames, int bits, void* __restrict__ out) {
  const int shift = 32 - bits;
  const int64_t total = n_frames * n_tracks;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = k / n_tracks;
    const int t = (int)(k - i * n_tracks);
    const int64_t r = row_of_track[t];
    long long q = 0;
    if (r >= 0 && i < len[r]) {
      const double v = (double)src[off[r] + i] * 2147483648.0;   // exact: a power-of-two scale
      // saturate, then round half to even like lrint / np.rint (a NaN sample falls through to llrint, as in libsndfile)
      q = v >= 2147483647.0 ? 2147483647ll : v <= -2147483648.0 ? -2147483648ll : llrint(v);
      q >>= shift;                                               // arithmetic shift: the top `bits` bits
    }
    if (bits == 16) reinterpret_cast<short*>(out)[k] = (short)q;
    else reinterpret_cast<int*>(out)[k] = (int)q;
  }
}

// fp32 device rows -> packed fp64 segments (K7 analysis spans of device-resident responses)
__global__ __launch_bounds__(256) void seg_from_float_kernel(const float* __restrict__ src, const int64_t* __restrict__ src_off,
                                                             double* __restrict__ dst, const int64_t* __restrict__ dst_off,
                                                             const int64_t* __restrict__ len) {
  const int b = blockIdx.y;
  const int64_t n = len[b];
  const float* in = src + src_off[b];
  double* out = dst + dst_off[b];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (double)in[i];
}

// K10: argmax of the full cross-correlation of B pairs of short segments (core/hrir.py:934-937, :946-949,
// scipy.signal.correlate(a, b, "full") then np.argmax).  corr[k] = sum_l a[l + k - (nb - 1)] b[l],
// k = 0 .. na + nb - 2, summed in fp64 in index order.  One workgroup per pair, both segments in LDS,
// a thread owns lags k = tid, tid + 256, ...; the first maximum wins, as in np.argmax.
// Sample = double (host segments, uploaded) or float (device-resident rows, converted exactly on load).
// A thread owns four ADJACENT lags (k = 4 T .. 4 T + 3) and walks l once for all of them: one LDS read of b[l] and one new
// a sample per step feed four fused multiply-adds (a register window slides over a).  l runs over a range that is
// uniform per WAVE (the union of its 256 lags' ranges; a sits in LDS between kXcorrPad zeros on either side, which cover the
// steps a lane runs beyond its own range and leave its sums unchanged), so the window's address is `uniform + 4 T`: with a
// stored as four interleaved planes (plane c holds a[4 m + c]) the lanes of a wave read CONSECUTIVE slots of one plane -
// no bank conflict - and eight steps issue their sixteen reads together.  Every lag's sum is still the fma chain over l in
// index order, so maxima and their order are those of a one-lag-per-thread form.  The lags of a pair are cut into slices of
// kXcorrThreads * 4 (grid.y): a workgroup leaves its slice's first maximum, xcorr_reduce_kernel picks the first over slices.
constexpr int kXcorrThreads = 256;
constexpr int kXcorrLags = 4;
constexpr int kXcorrPad = 260;                     // >= 64 lanes * 4 lags + the 3-sample window, a multiple of 4
__host__ __device__ inline long long xcorr_lds_doubles(long long na, long long nb) {
  return 4 * ((na + 2 * kXcorrPad + 3) / 4 + 1) + nb;
}
template <class Sample>
__global__ __launch_bounds__(kXcorrThreads) void xcorr_argmax_kernel(const Sample* __restrict__ a, const int64_t* __restrict__ a_off,
                                                                     const int64_t* __restrict__ a_len,
                                                                     const Sample* __restrict__ b, const int64_t* __restrict__ b_off,
                                                                     const int64_t* __restrict__ b_len,
                                                                     long long* __restrict__ part_k, double* __restrict__ part_val) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* plane = reinterpret_cast<double*>(smem_raw);
  const int p = blockIdx.x;
  const int na = (int)a_len[p], nb = (int)b_len[p];
  const int P = (na + 2 * kXcorrPad + 3) / 4 + 1;      // slots per plane
  double* sb = plane + 4 * P;
  for (int i = threadIdx.x; i < 4 * P; i += blockDim.x) {     // i' = 4 m + c <-> a[i' - pad], zero outside
    const int c = i / P, m = i - c * P, src = 4 * m + c - kXcorrPad;
    plane[i] = (src >= 0 && src < na) ? (double)a[a_off[p] + src] : 0.0;
  }
  for (int i = threadIdx.x; i < nb; i += blockDim.x) sb[i] = (double)b[b_off[p] + i];
  __syncthreads();
  double best = -__builtin_huge_val();
  int best_k = 0x7fffffff;
  const int nk = na + nb - 1;
  const int T = (int)blockIdx.y * kXcorrThreads + (int)threadIdx.x;           // lags 4 T .. 4 T + 3
  const int kw = __builtin_amdgcn_readfirstlane(((int)blockIdx.y * kXcorrThreads + ((int)threadIdx.x & ~63)) * kXcorrLags);
  if (kw < nk) {                                                               // wave-uniform
    int lo = (nb - 1) - (kw + 64 * kXcorrLags - 1);
    lo = lo < 0 ? 0 : lo;
    int hi = na + nb - 1 - kw;
    hi = hi > nb ? nb : hi;
    // a index of (l, lag 4 T + r) = l + 4 T + r - (nb - 1); padded: + kXcorrPad.  q = the wave-uniform part.
    const double* __restrict__ lane = plane + T;
    auto win = [&](int q) { return lane[(q & 3) * P + (q >> 2)]; };            // q + 4 T >= 0 inside the padded range
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int l = lo;
    int q0 = l - (nb - 1) + kXcorrPad;
    double w0 = win(q0), w1 = win(q0 + 1), w2 = win(q0 + 2);
    constexpr int U = 8;
    for (; l + U <= hi; l += U) {
      double bb[U], ww[U + 3];
      ww[0] = w0;
      ww[1] = w1;
      ww[2] = w2;
      const int q = l - (nb - 1) + kXcorrPad + 3;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bb[u] = sb[l + u];
        ww[u + 3] = win(q + u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc0 = fma(ww[u], bb[u], acc0);
        acc1 = fma(ww[u + 1], bb[u], acc1);
        acc2 = fma(ww[u + 2], bb[u], acc2);
        acc3 = fma(ww[u + 3], bb[u], acc3);
      }
      w0 = ww[U];
      w1 = ww[U + 1];
      w2 = ww[U + 2];
    }
    for (; l < hi; ++l) {
      const double w3 = win(l - (nb - 1) + kXcorrPad + 3);
      const double bv = sb[l];
      acc0 = fma(w0, bv, acc0);
      acc1 = fma(w1, bv, acc1);
      acc2 = fma(w2, bv, acc2);
      acc3 = fma(w3, bv, acc3);
      w0 = w1;
      w1 = w2;
      w2 = w3;
    }
    const int k0 = T * kXcorrLags;
    const double acc[kXcorrLags] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int r = 0; r < kXcorrLags; ++r)
      if (k0 + r < nk && acc[r] > best) {            // increasing k: the first maximum stays
        best = acc[r];
        best_k = k0 + r;
      }
  }
  // block argmax, ties to the smaller index
  __shared__ double rv[kXcorrThreads];
  __shared__ int rk[kXcorrThreads];
  rv[threadIdx.x] = best;
  rk[threadIdx.x] = best_k;
  __syncthreads();
  for (int s = kXcorrThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double ov = rv[threadIdx.x + s];
      const int ok = rk[threadIdx.x + s];
      if (ov > rv[threadIdx.x] || (ov == rv[threadIdx.x] && ok < rk[threadIdx.x])) {
        rv[threadIdx.x] = ov;
        rk[threadIdx.x] = ok;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part_k[(long long)p * gridDim.y + blockIdx.y] = rk[0];
    part_val[(long long)p * gridDim.y + blockIdx.y] = rv[0];
  }
}

// the first maximum over a pair's lag slices (slices are in lag order: a later slice wins only with a larger value)
__global__ __launch_bounds__(64) void xcorr_reduce_kernel(const long long* __restrict__ part_k, const double* __restrict__ part_val, int slices,
                                                          int n_pairs, long long* __restrict__ arg_out, double* __restrict__ val_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  double best = part_val[(long long)p * slices];
  long long k = part_k[(long long)p * slices];
  for (int y = 1; y < slices; ++y) {
    const double v = part_val[(long long)p * slices + y];
    if (v > best) {
      best = v;
      k = part_k[(long long)p * slices + y];
    }
  }
  arg_out[p] = k;
  if (val_out) val_out[p] = best;
}

// ---------------------------------------------------------------------------------------------
// K7: reductions of the decay analysis (core/decay.py:44-253, Lundeby) on fp64 segments kept on the
// device.  The reference works on e = (x / max|x|)^2 and takes np.mean over windows and ranges of it;
// the means below add in NumPy's pairwise order (numpy/_core/src/umath/loops_utils.h.src,
// @TYPE@_pairwise_sum: < 8 elements sequentially, <= 128 with eight strided accumulators, longer runs
// split at n/2 rounded down to a multiple of 8; runs beyond the 8192-element ufunc buffer are reduced
// piece by piece), so every level the host compares against a threshold has the bits the reference sees.
// ---------------------------------------------------------------------------------------------

// max |x| per segment (exact: max is order independent); bits of a non-negative double order like integers
__global__ __launch_bounds__(256) void seg_maxabs_kernel(const double* __restrict__ x, const int64_t* __restrict__ off,
                                                         const int64_t* __restrict__ len,
                                                         unsigned long long* __restrict__ maxbits) {
  const int b = blockIdx.y;
  const int64_t n = len[b];
  const double* row = x + off[b];
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmax(m, fabs(row[i]));
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s, 64));
  // one atomic per workgroup: the maxima of all rows share a cache line, and one L2 channel serves every operation on it
  __shared__ double s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(&maxbits[b], (unsigned long long)__double_as_longlong(fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]))));
}

// e = (x / top)^2 when top >= 1e-20 (core/decay.py:96-100), else x^2; in place
__global__ __launch_bounds__(256) void seg_square_kernel(double* __restrict__ x, const int64_t* __restrict__ off,
                                                         const int64_t* __restrict__ len,
                                                         const unsigned long long* __restrict__ maxbits) {
  const int b = blockIdx.y;
  const int64_t n = len[b];
  double* row = x + off[b];
  const double top = __longlong_as_double((long long)maxbits[b]);
  const bool norm = top >= 1e-20;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = norm ? row[i] / top : row[i];
    row[i] = v * v;
  }
}

// mean_out[q] = np.mean(e[seg][a:b]); an empty range gives NaN like NumPy.  One workgroup of 256 threads per query.
// np.add.reduce hands its inner loop the run in pieces of the ufunc buffer size (8192 elements) and adds the pieces'
// pairwise sums left to right (observed on NumPy 2.2: tests pin it against np.mean).  The pairwise sum of a piece is a
// fixed binary tree: a run longer than 128 splits at n/2 rounded down to a multiple of 8, a shorter one is summed on 8
// interleaved accumulators r_j = a[j] + a[j+8] + ..., combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 tail.  The tree is laid out level by level in LDS (wave 0), its leaves are summed by
// groups of 8 lanes - one accumulator each, the loads of a group contiguous - and the levels are folded back bottom-up
// with the left operand first, so every addition is the one NumPy performs.
constexpr int kSumNodes = 256;     // nodes per level of a piece's tree (<= 8192 / 57 leaves)
constexpr int kSumLevels = 8;      // 8192 -> 128 takes six splits

// np.mean(a[0:n]) by the 256 threads of a workgroup, every thread gets the value (all of them must call it)
__device__ inline double block_np_mean(const double* __restrict__ a, long long n) {
  const int tid = threadIdx.x, lane = tid & 63;
  __shared__ double s_mean;
  if (n <= 0) return __longlong_as_double(0x7ff8000000000000ll);
  __shared__ unsigned short s_off[kSumLevels][kSumNodes], s_len[kSumLevels][kSumNodes], s_child[kSumLevels][kSumNodes];
  __shared__ double s_sum[2][kSumNodes];
  __shared__ int s_levels, s_count[kSumLevels];
  double acc = 0.0;
  for (long long p0 = 0; p0 < n; p0 += 8192) {
    const int m = (int)((n - p0) < 8192 ? (n - p0) : 8192);
    const double* piece = a + p0;
    __syncthreads();                                       // the previous piece's tables are no longer read
    if (tid < 64) {                                        // wave 0 lays the tree out
      if (lane == 0) {
        s_off[0][0] = 0;
        s_len[0][0] = (unsigned short)m;
      }
      int cnt = 1, lev = 0;
      for (;; ++lev) {
        if (lane == 0) s_count[lev] = cnt;
        bool any = false;
        int carry = 0;
        for (int base = 0; base < cnt; base += 64) {
          const int k = base + lane;
          const bool valid = k < cnt;
          const int len = valid ? s_len[lev][k] : 0, o = valid ? s_off[lev][k] : 0;
          const bool split = len > 128;
          const int kids = valid ? (split ? 2 : 1) : 0;
          int incl = kids;                                 // inclusive scan over the wave
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
          }
          const int pos = carry + incl - kids;
          any = any || __any(split);
          if (valid && lev + 1 < kSumLevels) {
            s_child[lev][k] = (unsigned short)(pos | (split ? 0x8000 : 0));
            if (split) {
              int n2 = len / 2;
              n2 -= n2 % 8;
              s_off[lev + 1][pos] = (unsigned short)o;
              s_len[lev + 1][pos] = (unsigned short)n2;
              s_off[lev + 1][pos + 1] = (unsigned short)(o + n2);
              s_len[lev + 1][pos + 1] = (unsigned short)(len - n2);
            } else {
              s_off[lev + 1][pos] = (unsigned short)o;
              s_len[lev + 1][pos] = (unsigned short)len;
            }
          }
          carry += __shfl(incl, 63, 64);
        }
        if (!any || lev + 1 >= kSumLevels) break;
        cnt = carry;
      }
      if (lane == 0) s_levels = lev;                       // the leaves are the nodes of level `lev`
    }
    __syncthreads();
    const int lev = s_levels, leaves = s_count[lev];
    // leaf sums: 8 lanes per leaf, lane j owns accumulator r_j = a[j] + a[j + 8] + ... in order
    for (int leaf = tid >> 3; leaf < leaves; leaf += 32) {
      const int j = tid & 7, len = s_len[lev][leaf];
      const double* p = piece + s_off[lev][leaf];
      double res;
      if (len < 8) {
        res = 0.0;
        for (int i = 0; i < len; ++i) res += p[i];         // every lane of the group computes the same serial sum
      } else {
        const int body = len - (len % 8);
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (j + 8 * u) < body ? p[j + 8 * u] : 0.0;     // all loads in flight
        double r = v[0];
#pragma unroll
        for (int u = 1; u < 16; ++u)
          if (j + 8 * u < body) r += v[u];
        r += __shfl_xor(r, 1, 64);                         // (r0 + r1), (r2 + r3), ...
        r += __shfl_xor(r, 2, 64);                         // (r0 + r1) + (r2 + r3), ...
        r += __shfl_xor(r, 4, 64);                         // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
        res = r;
        for (int i = body; i < len; ++i) res += p[i];
      }
      if (j == 0) s_sum[0][leaf] = res;
    }
    __syncthreads();
    int cur = 0;
    for (int l = lev - 1; l >= 0; --l) {                   // fold the levels back: parent = left + right
      for (int k = tid; k < s_count[l]; k += 256) {
        const unsigned short c = s_child[l][k];
        const int pos = c & 0x7fff;
        s_sum[cur ^ 1][k] = (c & 0x8000) ? s_sum[cur][pos] + s_sum[cur][pos + 1] : s_sum[cur][pos];
      }
      cur ^= 1;
      __syncthreads();
    }
    if (tid == 0) acc = p0 == 0 ? s_sum[cur][0] : acc + s_sum[cur][0];
  }
  if (tid == 0) s_mean = acc / (double)n;
  __syncthreads();
  return s_mean;               // (the next call's first barrier comes before anything writes s_mean again)
}

__global__ __launch_bounds__(256) void seg_range_mean_kernel(const double* __restrict__ e, const int64_t* __restrict__ off,
                                                             const int64_t* __restrict__ q_seg,
                                                             const int64_t* __restrict__ q_a,
                                                             const int64_t* __restrict__ q_b, long long Q,
                                                             double* __restrict__ mean_out) {
  const long long q = blockIdx.x;
  const double m = block_np_mean(e + off[q_seg[q]] + q_a[q], q_b[q] - q_a[q]);
  if (threadIdx.x == 0) mean_out[q] = m;
}

// ---------------------------------------------------------------------------------------------
// K7b: Schroeder decay times (core/decay.py:263-340), one workgroup per response, fp64.
//   sch[i]   = 10 log10( sum_{m=i..K} e[m] / sum_{m<K} e[m] ),  e = (x[peak+m] / max|x[peak:]|)^2,  i < K
//   smooth   = 10 log10( running_mean((part / max|part|)^2, window) + 1e-18 ), part = x[peak-lead : peak+K+trail]
//   offset   = mean(sch[a:b] - smooth[a-skew : b-skew]) over the 10 %..90 % stretch both cover
//   EDT/RT20/RT30/RT60 = span / slope of the least-squares line through (t[i], sch[i]) between the first
//   samples at or below (top, bottom) dB, when bottom >= noise_floor + offset + 10; NaN otherwise.
// The sums are tree / blocked-scan reductions: unlike the knee search nothing here is compared against a
// threshold bit for bit (log10 and the reference's BLAS-based covariance are not reproducible to the last
// ulp either); the golden decay times agree to 1e-12.
// ---------------------------------------------------------------------------------------------
struct DecayJob {
  long long off;       // first sample of the response in x
  long long n;         // its length
  long long peak;      // peak_ind
  long long K;         // knee_point_ind - peak_ind
  long long window;    // window_size
  double noise_floor;  // dB
  long long scratch;   // offset of this job's 2 (n + 2) doubles in the scratch buffer
};

constexpr int kDecayThreads = 1024;     // one workgroup per response: its threads share the scans and the line fits

__device__ inline double block_sum(double v, double* red) {
  const int t = threadIdx.x;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  for (int s = kDecayThreads / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  return red[0];
}
__device__ inline double block_max(double v, double* red) {
  const int t = threadIdx.x;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  for (int s = kDecayThreads / 2; s > 0; s >>= 1) {
    if (t < s) red[t] = fmax(red[t], red[t + s]);
    __syncthreads();
  }
  return red[0];
}
__device__ inline long long block_min_ll(long long v, long long* red) {
  const int t = threadIdx.x;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  for (int s = kDecayThreads / 2; s > 0; s >>= 1) {
    if (t < s && red[t + s] < red[t]) red[t] = red[t + s];
    __syncthreads();
  }
  return red[0];
}
// inclusive scan of f(0..len) into out[0..len): tiles of kDecayThreads consecutive elements (coalesced loads and stores,
// f once per element), a shuffle scan inside every wave, the waves' totals and the running carry through LDS
template <class F>
__device__ inline void block_scan(F f, long long len, double* out, double* red) {
  constexpr int kWaves = kDecayThreads / 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  double carry = 0.0;
  for (long long base = 0; base < len; base += kDecayThreads) {
    const long long i = base + t;
    double v = i < len ? f(i) : 0.0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double u = __shfl_up(v, d, 64);
      if (lane >= d) v += u;
    }
    __syncthreads();
    if (lane == 63) red[w] = v;
    __syncthreads();
    double before = carry, total = 0.0;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) {
      const double r = red[k];
      if (k < w) before += r;
      total += r;
    }
    if (i < len) out[i] = v + before;
    carry += total;
  }
  __syncthreads();
}

// Sample = double (host responses, uploaded) or float (device-resident rows: converted exactly on load, so a row gives the
// bits its float64 copy would).  A job with window < 1 (a row without a knee search behind it) is left undefined.
template <class Sample>
__global__ __launch_bounds__(kDecayThreads) void decay_times_kernel(const Sample* __restrict__ x, const DecayJob* __restrict__ jobs,
                                                          double* __restrict__ scratch, double fs,
                                                          double* __restrict__ out) {
  __shared__ double red[kDecayThreads];
  __shared__ long long redl[kDecayThreads];
  const DecayJob jb = jobs[blockIdx.x];
  struct View {
    const Sample* __restrict__ p;
    __device__ __forceinline__ double operator[](long long i) const { return (double)p[i]; }
  };
  const View ir{x + jb.off};
  const long long n = jb.n, peak = jb.peak, K = jb.K;
  const int t = threadIdx.x;
  double* res = out + 4 * (long long)blockIdx.x;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  if (t < 4) res[t] = nan;
  if (K < 1 || peak < 0 || peak + K >= n + 1 || n < 2 || jb.window < 1) return;          // nothing to integrate: all undefined
  double* sch = scratch + jb.scratch;            // K + 1 values
  double* smo = sch + (n + 2);                   // prefix sums, then the smoothed level

  // ---- Schroeder backward integral
  double m = 0.0;
  for (long long i = peak + t; i < n; i += kDecayThreads) m = fmax(m, fabs(ir[i]));
  const double m1 = block_max(m, red);
  auto e = [&](long long i) {                    // i relative to the peak
    const double v = fabs(ir[peak + i] / m1);
    return v * v;
  };
  double part_sum = 0.0;
  for (long long i = t; i < K; i += kDecayThreads) part_sum += e(i);
  const double S = block_sum(part_sum, red);
  const long long last = (peak + K < n) ? K : K - 1;                     // analytical[K] exists unless the knee is the end
  // tmp[j] = sum_{m = last - j .. last} e[m] / S  (reversed order), stored so that sch[i] = tmp over m >= i
  block_scan([&](long long j) { return e(last - j) / S; }, last + 1, smo, red);
  // cumsum(...)[:0:-1] drops the first partial sum: len(schroeder) = last
  const long long Ks = last;
  for (long long i = t; i < Ks; i += kDecayThreads) sch[i] = 10.0 * log10(smo[last - i]);
  __syncthreads();

  // ---- moving average of the squared response around the same stretch
  const long long half = jb.window / 2;
  const long long lead = half < peak ? half : peak;
  long long trail = n - (peak + K);
  if (half < trail) trail = half;
  if (trail < 0) trail = 0;
  const long long skew = half - lead;
  const long long p0 = peak - lead, P = lead + K + trail, N = jb.window;
  m = 0.0;
  for (long long i = t; i < P; i += kDecayThreads) m = fmax(m, fabs(ir[p0 + i]));
  const double m2 = block_max(m, red);
  const long long Ls = (N >= 1 && P >= N) ? P - N + 1 : 0;               // len(running_mean)
  double offset = nan;
  long long a = (long long)((double)Ks * 0.1), b = (long long)((double)Ks * 0.9);
  if (a < skew) a = skew;
  if (b > skew + Ls) b = skew + Ls;
  if (Ls > 0 && a < b) {
    // c[i + 1] = c[i] + p2[i]; smo[] holds c[1..P]
    block_scan([&](long long i) { const double v = ir[p0 + i] / m2; return v * v; }, P, smo, red);
    double acc = 0.0;
    for (long long i = a + t; i < b; i += kDecayThreads) {
      const long long r = i - skew;                                      // index into the running mean
      const double hi = smo[r + N - 1], lo = r > 0 ? smo[r - 1] : 0.0;
      acc += sch[i] - 10.0 * log10((hi - lo) / (double)N + 1e-18);
    }
    offset = block_sum(acc, red) / (double)(b - a);
  }

  // ---- decay times
  const double stop = (double)n / fs, step = stop / (double)(n - 1);
  auto tt = [&](long long i) { return (i == n - 1) ? stop : (double)i * step; };
  const double tops[4] = {-1.0, -5.0, -5.0, -5.0}, bots[4] = {-10.0, -25.0, -35.0, -65.0}, spans[4] = {-10.0, -20.0, -30.0, -60.0};
  for (int k = 0; k < 4; ++k) {
    if (bots[k] < jb.noise_floor + offset + 10.0) continue;              // < 10 dB above the floor (a NaN offset passes, as in the reference)
    long long it = 0x7fffffffffffffffll, ib = 0x7fffffffffffffffll;
    for (long long i = t; i < Ks; i += kDecayThreads) {
      if (sch[i] <= tops[k] && i < it) it = i;
      if (sch[i] <= bots[k] && i < ib) ib = i;
    }
    it = block_min_ll(it, redl);
    ib = block_min_ll(ib, redl);
    if (it == 0x7fffffffffffffffll || ib == 0x7fffffffffffffffll || ib - it < 2) continue;
    const double cnt = (double)(ib - it);
    double sx = 0.0, sy = 0.0;
    for (long long i = it + t; i < ib; i += kDecayThreads) {
      sx += tt(i);
      sy += sch[i];
    }
    const double xm = block_sum(sx, red) / cnt;
    const double ym = block_sum(sy, red) / cnt;
    double sxy = 0.0, sxx = 0.0;
    for (long long i = it + t; i < ib; i += kDecayThreads) {
      const double dx = tt(i) - xm;
      sxy += dx * (sch[i] - ym);
      sxx += dx * dx;
    }
    const double cxy = block_sum(sxy, red), cxx = block_sum(sxx, red);
    if (t == 0) res[k] = spans[k] / (cxy / cxx);
  }
}

// ---------------------------------------------------------------------------------------------
// K11: cascaded second-order sections, scipy.signal.sosfilt(sos, x) with zero initial state
// (core/virtual_bass.py:121-176: Butterworth crossover / shelf filters over every response), fp64.
// SciPy's loop (scipy/signal/_sosfilt.pyx), per sample and section, direct form II transposed:
//     y  = b0 x + z0 ;  z0 = (b1 x - a1 y) + z1 ;  z1 = b2 x - a2 y ;  x <- y
// with separately rounded products and sums (the wheels' scalar code has no FMA), reproduced here with
// contraction switched off for the kernel (the _rn intrinsics alone still fuse under -ffp-contract=fast):
// results are bit-identical.
// One wave per row: the lanes load 64 samples at a time, every lane runs the (scalar) recurrence on the
// broadcast samples, lane i keeps output i, and the block is stored coalesced.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxSections = 8;       // per launch; longer cascades run as consecutive launches (same arithmetic)

__global__ __launch_bounds__(64) void sosfilt_kernel(const double* __restrict__ sos, int n_sections,
                                                     const double* __restrict__ x, double* __restrict__ y,
                                                     const int64_t* __restrict__ off, const int64_t* __restrict__ len) {
#pragma clang fp contract(off)          // products and sums round separately, as in SciPy's scalar loop
  const int row = blockIdx.x, lane = threadIdx.x;
  const int64_t n = len[row];
  const double* xr = x + off[row];
  double* yr = y + off[row];
  double b0[kMaxSections], b1[kMaxSections], b2[kMaxSections], a1[kMaxSections], a2[kMaxSections];
  double z0[kMaxSections], z1[kMaxSections];
#pragma unroll
  for (int s = 0; s < kMaxSections; ++s) {
    const bool on = s < n_sections;
    b0[s] = on ? sos[6 * s + 0] : 1.0;
    b1[s] = on ? sos[6 * s + 1] : 0.0;
    b2[s] = on ? sos[6 * s + 2] : 0.0;
    a1[s] = on ? sos[6 * s + 4] : 0.0;
    a2[s] = on ? sos[6 * s + 5] : 0.0;
    z0[s] = 0.0;
    z1[s] = 0.0;
  }
  for (int64_t base = 0; base < n; base += 64) {
    const double xv = (base + lane < n) ? xr[base + lane] : 0.0;
    double yv = 0.0;
    const int cnt = (n - base < 64) ? (int)(n - base) : 64;
    for (int i = 0; i < cnt; ++i) {
      double cur = __shfl(xv, i, 64);
#pragma unroll
      for (int s = 0; s < kMaxSections; ++s) {          // compile-time indices keep the state in registers
        if (s < n_sections) {
          const double out = b0[s] * cur + z0[s];
          z0[s] = (b1[s] * cur - a1[s] * out) + z1[s];
          z1[s] = b2[s] * cur - a2[s] * out;
          cur = out;
        }
      }
      if (lane == i) yv = cur;
    }
    if (base + lane < n) yr[base + lane] = yv;
  }
}

}  // namespace imp
