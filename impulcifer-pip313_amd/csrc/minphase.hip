// K6 - batched minimum-phase FIR design on the GPU, in fp64.
//
// Replaces, for B channels at once, the tail of FrequencyResponse.minimum_phase_impulse_response
// (reference autoeq/frequency_response.py:676-680):
//     ir = scipy.signal.firwin2(2n, f, gain, fs=fs)          # f = linspace(0, fs//2, n)
//     ir = scipy.signal.minimum_phase(ir, n_fft=len(ir))     # homomorphic, half=True  -> n taps
// as called per channel by core/parallel_workers.py:129 (n = 9 600 @48 kHz, 19 200 @96 kHz).
//
// firwin2 : fx = interp(linspace(0, nyq, 1 + 2^ceil(log2 2n)), f, gain) ; irfft(fx * linear-phase
//           shift)[:2n] * hamming(2n)
// minimum_phase : |FFT_2n| -> + 1e-7 min>0 -> 0.5 log -> IFFT -> causal cepstral window -> FFT -> exp
//           -> IFFT -> real[:n]
//
// Why fp64: the design forces a zero at Nyquist, so |H| there is rounding noise (1e-12) that the log
// turns into a -27 spike; SciPy's own result moves by 5e-9 of the FIR peak for a 1-ulp change of the
// input (tests/test_oracle_golden.py).  In fp32 that bin would be 1e-7 noise and the taps would move
// by ~1e-4.  The work is tiny (4 transforms of 19 200/38 400 points + one of 2^16/2^17 per channel),
// so it is laid out for simplicity: a batched Stockham autosort FFT, one launch per radix pass
// (radix 8/4/2/3/5/11, generic O(R^2) butterflies with exact table twiddles), ping-ponging two global
// buffers, and a few elementwise kernels.  HBM-bound streaming passes; no LDS, no MFMA.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>

#include "internal.h"
#include "fft64.hip.h"

typedef double2 cdbl;

namespace {

__device__ __forceinline__ cdbl zmul(cdbl a, cdbl b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// One Stockham pass of radix R over `batch` transforms of length N (blockIdx.y = transform).
//   n = current sub-transform length, s = N / n interleaved sub-transforms, m = n / R
//   a_k = x[q + s (p + k m)] ; b_j = sum_k a_k w_R^(j k) ; y[q + s (R p + j)] = b_j w_n^(p j)
// roots[k] = exp(-2 pi i k / N); dir = +1 uses the conjugates.
template <int R>
__global__ __launch_bounds__(256) void stockham_pass(const cdbl* __restrict__ x, cdbl* __restrict__ y,
                                                     const cdbl* __restrict__ roots, int N, int n, int s, int dir) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = n / R;
  if (i >= s * m) return;
  const int q = i % s, p = i / s;
  const long long base = (long long)blockIdx.y * N;
  cdbl a[R];
#pragma unroll
  for (int k = 0; k < R; ++k) a[k] = x[base + q + (long long)s * (p + k * m)];
  const int step_r = N / R;        // w_R = roots[step_r]
  const int step_n = N / n;        // w_n = roots[step_n]
  cdbl wr[R];                      // the R-th roots once per thread; (j k) % R is a compile-time index below
#pragma unroll
  for (int k = 0; k < R; ++k) {
    wr[k] = roots[k * step_r];
    if (dir > 0) wr[k].y = -wr[k].y;
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    cdbl acc = a[0];
#pragma unroll
    for (int k = 1; k < R; ++k) {
      const cdbl w = wr[(j * k) % R];
      const cdbl t = zmul(a[k], w);
      acc.x += t.x;
      acc.y += t.y;
    }
    cdbl tw = roots[(int)(((long long)p * j * step_n) % N)];
    if (dir > 0) tw.y = -tw.y;
    y[base + q + (long long)s * (R * p + j)] = zmul(acc, tw);
  }
}

// firwin2 front: fx = np.interp(x_i, f, gain) on the two uniform grids, times the linear-phase
// shift, extended to the Hermitian spectrum of length nirf = 2 (nfreqs - 1) for a complex IFFT.
__global__ __launch_bounds__(256) void firwin2_spectrum(const double* __restrict__ gain, cdbl* __restrict__ spec,
                                                        int n, int nfreqs, double nyq_f /* fs//2 grid end */,
                                                        double nyq /* fs/2 */, int numtaps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nfreqs) return;
  const double* g = gain + (long long)blockIdx.y * n;
  cdbl* out = spec + (long long)blockIdx.y * (2 * (nfreqs - 1));
  // np.linspace(0, stop, num): arange(num) * (stop / (num - 1)), last element forced to stop
  const double step_x = nyq / (double)(nfreqs - 1);
  const double step_f = nyq_f / (double)(n - 1);
  const double xi = (i == nfreqs - 1) ? nyq : (double)i * step_x;
  // np.interp: j = last index with f[j] <= x ; f[j] = j * step_f (f[n-1] = nyq_f)
  double val;
  if (xi >= nyq_f) {
    val = g[n - 1];
  } else {
    int j = (int)(xi / step_f);
    if (j > n - 2) j = n - 2;
    auto fj = [&](int jj) { return (jj == n - 1) ? nyq_f : (double)jj * step_f; };
    while (j > 0 && fj(j) > xi) --j;
    while (j < n - 2 && fj(j + 1) <= xi) ++j;
    const double slope = (g[j + 1] - g[j]) / (fj(j + 1) - fj(j));
    val = slope * (xi - fj(j)) + g[j];
  }
  // shift = exp(-(numtaps - 1)/2 * 1j * pi * x / nyq).  NumPy divides a complex array by a real
  // scalar as a multiplication by the reciprocal (scl = 1/nyq), so the phase is (c pi x) * (1/nyq);
  // a true division differs by 1 ulp of a ~3e4 rad angle (3.6e-12) in ~20 % of the bins, which is
  // enough to move the taps' alternating sum (= |H| at the forced Nyquist zero, 1e-11) by 25 %.
  const double scl = 1.0 / nyq;
  const double ang = (-((double)(numtaps - 1) / 2.0) * M_PI * xi) * scl;
  double sn, cs;
  sincos(ang, &sn, &cs);
  const cdbl v = make_double2(val * cs, val * sn);
  const int nirf = 2 * (nfreqs - 1);
  // irfft ignores the imaginary parts of the DC and Nyquist bins
  if (i == 0 || i == nfreqs - 1) {
    out[i] = make_double2(v.x, 0.0);
  } else {
    out[i] = v;
    out[nirf - i] = make_double2(v.x, -v.y);
  }
}

// ir[k] = Re(ifft)[k] / nirf * hamming(numtaps)[k], as a complex sequence of length numtaps
__global__ __launch_bounds__(256) void firwin2_window(const cdbl* __restrict__ time, cdbl* __restrict__ ir, int nirf,
                                                      int numtaps) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= numtaps) return;
  const double w = 0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(numtaps - 1));
  const double v = time[(long long)blockIdx.y * nirf + k].x / (double)nirf;
  ir[(long long)blockIdx.y * numtaps + k] = make_double2(v * w, 0.0);
}

// mag = |H| in place (.x), per-transform minimum of the strictly positive magnitudes.  The minima of all transforms sit
// in ONE cache line, and every operation on it - atomic or plain load - is served by one L2 channel at ~10 ns apiece:
// one per wave (4 800 at 16 x 19 200 points) took 43 - 60 us.  A workgroup now covers kMagPerThread x 256 points, folds
// its four waves in LDS and sends one atomic, after a plain read that lets all but a few skip even that.
constexpr int kMagPerThread = 8;
__global__ __launch_bounds__(256) void magnitude_and_min(cdbl* __restrict__ h, unsigned long long* __restrict__ minbits,
                                                         int N) {
  double m = INFINITY;
#pragma unroll
  for (int u = 0; u < kMagPerThread; ++u) {
    const int k = (blockIdx.x * kMagPerThread + u) * 256 + threadIdx.x;
    if (k < N) {
      cdbl* p = h + (long long)blockIdx.y * N + k;
      const double mag = hypot(p->x, p->y);
      *p = make_double2(mag, 0.0);
      if (mag > 0.0) m = fmin(m, mag);
    }
  }
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) m = fmin(m, __shfl_xor(m, sft, 64));
  __shared__ double s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmin(fmin(s_m[0], s_m[1]), fmin(s_m[2], s_m[3]));
    // positive doubles order like their bit patterns; a stale plain read only means an atomic that changes nothing
    if (m != INFINITY &&
        (unsigned long long)__double_as_longlong(m) < __atomic_load_n(&minbits[blockIdx.y], __ATOMIC_RELAXED))
      atomicMin(&minbits[blockIdx.y], (unsigned long long)__double_as_longlong(m));
  }
}

// x = 0.5 * log(mag + 1e-7 * min)
__global__ __launch_bounds__(256) void half_log(cdbl* __restrict__ h, const unsigned long long* __restrict__ minbits, int N) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const double mn = __longlong_as_double((long long)minbits[blockIdx.y]);
  cdbl* p = h + (long long)blockIdx.y * N + k;
  *p = make_double2(0.5 * log(p->x + 1e-7 * mn), 0.0);
}

// cepstrum (unnormalised IFFT output) -> real part / N * homomorphic window (1, 2.., 0..)
__global__ __launch_bounds__(256) void cepstral_window(cdbl* __restrict__ c, int N) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const int stop = N / 2;
  double w = 0.0;
  if (k == 0) w = 1.0;
  else if (k < stop) w = 2.0;
  else if (k == stop && (N & 1)) w = 1.0;
  cdbl* p = c + (long long)blockIdx.y * N + k;
  *p = make_double2(p->x / (double)N * w, 0.0);
}

__global__ __launch_bounds__(256) void complex_exp(cdbl* __restrict__ c, int N) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  cdbl* p = c + (long long)blockIdx.y * N + k;
  const double e = exp(p->x);
  double sn, cs;
  sincos(p->y, &sn, &cs);
  *p = make_double2(e * cs, e * sn);
}

__global__ __launch_bounds__(256) void take_real(const cdbl* __restrict__ c, double* __restrict__ out, int N, int ntaps) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ntaps) return;
  out[(long long)blockIdx.y * ntaps + k] = c[(long long)blockIdx.y * N + k].x / (double)N;
}

// The same elementwise steps as load hooks of the tile transform (fft64.hip.h): op(value read, transform, index within it).
struct WindowIn {                    // firwin2_window: the value comes from the [B][nirf] time-domain buffer
  int nirf, numtaps;
  __device__ __forceinline__ cdbl operator()(cdbl t, long long, long long k) const {
    const double w = 0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(numtaps - 1));
    const double v = t.x / (double)nirf;
    return make_double2(v * w, 0.0);
  }
};
struct HalfLogIn {                   // half_log
  const unsigned long long* __restrict__ minbits;
  __device__ __forceinline__ cdbl operator()(cdbl h, long long b, long long) const {
    const double mn = __longlong_as_double((long long)minbits[b]);
    return make_double2(0.5 * log(h.x + 1e-7 * mn), 0.0);
  }
};
struct CepstralIn {                  // cepstral_window
  int N;
  __device__ __forceinline__ cdbl operator()(cdbl c, long long, long long k) const {
    const int stop = N / 2;
    double w = 0.0;
    if (k == 0) w = 1.0;
    else if (k < stop) w = 2.0;
    else if (k == stop && (N & 1)) w = 1.0;
    return make_double2(c.x / (double)N * w, 0.0);
  }
};
struct ExpIn {                       // complex_exp
  __device__ __forceinline__ cdbl operator()(cdbl c, long long, long long) const {
    const double e = exp(c.x);
    double sn, cs;
    sincos(c.y, &sn, &cs);
    return make_double2(e * cs, e * sn);
  }
};

// ---- K2 when n is a product of the radices above (crop_tails leaves next_fast_len lengths): the DFT itself ----
__global__ __launch_bounds__(256) void real_to_complex(const double* __restrict__ x, cdbl* __restrict__ a, int n) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  a[(long long)blockIdx.y * n + m] = make_double2(x[(long long)blockIdx.y * n + m], 0.0);
}

// out = 20 log10 |X[k]| for k < half (no epsilon: -inf for exact zeros)
__global__ __launch_bounds__(256) void direct_post_db(const cdbl* __restrict__ X, double* __restrict__ out, int n, int half) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= half) return;
  const cdbl v = X[(long long)blockIdx.y * n + k];
  out[(long long)blockIdx.y * half + k] = 20.0 * log10(hypot(v.x, v.y));
}

// ---- K2: arbitrary-length DFT by Bluestein's chirp-z identity --------------------------------
// a[m] = x[m] c[m] (zero padded to Mfft), c[m] = exp(-i pi m^2 / n)
__global__ __launch_bounds__(256) void bluestein_pre(const double* __restrict__ x, const cdbl* __restrict__ chirp,
                                                     cdbl* __restrict__ a, int n, int mfft) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= mfft) return;
  cdbl v = make_double2(0.0, 0.0);
  if (m < n) {
    const double xv = x[(long long)blockIdx.y * n + m];
    const cdbl c = chirp[m];
    v = make_double2(xv * c.x, xv * c.y);
  }
  a[(long long)blockIdx.y * mfft + m] = v;
}

__global__ __launch_bounds__(256) void pointwise_mul(cdbl* __restrict__ a, const cdbl* __restrict__ b, int mfft) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= mfft) return;
  cdbl* p = a + (long long)blockIdx.y * mfft + m;
  *p = zmul(*p, b[m]);
}

// X[k] = c[k] * conv[k] / Mfft ; out = 20 log10 |X[k]| for k < half (no epsilon: -inf for exact zeros)
__global__ __launch_bounds__(256) void bluestein_post_db(const cdbl* __restrict__ conv, const cdbl* __restrict__ chirp,
                                                         double* __restrict__ out, int mfft, int half) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= half) return;
  const cdbl v = zmul(conv[(long long)blockIdx.y * mfft + k], chirp[k]);
  out[(long long)blockIdx.y * half + k] = 20.0 * log10(hypot(v.x, v.y) / (double)mfft);
}

// np.max of each row of out[B][half] (NaN if the row holds one, as np.max): one workgroup of 1024 per row, eight loads
// in flight per thread (256 threads walking the row one load at a time took 37 us for two rows of 32 448)
__global__ __launch_bounds__(1024) void rows_max_kernel(const double* __restrict__ out, int half, double* __restrict__ peak) {
  const double* row = out + (long long)blockIdx.x * half;
  double m = -INFINITY;
  bool nan = false;
  for (int k0 = 0; k0 < half; k0 += 8 * 1024) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u * 1024 + (int)threadIdx.x;
      v[u] = k < half ? row[k] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      nan = nan || v[u] != v[u];
      m = v[u] > m ? v[u] : m;
    }
  }
  __shared__ double s_m[1024];
  __shared__ int s_nan;
  if (threadIdx.x == 0) s_nan = 0;
  __syncthreads();
  s_m[threadIdx.x] = m;
  if (nan) s_nan = 1;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) s_m[threadIdx.x] = s_m[threadIdx.x] > s_m[threadIdx.x + st] ? s_m[threadIdx.x] : s_m[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) peak[blockIdx.x] = s_nan ? __longlong_as_double(0x7ff8000000000000ll) : s_m[0];
}

std::vector<int> factorise(int n) {
  std::vector<int> f;
  // Every pass is one launch at the launch floor (~6 us at these sizes): radix 8 passes shorten the power-of-two part
  // (78 VGPRs; the O(R^2) butterfly of radix 16 needs 256 and gains nothing more).  Whole slice, radix 4 / 8 / 16:
  // 3.47 / 3.38 / 3.45 ms.  Two-level butterflies (16 = 4 x 4, 25 = 5 x 5 inside the thread, 126 / 155 VGPRs) were
  // tried in round 3: 34 passes per slice instead of 47, and the same 300 us - 9.4 us per radix-16 pass against 6 - 7 us
  // per radix-8 pass.  IMPULSE_HIP_FFT_MAX_RADIX overrides.
  static const int max_radix = [] {
    const char* e = std::getenv("IMPULSE_HIP_FFT_MAX_RADIX");
    return e ? std::atoi(e) : 8;
  }();
  while (max_radix >= 16 && n % 16 == 0) { f.push_back(16); n /= 16; }
  while (max_radix >= 8 && n % 8 == 0) { f.push_back(8); n /= 8; }
  while (n % 4 == 0) { f.push_back(4); n /= 4; }
  while (n % 2 == 0) { f.push_back(2); n /= 2; }
  while (n % 3 == 0) { f.push_back(3); n /= 3; }
  while (n % 5 == 0) { f.push_back(5); n /= 5; }
  while (n % 11 == 0) { f.push_back(11); n /= 11; }      // 66-row convolution plans (filter spectrum preparation)
  if (n != 1) f.clear();
  return f;
}

}  // namespace

struct MinPhasePlan {
  int n = 0;            // taps out
  int numtaps = 0;      // 2n
  int nfreqs = 0;       // 1 + 2^ceil(log2 numtaps)
  int nirf = 0;         // 2 (nfreqs - 1)
  double nyq = 0, nyq_f = 0;
  std::vector<int> fac_tap, fac_irf;
  cdbl* roots_tap = nullptr;
  cdbl* roots_irf = nullptr;
  // work buffers for up to cap channels
  int64_t cap = 0;
  cdbl *a = nullptr, *b = nullptr;
  double* gain = nullptr;
  double* out = nullptr;
  unsigned long long* minbits = nullptr;
};

static int upload_roots(cdbl** dptr, int N, hipStream_t s) {
  std::vector<cdbl> h((size_t)N);
  for (int k = 0; k < N; ++k) {
    const double ang = -2.0 * M_PI * (double)k / (double)N;
    h[(size_t)k] = make_double2(std::cos(ang), std::sin(ang));
  }
  HIP_TRY(hipMalloc((void**)dptr, (size_t)N * sizeof(cdbl)));
  HIP_TRY(hipMemcpyAsync(*dptr, h.data(), (size_t)N * sizeof(cdbl), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

static void plan_free(MinPhasePlan* p) {
  if (!p) return;
  (void)hipFree(p->roots_tap);
  (void)hipFree(p->roots_irf);
  (void)hipFree(p->a);
  (void)hipFree(p->b);
  (void)hipFree(p->gain);
  (void)hipFree(p->out);
  (void)hipFree(p->minbits);
  delete p;
}

void minphase_plans_destroy(imp_ctx* ctx) {
  for (auto& kv : ctx->minphase_plans) plan_free(kv.second);
  ctx->minphase_plans.clear();
}

// one launch per radix pass through global memory (kept for lengths the tile transform does not take, and as its cross-check:
// IMPULSE_HIP_FFT64_GENERIC=1)
static int run_fft_passes(imp_ctx* ctx, const std::vector<int>& fac, const cdbl* roots, int N, int64_t B, int dir, cdbl** cur,
                          cdbl** other) {
  int n = N, s = 1;
  for (int r : fac) {
    const int threads = N / r;
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)B), block(256);
    switch (r) {
      case 16: hipLaunchKernelGGL(stockham_pass<16>, grid, block, 0, ctx->stream, *cur, *other, roots, N, n, s, dir); break;
      case 8: hipLaunchKernelGGL(stockham_pass<8>, grid, block, 0, ctx->stream, *cur, *other, roots, N, n, s, dir); break;
      case 4: hipLaunchKernelGGL(stockham_pass<4>, grid, block, 0, ctx->stream, *cur, *other, roots, N, n, s, dir); break;
      case 2: hipLaunchKernelGGL(stockham_pass<2>, grid, block, 0, ctx->stream, *cur, *other, roots, N, n, s, dir); break;
      case 3: hipLaunchKernelGGL(stockham_pass<3>, grid, block, 0, ctx->stream, *cur, *other, roots, N, n, s, dir); break;
      case 5: hipLaunchKernelGGL(stockham_pass<5>, grid, block, 0, ctx->stream, *cur, *other, roots, N, n, s, dir); break;
      case 11: hipLaunchKernelGGL(stockham_pass<11>, grid, block, 0, ctx->stream, *cur, *other, roots, N, n, s, dir); break;
      default: return fail(IMP_ERR_UNSUPPORTED, "radix %d", r);
    }
    HIP_TRY(hipGetLastError());
    std::swap(*cur, *other);
    n /= r;
    s *= r;
  }
  return IMP_OK;
}

// one pass of the tile transform (fft64.hip.h)
template <int T, class InOp, class OutOp>
static int fft64_launch(imp_ctx* ctx, const fft64::Args& a, InOp in_op, OutOp out_op) {
  auto kern = fft64::tile_kernel<T, InOp, OutOp>;
  const size_t lds = fft64::tile_lds(a.P, T);
  int rc = ctx_kernel_lds(ctx, reinterpret_cast<const void*>(kern), (size_t)160 * 1024);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)((a.n_groups + T - 1) / T)), dim3(256), lds, ctx->stream, a, in_op, out_op);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

template <class InOp, class OutOp>
static int fft64_pass(imp_ctx* ctx, fft64::Args a, int64_t B, InOp in_op, OutOp out_op) {
  a.n_groups = (long long)B * a.nvec;
  if ((double)a.n_groups * (double)a.nvec >= 4294967296.0)
    return fail(IMP_ERR_UNSUPPORTED, "fp64 transform batch of %lld vectors: beyond the kernel's 32-bit index arithmetic", (long long)a.n_groups);
  a.m_nvec = fft64::magic_of((unsigned)a.nvec);
  for (int st = 0, blk = a.P; st < a.nstages; ++st) {
    blk /= a.radix[st];
    a.m_blk[st] = fft64::magic_of((unsigned)blk);
  }
  switch (fft64::tile_vectors(a.P, a.n_groups)) {
    case 16: return fft64_launch<16>(ctx, a, in_op, out_op);
    case 8: return fft64_launch<8>(ctx, a, in_op, out_op);
    default: return fft64_launch<4>(ctx, a, in_op, out_op);
  }
}

static bool fft64_wanted() {
  static const bool generic = [] { const char* e = std::getenv("IMPULSE_HIP_FFT64_GENERIC"); return e && e[0] == '1'; }();
  return !generic;
}

// Batched N-point transforms of B rows ([B][N], roots = exp(-2 pi i k / N)): the tile transform in one or two launches
// where N splits into factors it holds (every length of the path does), else one launch per radix pass.  The result ends
// in *cur.  in_op is applied to every value read from *cur, out_op to every value of the result (elementwise hooks, so
// that the kernels between two transforms need no launch and no pass over memory of their own).
template <class InOp, class OutOp>
static int run_fft_ops(imp_ctx* ctx, const std::vector<int>& fac, const cdbl* roots, int N, int64_t B, int dir, cdbl** cur,
                       cdbl** other, InOp in_op, OutOp out_op, bool* used_tiles = nullptr, int64_t in_pitch = 0) {
  const fft64::Plan pl = fft64_wanted() ? fft64::make_plan(N) : fft64::Plan();
  if (used_tiles) *used_tiles = pl.ok;
  if (!pl.ok) return run_fft_passes(ctx, fac, roots, N, B, dir, cur, other);
  fft64::Args a = {};
  a.roots = roots;
  a.n_roots = N;
  a.dir = dir;
  a.in_batch = in_pitch > 0 ? in_pitch : N;              // rows of the input may be further apart than N (hooks that read a
  a.out_batch = N;                                       // longer buffer: the FIR design's window step)
  int rc;
  if (pl.P2 == 1) {
    a.in = *cur;
    a.out = *other;
    a.nvec = 1;
    a.P = N;
    a.in_vec = a.out_vec = N;
    a.in_elem = a.out_elem = 1;
    a.twiddle = 0;
    a.nstages = (int)pl.r1.size();
    for (int i = 0; i < a.nstages; ++i) a.radix[i] = pl.r1[(size_t)i];
    if ((rc = fft64_pass(ctx, a, B, in_op, out_op))) return rc;
    std::swap(*cur, *other);
    return IMP_OK;
  }
  // pass 1: the P2 columns, P1 points each, x w_N^(n2 k1) -> Y[k1][n2]
  a.in = *cur;
  a.out = *other;
  a.nvec = pl.P2;
  a.P = pl.P1;
  a.in_vec = 1;
  a.in_elem = pl.P2;
  a.out_vec = 1;
  a.out_elem = pl.P2;
  a.twiddle = 1;
  a.nstages = (int)pl.r1.size();
  for (int i = 0; i < a.nstages; ++i) a.radix[i] = pl.r1[(size_t)i];
  if ((rc = fft64_pass(ctx, a, B, in_op, fft64::NoOp{}))) return rc;
  // pass 2: the P1 rows of Y, P2 points each -> X[k1 + P1 k2]
  a.in = *other;
  a.out = *cur;
  a.in_batch = N;
  a.nvec = pl.P1;
  a.P = pl.P2;
  a.in_vec = pl.P2;
  a.in_elem = 1;
  a.out_vec = 1;
  a.out_elem = pl.P1;
  a.twiddle = 0;
  a.nstages = (int)pl.r2.size();
  for (int i = 0; i < a.nstages; ++i) a.radix[i] = pl.r2[(size_t)i];
  return fft64_pass(ctx, a, B, fft64::NoOp{}, out_op);
}

// batched FFT: result ends in *cur (either buf0 or buf1)
static int run_fft(imp_ctx* ctx, const std::vector<int>& fac, const cdbl* roots, int N, int64_t B, int dir, cdbl** cur,
                   cdbl** other) {
  return run_fft_ops(ctx, fac, roots, N, B, dir, cur, other, fft64::NoOp{}, fft64::NoOp{});
}

static int minphase_run(imp_ctx* ctx, const double* gain, const double* d_gain, int64_t B, int64_t n, double fs,
                        double* fir_out, int stage, double* d_fir_out = nullptr);

extern "C" int imp_minphase_fir(imp_ctx* ctx, const double* gain, int64_t B, int64_t n, double fs, double* fir_out) {
  return minphase_run(ctx, gain, nullptr, B, n, fs, fir_out, 2);
}

// the gains are already on the device (curves.hip: the FIR design grid computed from the equalisation curves there;
// its last column is zero by construction).  fir_out_host: the taps come back and the call waits for them;
// d_fir_out (device, [B][n]): they stay on the device and the call returns without waiting.
int minphase_fir_from_device_gain(imp_ctx* ctx, const double* d_gain, int64_t B, int64_t n, double fs, double* fir_out_host,
                                  double* d_fir_out) {
  return minphase_run(ctx, nullptr, d_gain, B, n, fs, fir_out_host, 2, d_fir_out);
}

extern "C" int imp_debug_minphase_stage(imp_ctx* ctx, const double* gain, int64_t B, int64_t n, double fs, int stage,
                                        double* out) {
  if (stage < 0 || stage > 1) return fail(IMP_ERR_INVALID, "stage must be 0 (firwin2 taps) or 1 (|FFT| of the taps)");
  return minphase_run(ctx, gain, nullptr, B, n, fs, out, stage);
}

// stage 0: out[B][2n] = firwin2 taps; stage 1: out[B][2n] = |FFT_2n(taps)|; stage 2: out[B][n] = FIR
static int minphase_run(imp_ctx* ctx, const double* gain, const double* d_gain, int64_t B, int64_t n, double fs,
                        double* fir_out, int stage, double* d_fir_out) {
  if (!ctx || (B && ((!gain && !d_gain) || (!fir_out && !d_fir_out)))) return fail(IMP_ERR_INVALID, "imp_minphase_fir: null argument");
  if (B < 0 || n < 2 || n > (1 << 20)) return fail(IMP_ERR_INVALID, "imp_minphase_fir: bad B or n");
  if (!(fs > 0)) return fail(IMP_ERR_INVALID, "imp_minphase_fir: fs must be positive");
  if (B == 0) return IMP_OK;
  for (int64_t b = 0; gain && b < B; ++b)
    if (gain[b * n + n - 1] != 0.0)
      return fail(IMP_ERR_INVALID, "A Type II filter must have zero gain at the Nyquist frequency (channel %lld)", (long long)b);
  IMP_CTX_LOCK(ctx);                               // plan buffers are shared by all callers of this context
  int rc = ctx_bind(ctx);
  if (rc) return rc;

  MinPhasePlan* p = nullptr;
  {
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    const auto key = std::make_pair((long long)n, (long long)std::llround(fs * 1000.0));
    auto it = ctx->minphase_plans.find(key);
    if (it != ctx->minphase_plans.end()) {
      p = it->second;
    } else {
      p = new (std::nothrow) MinPhasePlan();
      if (!p) return fail(IMP_ERR_ALLOC, "out of host memory");
      p->n = (int)n;
      p->numtaps = 2 * (int)n;
      int lg = 0;
      while ((1 << lg) < p->numtaps) ++lg;
      p->nfreqs = 1 + (1 << lg);
      p->nirf = 2 * (p->nfreqs - 1);
      p->nyq = 0.5 * fs;
      p->nyq_f = std::floor(fs / 2.0);                 // the reference's grid ends at fs // 2
      p->fac_tap = factorise(p->numtaps);
      p->fac_irf = factorise(p->nirf);
      if (p->fac_tap.empty() || p->fac_irf.empty()) {
        delete p;
        return fail(IMP_ERR_UNSUPPORTED, "2n = %d is not of the form 2^a 3^b 5^c", 2 * (int)n);
      }
      if ((rc = upload_roots(&p->roots_tap, p->numtaps, ctx->stream)) ||
          (rc = upload_roots(&p->roots_irf, p->nirf, ctx->stream))) {
        plan_free(p);
        return rc;
      }
      ctx->minphase_plans[key] = p;
    }
  }
  if (p->nyq_f != p->nyq)
    return fail(IMP_ERR_INVALID, "freq must start with 0 and end with fs/2. (odd fs %g: grid ends at %g)", fs, p->nyq_f);
  if (p->cap < B) {
    (void)hipFree(p->a); (void)hipFree(p->b); (void)hipFree(p->gain); (void)hipFree(p->out); (void)hipFree(p->minbits);
    p->a = p->b = nullptr; p->gain = p->out = nullptr; p->minbits = nullptr; p->cap = 0;
    const size_t len = (size_t)std::max(p->nirf, p->numtaps);
    if (hipMalloc((void**)&p->a, (size_t)B * len * sizeof(cdbl)) != hipSuccess ||
        hipMalloc((void**)&p->b, (size_t)B * len * sizeof(cdbl)) != hipSuccess ||
        hipMalloc((void**)&p->gain, (size_t)B * n * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->out, (size_t)B * n * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->minbits, (size_t)B * sizeof(unsigned long long)) != hipSuccess)
      return fail(IMP_ERR_ALLOC, "imp_minphase_fir: device allocation for %lld channels failed", (long long)B);
    p->cap = B;
  }
  hipStream_t s = ctx->stream;
  if (gain) HIP_TRY(hipMemcpyAsync(p->gain, gain, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, s));
  else HIP_TRY(hipMemcpyAsync(p->gain, d_gain, (size_t)B * n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemsetAsync(p->minbits, 0xFF, (size_t)B * sizeof(unsigned long long), s));
  auto grid_for = [&](int count) { return dim3((unsigned)((count + 255) / 256), (unsigned)B); };
  cdbl *cur = p->a, *oth = p->b;

  // firwin2
  hipLaunchKernelGGL(firwin2_spectrum, grid_for(p->nfreqs), dim3(256), 0, s, p->gain, cur, p->n, p->nfreqs, p->nyq_f,
                     p->nyq, p->numtaps);
  HIP_TRY(hipGetLastError());
  if ((rc = run_fft(ctx, p->fac_irf, p->roots_irf, p->nirf, B, +1, &cur, &oth))) return rc;
  // minimum_phase (homomorphic, half = True)
  const int N = p->numtaps;
  auto dump_real = [&](const cdbl* src) -> int {       // debug stages: real parts of [B][N]
    std::vector<cdbl> h((size_t)B * N);
    HIP_TRY(hipMemcpyAsync(h.data(), src, h.size() * sizeof(cdbl), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < h.size(); ++i) fir_out[i] = h[i].x;
    return IMP_OK;
  };
  // The elementwise steps between the transforms are the LOAD hooks of the transform that follows them (same arithmetic,
  // same values - one launch and one pass over memory less each); the debug stages and lengths the tile transform does
  // not take keep them as kernels of their own.
  const bool fused = stage == 2 && fft64_wanted() && fft64::make_plan(N).ok;
  if (fused) {
    if ((rc = run_fft_ops(ctx, p->fac_tap, p->roots_tap, N, B, -1, &cur, &oth, WindowIn{p->nirf, p->numtaps}, fft64::NoOp{}, nullptr,
                          p->nirf)))
      return rc;
    hipLaunchKernelGGL(magnitude_and_min, dim3((unsigned)((N + 256 * kMagPerThread - 1) / (256 * kMagPerThread)), (unsigned)B),
                       dim3(256), 0, s, cur, p->minbits, N);
    HIP_TRY(hipGetLastError());
    if ((rc = run_fft_ops(ctx, p->fac_tap, p->roots_tap, N, B, +1, &cur, &oth, HalfLogIn{p->minbits}, fft64::NoOp{}))) return rc;
    if ((rc = run_fft_ops(ctx, p->fac_tap, p->roots_tap, N, B, -1, &cur, &oth, CepstralIn{N}, fft64::NoOp{}))) return rc;
    if ((rc = run_fft_ops(ctx, p->fac_tap, p->roots_tap, N, B, +1, &cur, &oth, ExpIn{}, fft64::NoOp{}))) return rc;
  } else {
    hipLaunchKernelGGL(firwin2_window, grid_for(p->numtaps), dim3(256), 0, s, cur, oth, p->nirf, p->numtaps);
    HIP_TRY(hipGetLastError());
    std::swap(cur, oth);
    if (stage == 0) return dump_real(cur);
    if ((rc = run_fft(ctx, p->fac_tap, p->roots_tap, N, B, -1, &cur, &oth))) return rc;
    hipLaunchKernelGGL(magnitude_and_min, dim3((unsigned)((N + 256 * kMagPerThread - 1) / (256 * kMagPerThread)), (unsigned)B),
                       dim3(256), 0, s, cur, p->minbits, N);
    HIP_TRY(hipGetLastError());
    if (stage == 1) return dump_real(cur);
    hipLaunchKernelGGL(half_log, grid_for(N), dim3(256), 0, s, cur, p->minbits, N);
    HIP_TRY(hipGetLastError());
    if ((rc = run_fft(ctx, p->fac_tap, p->roots_tap, N, B, +1, &cur, &oth))) return rc;
    hipLaunchKernelGGL(cepstral_window, grid_for(N), dim3(256), 0, s, cur, N);
    HIP_TRY(hipGetLastError());
    if ((rc = run_fft(ctx, p->fac_tap, p->roots_tap, N, B, -1, &cur, &oth))) return rc;
    hipLaunchKernelGGL(complex_exp, grid_for(N), dim3(256), 0, s, cur, N);
    HIP_TRY(hipGetLastError());
    if ((rc = run_fft(ctx, p->fac_tap, p->roots_tap, N, B, +1, &cur, &oth))) return rc;
  }
  hipLaunchKernelGGL(take_real, grid_for(p->n), dim3(256), 0, s, cur, d_fir_out ? d_fir_out : p->out, N, p->n);
  HIP_TRY(hipGetLastError());
  if (fir_out) {
    HIP_TRY(hipMemcpyAsync(fir_out, d_fir_out ? d_fir_out : p->out, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return IMP_OK;
}


// Test hook: the batched fp64 transform the fp64 kernels share (K6, K2, filter spectra), on host data.
extern "C" int imp_debug_fft64(imp_ctx* ctx, const double* x, int64_t B, int64_t N, int dir, double* y, int* used_tiles) {
  if (!ctx || !x || !y) return fail(IMP_ERR_INVALID, "imp_debug_fft64: null argument");
  if (B < 1 || N < 2 || N > (1 << 22) || (dir != 1 && dir != -1)) return fail(IMP_ERR_INVALID, "imp_debug_fft64: bad B, N or dir");
  const std::vector<int> fac = factorise((int)N);
  if (fac.empty()) return fail(IMP_ERR_UNSUPPORTED, "imp_debug_fft64: N = %lld is not 2^a 3^b 5^c 11^d", (long long)N);
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  cdbl *a = nullptr, *b = nullptr, *roots = nullptr;
  const size_t bytes = (size_t)B * N * sizeof(cdbl);
  hipStream_t s = ctx->stream;
  auto cleanup = [&](int code) {
    (void)hipStreamSynchronize(s);
    (void)ctx_block_put(ctx, a);
    (void)ctx_block_put(ctx, b);
    (void)hipFree(roots);
    return code;
  };
  if (ctx_block_get(ctx, bytes, (void**)&a) || ctx_block_get(ctx, bytes, (void**)&b)) return cleanup(IMP_ERR_ALLOC);
  if ((rc = upload_roots(&roots, (int)N, s))) return cleanup(rc);
  if (hipMemcpyAsync(a, x, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return cleanup(fail(IMP_ERR_HIP, "imp_debug_fft64: upload failed"));
  cdbl *cur = a, *oth = b;
  bool tiles = false;
  if ((rc = run_fft_ops(ctx, fac, roots, (int)N, B, dir, &cur, &oth, fft64::NoOp{}, fft64::NoOp{}, &tiles))) return cleanup(rc);
  if (used_tiles) *used_tiles = tiles ? 1 : 0;
  if (hipMemcpyAsync(y, cur, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return cleanup(fail(IMP_ERR_HIP, "imp_debug_fft64: download failed"));
  return cleanup(IMP_OK);
}

// ------------------------------------------------------------------------------------------------
// K2: magnitude response of arbitrary-length rows (core/audio_io.py:100-113), fp64 Bluestein
// ------------------------------------------------------------------------------------------------
struct MagPlan {
  int n = 0, mfft = 0;
  bool direct = false;          // n factorises over the Stockham radices: one n-point transform, no chirp
  std::vector<int> fac;
  cdbl *roots = nullptr, *chirp = nullptr, *bhat = nullptr;
  int64_t cap = 0;
  cdbl *a = nullptr, *b = nullptr;
  double *x = nullptr, *out = nullptr;
};

static std::map<long long, MagPlan*>& mag_plans(imp_ctx* ctx) { return ctx->magnitude_plans; }

static void mag_plan_free(MagPlan* p) {
  if (!p) return;
  (void)hipFree(p->roots); (void)hipFree(p->chirp); (void)hipFree(p->bhat);
  (void)hipFree(p->a); (void)hipFree(p->b); (void)hipFree(p->x); (void)hipFree(p->out);
  delete p;
}

void magnitude_plans_destroy(imp_ctx* ctx) {
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  auto& m = mag_plans(ctx);
  for (auto& kv : m) mag_plan_free(kv.second);
  m.clear();
}

namespace {
// out[g][i] = sum over the rows of group g (in row order, zero beyond a row's end) - np.sum(np.vstack(padded), axis=0)
__global__ __launch_bounds__(256) void rows_group_sum_kernel(const float* __restrict__ src, const int64_t* __restrict__ off,
                                                             const int64_t* __restrict__ len, const int64_t* __restrict__ group,
                                                             int B, double* __restrict__ out, int64_t n) {
  const int g = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int b = 0; b < B; ++b)
      if (group[b] == g && i < len[b]) acc += (double)src[off[b] + i];
    out[(int64_t)g * n + i] = acc;
  }
}

}  // namespace

static int magnitude_db_core(imp_ctx* ctx, const double* x, const float* d_rows, const int64_t* off, const int64_t* len,
                             const int64_t* group, int64_t n_rows, int64_t B, int64_t n, double* db_out, bool peak_only);

extern "C" int imp_magnitude_db(imp_ctx* ctx, const double* x, int64_t B, int64_t n, double* db_out) {
  return magnitude_db_core(ctx, x, nullptr, nullptr, nullptr, nullptr, 0, B, n, db_out, false);
}

// Device-resident rows (fp32 at d_rows + off[r], len[r] samples, r < n_rows) are summed per group (group[r] in
// [0, n_groups), rows of a group added in row order in fp64, zero beyond a row's end: np.sum(np.vstack(padded), axis=0)
// of core/hrir.py:496-503) and the magnitude response of every sum (n points) is returned: HRIR.normalize without
// bringing the responses back to the host.
extern "C" int imp_magnitude_db_sum_device(imp_ctx* ctx, const float* d_rows, const int64_t* off, const int64_t* len,
                                           const int64_t* group, int64_t n_rows, int64_t n_groups, int64_t n,
                                           double* db_out) {
  if (!ctx || !d_rows || !off || !len || !group || n_rows < 1 || n_groups < 1)
    return fail(IMP_ERR_INVALID, "imp_magnitude_db_sum_device: bad argument");
  for (int64_t r = 0; r < n_rows; ++r)
    if (off[r] < 0 || len[r] < 0 || len[r] > n || group[r] < 0 || group[r] >= n_groups)
      return fail(IMP_ERR_INVALID, "imp_magnitude_db_sum_device: row %lld out of range", (long long)r);
  return magnitude_db_core(ctx, nullptr, d_rows, off, len, group, n_rows, n_groups, n, db_out, false);
}

// the maximum of each of those spectra only (HRIR.normalize with peak_target, core/hrir.py:505: np.max of the stacked
// spectra): peak_db_out[n_groups]; NaN if a spectrum holds one, -inf for an all-zero sum - what np.max returns
extern "C" int imp_magnitude_db_sum_peak_device(imp_ctx* ctx, const float* d_rows, const int64_t* off, const int64_t* len,
                                                const int64_t* group, int64_t n_rows, int64_t n_groups, int64_t n,
                                                double* peak_db_out) {
  if (!ctx || !d_rows || !off || !len || !group || n_rows < 1 || n_groups < 1)
    return fail(IMP_ERR_INVALID, "imp_magnitude_db_sum_peak_device: bad argument");
  for (int64_t r = 0; r < n_rows; ++r)
    if (off[r] < 0 || len[r] < 0 || len[r] > n || group[r] < 0 || group[r] >= n_groups)
      return fail(IMP_ERR_INVALID, "imp_magnitude_db_sum_peak_device: row %lld out of range", (long long)r);
  return magnitude_db_core(ctx, nullptr, d_rows, off, len, group, n_rows, n_groups, n, peak_db_out, true);
}

static int magnitude_db_core(imp_ctx* ctx, const double* x, const float* d_rows, const int64_t* off, const int64_t* len,
                             const int64_t* group, int64_t n_rows, int64_t B, int64_t n, double* db_out, bool peak_only) {
  if (!ctx || (B && n && ((!x && !d_rows) || !db_out))) return fail(IMP_ERR_INVALID, "imp_magnitude_db: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0 || n < 0 || n > (1 << 22)) return fail(IMP_ERR_INVALID, "imp_magnitude_db: bad B or n");
  if (B == 0 || n == 0) return IMP_OK;
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  const int half = (int)((n + 1) / 2);
  MagPlan* p = nullptr;
  {
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    auto& plans = mag_plans(ctx);
    auto it = plans.find((long long)n);
    if (it != plans.end()) {
      p = it->second;
    } else {
      p = new (std::nothrow) MagPlan();
      if (!p) return fail(IMP_ERR_ALLOC, "out of host memory");
      p->n = (int)n;
      if (n >= 2 && !ctx->k2_bluestein_only && !factorise((int)n).empty()) {
        p->direct = true;
        p->mfft = (int)n;
        p->fac = factorise((int)n);
        if (upload_roots(&p->roots, (int)n, ctx->stream) != IMP_OK || hipStreamSynchronize(ctx->stream) != hipSuccess) {
          mag_plan_free(p);
          return fail(IMP_ERR_HIP, "imp_magnitude_db: plan set-up for n = %lld failed", (long long)n);
        }
        plans[(long long)n] = p;
      }
    }
    if (!p->direct && !p->chirp) {
      int mf = 1;
      while (mf < 2 * (int)n - 1) mf <<= 1;
      if (mf < 4) mf = 4;
      p->mfft = mf;
      p->fac = factorise(mf);
      // chirp c[m] = exp(-i pi m^2 / n), phase reduced exactly: m^2 mod 2n
      std::vector<cdbl> c((size_t)n), bb((size_t)mf, make_double2(0.0, 0.0));
      for (int64_t m = 0; m < n; ++m) {
        const double ang = -M_PI * (double)((m * m) % (2 * n)) / (double)n;
        c[(size_t)m] = make_double2(std::cos(ang), std::sin(ang));
      }
      // b[j] = conj(c[|j|]) placed circularly at j mod mfft, j in (-n, n)
      for (int64_t j = 0; j < n; ++j) {
        const cdbl v = make_double2(c[(size_t)j].x, -c[(size_t)j].y);
        bb[(size_t)j] = v;
        if (j) bb[(size_t)(mf - j)] = v;
      }
      hipStream_t s = ctx->stream;
      bool ok = upload_roots(&p->roots, mf, s) == IMP_OK &&
                hipMalloc((void**)&p->chirp, (size_t)n * sizeof(cdbl)) == hipSuccess &&
                hipMalloc((void**)&p->bhat, (size_t)mf * sizeof(cdbl)) == hipSuccess &&
                hipMalloc((void**)&p->b, (size_t)mf * sizeof(cdbl)) == hipSuccess &&
                hipMemcpyAsync(p->chirp, c.data(), (size_t)n * sizeof(cdbl), hipMemcpyHostToDevice, s) == hipSuccess &&
                hipMemcpyAsync(p->bhat, bb.data(), (size_t)mf * sizeof(cdbl), hipMemcpyHostToDevice, s) == hipSuccess;
      if (ok) {
        cdbl *cur = p->bhat, *oth = p->b;
        ok = run_fft(ctx, p->fac, p->roots, mf, 1, -1, &cur, &oth) == IMP_OK && hipStreamSynchronize(s) == hipSuccess;
        if (ok && cur != p->bhat) std::swap(p->bhat, p->b);       // result may sit in the other buffer
      }
      if (p->b) { (void)hipFree(p->b); p->b = nullptr; }
      if (!ok) {
        mag_plan_free(p);
        return fail(IMP_ERR_HIP, "imp_magnitude_db: plan set-up for n = %lld failed", (long long)n);
      }
      plans[(long long)n] = p;
    }
  }
  if (p->cap < B) {
    (void)hipFree(p->a); (void)hipFree(p->b); (void)hipFree(p->x); (void)hipFree(p->out);
    p->a = p->b = nullptr; p->x = p->out = nullptr; p->cap = 0;
    if (hipMalloc((void**)&p->a, (size_t)B * p->mfft * sizeof(cdbl)) != hipSuccess ||
        hipMalloc((void**)&p->b, (size_t)B * p->mfft * sizeof(cdbl)) != hipSuccess ||
        hipMalloc((void**)&p->x, (size_t)B * n * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->out, (size_t)B * half * sizeof(double)) != hipSuccess)
      return fail(IMP_ERR_ALLOC, "imp_magnitude_db: device allocation failed");
    p->cap = B;
  }
  hipStream_t s = ctx->stream;
  if (x) {
    HIP_TRY(hipMemcpyAsync(p->x, x, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, s));
  } else {
    // row tables through the staging ring: one copy in stream order, no wait before the transform
    const size_t meta = (size_t)n_rows * sizeof(int64_t);
    int64_t *h_meta = nullptr, *d_meta = nullptr;
    if ((rc = ctx_stage(ctx, 3 * meta, (void**)&h_meta, (void**)&d_meta))) return rc;
    std::memcpy(h_meta, off, meta);
    std::memcpy(h_meta + n_rows, len, meta);
    std::memcpy(h_meta + 2 * n_rows, group, meta);
    if ((rc = ctx_stage_push(ctx, h_meta, d_meta, 3 * meta))) return rc;
    hipLaunchKernelGGL(rows_group_sum_kernel, dim3((unsigned)std::min<int64_t>(256, (n + 255) / 256), (unsigned)B),
                       dim3(256), 0, s, d_rows, d_meta, d_meta + n_rows, d_meta + 2 * n_rows, (int)n_rows, p->x, n);
    if (hipGetLastError() != hipSuccess) return fail(IMP_ERR_HIP, "imp_magnitude_db_sum_device: row sum failed");
  }
  auto grid_for = [&](int count) { return dim3((unsigned)((count + 255) / 256), (unsigned)B); };
  cdbl *cur = p->a, *oth = p->b;
  if (p->direct) {                                       // n = 2^a 3^b 5^c 11^d: the n-point transform itself
    hipLaunchKernelGGL(real_to_complex, grid_for(p->n), dim3(256), 0, s, p->x, cur, p->n);
    HIP_TRY(hipGetLastError());
    if ((rc = run_fft(ctx, p->fac, p->roots, p->n, B, -1, &cur, &oth))) return rc;
    hipLaunchKernelGGL(direct_post_db, grid_for(half), dim3(256), 0, s, cur, p->out, p->n, half);
    HIP_TRY(hipGetLastError());
  } else {
    hipLaunchKernelGGL(bluestein_pre, grid_for(p->mfft), dim3(256), 0, s, p->x, p->chirp, cur, p->n, p->mfft);
    HIP_TRY(hipGetLastError());
    if ((rc = run_fft(ctx, p->fac, p->roots, p->mfft, B, -1, &cur, &oth))) return rc;
    hipLaunchKernelGGL(pointwise_mul, grid_for(p->mfft), dim3(256), 0, s, cur, p->bhat, p->mfft);
    HIP_TRY(hipGetLastError());
    if ((rc = run_fft(ctx, p->fac, p->roots, p->mfft, B, +1, &cur, &oth))) return rc;
    hipLaunchKernelGGL(bluestein_post_db, grid_for(half), dim3(256), 0, s, cur, p->chirp, p->out, p->mfft, half);
    HIP_TRY(hipGetLastError());
  }
  if (peak_only) {                                       // db_out[B]: the maximum of each spectrum (p->x is free again)
    hipLaunchKernelGGL(rows_max_kernel, dim3((unsigned)B), dim3(1024), 0, s, p->out, half, p->x);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(db_out, p->x, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, s));
  } else {
    HIP_TRY(hipMemcpyAsync(db_out, p->out, (size_t)B * half * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}


// ------------------------------------------------------------------------------------------------
// K2 for imp_slice (HRIR.normalize, core/hrir.py:496-505): np.max of the magnitude response of each ear's sum, for M
// measurements whose row length n_m is known ONLY ON THE DEVICE (crop_tails decided it there).  Same chirp-z identity as
// above, but the chirp of each measurement is formed on the device from its n_m and the convolution length is fixed by
// the slice's capacity (mfft >= 2 n_max - 1), so the launch sequence does not depend on any n_m: nothing is read back.
// The two ear sums are real: they go through ONE complex transform, z = x_L + i x_R, and are separated afterwards.
// ------------------------------------------------------------------------------------------------
struct SliceNorm {
  int n_max = 0, mfft = 0, half_max = 0;
  int64_t m_cap = 0;
  std::vector<int> fac;
  cdbl* roots = nullptr;
  double* x = nullptr;       // [2 m_cap][n_max]   ear sums
  cdbl* chirp = nullptr;     // [m_cap][n_max]
  cdbl* bhat = nullptr;      // [m_cap][mfft]      transform of the conjugate chirp
  cdbl* bwork = nullptr;
  cdbl *a = nullptr, *b = nullptr;     // [m_cap][mfft]: the two ears of a measurement share a transform
  double* out = nullptr;     // [2 m_cap][half_max]
};

namespace {

// x[2 m + ear][i] = sum over the pairs q of rows[(m R + 2 q + ear) pitch + i], i < n_m, added in row order in fp64
// (np.sum(np.vstack(...), axis=0) of core/hrir.py:496-503; the rows of a measurement are equally long after crop_tails)
__global__ __launch_bounds__(256) void sn_sum_kernel(const float* __restrict__ rows, long long pitch, int rows_per_meas,
                                                     const long long* __restrict__ n_of, double* __restrict__ x, int n_max) {
  const int g = blockIdx.y, m = g >> 1, ear = g & 1;
  long long n = n_of[m];
  n = n < n_max ? n : n_max;
  const float* base = rows + ((long long)m * rows_per_meas + ear) * pitch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int q = 0; 2 * q < rows_per_meas; ++q) acc += (double)base[(long long)(2 * q) * pitch + i];
    x[(long long)g * n_max + i] = acc;
  }
}

// chirp[m][j] = exp(-i pi j^2 / n_m) (phase reduced exactly: j^2 mod 2 n_m), j < n_m; bb[m][j] = conj chirp[|j|] placed
// circularly at j mod mfft, zero elsewhere
__global__ __launch_bounds__(256) void sn_chirp_kernel(const long long* __restrict__ n_of, cdbl* __restrict__ chirp,
                                                       cdbl* __restrict__ bb, int n_max, int mfft) {
  const int m = blockIdx.y;
  long long n = n_of[m];
  n = n < n_max ? n : n_max;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= mfft) return;
  auto c_at = [&](long long k) {
    const double ang = -M_PI * (double)((k * k) % (2 * n)) / (double)n;
    double sn, cs;
    sincos(ang, &sn, &cs);
    return make_double2(cs, sn);
  };
  cdbl v = make_double2(0.0, 0.0);
  if (j < n) {
    const cdbl c = c_at(j);
    chirp[(long long)m * n_max + j] = c;
    v = make_double2(c.x, -c.y);
  } else if (j > 0 && mfft - j < n) {
    const cdbl c = c_at(mfft - j);
    v = make_double2(c.x, -c.y);
  }
  bb[(long long)m * mfft + j] = v;
}

// the two ears of a measurement travel as ONE complex signal z = x_L + i x_R (both real): a[m][j] = z[j] chirp[j]
__global__ __launch_bounds__(256) void sn_pre_kernel(const double* __restrict__ x, const cdbl* __restrict__ chirp,
                                                     const long long* __restrict__ n_of, cdbl* __restrict__ a, int n_max,
                                                     int mfft) {
  const int m = blockIdx.y;
  long long n = n_of[m];
  n = n < n_max ? n : n_max;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= mfft) return;
  cdbl v = make_double2(0.0, 0.0);
  if (j < n) {
    const cdbl z = make_double2(x[(long long)(2 * m) * n_max + j], x[(long long)(2 * m + 1) * n_max + j]);
    v = zmul(z, chirp[(long long)m * n_max + j]);
  }
  a[(long long)m * mfft + j] = v;
}

__global__ __launch_bounds__(256) void sn_mul_kernel(cdbl* __restrict__ a, const cdbl* __restrict__ bhat, int mfft) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= mfft) return;
  cdbl* p = a + (long long)blockIdx.y * mfft + j;
  *p = zmul(*p, bhat[(long long)blockIdx.y * mfft + j]);
}

// Z[k] = chirp[k] conv[k] / mfft is the n-point transform of z; the ears' transforms are its Hermitian parts:
// X_L[k] = (Z[k] + conj Z[n - k]) / 2, X_R[k] = (Z[k] - conj Z[n - k]) / 2i; out[2 m + ear][k] = 20 log10 |X_ear[k]|, k < ceil(n / 2)
__global__ __launch_bounds__(256) void sn_post_kernel(const cdbl* __restrict__ conv, const cdbl* __restrict__ chirp,
                                                      const long long* __restrict__ n_of, double* __restrict__ out, int n_max,
                                                      int mfft, int half_max) {
  const int m = blockIdx.y;
  long long n = n_of[m];
  n = n < n_max ? n : n_max;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (n + 1) / 2) return;
  const cdbl* cv = conv + (long long)m * mfft;
  const cdbl* ch = chirp + (long long)m * n_max;
  const int kk = k == 0 ? 0 : (int)(n - k);
  const cdbl zk = zmul(cv[k], ch[k]), zn = zmul(cv[kk], ch[kk]);
  const double s = 0.5 / (double)mfft;
  const cdbl xl = make_double2((zk.x + zn.x) * s, (zk.y - zn.y) * s);
  const cdbl xr = make_double2((zk.y + zn.y) * s, (zn.x - zk.x) * s);
  out[(long long)(2 * m) * half_max + k] = 20.0 * log10(hypot(xl.x, xl.y));
  out[(long long)(2 * m + 1) * half_max + k] = 20.0 * log10(hypot(xr.x, xr.y));
}

// np.max of out[g][0 : ceil(n_m / 2)] (NaN if the row holds one; -inf for an empty row)
__global__ __launch_bounds__(1024) void sn_rows_max_kernel(const double* __restrict__ out, const long long* __restrict__ n_of,
                                                           int n_max, int half_max, double* __restrict__ peak) {
  const int g = blockIdx.x;
  long long n = n_of[g >> 1];
  n = n < n_max ? n : n_max;
  const int half = (int)((n + 1) / 2);
  const double* row = out + (long long)g * half_max;
  double m = -INFINITY;
  bool nan = false;
  for (int k0 = 0; k0 < half; k0 += 8 * 1024) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u * 1024 + (int)threadIdx.x;
      v[u] = k < half ? row[k] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      nan = nan || v[u] != v[u];
      m = v[u] > m ? v[u] : m;
    }
  }
  __shared__ double s_m[1024];
  __shared__ int s_nan;
  if (threadIdx.x == 0) s_nan = 0;
  __syncthreads();
  s_m[threadIdx.x] = m;
  if (nan) s_nan = 1;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) s_m[threadIdx.x] = s_m[threadIdx.x] > s_m[threadIdx.x + st] ? s_m[threadIdx.x] : s_m[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) peak[g] = s_nan ? __longlong_as_double(0x7ff8000000000000ll) : s_m[0];
}

}  // namespace

void slice_norm_destroy(SliceNorm* p) {
  if (!p) return;
  (void)hipFree(p->roots); (void)hipFree(p->x); (void)hipFree(p->chirp); (void)hipFree(p->bhat); (void)hipFree(p->bwork);
  (void)hipFree(p->a); (void)hipFree(p->b); (void)hipFree(p->out);
  delete p;
}

int slice_norm_create(imp_ctx* ctx, int64_t n_max, int64_t m_cap, SliceNorm** out) {
  *out = nullptr;
  if (n_max < 1 || n_max > (1 << 22) || m_cap < 1) return fail(IMP_ERR_INVALID, "slice normalisation: bad n_max / capacity");
  SliceNorm* p = new (std::nothrow) SliceNorm();
  if (!p) return fail(IMP_ERR_ALLOC, "out of host memory");
  p->n_max = (int)n_max;
  p->half_max = (int)((n_max + 1) / 2);
  p->m_cap = m_cap;
  int mf = 4;
  while (mf < 2 * (int)n_max - 1) mf <<= 1;
  p->mfft = mf;
  p->fac = factorise(mf);
  const size_t M = (size_t)m_cap;
  bool ok = upload_roots(&p->roots, mf, ctx->stream) == IMP_OK &&
            hipMalloc((void**)&p->x, 2 * M * (size_t)n_max * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&p->chirp, M * (size_t)n_max * sizeof(cdbl)) == hipSuccess &&
            hipMalloc((void**)&p->bhat, M * (size_t)mf * sizeof(cdbl)) == hipSuccess &&
            hipMalloc((void**)&p->bwork, M * (size_t)mf * sizeof(cdbl)) == hipSuccess &&
            hipMalloc((void**)&p->a, M * (size_t)mf * sizeof(cdbl)) == hipSuccess &&
            hipMalloc((void**)&p->b, M * (size_t)mf * sizeof(cdbl)) == hipSuccess &&
            hipMalloc((void**)&p->out, 2 * M * (size_t)p->half_max * sizeof(double)) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    slice_norm_destroy(p);
    return fail(IMP_ERR_ALLOC, "slice normalisation: device allocation failed (n_max %lld, %lld measurements)", (long long)n_max,
                (long long)m_cap);
  }
  *out = p;
  return IMP_OK;
}

int64_t slice_norm_mfft(const SliceNorm* p) { return p->mfft; }

// d_rows: the equalised rows [M R][pitch]; d_n[m] = their length; d_peak_db[2 m + ear] receives the maxima.  Asynchronous
// on the context's stream.
int slice_norm_run(imp_ctx* ctx, SliceNorm* p, const float* d_rows, int64_t pitch, int rows_per_meas, const long long* d_n,
                   int64_t M, double* d_peak_db) {
  if (M < 1 || M > p->m_cap) return fail(IMP_ERR_INVALID, "slice normalisation: %lld measurements exceed the capacity %lld",
                                         (long long)M, (long long)p->m_cap);
  hipStream_t s = ctx->stream;
  const int mf = p->mfft, nm = p->n_max;
  auto grid = [&](int count, int64_t rows) { return dim3((unsigned)((count + 255) / 256), (unsigned)rows); };
  hipLaunchKernelGGL(sn_sum_kernel, dim3((unsigned)std::min<int>(256, (nm + 255) / 256), (unsigned)(2 * M)), dim3(256), 0, s, d_rows,
                     (long long)pitch, rows_per_meas, d_n, p->x, nm);
  hipLaunchKernelGGL(sn_chirp_kernel, grid(mf, M), dim3(256), 0, s, d_n, p->chirp, p->bhat, nm, mf);
  HIP_TRY(hipGetLastError());
  int rc;
  cdbl *cur = p->bhat, *oth = p->bwork;
  if ((rc = run_fft(ctx, p->fac, p->roots, mf, M, -1, &cur, &oth))) return rc;
  const cdbl* bhat = cur;                                  // (either buffer: both belong to the plan)
  hipLaunchKernelGGL(sn_pre_kernel, grid(mf, M), dim3(256), 0, s, p->x, p->chirp, d_n, p->a, nm, mf);
  HIP_TRY(hipGetLastError());
  cdbl *c2 = p->a, *o2 = p->b;
  if ((rc = run_fft(ctx, p->fac, p->roots, mf, M, -1, &c2, &o2))) return rc;
  hipLaunchKernelGGL(sn_mul_kernel, grid(mf, M), dim3(256), 0, s, c2, bhat, mf);
  HIP_TRY(hipGetLastError());
  if ((rc = run_fft(ctx, p->fac, p->roots, mf, M, +1, &c2, &o2))) return rc;
  hipLaunchKernelGGL(sn_post_kernel, grid(p->half_max, M), dim3(256), 0, s, c2, p->chirp, d_n, p->out, nm, mf, p->half_max);
  hipLaunchKernelGGL(sn_rows_max_kernel, dim3((unsigned)(2 * M)), dim3(1024), 0, s, p->out, d_n, nm, p->half_max, d_peak_db);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

// ------------------------------------------------------------------------------------------------
// Filter spectra of convolution plans (alpha/beta planes), fp64 on the device.
// The same arithmetic as host_rfft + host_alpha_beta in impulse_hip.hip (kept there as the debug /
// cross-check path): one packed Nc-point complex FFT per filter, real-FFT unpack, then
//   alpha = (H_k (1+s) + G_k (1-s)) / (2 Nc), beta = i c (H_k - G_k) / (2 Nc),  G_k = conj H[Nc-k],
//   s + i c ... = sin/cos(-pi k / Nc),
// rounded once to fp32 in the row pass's register order.  A 16-filter equalisation plan costs ~8 ms per
// filter on one host core; here the whole batch is a few launches.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void roots_kernel(cdbl* __restrict__ roots, int N) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  double sn, cs;
  sincospi(-2.0 * (double)k / (double)N, &sn, &cs);
  roots[k] = make_double2(cs, sn);
}

// z[f][n] = h[f][2n] + i h[f][2n+1], zero beyond M
__global__ __launch_bounds__(256) void pack_filter_kernel(const double* __restrict__ h, cdbl* __restrict__ z, long long M,
                                                          long long ld, int Nc) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Nc) return;
  const double* row = h + (long long)blockIdx.y * ld;
  const long long i = 2ll * n;
  z[(long long)blockIdx.y * Nc + n] = make_double2(i < M ? row[i] : 0.0, i + 1 < M ? row[i + 1] : 0.0);
}

__device__ __forceinline__ cdbl zconj(cdbl a) { return make_double2(a.x, -a.y); }

// H[k] of the real filter from the packed transform z (k in [0, Nc])
__device__ __forceinline__ cdbl unpack_bin(const cdbl* __restrict__ z, int k, int Nc) {
  const cdbl zk = z[k % Nc];
  const cdbl zm = zconj(z[(Nc - k) % Nc]);
  const cdbl E = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y));
  const cdbl d = make_double2(zk.x - zm.x, zk.y - zm.y);
  const cdbl O = make_double2(0.5 * d.y, -0.5 * d.x);                 // -i/2 (zk - zm)
  double sn, cs;
  sincospi(-(double)k / (double)Nc, &sn, &cs);
  const cdbl wO = zmul(make_double2(cs, sn), O);
  return make_double2(E.x + wO.x, E.y + wO.y);
}

__global__ __launch_bounds__(256) void alpha_beta_kernel(const cdbl* __restrict__ zall, float4* __restrict__ ab, int Nc,
                                                         int N1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // position in the plane: k1*4096 + q*256 + u
  if (idx >= Nc) return;
  const cdbl* z = zall + (long long)blockIdx.y * Nc;
  const int k1 = idx >> 12, r = idx & 4095, q = r >> 8, u = r & 255;
  const int k2 = (u >> 4) + 16 * (u & 15) + 256 * q;
  const long long k = (long long)k1 + (long long)N1 * k2;
  const double inv = 1.0 / (double)Nc;
  float4 o;
  if (k == 0) {
    const cdbl z0 = z[0];
    o = make_float4((float)((z0.x + z0.y) * inv), 0.f, (float)((z0.x - z0.y) * inv), 0.f);
  } else {
    const cdbl Hk = unpack_bin(z, (int)k, Nc);
    const cdbl Gk = zconj(unpack_bin(z, Nc - (int)k, Nc));
    double sn, cs;
    sincospi(-(double)k / (double)Nc, &sn, &cs);
    const double a = 0.5 * inv;
    const cdbl alpha = make_double2(a * (Hk.x * (1.0 + sn) + Gk.x * (1.0 - sn)), a * (Hk.y * (1.0 + sn) + Gk.y * (1.0 - sn)));
    const cdbl dd = make_double2(Hk.x - Gk.x, Hk.y - Gk.y);
    const cdbl beta = make_double2(-a * cs * dd.y, a * cs * dd.x);    // i (a c) (Hk - Gk)
    o = make_float4((float)alpha.x, (float)alpha.y, (float)beta.x, (float)beta.y);
  }
  ab[(long long)blockIdx.y * Nc + idx] = o;
}

// pair mode: hs[k1][q*256 + u] = H[k1 + N1 k2] / Nc, k2 = (u >> 4) + 16 (u & 15) + 256 q, over all Nc bins of the
// Nc-point transform of the real filter; z is its packed (Nc / 2)-point transform, H[Nc - k] = conj H[k]
__global__ __launch_bounds__(256) void pair_spectrum_kernel(const cdbl* __restrict__ z, float2* __restrict__ hs, int Nc, int N1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Nc) return;
  const int k1 = idx >> 12, r = idx & 4095, q = r >> 8, u = r & 255;
  const int k2 = (u >> 4) + 16 * (u & 15) + 256 * q;
  const long long k = (long long)k1 + (long long)N1 * k2;
  const int half = Nc / 2;
  cdbl H = k <= half ? unpack_bin(z, (int)k, half) : zconj(unpack_bin(z, Nc - (int)k, half));
  const double inv = 1.0 / (double)Nc;
  hs[idx] = make_float2((float)(H.x * inv), (float)(H.y * inv));
}

}  // namespace

void fft_roots_destroy(imp_ctx* ctx) {
  for (auto& kv : ctx->fft_roots) (void)hipFree(kv.second);
  ctx->fft_roots.clear();
}

int spectrum_alpha_beta_device(imp_ctx* ctx, const double* filters, int64_t M, int64_t n_filters, int64_t filter_ld,
                               int64_t Nc, int N1, float4* d_ab, bool filters_on_device) {
  const std::vector<int> fac = factorise((int)Nc);
  if (fac.empty()) return fail(IMP_ERR_UNSUPPORTED, "spectrum length %lld is not 2^a 3^b 5^c 11^d", (long long)Nc);
  hipStream_t s = ctx->stream;
  cdbl* roots = nullptr;
  auto it = ctx->fft_roots.find((long long)Nc);
  if (it != ctx->fft_roots.end()) {
    roots = (cdbl*)it->second;
  } else {
    HIP_TRY(hipMalloc((void**)&roots, (size_t)Nc * sizeof(cdbl)));
    hipLaunchKernelGGL(roots_kernel, dim3((unsigned)((Nc + 255) / 256)), dim3(256), 0, s, roots, (int)Nc);
    HIP_TRY(hipGetLastError());
    ctx->fft_roots[(long long)Nc] = roots;
  }
  // filters go through in chunks of <= 64 MiB per ping-pong buffer
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_filters, ((int64_t)64 << 20) / (Nc * (int64_t)sizeof(cdbl))));
  cdbl *a = nullptr, *b = nullptr;
  double* d_h = nullptr;
  auto cleanup = [&](int code) {
    (void)hipStreamSynchronize(s);
    (void)ctx_block_put(ctx, a);
    (void)ctx_block_put(ctx, b);
    (void)ctx_block_put(ctx, d_h);
    return code;
  };
  // (filters already on the device are read where they are and nothing below waits: the blocks go back to the pool in
  // stream order)
  auto cleanup_async = [&](int code) {
    (void)ctx_block_put(ctx, a);
    (void)ctx_block_put(ctx, b);
    return code;
  };
  if (ctx_block_get(ctx, (size_t)chunk * Nc * sizeof(cdbl), (void**)&a) ||
      ctx_block_get(ctx, (size_t)chunk * Nc * sizeof(cdbl), (void**)&b) ||
      (!filters_on_device && ctx_block_get(ctx, (size_t)chunk * M * sizeof(double), (void**)&d_h)))
    return cleanup(fail(IMP_ERR_ALLOC, "device buffers for the filter spectra (%lld filters of %lld points)",
                        (long long)chunk, (long long)Nc));
  for (int64_t f0 = 0; f0 < n_filters; f0 += chunk) {
    const int64_t nf = std::min(chunk, n_filters - f0);
    if (!filters_on_device &&
        hipMemcpy2DAsync(d_h, (size_t)M * sizeof(double), filters + f0 * filter_ld, (size_t)filter_ld * sizeof(double),
                         (size_t)M * sizeof(double), (size_t)nf, hipMemcpyHostToDevice, s) != hipSuccess)
      return cleanup(fail(IMP_ERR_HIP, "filter upload failed"));
    const dim3 grid((unsigned)((Nc + 255) / 256), (unsigned)nf);
    if (filters_on_device)
      hipLaunchKernelGGL(pack_filter_kernel, grid, dim3(256), 0, s, filters + f0 * filter_ld, a, (long long)M, (long long)filter_ld, (int)Nc);
    else
      hipLaunchKernelGGL(pack_filter_kernel, grid, dim3(256), 0, s, (const double*)d_h, a, (long long)M, (long long)M, (int)Nc);
    cdbl *cur = a, *oth = b;
    int rc = run_fft(ctx, fac, roots, (int)Nc, nf, -1, &cur, &oth);
    if (rc) return cleanup(rc);
    hipLaunchKernelGGL(alpha_beta_kernel, grid, dim3(256), 0, s, cur, d_ab + f0 * Nc, (int)Nc, N1);
    if (hipGetLastError() != hipSuccess) return cleanup(fail(IMP_ERR_HIP, "alpha/beta launch failed"));
    // the host rows of this chunk may be reused by the caller after return: drain before the next upload
    if (!filters_on_device && hipStreamSynchronize(s) != hipSuccess) return cleanup(fail(IMP_ERR_HIP, "filter spectrum: stream error"));
  }
  return filters_on_device ? cleanup_async(IMP_OK) : cleanup(IMP_OK);
}

int spectrum_pair_device(imp_ctx* ctx, const double* filter, int64_t M, int64_t Nc, int N1, cf* d_hs) {
  if (Nc % 2) return fail(IMP_ERR_UNSUPPORTED, "pair spectrum: odd circular length %lld", (long long)Nc);
  const int64_t half = Nc / 2;
  const std::vector<int> fac = factorise((int)half);
  if (fac.empty()) return fail(IMP_ERR_UNSUPPORTED, "spectrum length %lld is not 2^a 3^b 5^c 11^d", (long long)half);
  if (M > Nc) return fail(IMP_ERR_INVALID, "filter of %lld taps longer than the circular length %lld", (long long)M, (long long)Nc);
  hipStream_t s = ctx->stream;
  cdbl* roots = nullptr;
  auto it = ctx->fft_roots.find((long long)half);
  if (it != ctx->fft_roots.end()) {
    roots = (cdbl*)it->second;
  } else {
    HIP_TRY(hipMalloc((void**)&roots, (size_t)half * sizeof(cdbl)));
    hipLaunchKernelGGL(roots_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, s, roots, (int)half);
    HIP_TRY(hipGetLastError());
    ctx->fft_roots[(long long)half] = roots;
  }
  cdbl *a = nullptr, *b = nullptr;
  double* d_h = nullptr;
  auto cleanup = [&](int code) {
    (void)hipStreamSynchronize(s);
    (void)ctx_block_put(ctx, a);
    (void)ctx_block_put(ctx, b);
    (void)ctx_block_put(ctx, d_h);
    return code;
  };
  if (ctx_block_get(ctx, (size_t)half * sizeof(cdbl), (void**)&a) || ctx_block_get(ctx, (size_t)half * sizeof(cdbl), (void**)&b) ||
      ctx_block_get(ctx, (size_t)M * sizeof(double), (void**)&d_h))
    return cleanup(fail(IMP_ERR_ALLOC, "device buffers for the pair spectrum (%lld points)", (long long)Nc));
  if (hipMemcpyAsync(d_h, filter, (size_t)M * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess)
    return cleanup(fail(IMP_ERR_HIP, "filter upload failed"));
  hipLaunchKernelGGL(pack_filter_kernel, dim3((unsigned)((half + 255) / 256), 1), dim3(256), 0, s, d_h, a, (long long)M,
                     (long long)M, (int)half);
  cdbl *cur = a, *oth = b;
  int rc = run_fft(ctx, fac, roots, (int)half, 1, -1, &cur, &oth);
  if (rc) return cleanup(rc);
  hipLaunchKernelGGL(pair_spectrum_kernel, dim3((unsigned)((Nc + 255) / 256)), dim3(256), 0, s, cur,
                     reinterpret_cast<float2*>(d_hs), (int)Nc, N1);
  if (hipGetLastError() != hipSuccess) return cleanup(fail(IMP_ERR_HIP, "pair spectrum launch failed"));
  return cleanup(IMP_OK);
}
