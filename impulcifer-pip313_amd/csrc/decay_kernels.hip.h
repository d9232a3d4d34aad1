// K7c: the Lundeby knee search of core/decay.py:44-260 carried out on the device, for callers that need the knee INDEX
// only (HRIR.crop_tails, core/hrir.py:585-611) and whose responses already live on the device.
//
// The host search (impulse_hip/decay.py) keeps its control flow in NumPy because two of its primitives are not
// reproducible bit for bit anywhere else: np.log10 (SVML / libm, differs from every other log10 in the last place) and
// the BLAS dot product inside scipy.stats.linregress.  The time grids and the integer truncations are plain IEEE
// arithmetic that the device repeats exactly (this header is compiled with fp contraction off); the window means are
// tree sums within ~1e-14 dB of NumPy's pairwise ones (block_fast_mean below).  So the device search differs from the
// host search only through quantities that are a few ulp apart, and the knee it returns is an INTEGER: it can only come
// out different if one of
// the search's decisions (a comparison against a threshold, an int() truncation, an argmin on the time grid) falls within
// the width of those few ulp.  Every decision is therefore taken with a guard band derived from explicit error bounds
// (el for levels, es / ewd for slopes and window durations, propagated to times); a row with any decision inside its
// band is flagged KNEE_GUARD and the host search decides that row instead.  A clear row's knee is the host's knee by
// construction; the floor it reports is within a few ulp of the host's (crop_tails does not use it).
#pragma once
#pragma clang fp contract(off)

namespace imp {

constexpr int kKneeMaxWindows = 2048;     // second-round windows the device search takes (typical: 20 - 60)
constexpr int kKneeMaxFit = 128;          // points of one line fit (NumPy's pairwise mean is one block up to here)
constexpr int kKneeRound1 = 68;           // 66 windows of 30 ms in 2 s, the noise tail, one spare
enum { KNEE_OK = 0, KNEE_GUARD = 1, KNEE_RANGE = 2 };
// KneeRow::why (diagnostics, tools/knee_stats.py): stage 1: 1 floor + 10 dB crossing, 2 first fit, 3 window count; stage 2: 4 knee window,
// 5 level - 5 dB crossing, 6 tenth of the span, 7 noise range start, 8 noise range end, 9 floor + 8 dB, 10 floor + 28 dB, 11 late fit,
// 12 new knee window, 13 knee sample
#define KNEE_WHY(code) do { if (unsure && !r.why) r.why = (code); } while (0)

struct KneeRow {
  long long src_off;     // first sample of the response in the fp32 rows
  long long n;           // its length
  long long peak;        // result: peak index
  long long knee;        // result: knee index
  long long window;      // result: window size
  double floor;          // result: noise floor, dB
  double wd, knee_time, ekt, ewd;     // round-1 outcome: window duration, knee time and their error bounds
  int n_sq;              // analysis span: [peak, peak + n_sq)
  int n_win, w;          // the round being measured: windows [i w, (i + 1) w), i < n_win
  int tail_a, tail_b;    // round 1: the noise tail (one more mean, after the windows)
  int done;              // results are final
  int flags;             // KNEE_*
  int why;               // diagnostic: which decision put the row in its guard band first (WHY_* below), 0 = none
};

constexpr double kKneeU = 1.1102230246251565e-16;     // 2^-53
constexpr double kKneeEl = 1e-12;                     // dB: |device level - NumPy level| (log10 a few ulp apart, |level| <= 200: 3e-13; the mean's summation order: 1e-14)

__device__ inline double knee_db(double mean) { return 10 * log10(fmax(mean, 1e-20)); }

// mean(e[0:n]), e[i] = (x[i] / top)^2 (x[i]^2 when top < 1e-20: core/decay.py:96-100), straight from the fp32 row, by
// the 256 threads of a workgroup, on every thread.  NOT NumPy's summation order (block_np_mean is, at several barriers
// and a serial tree lay-out per 8 192 samples): a tree sum of non-negative fp64 terms is within ~log2(n) ulp of any
// other order - 1e-14 dB on a level, a hundredth of kKneeEl, which bounds |device level - host level| for every
// decision below.  The levels of the device search need that bound, not the host's bits.
__device__ inline double block_fast_mean_sq(const float* __restrict__ x, double top, long long n) {
  __shared__ double s_part[4];
  if (n <= 0) return __longlong_as_double(0x7ff8000000000000ll);
  const bool norm = top >= 1e-20;
  auto e = [&](long long i) {
    const double v = norm ? (double)x[i] / top : (double)x[i];
    return v * v;
  };
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  long long i = threadIdx.x;
  for (; i + 3 * 256 < n; i += 4 * 256) {                 // four loads in flight
    acc[0] += e(i);
    acc[1] += e(i + 256);
    acc[2] += e(i + 512);
    acc[3] += e(i + 768);
  }
  for (; i < n; i += 256) acc[0] += e(i);
  double v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft, 64);
  __syncthreads();                                        // the previous call's s_part is no longer read
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (double)n;
}

// NumPy's pairwise sum of n <= 128 values f(0) .. f(n - 1) (numpy/_core/src/umath/loops_utils.h.src)
template <typename F>
__device__ inline double np_sum_small(F f, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += f(i);
    return res;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f(j);
  int i;
  for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += f(i + j);
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += f(i);
  return res;
}

// np.linspace(0, n / fs, n) without the array (impulse_hip/decay.py _Grid)
struct KneeGrid {
  int n;
  double stop, step;
  __device__ KneeGrid(int n_, double fs) : n(n_), stop(n_ / fs), step(n_ > 1 ? (n_ / fs) / (n_ - 1) : 0.0) {}
  __device__ double at(int i) const { return n == 1 ? 0.0 : (i == n - 1 ? stop : i * step); }
  __device__ int nearest(double time) const {
    if (n <= 1 || step == 0.0) return 0;
    const int guess = (int)fmin(fmax(time / step, 0.0), (double)(n - 1));
    int best = -1;
    double best_d = 0.0;
    const int hi = guess + 3 < n ? guess + 3 : n;
    for (int i = guess - 2 > 0 ? guess - 2 : 0; i < hi; ++i) {
      const double d = fabs(at(i) - time);
      if (best < 0 || d < best_d) {
        best = i;
        best_d = d;
      }
    }
    return best;
  }
  // nearest(time) when time is known to +- g: sets `unsure` if the answer depends on where in the band it is
  __device__ int nearest_guarded(double time, double g, bool& unsure) const {
    const int c = nearest(time);
    if (nearest(time - g) != c || nearest(time + g) != c) unsure = true;
    return c;
  }
};

// first i < n with lev[i] <= level, -1 if none; `unsure` if a level at or before the answer is within g of the threshold
__device__ inline int knee_first_le(const double* lev, int n, double level, double g, bool& unsure) {
  for (int i = 0; i < n; ++i) {
    const double d = lev[i] - level;
    if (fabs(d) <= g) unsure = true;
    if (d <= 0.0) return i;
  }
  return -1;
}

// t_win[i] = i wd + wd / 2 (np.arange(n) * wd + wd / 2)
__device__ inline double knee_twin(int i, double wd) { return i * wd + wd / 2; }

// first i < n with t_win[i] >= time, -1 if none
__device__ inline int knee_first_ge(int n, double wd, double time, double g, bool& unsure) {
  for (int i = 0; i < n; ++i) {
    const double d = knee_twin(i, wd) - time;
    if (fabs(d) <= g) unsure = true;
    if (d >= 0.0) return i;
  }
  return -1;
}

// scipy.stats.linregress(t_win[lo:hi], lev[lo:hi]) as impulse_hip/decay.py _fit_line forms it: row means by NumPy's
// pairwise sum, centring, then the two dot products (BLAS on the host: any order, with or without fma - the bound es
// covers every order), scaling by 1 / m.  es: relative bound on |slope - host slope|, including the effect of levels
// el apart and of a window duration ewd (relative) apart.
struct KneeFit {
  double slope, icpt, es, e_icpt;
  bool ok;
};
__device__ inline KneeFit knee_fit(const double* lev, int lo, int hi, double wd, double ewd) {
  KneeFit f;
  const int m = hi - lo;
  const double xm = np_sum_small([&](int i) { return knee_twin(lo + i, wd); }, m) / m;
  const double ym = np_sum_small([&](int i) { return lev[lo + i]; }, m) / m;
  double c00 = 0.0, c01 = 0.0, a01 = 0.0, ax = 0.0;
  for (int i = 0; i < m; ++i) {
    const double xc = knee_twin(lo + i, wd) - xm, yc = lev[lo + i] - ym;
    c00 += xc * xc;
    c01 += xc * yc;
    a01 += fabs(xc * yc);
    ax += fabs(xc);
  }
  const double rn = 1.0 / m;
  c00 *= rn;
  c01 *= rn;
  f.slope = c01 / c00;
  f.icpt = ym - f.slope * xm;
  const double kappa = a01 * rn / fabs(c01);            // conditioning of the cross term
  f.ok = c00 > 0.0 && fabs(f.slope) >= 1.0 && kappa < 1e3 && f.slope == f.slope;     // dB/s: flatter fits go to the host
  f.es = 64.0 * m * kKneeU * (1.0 + kappa) + ewd + 4.0 * kKneeEl * (ax * rn) / (c00 * fabs(f.slope));
  f.e_icpt = 2.0 * fabs(xm * f.slope) * f.es + 2.0 * kKneeEl;      // slope off by es, xm by ewd <= es, ym by el
  return f;
}

// ---- the launches, in stream order --------------------------------------------------------------------------------

// 1. analysis spans from the peak search's results (impulse_hip/decay.py _knee_searches) and the first round's windows
__global__ __launch_bounds__(64) void knee_span_kernel(const RowPeak* __restrict__ res, const int64_t* __restrict__ off,
                                                       const int64_t* __restrict__ len, long long two_fs, double fs,
                                                       KneeRow* __restrict__ rows, unsigned long long* __restrict__ maxbits) {
  if (threadIdx.x) return;
  const int b = blockIdx.x;
  maxbits[b] = 0ull;
  KneeRow r = {};
  r.src_off = off[b];
  r.n = len[b];
  const float top = __uint_as_float(res[b].maxabs_bits);
  long long pk;
  if (r.n == 0 || !(top >= 1e-20f)) pk = 0;
  else if (res[b].first_peak != ~0ull) pk = (long long)res[b].first_peak;
  else pk = (long long)res[b].first_max;
  if (r.n < 10) {
    r.peak = 0;
    r.knee = r.n;
    r.floor = -200.0;
    r.window = r.n > 0 ? r.n : 1;
    r.done = 1;
    rows[b] = r;
    return;
  }
  const long long end = pk + two_fs < r.n ? pk + two_fs : r.n;
  long long seg = end - pk;
  if (pk >= end) {
    pk = pk < 0 ? 0 : pk;
    pk = pk < r.n - 1 ? pk : r.n - 1;
    seg = 1;
  }
  r.peak = pk;
  r.n_sq = (int)seg;
  const double wd = 0.03;
  const int n = fs > 0 ? (int)(r.n_sq / fs / wd) : 0;
  if (n == 0) {                                          // one mean over the whole span decides
    r.n_win = 0;
    r.w = 0;
    r.tail_a = 0;
    r.tail_b = r.n_sq;
  } else {
    const int w0 = (int)(r.n_sq / (double)n) > 1 ? (int)(r.n_sq / (double)n) : 1;
    const int tail_from = (int)(r.n_sq * 0.9);
    r.n_win = n;
    r.w = w0;
    r.tail_a = tail_from < r.n_sq ? tail_from : 0;
    r.tail_b = r.n_sq;
  }
  if (r.n_win + 1 > kKneeRound1) r.flags |= KNEE_RANGE;   // (two_fs / fs / 0.03 = 66.7: cannot happen)
  rows[b] = r;
}

__global__ __launch_bounds__(256) void knee_maxabs_kernel(const float* __restrict__ x, const KneeRow* __restrict__ rows,
                                                          unsigned long long* __restrict__ maxbits) {
  const int b = blockIdx.y;
  const KneeRow& r = rows[b];
  if (r.done) return;
  const float* seg = x + r.src_off + r.peak;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < r.n_sq; i += (long long)gridDim.x * 256)
    m = fmaxf(m, fabsf(seg[i]));
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
  __shared__ float s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  // one atomic per workgroup (the maxima of all rows share a cache line); bits of a non-negative double order like integers
  if (threadIdx.x == 0)
    atomicMax(&maxbits[b], (unsigned long long)__double_as_longlong((double)fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]))));
}

// 2 / 4. window means of the current round: means[b][i], i < n_win, and the tail's mean at [n_win] in round 1
__global__ __launch_bounds__(256) void knee_windows_kernel(const KneeRow* __restrict__ rows, const float* __restrict__ x,
                                                           const unsigned long long* __restrict__ maxbits,
                                                           double* __restrict__ means, int mean_pitch, int with_tail) {
  const int b = blockIdx.y;
  const KneeRow& r = rows[b];
  if (r.done || r.flags) return;
  const int count = r.n_win + (with_tail ? 1 : 0);
  const float* seg = x + r.src_off + r.peak;
  const double top = __longlong_as_double((long long)maxbits[b]);
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const long long a = q < r.n_win ? (long long)q * r.w : r.tail_a;
    const long long z = q < r.n_win ? a + r.w : r.tail_b;
    const double m = block_fast_mean_sq(seg + a, top, z - a);
    if (threadIdx.x == 0) means[(long long)b * mean_pitch + q] = m;
  }
}

// 3. first round: 30 ms window levels, the line through the decay, the knee estimate, the second round's windows
// (core/decay.py:103-160).  One wave per response; every lane runs the same scalar flow.
__global__ __launch_bounds__(64) void knee_stage1_kernel(KneeRow* __restrict__ rows, const double* __restrict__ means,
                                                         int mean_pitch, double fs) {
  const int b = blockIdx.x, lane = threadIdx.x;
  KneeRow r = rows[b];
  if (r.done || r.flags) return;
  __shared__ double lev[kKneeRound1];
  const double* mu = means + (long long)b * mean_pitch;
  const int n = r.n_win;
  for (int i = lane; i <= n; i += 64) lev[i] = knee_db(mu[i]);
  __syncthreads();
  bool unsure = false;
  auto finish = [&](long long knee_off, double floor, long long window) {
    r.knee = r.peak + knee_off;
    r.floor = floor;
    r.window = window;
    r.done = 1;
    if (unsure) r.flags |= KNEE_GUARD;
    if (lane == 0) rows[b] = r;
  };
  const double floor = lev[n];
  if (n == 0) return finish(r.n_sq, floor, r.n_sq > 1 ? r.n_sq : 1);
  const int hit = knee_first_le(lev, n, floor + 10.0, 4 * kKneeEl, unsure);
  KNEE_WHY(1);
  int stop = hit > 0 ? hit : n;
  if (stop < 2) {
    if (n < 2) return finish(r.n_sq, floor, r.w);
    stop = n;
  }
  if (stop > kKneeMaxFit) {
    r.flags |= KNEE_RANGE;
    if (lane == 0) rows[b] = r;
    return;
  }
  const KneeFit f = knee_fit(lev, 0, stop, 0.03, 0.0);
  if (!f.ok) unsure = true;                               // (includes the reference's NaN / |slope| < 1e-20 exits)
  KNEE_WHY(2);
  if (unsure) {
    r.flags |= KNEE_GUARD;
    if (lane == 0) rows[b] = r;
    return;
  }
  const KneeGrid grid(r.n_sq, fs);
  const double t_first = grid.at(0), t_last = grid.at(r.n_sq - 1);
  const double raw = (floor - f.icpt) / f.slope;
  const double knee_time = fmin(fmax(raw, t_first), t_last);
  const double ekt = 4.0 * ((kKneeEl + f.e_icpt) / fabs(f.slope) + fabs(raw) * f.es);
  const double per10 = fabs(f.slope) * 3;
  const double wd = 10 / per10;                           // per10 >= 3: the reference's t_last / 3 branch cannot be meant
  const double ewd = f.es + 4 * kKneeU;
  const double v = r.n_sq / fs / wd;
  if (!(v <= (double)kKneeMaxWindows)) {                  // thousands of windows: a decay of microseconds, host's case
    r.flags |= KNEE_RANGE;
    if (lane == 0) rows[b] = r;
    return;
  }
  const double vg = 4.0 * v * ewd + 1e-12;
  if (v - __builtin_floor(v) <= vg || __builtin_ceil(v) - v <= vg) unsure = true;
  KNEE_WHY(3);
  int n2 = (int)v;
  n2 = n2 > 1 ? n2 : 1;
  const int w2 = (int)(r.n_sq / (double)n2) > 1 ? (int)(r.n_sq / (double)n2) : 1;
  r.n_win = n2;
  r.w = w2;
  r.wd = wd;
  r.ewd = ewd;
  r.knee_time = knee_time;
  r.ekt = ekt;
  r.floor = floor;
  if (unsure) r.flags |= KNEE_GUARD;
  if (lane == 0) rows[b] = r;
}

// 5. second round: levels at three windows per 10 dB, then up to five refinements of noise floor, late slope and knee
// (core/decay.py:163-253).  One workgroup of 256 per response: the refinement's range means are block_np_mean calls
// that all threads reach together - every thread runs the same scalar flow on the same LDS levels.
__global__ __launch_bounds__(256) void knee_stage2_kernel(KneeRow* __restrict__ rows, const double* __restrict__ means,
                                                          int mean_pitch, const float* __restrict__ x,
                                                          const unsigned long long* __restrict__ maxbits, double fs) {
  const int b = blockIdx.x, tid = threadIdx.x;
  KneeRow r = rows[b];
  if (r.done || r.flags) return;
  __shared__ double lev[kKneeMaxWindows];
  const double* mu = means + (long long)b * mean_pitch;
  const float* seg = x + r.src_off + r.peak;
  const double top = __longlong_as_double((long long)maxbits[b]);
  const int n = r.n_win;
  const double wd = r.wd, ewd = r.ewd;
  for (int i = tid; i < n; i += 256) lev[i] = knee_db(mu[i]);
  __syncthreads();
  bool unsure = false, range = false;
  const KneeGrid grid(r.n_sq, fs);
  const double total = grid.at(r.n_sq - 1);
  const double t_end = knee_twin(n - 1, wd);
  const double gt = 4.0 * (n * wd) * ewd + 1e-15;         // |t_win[i] - host t_win[i]| for every i
  double knee_time = r.knee_time, ekt = r.ekt, floor = r.floor;
  int k_idx = knee_first_ge(n, wd, knee_time, gt + ekt, unsure);
  KNEE_WHY(4);
  if (k_idx < 0) {
    k_idx = n - 1;
    knee_time = t_end;
    ekt = gt;
  }
  double k_level = lev[k_idx];
  for (int it = 0; it < 5 && !unsure; ++it) {
    const int i0 = knee_first_le(lev, n, k_level - 5, 4 * kKneeEl, unsure);
    KNEE_WHY(5);
    if (i0 < 0) break;
    const double tenth = 0.1 * total;
    const double t0 = fmax(knee_twin(i0, wd), tenth);
    // t0 > t_win[-1] can only come from the 0.1 total branch (t_win[i0] <= t_win[-1] on both sides, same expression)
    if (fabs(tenth - t_end) <= gt) unsure = true;
    KNEE_WHY(6);
    if (tenth > t_end) break;
    const int a = grid.nearest_guarded(t0, gt, unsure);
    KNEE_WHY(7);
    const int z = grid.nearest_guarded(fmin(t0 + knee_time, total), gt + ekt, unsure);
    KNEE_WHY(8);
    if (a >= z) break;
    if (unsure) break;                                    // (uniform: every thread holds the same flags)
    floor = knee_db(block_fast_mean_sq(seg + a, top, (long long)(z - a)));
    int hi = knee_first_le(lev, n, floor + 8, 4 * kKneeEl, unsure);
    KNEE_WHY(9);
    int lo = knee_first_le(lev, n, floor + 28, 4 * kKneeEl, unsure);
    KNEE_WHY(10);
    if (hi < 0 || lo < 0) break;
    hi -= 1;
    lo = lo - 1 > 0 ? lo - 1 : 0;
    if (hi <= lo + 1) break;
    if (hi - lo > kKneeMaxFit) {
      range = true;
      break;
    }
    const KneeFit f = knee_fit(lev, lo, hi, wd, ewd);
    if (!f.ok) {
      unsure = true;
      KNEE_WHY(11);
      break;
    }
    const double raw = (floor - f.icpt) / f.slope;
    const double t_new = fmin(fmax(raw, knee_twin(0, wd)), t_end);
    const double etn = 4.0 * ((kKneeEl + f.e_icpt) / fabs(f.slope) + fabs(raw) * f.es) + gt;
    // np.clip pins t_new to t_win[0] / t_win[-1] on both sides when the raw intersection lies beyond them by more than its
    // error bound: the answer is then the first / last window whatever the last ulps are (the comparison against the
    // window's own time would otherwise read as "within the band" - a knee at the end of the span is the common case of a
    // response that has not reached its floor within 2 s).  raw - etn > t_win[n - 2] + gt: every value the host may hold
    // lies above window n - 2, clipped or not.
    int new_idx;
    if (n >= 2 && raw - etn > knee_twin(n - 2, wd) + gt) {
      new_idx = n - 1;
    } else if (raw + etn < knee_twin(0, wd) - gt) {
      new_idx = 0;
    } else {
      new_idx = knee_first_ge(n, wd, t_new, gt + etn, unsure);
      KNEE_WHY(12);
      if (new_idx < 0) new_idx = n - 1;
    }
    const bool same = new_idx == k_idx;
    k_idx = new_idx;
    knee_time = knee_twin(k_idx, wd);
    ekt = gt;
    if (same) break;
    k_level = lev[k_idx];
  }
  const int knee_off = grid.nearest_guarded(knee_time, ekt, unsure);
  KNEE_WHY(13);
  r.knee = r.peak + knee_off;
  r.floor = floor;
  r.window = r.w;
  r.done = 1;
  if (unsure) r.flags |= KNEE_GUARD;
  if (range) r.flags |= KNEE_RANGE;
  if (tid == 0) rows[b] = r;
}

}  // namespace imp
