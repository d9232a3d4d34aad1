// K12: equalisation-curve conditioning on the device, fp64, one workgroup per curve.
//
// Replaces the per-channel Python/SciPy work of the reference's EQ worker front half
//   autoeq/frequency_response.py:1060-1105  _smoothen_fractional_octave  (scipy.signal.savgol_filter x 2 + sigmoid blend)
//   autoeq/frequency_response.py:1181-1239  smoothen_heavy_light
//   autoeq/frequency_response.py:1241-1310  equalize (gain-limited inversion, kinks bridged by a quadratic spline)
//   autoeq/frequency_response.py:651-674    the gain grid handed to firwin2 (log-linear interpolation, dB -> linear)
// for all speaker-ear curves of a measurement at once (core/parallel_workers.py:69-131 runs them one by one in a pool).
// Curves are ~800-point dB arrays on the shared 1 % log grid; a curve lives in LDS for the whole chain.
//
// What is a TABLE (computed once per grid on the host in fp64, it depends on the grid only) and what is ARITHMETIC
// (done here per curve):
//   tables      log10 f, Savitzky-Golay windows (interior coefficients in closed form, edge-fit matrices from the
//               discrete orthogonal polynomials), the logistic blend weights per (f_lower, f_upper), log10 of the FIR
//               design grid;
//   arithmetic  every dot product, blend, max, clip, the spline system and its evaluation, the interpolation onto
//               the FIR grid and 10^(x/20).
// savgol_filter(mode='interp', polyorder=2) is a fixed linear operator per window: y[i] = sum_k c[k] x[i+k] inside,
// a least-squares parabola through the first / last `window` points at the edges.  The interior sum runs in
// scipy.ndimage.correlate1d's order (centre tap, then symmetric pairs from the outermost inwards) without fused
// multiply-adds, so it agrees with SciPy to the last bits of the coefficients.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <new>
#include <tuple>
#include <vector>

#include "internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int kMaxPoints = 2048;        // grid points one workgroup holds in LDS (the 1 % grid has 783 at 48 kHz, 922 at 192 kHz)
constexpr int kT = 256;

struct SgTable {        // device
  int w;                // odd window length
  const double* coeff;  // [w]
  const double* edge;   // [w/2][w]: fitted value at position i < w/2 from the first w samples
};

__device__ __forceinline__ double sg_point(const double* x, int n, int i, const SgTable& t) {
  const int m = t.w >> 1;
  if (i < m) {
    const double* e = t.edge + (size_t)i * t.w;
    double acc = 0.0;
    for (int j = 0; j < t.w; ++j) acc += e[j] * x[j];
    return acc;
  }
  if (i >= n - m) {
    const double* e = t.edge + (size_t)(n - 1 - i) * t.w;      // mirror image of the left edge
    double acc = 0.0;
    for (int j = 0; j < t.w; ++j) acc += e[j] * x[n - 1 - j];
    return acc;
  }
  double acc = x[i] * t.coeff[m];
  for (int jj = m; jj >= 1; --jj) acc += (x[i - jj] + x[i + jj]) * t.coeff[m - jj];
  return acc;
}

// y = sg(x, wn) * (k * -1 + 1) + sg(x, wt) * k      (autoeq :1103-1105)
__device__ __forceinline__ void smooth_into(const double* x, double* y, int n, const SgTable& tn, const SgTable& tt,
                                            const double* k) {
  for (int i = threadIdx.x; i < n; i += kT) {
    const double yn = sg_point(x, n, i, tn), yt = sg_point(x, n, i, tt);
    y[i] = yn * (k[i] * -1.0 + 1.0) + yt * k[i];
  }
}

__global__ __launch_bounds__(kT) void curves_smooth_kernel(const double* __restrict__ xin, double* __restrict__ yout, int n,
                                                          SgTable tn, SgTable tt, const double* __restrict__ k) {
  extern __shared__ double lds[];
  double* x = lds;
  const size_t base = (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += kT) x[i] = xin[base + i];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += kT) {
    const double yn = sg_point(x, n, i, tn), yt = sg_point(x, n, i, tt);
    yout[base + i] = yn * (k[i] * -1.0 + 1.0) + yt * k[i];
  }
}

struct EqArgs {
  const double* error;     // [B][n]
  double* error_smoothed;  // [B][n] or null
  double* equalization;    // [B][n]
  int* spline_used;        // [B] or null: 1 if the kink-bridging spline ran for the curve
  int n;
  int smoothen_first;      // 1: smoothen_heavy_light before the inversion; 0: `error` is used as it is
  SgTable w6, w3, w130;    // 1/6, 1/3 and 1.3 octaves
  const double* k_light;   // logistic 100 Hz .. 10 kHz
  const double* k_heavy;   // logistic 1 kHz .. 6 kHz
  const double* limit;     // max gain per bin:   sigmoid(a_normal = max_gain, a_treble = treble_max_gain)
  const double* gain_k;    // gain factor per bin: sigmoid(a_normal = 1,       a_treble = treble_gain_k)
  const double* log10f;    // [n]
  int kink_half;           // (window(1/12 octave) - 1) / 2
  int smoothen_kinks;
};

// Quadratic interpolating spline with FITPACK's knots (fpcurf, s = 0, k = 2): t = x0 x0 x0, midpoints of
// (x1,x2) ... (x_{m-3},x_{m-2}), x_{m-1} x3.  With these knots the collocation matrix is tridiagonal (data point i
// sees B-splines i-1, i, i+1) and totally positive, so plain elimination is stable.
__device__ __forceinline__ double knot(const double* xk, int m, int j) {      // t[j], 0 <= j < m + 3
  if (j <= 2) return xk[0];
  if (j >= m) return xk[m - 1];
  return (xk[j - 2] + xk[j - 1]) * 0.5;
}

__device__ __forceinline__ void bspline3(const double* xk, int m, int s, double x, double (&N)[3]) {
  // degree-2 B-splines s-2, s-1, s on the span [t_s, t_{s+1}) (also its polynomial continuation outside)
  const double tm1 = knot(xk, m, s - 1), t0 = knot(xk, m, s), t1 = knot(xk, m, s + 1), t2 = knot(xk, m, s + 2);
  const double a1 = (t1 - x) / (t1 - t0), b1 = (x - t0) / (t1 - t0);
  N[0] = (t1 - x) / (t1 - tm1) * a1;
  N[1] = (x - tm1) / (t1 - tm1) * a1 + (t2 - x) / (t2 - t0) * b1;
  N[2] = (x - t0) / (t2 - t0) * b1;
}

__global__ __launch_bounds__(kT) void curves_eq_kernel(EqArgs a) {
  extern __shared__ double lds[];
  const int n = a.n;
  double* e = lds;                 // input error, later the clipped gain
  double* p = e + kMaxPoints;      // light / kept abscissae
  double* q = p + kMaxPoints;      // heavy / kept ordinates
  double* r = q + kMaxPoints;      // smoothed error / spline coefficients
  double* sub = r + kMaxPoints;    // tridiagonal system
  double* dia = sub + kMaxPoints;
  double* sup = dia + kMaxPoints;
  __shared__ int s_cnt;
  __shared__ int s_keep[kMaxPoints];
  const size_t base = (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += kT) e[i] = a.error[base + i];
  __syncthreads();
  if (a.smoothen_first) {
    smooth_into(e, p, n, a.w6, a.w3, a.k_light);                 // light: 1/6 octave, treble 1/3 from 100 Hz .. 10 kHz
    smooth_into(e, q, n, a.w3, a.w130, a.k_heavy);               // heavy: 1/3 octave, treble 1.3 from 1 .. 6 kHz
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kT) p[i] = p[i] > q[i] ? p[i] : q[i];       // np.max of the stacked pair
    __syncthreads();
    smooth_into(p, r, n, a.w3, a.w3, a.k_light);                 // once more at 1/3 octave
    __syncthreads();
  } else {
    for (int i = threadIdx.x; i < n; i += kT) r[i] = e[i];
    __syncthreads();
  }
  if (a.error_smoothed)
    for (int i = threadIdx.x; i < n; i += kT) a.error_smoothed[base + i] = r[i];
  // gain-limited inversion
  for (int i = threadIdx.x; i < n; i += kT) {
    const double g = -r[i] * a.gain_k[i];
    const bool clipped = g > a.limit[i];
    e[i] = clipped ? a.limit[i] : g;
    s_keep[i] = clipped ? 3 : 1;                                 // bit 1: clipped
  }
  __syncthreads();
  if (!a.smoothen_kinks) {
    for (int i = threadIdx.x; i < n; i += kT) a.equalization[base + i] = e[i];
    return;
  }
  // samples within kink_half of a clip on/off transition are dropped (never the last two samples)
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += kT) {
    bool doomed = false;
    const int lo = i - a.kink_half < 1 ? 1 : i - a.kink_half, hi = i + a.kink_half > n - 1 ? n - 1 : i + a.kink_half;
    for (int j = lo; j <= hi && !doomed; ++j) doomed = (s_keep[j] & 2) != (s_keep[j - 1] & 2);
    if (i >= n - 2) doomed = false;
    if (doomed) {
      s_keep[i] |= 4;
      atomicAdd(&s_cnt, 1);
    }
  }
  __syncthreads();
  const int n_doomed = s_cnt;
  if (a.spline_used && threadIdx.x == 0) a.spline_used[blockIdx.x] = n_doomed > 0;
  if (n_doomed == 0) {
    // the reference still runs the interpolating spline through ALL points and reads it back at those points:
    // the data again, to rounding
    for (int i = threadIdx.x; i < n; i += kT) a.equalization[base + i] = e[i];
    return;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // compact the kept points, set up and solve the tridiagonal collocation system
    int m = 0;
    for (int i = 0; i < n; ++i)
      if (!(s_keep[i] & 4)) {
        p[m] = a.log10f[i];
        q[m] = e[i];
        ++m;
      }
    s_cnt = m;
    for (int i = 0; i < m; ++i) {
      if (i == 0) {
        sub[i] = 0.0; dia[i] = 1.0; sup[i] = 0.0;
      } else if (i == m - 1) {
        sub[i] = 0.0; dia[i] = 1.0; sup[i] = 0.0;
      } else {
        const int s = i <= 1 ? 2 : (i >= m - 2 ? m - 1 : i + 1);
        double N[3];
        bspline3(p, m, s, p[i], N);                              // B-splines s-2 .. s  =  columns i-1 .. i+1
        sub[i] = N[0]; dia[i] = N[1]; sup[i] = N[2];
      }
    }
    for (int i = 1; i < m; ++i) {                                // forward elimination
      const double f = sub[i] / dia[i - 1];
      dia[i] -= f * sup[i - 1];
      q[i] -= f * q[i - 1];
    }
    r[m - 1] = q[m - 1] / dia[m - 1];
    for (int i = m - 2; i >= 0; --i) r[i] = (q[i] - sup[i] * r[i + 1]) / dia[i];
  }
  __syncthreads();
  const int m = s_cnt;
  for (int i = threadIdx.x; i < n; i += kT) {
    const double x = a.log10f[i];
    // span: largest s in [2, m-1] with t_s <= x (first / last span beyond the ends: polynomial continuation)
    int lo = 2, hi = m - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (knot(p, m, mid) <= x) lo = mid; else hi = mid - 1;
    }
    double N[3];
    bspline3(p, m, lo, x, N);
    a.equalization[base + i] = r[lo - 2] * N[0] + r[lo - 1] * N[1] + r[lo] * N[2];
  }
}

// Gain grid of the FIR design (autoeq :651-674): equalization interpolated linearly in log10 f (linear extrapolation
// beyond the grid) onto linspace(0, fs//2, ntaps); flat below max(f[0], f_res/2); optional normalisation; dB doubled
// (the homomorphic step halves them); linear; zero at Nyquist.
struct GainArgs {
  const double* eq;        // [B][n]
  double* gain;            // [B][ntaps]
  const double* log10f;    // [n]
  const double* log10q;    // [ntaps]: log10 of the design grid (0 Hz read at 0.001 Hz)
  int n, ntaps;
  int n_flat;              // design-grid points <= f_min
  double log10_fmin;
  int normalize;
};

__device__ __forceinline__ double interp_log(const double* xk, const double* yk, int n, double xq) {
  int lo = 0, hi = n - 2;                                       // segment index: last i with xk[i] <= xq, clamped
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (xk[mid] <= xq) lo = mid; else hi = mid - 1;
  }
  const double t = (xq - xk[lo]) / (xk[lo + 1] - xk[lo]);
  return yk[lo] + t * (yk[lo + 1] - yk[lo]);
}

__global__ __launch_bounds__(kT) void curves_fir_gain_kernel(GainArgs a) {
  extern __shared__ double lds[];
  double* y = lds;
  __shared__ double s_red[kT];
  const size_t base = (size_t)blockIdx.x * a.n;
  for (int i = threadIdx.x; i < a.n; i += kT) y[i] = a.eq[base + i];
  __syncthreads();
  const double flat = interp_log(a.log10f, y, a.n, a.log10_fmin);
  double shift = 0.0;
  if (a.normalize) {
    double mx = -INFINITY;
    for (int j = threadIdx.x; j < a.ntaps; j += kT) {
      const double v = j < a.n_flat ? flat : interp_log(a.log10f, y, a.n, a.log10q[j]);
      mx = v > mx ? v : mx;
    }
    s_red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = kT / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) s_red[threadIdx.x] = s_red[threadIdx.x] > s_red[threadIdx.x + s] ? s_red[threadIdx.x] : s_red[threadIdx.x + s];
      __syncthreads();
    }
    shift = s_red[0];
  }
  double* out = a.gain + (size_t)blockIdx.x * a.ntaps;
  // blockIdx.y: the channel's taps are shared out over gridDim.y workgroups (1 when normalising: the maximum above
  // is per channel) - a binary search and a pow() per tap, 16 workgroups of 9 600 taps would leave the chip idle
  for (int j = blockIdx.y * kT + threadIdx.x; j < a.ntaps; j += kT * gridDim.y) {
    double v = j < a.n_flat ? flat : interp_log(a.log10f, y, a.n, a.log10q[j]);
    if (a.normalize) {
      v -= shift;
      v -= 0.5;
    }
    v *= 2.0;
    out[j] = j == a.ntaps - 1 ? 0.0 : pow(10.0, v / 20.0);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct SgHost {
  int w = 0;
  double* d_coeff = nullptr;
  double* d_edge = nullptr;
};

struct imp_curves {
  imp_ctx* ctx = nullptr;
  int n = 0;
  std::vector<double> freq, log10f;
  double* d_log10f = nullptr;
  std::map<int, SgHost> windows;                                              // by window length
  std::map<std::tuple<double, double, double, double>, double*> sigmoids;     // (f_lower, f_upper, a_normal, a_treble)
  struct FirGrid { int ntaps = 0; int n_flat = 0; double log10_fmin = 0; double* d_log10q = nullptr; };
  std::map<std::tuple<long long, long long>, FirGrid> fir_grids;              // (fs * 1000, f_res * 1000)
  double *d_a = nullptr, *d_b = nullptr, *d_c = nullptr;                      // [cap][n] work curves
  int* d_flags = nullptr;
  int64_t cap = 0;
  double* d_gain = nullptr;                                                   // [gain_cap] FIR design gains
  size_t gain_cap = 0;
};

static int curves_upload(const std::vector<double>& h, double** d, hipStream_t s) {
  HIP_TRY(hipMalloc((void**)d, std::max<size_t>(h.size(), 1) * sizeof(double)));
  if (!h.empty()) HIP_TRY(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

// autoeq :1033-1050: odd window (in grid points) that covers `octaves`
static int curves_window_size(const imp_curves* c, double octaves) {
  double sum = 0.0;
  for (int i = 1; i < c->n; ++i) sum += c->freq[(size_t)i] / c->freq[(size_t)i - 1];     // Python's sum(): left to right
  const double step = sum / (double)(c->n - 1);
  long w = std::lrint(std::log(std::pow(2.0, octaves)) / std::log(step));                 // round(): half to even
  if (w % 2 == 0) w += 1;
  return (int)w;
}

static int curves_window(imp_curves* c, double octaves, SgTable* out) {
  const int w = curves_window_size(c, octaves);
  if (w < 3 || w > c->n)
    return fail(IMP_ERR_INVALID, "smoothing window of %d points for %g octaves does not fit a grid of %d points "
                "(savgol_filter: polyorder 2 needs >= 3 points, mode='interp' needs window <= size)", w, octaves, c->n);
  auto it = c->windows.find(w);
  if (it == c->windows.end()) {
    const int m = w / 2;
    // least-squares parabola, value at the window centre: c_k = (3 (3 m^2 + 3 m - 1) - 15 k^2) / ((2m-1)(2m+1)(2m+3))
    std::vector<double> coeff((size_t)w), edge((size_t)m * w);
    const double dm = (double)m, den = (2 * dm - 1) * (2 * dm + 1) * (2 * dm + 3);
    for (int k = -m; k <= m; ++k) coeff[(size_t)(k + m)] = (3.0 * (3 * dm * dm + 3 * dm - 1) - 15.0 * (double)k * k) / den;
    // edge fit: projection on the discrete orthogonal polynomials 1, t, t^2 - (w^2-1)/12 over t = j - (w-1)/2
    const double dw = (double)w, s0 = dw, s1 = dw * (dw * dw - 1) / 12.0, s2 = dw * (dw * dw - 1) * (dw * dw - 4) / 180.0;
    const double mu2 = (dw * dw - 1) / 12.0;
    for (int i = 0; i < m; ++i) {
      const double ti = (double)i - (dw - 1) / 2;
      for (int j = 0; j < w; ++j) {
        const double tj = (double)j - (dw - 1) / 2;
        edge[(size_t)i * w + j] = 1.0 / s0 + ti * tj / s1 + (ti * ti - mu2) * (tj * tj - mu2) / s2;
      }
    }
    SgHost h;
    h.w = w;
    int rc;
    if ((rc = curves_upload(coeff, &h.d_coeff, c->ctx->stream))) return rc;
    if ((rc = curves_upload(edge, &h.d_edge, c->ctx->stream))) return rc;
    it = c->windows.emplace(w, h).first;
  }
  out->w = w;
  out->coeff = it->second.d_coeff;
  out->edge = it->second.d_edge;
  return IMP_OK;
}

// autoeq :1052-1058 (scipy.special.expit(x) = 1 / (1 + exp(-x)))
static int curves_sigmoid(imp_curves* c, double f_lower, double f_upper, double a_normal, double a_treble, const double** out) {
  const auto key = std::make_tuple(f_lower, f_upper, a_normal, a_treble);
  auto it = c->sigmoids.find(key);
  if (it == c->sigmoids.end()) {
    if (!(f_lower > 0) || !(f_upper > f_lower)) return fail(IMP_ERR_INVALID, "sigmoid: need 0 < f_lower < f_upper");
    std::vector<double> v((size_t)c->n);
    double f_center = std::sqrt(f_upper / f_lower) * f_lower;
    const double half_range = std::log10(f_upper) - std::log10(f_center);
    f_center = std::log10(f_center);
    for (int i = 0; i < c->n; ++i) {
      const double a = 1.0 / (1.0 + std::exp(-((c->log10f[(size_t)i] - f_center) / (half_range / 4))));
      v[(size_t)i] = a * -(a_normal - a_treble) + a_normal;
    }
    double* d = nullptr;
    int rc = curves_upload(v, &d, c->ctx->stream);
    if (rc) return rc;
    it = c->sigmoids.emplace(key, d).first;
  }
  *out = it->second;
  return IMP_OK;
}

static int curves_reserve(imp_curves* c, int64_t B) {
  if (c->cap >= B) return IMP_OK;
  (void)hipFree(c->d_a); (void)hipFree(c->d_b); (void)hipFree(c->d_c); (void)hipFree(c->d_flags);
  c->d_a = c->d_b = c->d_c = nullptr;
  c->d_flags = nullptr;
  c->cap = 0;
  const size_t bytes = (size_t)B * c->n * sizeof(double);
  if (hipMalloc((void**)&c->d_a, bytes) != hipSuccess || hipMalloc((void**)&c->d_b, bytes) != hipSuccess ||
      hipMalloc((void**)&c->d_c, bytes) != hipSuccess || hipMalloc((void**)&c->d_flags, (size_t)B * sizeof(int)) != hipSuccess)
    return fail(IMP_ERR_ALLOC, "imp_curves: device allocation for %lld curves failed", (long long)B);
  c->cap = B;
  return IMP_OK;
}

extern "C" void imp_curves_destroy(imp_curves* c) {
  if (!c) return;
  IMP_CTX_LOCK(c->ctx);
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  (void)hipFree(c->d_log10f);
  for (auto& kv : c->windows) {
    (void)hipFree(kv.second.d_coeff);
    (void)hipFree(kv.second.d_edge);
  }
  for (auto& kv : c->sigmoids) (void)hipFree(kv.second);
  for (auto& kv : c->fir_grids) (void)hipFree(kv.second.d_log10q);
  (void)hipFree(c->d_a); (void)hipFree(c->d_b); (void)hipFree(c->d_c); (void)hipFree(c->d_flags); (void)hipFree(c->d_gain);
  delete c;
}

extern "C" int imp_curves_create(imp_ctx* ctx, const double* frequency, int64_t n, imp_curves** out) {
  if (!ctx || !frequency || !out) return fail(IMP_ERR_INVALID, "imp_curves_create: null argument");
  *out = nullptr;
  if (n < 8 || n > kMaxPoints) return fail(IMP_ERR_INVALID, "imp_curves_create: grid of %lld points (8 .. %d supported)", (long long)n, kMaxPoints);
  for (int64_t i = 0; i < n; ++i)
    if (!(frequency[i] > 0) || (i && !(frequency[i] > frequency[i - 1])))
      return fail(IMP_ERR_INVALID, "imp_curves_create: frequencies must be positive and strictly increasing");
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  imp_curves* c = new (std::nothrow) imp_curves();
  if (!c) return fail(IMP_ERR_ALLOC, "out of host memory");
  c->ctx = ctx;
  c->n = (int)n;
  c->freq.assign(frequency, frequency + n);
  c->log10f.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) c->log10f[(size_t)i] = std::log10(frequency[i]);
  if ((rc = curves_upload(c->log10f, &c->d_log10f, ctx->stream))) {
    imp_curves_destroy(c);
    return rc;
  }
  *out = c;
  return IMP_OK;
}

extern "C" int imp_curves_window_size(imp_curves* c, double octaves, int* window) {
  if (!c || !window) return fail(IMP_ERR_INVALID, "imp_curves_window_size: null argument");
  *window = curves_window_size(c, octaves);
  return IMP_OK;
}

static constexpr size_t kEqLds = 7 * kMaxPoints * sizeof(double);

extern "C" int imp_curves_smooth(imp_curves* c, const double* x, int64_t B, double window_oct, double treble_window_oct,
                                 double treble_f_lower, double treble_f_upper, double* y) {
  if (!c || (B && (!x || !y))) return fail(IMP_ERR_INVALID, "imp_curves_smooth: null argument");
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  for (int64_t i = 0; i < B * c->n; ++i)
    if (std::isnan(x[i])) return fail(IMP_ERR_INVALID, "NaN values present, cannot smoothen!");
  IMP_CTX_LOCK(c->ctx);
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  SgTable tn, tt;
  const double* k = nullptr;
  if ((rc = curves_window(c, window_oct, &tn)) || (rc = curves_window(c, treble_window_oct, &tt)) ||
      (rc = curves_sigmoid(c, treble_f_lower, treble_f_upper, 0.0, 1.0, &k)) || (rc = curves_reserve(c, B)))
    return rc;
  hipStream_t s = c->ctx->stream;
  const size_t bytes = (size_t)B * c->n * sizeof(double);
  HIP_TRY(hipMemcpyAsync(c->d_a, x, bytes, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(curves_smooth_kernel, dim3((unsigned)B), dim3(kT), (size_t)c->n * sizeof(double), s, c->d_a, c->d_b, c->n,
                     tn, tt, k);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(y, c->d_b, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

// error (host) -> d_b = error_smoothed, d_c = equalization on the device
static int curves_equalization_device(imp_curves* c, const double* error, int64_t B, int smoothen_first, double max_gain,
                                      double treble_f_lower, double treble_f_upper, double treble_max_gain,
                                      double treble_gain_k, int smoothen_kinks) {
  for (int64_t i = 0; i < B * c->n; ++i)
    if (std::isnan(error[i]))
      return fail(IMP_ERR_INVALID, smoothen_first ? "NaN values present, cannot smoothen!"
                                                  : "NaN values detected during equalization, interpolating data with default parameters.");
  int rc;
  EqArgs a{};
  if ((rc = curves_window(c, 1.0 / 6, &a.w6)) || (rc = curves_window(c, 1.0 / 3, &a.w3)) || (rc = curves_window(c, 1.3, &a.w130)) ||
      (rc = curves_sigmoid(c, 100.0, 10000.0, 0.0, 1.0, &a.k_light)) || (rc = curves_sigmoid(c, 1000.0, 6000.0, 0.0, 1.0, &a.k_heavy)) ||
      (rc = curves_sigmoid(c, treble_f_lower, treble_f_upper, max_gain, treble_max_gain, &a.limit)) ||
      (rc = curves_sigmoid(c, treble_f_lower, treble_f_upper, 1.0, treble_gain_k, &a.gain_k)) || (rc = curves_reserve(c, B)))
    return rc;
  if ((rc = ctx_kernel_lds(c->ctx, reinterpret_cast<const void*>(curves_eq_kernel), kEqLds))) return rc;
  a.error = c->d_a;
  a.error_smoothed = c->d_b;
  a.equalization = c->d_c;
  a.spline_used = c->d_flags;
  a.n = c->n;
  a.smoothen_first = smoothen_first;
  a.log10f = c->d_log10f;
  a.kink_half = (curves_window_size(c, 1.0 / 12) - 1) / 2;
  a.smoothen_kinks = smoothen_kinks;
  hipStream_t s = c->ctx->stream;
  HIP_TRY(hipMemcpyAsync(c->d_a, error, (size_t)B * c->n * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(curves_eq_kernel, dim3((unsigned)B), dim3(kT), kEqLds, s, a);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

extern "C" int imp_curves_equalization(imp_curves* c, const double* error, int64_t B, int smoothen_first, double max_gain,
                                       double treble_f_lower, double treble_f_upper, double treble_max_gain,
                                       double treble_gain_k, int smoothen_kinks, double* error_smoothed_out,
                                       double* equalization_out, int* spline_used_out) {
  if (!c || (B && (!error || !equalization_out))) return fail(IMP_ERR_INVALID, "imp_curves_equalization: null argument");
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  IMP_CTX_LOCK(c->ctx);
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  if ((rc = curves_equalization_device(c, error, B, smoothen_first, max_gain, treble_f_lower, treble_f_upper, treble_max_gain,
                                       treble_gain_k, smoothen_kinks)))
    return rc;
  hipStream_t s = c->ctx->stream;
  const size_t bytes = (size_t)B * c->n * sizeof(double);
  if (error_smoothed_out) HIP_TRY(hipMemcpyAsync(error_smoothed_out, c->d_b, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(equalization_out, c->d_c, bytes, hipMemcpyDeviceToHost, s));
  if (spline_used_out) HIP_TRY(hipMemcpyAsync(spline_used_out, c->d_flags, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

// scipy.fftpack.next_fast_len: smallest 2^a 3^b 5^c >= n
static int64_t curves_next_fast_len(int64_t n) {
  for (int64_t m = std::max<int64_t>(n, 1);; ++m) {
    int64_t r = m;
    for (int64_t p : {2, 3, 5})
      while (r % p == 0) r /= p;
    if (r == 1) return m;
  }
}

static int curves_fir_grid(imp_curves* c, double fs, double f_res, imp_curves::FirGrid* out) {
  const auto key = std::make_tuple((long long)std::llround(fs * 1000.0), (long long)std::llround(f_res * 1000.0));
  auto it = c->fir_grids.find(key);
  if (it == c->fir_grids.end()) {
    imp_curves::FirGrid g;
    const double half_res = f_res / 2;                                          // autoeq :652
    const double f_min = std::max(c->freq[0], half_res);
    const double nyq = std::floor(fs / 2.0);                                    // fs // 2
    g.ntaps = (int)curves_next_fast_len((int64_t)std::lrint(nyq / half_res));   // round(fs // 2 / f_res)
    if (g.ntaps < 2 || g.ntaps > (1 << 20)) return fail(IMP_ERR_INVALID, "FIR design grid of %d points", g.ntaps);
    std::vector<double> lq((size_t)g.ntaps);
    const double step = nyq / (double)(g.ntaps - 1);                            // np.linspace(0, nyq, ntaps)
    g.n_flat = 0;
    for (int j = 0; j < g.ntaps; ++j) {
      double f = j == g.ntaps - 1 ? nyq : (double)j * step;
      if (f <= f_min) g.n_flat = j + 1;
      if (f == 0.0) f = 0.001;
      lq[(size_t)j] = std::log10(f);
    }
    g.log10_fmin = std::log10(f_min);
    int rc = curves_upload(lq, &g.d_log10q, c->ctx->stream);
    if (rc) return rc;
    it = c->fir_grids.emplace(key, g).first;
  }
  *out = it->second;
  return IMP_OK;
}

int minphase_fir_from_device_gain(imp_ctx* ctx, const double* d_gain, int64_t B, int64_t n, double fs, double* fir_out_host,
                                  double* d_fir_out);

extern "C" int imp_curves_fir_taps(imp_curves* c, double fs, double f_res, int64_t* ntaps) {
  if (!c || !ntaps) return fail(IMP_ERR_INVALID, "imp_curves_fir_taps: null argument");
  IMP_CTX_LOCK(c->ctx);
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  imp_curves::FirGrid g;
  if ((rc = curves_fir_grid(c, fs, f_res, &g))) return rc;
  *ntaps = g.ntaps;
  return IMP_OK;
}

// equalization curves (host, or the device result of the previous step when eq == NULL) -> minimum-phase FIRs
static int curves_fir_from(imp_curves* c, const double* d_eq, int64_t B, double fs, double f_res, int normalize,
                           double* gain_out, double* fir_out, double* d_fir_out = nullptr) {
  imp_curves::FirGrid g;
  int rc = curves_fir_grid(c, fs, f_res, &g);
  if (rc) return rc;
  const size_t need = (size_t)B * g.ntaps;
  if (c->gain_cap < need) {
    (void)hipFree(c->d_gain);
    c->d_gain = nullptr;
    c->gain_cap = 0;
    if (hipMalloc((void**)&c->d_gain, need * sizeof(double)) != hipSuccess) return fail(IMP_ERR_ALLOC, "imp_curves: gain buffer");
    c->gain_cap = need;
  }
  GainArgs a{};
  a.eq = d_eq;
  a.gain = c->d_gain;
  a.log10f = c->d_log10f;
  a.log10q = g.d_log10q;
  a.n = c->n;
  a.ntaps = g.ntaps;
  a.n_flat = g.n_flat;
  a.log10_fmin = g.log10_fmin;
  a.normalize = normalize;
  hipStream_t s = c->ctx->stream;
  const unsigned tiles = normalize ? 1u : (unsigned)std::max(1, std::min(64, (g.ntaps + 2 * kT - 1) / (2 * kT)));
  hipLaunchKernelGGL(curves_fir_gain_kernel, dim3((unsigned)B, tiles), dim3(kT), (size_t)c->n * sizeof(double), s, a);
  HIP_TRY(hipGetLastError());
  if (gain_out) HIP_TRY(hipMemcpyAsync(gain_out, c->d_gain, need * sizeof(double), hipMemcpyDeviceToHost, s));
  if (fir_out || d_fir_out) return minphase_fir_from_device_gain(c->ctx, c->d_gain, B, g.ntaps, fs, fir_out, d_fir_out);
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

extern "C" int imp_curves_fir(imp_curves* c, const double* equalization, int64_t B, double fs, double f_res, int normalize,
                              double* gain_out, double* fir_out) {
  if (!c || (B && !equalization)) return fail(IMP_ERR_INVALID, "imp_curves_fir: null argument");
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  IMP_CTX_LOCK(c->ctx);
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  if ((rc = curves_reserve(c, B))) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_c, equalization, (size_t)B * c->n * sizeof(double), hipMemcpyHostToDevice, c->ctx->stream));
  return curves_fir_from(c, c->d_c, B, fs, f_res, normalize, gain_out, fir_out);
}

extern "C" int imp_curves_equalization_fir(imp_curves* c, const double* error, int64_t B, int smoothen_first, double max_gain,
                                           double treble_f_lower, double treble_f_upper, double treble_max_gain,
                                           double treble_gain_k, int smoothen_kinks, double fs, double f_res, int normalize,
                                           double* equalization_out, double* fir_out) {
  if (!c || (B && (!error || !fir_out))) return fail(IMP_ERR_INVALID, "imp_curves_equalization_fir: null argument");
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  IMP_CTX_LOCK(c->ctx);
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  if ((rc = curves_equalization_device(c, error, B, smoothen_first, max_gain, treble_f_lower, treble_f_upper, treble_max_gain,
                                       treble_gain_k, smoothen_kinks)))
    return rc;
  if (equalization_out)
    HIP_TRY(hipMemcpyAsync(equalization_out, c->d_c, (size_t)B * c->n * sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream));
  return curves_fir_from(c, c->d_c, B, fs, f_res, normalize, nullptr, fir_out);
}

// the whole worker with the FIRs LEFT ON THE DEVICE: *d_fir_out = [B][ntaps] fp64 in a block of the context's pool (the
// caller hands it back with imp_free); nothing waits for the device unless equalization_out is asked for
extern "C" int imp_curves_equalization_fir_device(imp_curves* c, const double* error, int64_t B, int smoothen_first, double max_gain,
                                                  double treble_f_lower, double treble_f_upper, double treble_max_gain,
                                                  double treble_gain_k, int smoothen_kinks, double fs, double f_res, int normalize,
                                                  double* equalization_out, void** d_fir_out, int64_t* ntaps_out) {
  if (!c || !d_fir_out || (B && !error)) return fail(IMP_ERR_INVALID, "imp_curves_equalization_fir_device: null argument");
  *d_fir_out = nullptr;
  if (B < 1) return fail(IMP_ERR_INVALID, "B < 1");
  IMP_CTX_LOCK(c->ctx);
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  imp_curves::FirGrid g;
  if ((rc = curves_fir_grid(c, fs, f_res, &g))) return rc;
  if (ntaps_out) *ntaps_out = g.ntaps;
  void* d_fir = nullptr;
  if ((rc = ctx_block_get(c->ctx, (size_t)B * g.ntaps * sizeof(double), &d_fir))) return rc;
  rc = curves_equalization_device(c, error, B, smoothen_first, max_gain, treble_f_lower, treble_f_upper, treble_max_gain,
                                  treble_gain_k, smoothen_kinks);
  if (!rc && equalization_out) {
    if (hipMemcpyAsync(equalization_out, c->d_c, (size_t)B * c->n * sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream) != hipSuccess ||
        hipStreamSynchronize(c->ctx->stream) != hipSuccess)
      rc = fail(IMP_ERR_HIP, "imp_curves_equalization_fir_device: download failed");
  }
  if (!rc) rc = curves_fir_from(c, c->d_c, B, fs, f_res, normalize, nullptr, nullptr, (double*)d_fir);
  if (rc) {
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)ctx_block_put(c->ctx, d_fir);
    return rc;
  }
  *d_fir_out = d_fir;
  return IMP_OK;
}
