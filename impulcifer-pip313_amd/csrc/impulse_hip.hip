// libimpulse_hip.so - host side of the C ABI declared in include/impulse_hip.h.
// Built with: hipcc --offload-arch=gfx950 -O3 -shared -fPIC impulse_hip.hip -o libimpulse_hip.so
// gfx950 only; no torch, no rocFFT/hipFFT.  The one collective of the path - the broadcast of the prepared filter
// spectrum (imp_plan_spectrum) - is comm.hip's imp_comm_broadcast over RCCL (librccl opened with dlopen on first use).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "internal.h"
#include "conv_kernels.hip.h"
#include "ir_kernels.hip.h"
#include "decay_kernels.hip.h"      // it switches fp contraction off for what follows
#include "slice_kernels.hip.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int imp_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

int ctx_bind(imp_ctx* ctx) {
  HIP_TRY(hipSetDevice(ctx->device));
  return IMP_OK;
}

int ctx_kernel_lds(imp_ctx* ctx, const void* kernel, size_t bytes) {
  if (bytes == 0 || ctx->lds_opt_in.count(kernel)) return IMP_OK;
  HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  ctx->lds_opt_in.insert(kernel);
  return IMP_OK;
}

static int upload_table(cf** dptr, const std::vector<cf>& h, hipStream_t s) {
  HIP_TRY(hipMalloc((void**)dptr, h.size() * sizeof(cf)));
  HIP_TRY(hipMemcpyAsync(*dptr, h.data(), h.size() * sizeof(cf), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

static cf unit_root(int64_t num, int64_t den) {          // exp(-2 pi i num/den), num reduced mod den
  const double ang = -2.0 * M_PI * (double)(num % den) / (double)den;
  return make_float2((float)std::cos(ang), (float)std::sin(ang));
}

static int ctx_row_tables(imp_ctx* ctx) {
  std::vector<cf> t1(16 * 256), t2(16 * 16), t4(16 * 256);
  for (int a = 0; a < 16; ++a)
    for (int t = 0; t < 256; ++t) {
      t1[(size_t)a * 256 + t] = unit_root((int64_t)t * a, 4096);
      t4[(size_t)a * 256 + t] = unit_root((int64_t)(16 * a + (t & 15)) * (t >> 4), 4096);
    }
  for (int q = 0; q < 16; ++q)
    for (int l = 0; l < 16; ++l) t2[(size_t)q * 16 + l] = unit_root((int64_t)16 * l * q, 4096);
  int rc;
  if ((rc = upload_table(&ctx->tw_t1, t1, ctx->stream))) return rc;
  if ((rc = upload_table(&ctx->tw_t2, t2, ctx->stream))) return rc;
  if ((rc = upload_table(&ctx->tw_t4, t4, ctx->stream))) return rc;
  return IMP_OK;
}

static int ctx_twiddles(imp_ctx* ctx, int n1, TwSet* out) {
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  auto it = ctx->tw_by_n1.find(n1);
  if (it != ctx->tw_by_n1.end()) {
    *out = it->second;
    return IMP_OK;
  }
  const int64_t nc = (int64_t)n1 * imp::kN2;
  TwSet t;
  int rc;
  {
    std::vector<cf> full((size_t)nc);
    for (int64_t k1 = 0; k1 < n1; ++k1)
      for (int64_t n2 = 0; n2 < imp::kN2; ++n2) full[(size_t)(k1 * imp::kN2 + n2)] = unit_root(k1 * n2, nc);
    if ((rc = upload_table(&t.full, full, ctx->stream))) return rc;
  }
  {
    std::vector<cf> hi((size_t)(4 * n1));
    for (int64_t m = 0; m < 4 * n1; ++m) hi[(size_t)m] = unit_root(1024 * m, nc);
    if ((rc = upload_table(&t.hi, hi, ctx->stream))) return rc;
  }
  ctx->tw_by_n1[n1] = t;
  *out = t;
  return IMP_OK;
}

static int ctx_scratch(imp_ctx* ctx, size_t bytes, void** out) {
  if (ctx->scratch_bytes < bytes) {
    if (ctx->scratch) HIP_TRY(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = std::max(bytes, (size_t)1 << 20);
    HIP_TRY(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return IMP_OK;
}

// Small tables (row offsets, lengths, window parameters) on their way to a kernel: a ring in pinned host memory mirrored by a
// ring in device memory.  ctx_stage hands out the same offset in both, the caller fills the host side and ctx_stage_push
// sends it with ONE asynchronous copy in stream order - no pageable staging, no wait, nothing for the next call to
// overwrite (the ring only wraps after the stream has drained).
int ctx_stage(imp_ctx* ctx, size_t bytes, void** host, void** dev) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (ctx->stage_cap < bytes) {
    if (ctx->stage_host) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      (void)hipHostFree(ctx->stage_host);
      (void)hipFree(ctx->stage_dev);
      ctx->stage_host = ctx->stage_dev = nullptr;
      ctx->stage_cap = ctx->stage_pos = 0;
    }
    const size_t want = std::max(4 * bytes, (size_t)1 << 20);
    HIP_TRY(hipHostMalloc((void**)&ctx->stage_host, want, hipHostMallocDefault));
    if (hipMalloc((void**)&ctx->stage_dev, want) != hipSuccess) {
      (void)hipHostFree(ctx->stage_host);
      ctx->stage_host = nullptr;
      return fail(IMP_ERR_ALLOC, "staging ring: device allocation of %zu bytes failed", want);
    }
    ctx->stage_cap = want;
    ctx->stage_pos = 0;
  }
  if (ctx->stage_pos + bytes > ctx->stage_cap) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stage_pos = 0;
  }
  *host = ctx->stage_host + ctx->stage_pos;
  *dev = ctx->stage_dev + ctx->stage_pos;
  ctx->stage_pos += bytes;
  return IMP_OK;
}
int ctx_stage_push(imp_ctx* ctx, const void* host, void* dev, size_t bytes) {
  HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return IMP_OK;
}

extern "C" const char* imp_version(void) { return "impulse_hip 0.1.0 (gfx950)"; }
extern "C" const char* imp_last_error(void) { return g_last_error.c_str(); }

extern "C" int imp_device_count(int* n) {
  if (!n) return fail(IMP_ERR_INVALID, "imp_device_count: null output");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return fail(IMP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *n = c;
  return IMP_OK;
}

int ctx_new_stream(imp_ctx* ctx, hipStream_t* out) {
  (void)ctx;
  HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return IMP_OK;
}

// a stream ctx_new_stream made and nothing uses any more
static void ctx_release_stream(imp_ctx*, hipStream_t st) {
  (void)hipStreamSynchronize(st);
  (void)hipStreamDestroy(st);
}

extern "C" int imp_ctx_create(int device_id, imp_ctx** out) {
  if (!out) return fail(IMP_ERR_INVALID, "imp_ctx_create: null output");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(IMP_ERR_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
  if (device_id < 0 || device_id >= n)
    return fail(IMP_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(IMP_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device_id,
                prop.gcnArchName);
  imp_ctx* ctx = new (std::nothrow) imp_ctx();
  if (!ctx) return fail(IMP_ERR_ALLOC, "out of host memory");
  ctx->k2_bluestein_only = std::getenv("IMPULSE_HIP_K2_BLUESTEIN") != nullptr;
  ctx->device = device_id;
  if (ctx_new_stream(ctx, &ctx->stream)) {
    delete ctx;
    return IMP_ERR_HIP;
  }
  if (const char* mb = std::getenv("IMPULSE_HIP_POOL_MB")) ctx->free_cap = (size_t)std::max(0ll, std::atoll(mb)) << 20;
  int rc = ctx_row_tables(ctx);
  if (rc) {
    ctx_release_stream(ctx, ctx->stream);
    delete ctx;
    return rc;
  }
  *out = ctx;
  return IMP_OK;
}

// The staging ring and the stream-ordered imp_free are correct only while all of a context's work is on ONE stream: the
// outgoing stream is drained before the swap, whoever owns it, so nothing queued there can still read the ring or a block
// that imp_free has handed back; the ring starts over on the new stream.
extern "C" int imp_ctx_set_stream(imp_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(IMP_ERR_INVALID, "null ctx");
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (ctx->stream) HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto st : ctx->side_streams) HIP_TRY(hipStreamSynchronize(st));
  if (ctx->own_stream && ctx->stream) ctx_release_stream(ctx, ctx->stream);
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  ctx->stage_pos = 0;
  return IMP_OK;
}

static void pool_release(imp_ctx* ctx);      // defined with imp_malloc

extern "C" int imp_ctx_synchronize(imp_ctx* ctx) {
  if (!ctx) return fail(IMP_ERR_INVALID, "null ctx");
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto st : ctx->side_streams) HIP_TRY(hipStreamSynchronize(st));
  return IMP_OK;
}

extern "C" void imp_ctx_destroy(imp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->tw_by_n1) {
    (void)hipFree(kv.second.full);
    (void)hipFree(kv.second.hi);
  }
  if (ctx->tw_t1) (void)hipFree(ctx->tw_t1);
  if (ctx->tw_t2) (void)hipFree(ctx->tw_t2);
  if (ctx->tw_t4) (void)hipFree(ctx->tw_t4);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
  if (ctx->stage_dev) (void)hipFree(ctx->stage_dev);
  pool_release(ctx);
  for (auto& kv : ctx->live_blocks) (void)hipFree(kv.first);      // blocks the caller never handed back
  minphase_plans_destroy(ctx);
  fft_roots_destroy(ctx);
  magnitude_plans_destroy(ctx);
  for (auto st : ctx->side_streams) ctx_release_stream(ctx, st);
  if (ctx->own_stream && ctx->stream) ctx_release_stream(ctx, ctx->stream);
  delete ctx;
}

static void pool_release(imp_ctx* ctx) {
  for (auto& kv : ctx->free_blocks) (void)hipFree(kv.second);
  ctx->free_blocks.clear();
  ctx->free_bytes = 0;
}

// pooled device blocks: imp_malloc / imp_free and the library's own short-lived buffers (filter-spectrum work arrays,
// segment sets) share one pool per context
int ctx_block_get(imp_ctx* ctx, size_t bytes, void** dptr) {
  *dptr = nullptr;
  if (bytes == 0) bytes = 1;
  // a kept block of this size, or up to a quarter larger
  auto it = ctx->free_blocks.lower_bound(bytes);
  if (it != ctx->free_blocks.end() && it->first <= bytes + bytes / 4) {
    *dptr = it->second;
    ctx->live_blocks[it->second] = it->first;
    ctx->free_bytes -= it->first;
    ctx->free_blocks.erase(it);
    return IMP_OK;
  }
  hipError_t e = hipMalloc(dptr, bytes);
  if (e != hipSuccess && !ctx->free_blocks.empty()) {        // give the kept blocks back and try once more
    (void)hipGetLastError();
    pool_release(ctx);
    e = hipMalloc(dptr, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *dptr = nullptr;
    return fail(IMP_ERR_ALLOC, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  }
  ctx->live_blocks[*dptr] = bytes;
  return IMP_OK;
}

// the caller guarantees that nothing in flight still uses the block; returns false for a foreign pointer
bool ctx_block_put(imp_ctx* ctx, void* dptr) {
  if (!dptr) return true;
  auto it = ctx->live_blocks.find(dptr);
  if (it == ctx->live_blocks.end()) return false;
  const size_t bytes = it->second;
  ctx->live_blocks.erase(it);
  if (ctx->free_bytes + bytes <= ctx->free_cap) {
    ctx->free_blocks.emplace(bytes, dptr);
    ctx->free_bytes += bytes;
  } else {
    (void)hipFree(dptr);
  }
  return true;
}

extern "C" int imp_malloc(imp_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return fail(IMP_ERR_INVALID, "imp_malloc: null argument");
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  *dptr = nullptr;
  if (bytes == 0) return IMP_OK;
  return ctx_block_get(ctx, bytes, dptr);
}

extern "C" int imp_free(imp_ctx* ctx, void* dptr) {
  if (!ctx) return fail(IMP_ERR_INVALID, "null ctx");
  if (!dptr) return IMP_OK;
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  // Whoever gets the block next uses it in the order of the context's stream - every entry point of this library
  // queues its work (copies included) there - so a context with that one stream hands the block back without waiting:
  // what is still in flight on it finishes before anything queued later starts.  Contexts with overlap lanes (work on
  // side streams) drain first, as hipFree would.  IMPULSE_HIP_FREE_SYNC=1: always drain.
  static const bool always_sync = [] {
    const char* e = std::getenv("IMPULSE_HIP_FREE_SYNC");
    return e && e[0] == '1';
  }();
  if (always_sync || !ctx->side_streams.empty()) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto st : ctx->side_streams) HIP_TRY(hipStreamSynchronize(st));
  }
  if (!ctx_block_put(ctx, dptr)) return fail(IMP_ERR_INVALID, "imp_free: %p did not come from imp_malloc on this context", dptr);
  return IMP_OK;
}

extern "C" int imp_memcpy_h2d(imp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(IMP_ERR_INVALID, "imp_memcpy_h2d: null argument");
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!bytes) return IMP_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return IMP_OK;
}

extern "C" int imp_memcpy_d2h(imp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(IMP_ERR_INVALID, "imp_memcpy_d2h: null argument");
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!bytes) return IMP_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return IMP_OK;
}

extern "C" int imp_memcpy_d2d(imp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(IMP_ERR_INVALID, "imp_memcpy_d2d: null argument");
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!bytes) return IMP_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return IMP_OK;
}

extern "C" int imp_memset(imp_ctx* ctx, void* dptr, int value, size_t bytes) {
  if (!ctx || (!dptr && bytes)) return fail(IMP_ERR_INVALID, "imp_memset: null argument");
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  if (!bytes) return IMP_OK;
  HIP_TRY(hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return IMP_OK;
}

// ------------------------------------------------------------------------------------------------
// host fp64 FFT (plan preparation only): iterative radix-2, power-of-two sizes
// ------------------------------------------------------------------------------------------------
typedef std::complex<double> cd;

// Recursive decimation-in-time FFT for lengths 2^a 3^b 5^c (plan preparation only, fp64).
// w holds exp(-2 pi i k / N) for the top-level N; stride selects the sub-transform's roots.
static void host_fft_rec(const cd* in, cd* out, size_t n, size_t in_stride, const std::vector<cd>& w, size_t w_stride,
                         cd* scratch) {
  if (n == 1) {
    out[0] = in[0];
    return;
  }
  const size_t p = (n % 2 == 0) ? 2 : (n % 3 == 0) ? 3 : (n % 5 == 0) ? 5 : 11;
  const size_t m = n / p;
  for (size_t r = 0; r < p; ++r)
    host_fft_rec(in + r * in_stride, scratch + r * m, m, in_stride * p, w, w_stride * p, out + r * m);
  // scratch[r*m + k] = DFT_m of the r-th decimated sequence; combine
  const size_t wn = w.size();
  for (size_t k = 0; k < m; ++k) {
    cd t[11];
    for (size_t r = 0; r < p; ++r) t[r] = scratch[r * m + k] * w[(r * k * w_stride) % wn];
    for (size_t q = 0; q < p; ++q) {
      cd acc = t[0];
      for (size_t r = 1; r < p; ++r) acc += t[r] * w[(r * q * m * w_stride) % wn];
      out[k + q * m] = acc;
    }
  }
}

static bool host_fft(std::vector<cd>& a) {
  const size_t n = a.size();
  size_t r = n;
  for (size_t p : {2u, 3u, 5u, 11u})
    while (r % p == 0) r /= p;
  if (r != 1) return false;
  std::vector<cd> w(n), out(n), scratch(n);
  for (size_t k = 0; k < n; ++k) {
    const double ang = -2.0 * M_PI * (double)k / (double)n;
    w[k] = cd(std::cos(ang), std::sin(ang));
  }
  host_fft_rec(a.data(), out.data(), n, 1, w, 1, scratch.data());
  a.swap(out);
  return true;
}

// H[0..Nc] = rfft(h zero-padded to nfft = 2 Nc) via one Nc-point complex FFT
static void host_rfft(const double* h, int64_t M, int64_t Nc, std::vector<cd>& H) {
  std::vector<cd> z((size_t)Nc, cd(0, 0));
  for (int64_t n = 0; 2 * n < M; ++n) {
    double re = h[2 * n];
    double im = (2 * n + 1 < M) ? h[2 * n + 1] : 0.0;
    z[(size_t)n] = cd(re, im);
  }
  host_fft(z);
  H.assign((size_t)Nc + 1, cd(0, 0));
  const double nfft = 2.0 * (double)Nc;
  for (int64_t k = 0; k <= Nc; ++k) {
    cd zk = z[(size_t)(k % Nc)];
    cd zm = std::conj(z[(size_t)((Nc - k) % Nc)]);
    cd E = 0.5 * (zk + zm);
    cd O = cd(0, -0.5) * (zk - zm);
    double ang = -2.0 * M_PI * (double)k / nfft;
    H[(size_t)k] = E + cd(std::cos(ang), std::sin(ang)) * O;
  }
}

// alpha/beta planes in the row kernel's register order, fp32 (see conv_kernels.hip.h)
static void host_alpha_beta(const std::vector<cd>& H, int64_t Nc, int N1, float4* ab) {
  const double nfft = 2.0 * (double)Nc;
  const double inv = 1.0 / (double)Nc;
  for (int k1 = 0; k1 < N1; ++k1) {
    for (int k2 = 0; k2 < imp::kN2; ++k2) {
      const int64_t k = (int64_t)k1 + (int64_t)N1 * k2;
      const int u = 16 * (k2 & 15) + ((k2 >> 4) & 15);
      const int q = k2 >> 8;
      float4 o;
      if (k == 0) {
        o = make_float4((float)(H[0].real() * inv), 0.f, (float)(H[(size_t)Nc].real() * inv), 0.f);
      } else {
        cd Hk = H[(size_t)k];
        cd Gk = std::conj(H[(size_t)(Nc - k)]);
        double ang = -2.0 * M_PI * (double)k / nfft;
        double s = std::sin(ang), c = std::cos(ang);
        cd alpha = 0.5 * inv * (Hk * (1.0 + s) + Gk * (1.0 - s));
        cd beta = cd(0, 0.5 * inv * c) * (Hk - Gk);
        o = make_float4((float)alpha.real(), (float)alpha.imag(), (float)beta.real(), (float)beta.imag());
      }
      ab[(size_t)k1 * imp::kN2 + (size_t)q * 256 + (size_t)u] = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// convolution plan
// ------------------------------------------------------------------------------------------------
static constexpr int kMaxTimed = 8192;

struct imp_plan {
  imp_ctx* ctx = nullptr;
  int64_t L = 0, M = 0, n_filters = 1;
  int mode = IMP_MODE_SAME;
  int64_t out_start = 0, out_len = 0;
  int64_t nfft = 0, Nc = 0;
  // overlap-add (need > 2^21 points): pieces of ola_lb input samples x ola_mp filter taps, each a 'full' transform
  bool ola = false;
  int64_t ola_lb = 0, ola_mp = 0, ola_blocks = 1, ola_parts = 1;
  int N1 = 0, R2 = 0, F = 16;       // N1 = F * R2 rows; F = rows per thread in the column passes
  int64_t ws_channels = 0;           // channels the workspace holds in total
  int lanes = 1;                     // launch groups in flight (imp_plan_set_overlap)
  int64_t group_counter_lane = 0;    // round-robin lane assignment of launch groups
  hipStream_t cur_stream = nullptr;  // lane the kernels of the current launch group go to
  cf* cur_ws = nullptr;
  TwSet tw;
  float4* ab = nullptr;    // [n_filters][N1][4096]
  cf* ws = nullptr;        // [ws_channels][N1][4096] (pair mode: [ws_channels / 2][N1][4096])
  // fused FIR plan (conv_kernels.hip.h fir_block_kernel): overlap-save blocks of 32 768 samples, one launch, no workspace
  bool fused = false;
  int f_kp = 0, f_valid = 0;        // history samples a block starts with (taps - 1, even), outputs per block
  int64_t f_first = 0, f_blocks = 0; // first block that touches the kept window, blocks per channel
  // pair mode (conv_kernels.hip.h): two channels per transform, z = x_L + i x_R; Nc = nfft = circular length in samples
  bool paired = false;
  cf* hs = nullptr;        // [N1][4096]: H / Nc in the register order of rows_single_kernel
  // staging for the host-buffer entry points
  float* d_in = nullptr;
  float* d_out = nullptr;
  size_t d_in_bytes = 0, d_out_bytes = 0;
  // set by imp_chain around a launch: pass C also leaves K3's chunk maxima (ir_kernels.hip.h StoreRealCropMax)
  unsigned* tile_max = nullptr;     // [channels][column tiles][N1]
  // timing
  int timing = 0;                    // 0 = off, n = record every n-th launch group
  int64_t group_counter = 0;
  bool group_timed = false;
  std::vector<hipEvent_t> events;   // 4 per launch group
  int64_t timed = 0;
  double acc_ms[3] = {0, 0, 0};
  int64_t acc_launches = 0;
};

template <int R2, int DIR, class Load, class Store>
static int launch_cols(imp_plan* p, int64_t nchan, Load ld, Store st) {
  using Cfg = imp::ColsCfg<R2>;
  auto kern = imp::cols_kernel<R2, DIR, Load, Store>;
  int rc_attr = ctx_kernel_lds(p->ctx, reinterpret_cast<const void*>(kern), Cfg::lds_bytes);
  if (rc_attr) return rc_attr;
  const int tiles = imp::kN2 / Cfg::TC;
  imp::Twiddles tw{p->tw.full, p->tw.hi, p->ctx->tw_t1, p->ctx->tw_t2, p->ctx->tw_t4};
  dim3 grid((unsigned)(nchan * tiles)), block(Cfg::T);
  hipLaunchKernelGGL(kern, grid, block, Cfg::lds_bytes, p->cur_stream, ld, st, tw, (int)nchan, (int)p->N1);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

template <int F, int R2, int DIR, class Load, class Store>
static int launch_cols_mixed(imp_plan* p, int64_t nchan, Load ld, Store st) {
  using Cfg = imp::MixCfg<F, R2>;
  auto kern = imp::cols_mixed_kernel<F, R2, DIR, Load, Store>;
  int rc_attr = ctx_kernel_lds(p->ctx, reinterpret_cast<const void*>(kern), Cfg::lds_bytes);
  if (rc_attr) return rc_attr;
  const int tiles = imp::kN2 / Cfg::TC;
  imp::Twiddles tw{p->tw.full, p->tw.hi, p->ctx->tw_t1, p->ctx->tw_t2, p->ctx->tw_t4};
  dim3 grid((unsigned)(nchan * tiles)), block(Cfg::T);
  hipLaunchKernelGGL(kern, grid, block, Cfg::lds_bytes, p->cur_stream, ld, st, tw, (int)nchan, (int)p->N1);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

template <int F, int DIR, class Load, class Store>
static int launch_cols_small(imp_plan* p, int64_t nchan, Load ld, Store st) {
  imp::Twiddles tw{p->tw.full, p->tw.hi, p->ctx->tw_t1, p->ctx->tw_t2, p->ctx->tw_t4};
  dim3 grid((unsigned)(nchan * (imp::kN2 / 256))), block(256);
  hipLaunchKernelGGL((imp::cols_small_kernel<F, DIR, Load, Store>), grid, block, 0, p->cur_stream, ld, st, tw, (int)nchan,
                     (int)p->N1);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

// column tiles per row of the kernel launch_cols_any picks for this plan (kN2 / its TC)
static int plan_col_tiles(const imp_plan* p) {
  if (p->R2 == 1 && (p->F == 4 || p->F == 8)) return imp::kN2 / 256;
  const bool pow2 = p->F == 16 && (p->R2 == 1 || p->R2 == 2 || p->R2 == 4 || p->R2 == 8 || p->R2 == 16);
  const int tc = pow2 ? (p->R2 >= 16 ? 32 : 64) : (p->F * p->R2 >= 192 ? 32 : 64);
  return imp::kN2 / tc;
}

template <int DIR, class Load, class Store>
static int launch_cols_any(imp_plan* p, int64_t nchan, Load ld, Store st) {
  if (p->R2 == 1 && p->F == 4) return launch_cols_small<4, DIR>(p, nchan, ld, st);
  if (p->R2 == 1 && p->F == 8) return launch_cols_small<8, DIR>(p, nchan, ld, st);
  if (p->F == 11 && p->R2 == 6) return launch_cols_mixed<11, 6, DIR>(p, nchan, ld, st);
  if (p->F == 11 && p->R2 == 12) return launch_cols_mixed<11, 12, DIR>(p, nchan, ld, st);
  if (p->F == 8) {
    switch (p->R2) {
      case 3: return launch_cols_mixed<8, 3, DIR>(p, nchan, ld, st);
      case 5: return launch_cols_mixed<8, 5, DIR>(p, nchan, ld, st);
      case 9: return launch_cols_mixed<8, 9, DIR>(p, nchan, ld, st);
    }
    return fail(IMP_ERR_UNSUPPORTED, "unsupported column factorisation 8 x %d", p->R2);
  }
  switch (p->R2) {
    case 1: return launch_cols<1, DIR>(p, nchan, ld, st);
    case 2: return launch_cols<2, DIR>(p, nchan, ld, st);
    case 4: return launch_cols<4, DIR>(p, nchan, ld, st);
    case 8: return launch_cols<8, DIR>(p, nchan, ld, st);
    case 16: return launch_cols<16, DIR>(p, nchan, ld, st);
    case 3: return launch_cols_mixed<16, 3, DIR>(p, nchan, ld, st);
    case 5: return launch_cols_mixed<16, 5, DIR>(p, nchan, ld, st);
    case 6: return launch_cols_mixed<16, 6, DIR>(p, nchan, ld, st);
    case 9: return launch_cols_mixed<16, 9, DIR>(p, nchan, ld, st);
    case 10: return launch_cols_mixed<16, 10, DIR>(p, nchan, ld, st);
    case 12: return launch_cols_mixed<16, 12, DIR>(p, nchan, ld, st);
    case 18: return launch_cols_mixed<16, 18, DIR>(p, nchan, ld, st);
    case 24: return launch_cols_mixed<16, 24, DIR>(p, nchan, ld, st);
  }
  return fail(IMP_ERR_UNSUPPORTED, "unsupported column radix %d", p->R2);
}

static constexpr size_t kRowsLds = sizeof(cf) * 2 * 16 * imp::kRowPad;

static int launch_rows(imp_plan* p, int64_t nchan, int64_t first_chan, int64_t part = 0) {
  // experiment switch: a larger LDS request leaves ONE row pair per CU instead of two
  static const size_t rows_lds = [] {
    const char* e = std::getenv("IMPULSE_HIP_ROWS_LDS");
    return e ? std::max((size_t)std::atoll(e), (size_t)kRowsLds) : (size_t)kRowsLds;
  }();
  int rc_attr = ctx_kernel_lds(p->ctx, reinterpret_cast<const void*>(imp::rows_kernel), rows_lds);
  if (rc_attr) return rc_attr;
  imp::RowsArgs a;
  a.ws = p->cur_ws;
  const int64_t plane = (int64_t)p->N1 * imp::kN2;
  a.ab = p->ab + (p->n_filters > 1 ? first_chan * p->ola_parts * plane : 0) + part * plane;
  a.ab_chan_stride = p->n_filters > 1 ? p->ola_parts * plane : 0;
  a.n1_total = p->N1;
  a.npairs = p->N1 / 2;
  a.nchan = (int)nchan;
  imp::Twiddles tw{p->tw.full, p->tw.hi, p->ctx->tw_t1, p->ctx->tw_t2, p->ctx->tw_t4};
  // the XCD-aware work mapping deals pairs in eights: pad, the surplus workgroups exit at once
  dim3 grid((unsigned)(nchan * ((a.npairs + 7) / 8 * 8))), block(512);
  hipLaunchKernelGGL(imp::rows_kernel, grid, block, rows_lds, p->cur_stream, a, tw);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

static constexpr size_t kRowsSingleLds = sizeof(cf) * 16 * imp::kRowPad;

static int launch_rows_single(imp_plan* p, int64_t npairs) {
  int rc_attr = ctx_kernel_lds(p->ctx, reinterpret_cast<const void*>(imp::rows_single_kernel), kRowsSingleLds);
  if (rc_attr) return rc_attr;
  imp::RowsPairArgs a;
  a.ws = p->cur_ws;
  a.hs = p->hs;
  a.n1_total = p->N1;
  a.npairs = (int)npairs;
  imp::Twiddles tw{p->tw.full, p->tw.hi, p->ctx->tw_t1, p->ctx->tw_t2, p->ctx->tw_t4};
  // the XCD-aware work mapping deals rows in eights: pad, the surplus workgroups exit at once
  dim3 grid((unsigned)(npairs * ((p->N1 + 7) / 8 * 8))), block(256);
  hipLaunchKernelGGL(imp::rows_single_kernel, grid, block, kRowsSingleLds, p->cur_stream, a, tw);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

// taps a fused FIR plan takes: a block then still yields 8 191 or more of its 32 768 samples
static constexpr int64_t kFusedMaxTaps = 24577;

static int plan_geometry(imp_plan* p, int64_t M, int64_t n_filters, int64_t L, int mode, int64_t ws_channels,
                         bool paired = false, bool allow_fused = true) {
  if (M < 1 || L < 1) return fail(IMP_ERR_INVALID, "M and L must be >= 1 (M=%lld L=%lld)", (long long)M, (long long)L);
  if (n_filters < 1) return fail(IMP_ERR_INVALID, "n_filters must be >= 1");
  if (mode != IMP_MODE_SAME && mode != IMP_MODE_FULL) return fail(IMP_ERR_INVALID, "mode must be IMP_MODE_SAME or IMP_MODE_FULL");
  // Circular length the transform must cover.  'full' needs every sample of the linear
  // convolution: L+M-1.  'same' keeps only [s0, s0+L) with s0 = (M-1)/2, so wrap-around may land in
  // the discarded part: P >= L + ceil((M-1)/2) = L + M/2 is enough (and P >= M so the filter fits).
  const int64_t full = L + M - 1;
  const int64_t need = (mode == IMP_MODE_SAME) ? std::max(L + M / 2, M) : full;
  // N1 = F * R2 rows of 4096 complex points, ascending; F = rows per thread in the column passes
  // pair mode: a row holds 4096 samples of both channels (mono: 8192 of one), so a pair plan has twice the rows of the
  // mono plan for the same lengths; 11 x 12 = 132 rows is the 7.1 / 6.15 s configuration there (mono: 11 x 6 = 66)
  static const struct { int f, r2; bool pair_only; } kShapes[] = {
      {4, 1, false},  {8, 1, false},  {16, 1, false}, {8, 3, false},  {16, 2, false},  {8, 5, false},
      {16, 3, false}, {16, 4, false}, {11, 6, false}, {8, 9, false},  {16, 5, false},  {16, 6, false},
      {16, 8, false}, {11, 12, true}, {16, 9, false}, {16, 10, false}, {16, 12, false}, {16, 16, false}, {16, 18, true},
      {16, 24, true}};
  const int64_t samples_per_row = paired ? imp::kN2 : 2 * imp::kN2;
  if (paired && n_filters != 1) return fail(IMP_ERR_INVALID, "pair mode needs ONE filter shared by both channels of a pair");
  // Short filters (every FIR of the path: 9 600 taps at 48 kHz, 19 200 at 96 kHz): one launch of overlap-save blocks that stay
  // in a CU's registers and LDS instead of a three-launch transform over the whole input (IMPULSE_HIP_NO_FUSED_FIR=1 or
  // imp_conv_plan_create_ex(..., IMP_PLAN_NO_FUSED) keep the three-launch plan)
  p->fused = allow_fused && !paired && M <= kFusedMaxTaps && std::getenv("IMPULSE_HIP_NO_FUSED_FIR") == nullptr;
  if (p->fused && (L >= ((int64_t)1 << 29) || full >= ((int64_t)1 << 29)))
    return fail(IMP_ERR_UNSUPPORTED, "L = %lld beyond the 2^29 samples one channel's buffer range covers", (long long)L);
  if (p->fused) {
    p->ola = false;
    p->paired = false;
    p->L = L;
    p->M = M;
    p->n_filters = n_filters;
    p->mode = mode;
    p->R2 = 1;
    p->F = 4;
    p->N1 = 4;
    p->Nc = 4 * imp::kN2;
    p->nfft = 2 * p->Nc;
    p->out_start = mode == IMP_MODE_SAME ? (M - 1) / 2 : 0;
    p->out_len = mode == IMP_MODE_SAME ? L : full;
    p->f_kp = (int)((M - 1 + 1) & ~(int64_t)1);
    p->f_valid = (int)(p->nfft - p->f_kp);
    p->f_first = p->out_start / p->f_valid;
    p->f_blocks = (p->out_start + p->out_len - 1) / p->f_valid - p->f_first + 1;
    p->ws_channels = ws_channels > 0 ? ws_channels : 64;
    return IMP_OK;
  }
  int r2 = 0, f1 = 16;
  const char* min_rows_env = std::getenv("IMPULSE_HIP_MIN_ROWS");             // experiments: 16 = the round-1 smallest plan
  const int min_rows = min_rows_env ? atoi(min_rows_env) : 0;
  const bool pow2_only = std::getenv("IMPULSE_HIP_POW2_ONLY") != nullptr;     // debug/experiments
  const bool no_wrap = std::getenv("IMPULSE_HIP_NO_WRAP") != nullptr;
  const int64_t need_eff = no_wrap ? full : need;
  for (const auto& sh : kShapes) {
    const int n1 = sh.f * sh.r2;
    if (sh.pair_only && !paired) continue;
    if (n1 >= min_rows && (!pow2_only || (n1 & (n1 - 1)) == 0) && (int64_t)n1 * samples_per_row >= need_eff) {
      r2 = sh.r2;
      f1 = sh.f;
      break;
    }
  }
  p->ola = false;
  p->paired = paired;
  if (!r2 && paired)
    return fail(IMP_ERR_UNSUPPORTED, "pair mode covers circular lengths up to 1 572 864 samples (384 rows); this plan needs %lld",
                (long long)need_eff);
  if (!r2) {
    // Longer than one two-level transform (2^21 points): overlap-add.  The input is cut into blocks and, when
    // the filter itself is longer than 2^20 taps, the filter into partitions; every (block, partition) piece is
    // a 'full' convolution of at most 2^21 points through the same three passes, added into the result.
    if (L >= ((int64_t)1 << 29) || M >= ((int64_t)1 << 29))
      return fail(IMP_ERR_UNSUPPORTED, "L = %lld / M = %lld beyond the 2^29 samples one channel's buffer range covers",
                  (long long)L, (long long)M);
    const int64_t cap = (int64_t)1 << 21;
    p->ola = true;
    p->ola_mp = std::min<int64_t>(M, cap / 2);
    p->ola_lb = cap - p->ola_mp + 1;                       // Lb + Mp - 1 = 2^21
    p->ola_parts = (M + p->ola_mp - 1) / p->ola_mp;
    p->ola_blocks = (L + p->ola_lb - 1) / p->ola_lb;
    r2 = 16;
    f1 = 16;
  }
  p->L = L;
  p->M = M;
  p->n_filters = n_filters;
  p->mode = mode;
  p->R2 = r2;
  p->F = f1;
  p->N1 = f1 * r2;
  p->Nc = (int64_t)p->N1 * imp::kN2;
  p->nfft = paired ? p->Nc : 2 * p->Nc;
  if (mode == IMP_MODE_SAME) {
    // scipy.signal._signaltools._centered: start = (full - L) // 2 with full = L + M - 1
    p->out_start = (M - 1) / 2;
    p->out_len = L;
  } else {
    p->out_start = 0;
    p->out_len = full;
  }
  if (ws_channels <= 0) {
    // keep the workspace within ~100 MiB so A->B->C hand-offs stay in the 256 MiB Infinity Cache
    // (measured on C5/C3: groups whose workspace + inputs + outputs exceed the cache lose 5-20 %)
    ws_channels = std::max<int64_t>(1, ((int64_t)100 << 20) / (p->Nc * (int64_t)sizeof(cf))) * (paired ? 2 : 1);
  }
  if (paired) ws_channels += ws_channels & 1;            // whole pairs
  p->ws_channels = ws_channels;
  return IMP_OK;
}

static int plan_alloc(imp_plan* p) {
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  if ((rc = ctx_twiddles(p->ctx, p->N1, &p->tw))) return rc;
  const size_t plane = (size_t)p->N1 * imp::kN2;
  hipError_t e;
  if (p->paired) e = hipMalloc((void**)&p->hs, plane * sizeof(cf));
  else e = hipMalloc((void**)&p->ab, plane * (size_t)(p->n_filters * p->ola_parts) * sizeof(float4));
  if (e != hipSuccess) return fail(IMP_ERR_ALLOC, "hipMalloc(spectrum): %s", hipGetErrorString(e));
  e = hipMalloc((void**)&p->ws, p->fused ? 256 : plane * (size_t)(p->paired ? p->ws_channels / 2 : p->ws_channels) * sizeof(cf));
  if (e != hipSuccess) return fail(IMP_ERR_ALLOC, "hipMalloc(workspace): %s", hipGetErrorString(e));
  return IMP_OK;
}

static int plan_sync_lanes(imp_plan* p);

extern "C" void imp_plan_destroy(imp_plan* p) {
  if (!p) return;
  IMP_CTX_LOCK(p->ctx);
  (void)hipSetDevice(p->ctx->device);
  (void)plan_sync_lanes(p);
  if (p->ab) (void)hipFree(p->ab);
  if (p->hs) (void)hipFree(p->hs);
  if (p->ws) (void)hipFree(p->ws);
  if (p->d_in) (void)hipFree(p->d_in);
  if (p->d_out) (void)hipFree(p->d_out);
  for (auto ev : p->events) (void)hipEventDestroy(ev);
  delete p;
}

static int plan_create_empty_impl(imp_ctx* ctx, int64_t M, int64_t n_filters, int64_t L, int mode, int64_t ws_channels,
                                  bool paired, imp_plan** out, bool allow_fused = true) {
  if (!ctx || !out) return fail(IMP_ERR_INVALID, "imp_conv_plan_create_empty: null argument");
  IMP_CTX_LOCK(ctx);
  *out = nullptr;
  imp_plan* p = new (std::nothrow) imp_plan();
  if (!p) return fail(IMP_ERR_ALLOC, "out of host memory");
  p->ctx = ctx;
  int rc = plan_geometry(p, M, n_filters, L, mode, ws_channels, paired, allow_fused);
  if (!rc) rc = plan_alloc(p);
  if (rc) {
    imp_plan_destroy(p);
    return rc;
  }
  *out = p;
  return IMP_OK;
}

extern "C" int imp_conv_plan_create_empty(imp_ctx* ctx, int64_t M, int64_t n_filters, int64_t L, int mode,
                                          int64_t ws_channels, imp_plan** out) {
  return plan_create_empty_impl(ctx, M, n_filters, L, mode, ws_channels, false, out);
}

extern "C" int imp_conv_plan_create_empty_paired(imp_ctx* ctx, int64_t M, int64_t L, int mode, int64_t ws_channels,
                                                 imp_plan** out) {
  return plan_create_empty_impl(ctx, M, 1, L, mode, ws_channels, true, out);
}

// the plan's spectrum planes from host filters (fp64 on the device, rounded once); work in flight must be drained.
// on_device: the filters are device memory (mono / fused plans without overlap-add): nothing is uploaded and nothing waits.
static int plan_fill_spectrum(imp_plan* p, const double* filter, int64_t filter_ld, bool on_device = false) {
  imp_ctx* ctx = p->ctx;
  const int64_t M = p->M, n_filters = p->n_filters;
  const int64_t ld = n_filters > 1 ? filter_ld : M;
  if (on_device) {
    if (p->paired || p->ola) return fail(IMP_ERR_UNSUPPORTED, "device filters: mono and fused plans without overlap-add");
    return spectrum_alpha_beta_device(ctx, filter, M, n_filters, ld, p->Nc, p->N1, p->ab, true);
  }
  if (p->paired) return spectrum_pair_device(ctx, filter, M, p->Nc, p->N1, p->hs);
  const size_t plane = (size_t)p->N1 * imp::kN2;
  // IMPULSE_HIP_HOST_SPECTRUM=1 keeps the fp64 host preparation (the cross-check path of the tests)
  const char* host_env = std::getenv("IMPULSE_HIP_HOST_SPECTRUM");
  if (host_env && host_env[0] == '1' && !p->ola) {
    std::vector<float4> ab(plane);
    std::vector<cd> H;
    for (int64_t f = 0; f < n_filters; ++f) {
      host_rfft(filter + f * ld, M, p->Nc, H);
      host_alpha_beta(H, p->Nc, p->N1, ab.data());
      hipError_t e = hipMemcpyAsync(p->ab + (size_t)f * plane, ab.data(), plane * sizeof(float4),
                                    hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) return fail(IMP_ERR_HIP, "spectrum upload: %s", hipGetErrorString(e));
    }
    return IMP_OK;
  }
  if (!p->ola) return spectrum_alpha_beta_device(ctx, filter, M, n_filters, ld, p->Nc, p->N1, p->ab);
  // overlap-add: plane (f, j) = partition j of filter f
  for (int64_t j = 0; j < p->ola_parts; ++j) {
    const int64_t m0 = j * p->ola_mp, mj = std::min(p->ola_mp, M - m0);
    for (int64_t f = 0; f < n_filters; ++f) {
      int rc = spectrum_alpha_beta_device(ctx, filter + f * ld + m0, mj, 1, mj, p->Nc, p->N1,
                                          p->ab + (size_t)(f * p->ola_parts + j) * plane);
      if (rc) return rc;
    }
  }
  return IMP_OK;
}

static int plan_create_impl(imp_ctx* ctx, const double* filter, int64_t M, int64_t n_filters, int64_t filter_ld, int64_t L,
                            int mode, int64_t ws_channels, bool paired, imp_plan** out, bool allow_fused = true) {
  if (!ctx || !out) return fail(IMP_ERR_INVALID, "imp_conv_plan_create: null argument");
  if (!filter) return fail(IMP_ERR_INVALID, "imp_conv_plan_create: null filter");
  IMP_CTX_LOCK(ctx);
  if (n_filters > 1 && filter_ld < M) return fail(IMP_ERR_INVALID, "filter_ld < M");
  imp_plan* p = nullptr;
  int rc = plan_create_empty_impl(ctx, M, n_filters, L, mode, ws_channels, paired, &p, allow_fused);
  if (rc) return rc;
  if ((rc = plan_fill_spectrum(p, filter, filter_ld))) {
    imp_plan_destroy(p);
    return rc;
  }
  *out = p;
  return IMP_OK;
}

extern "C" int imp_conv_plan_create(imp_ctx* ctx, const double* filter, int64_t M, int64_t n_filters,
                                    int64_t filter_ld, int64_t L, int mode, int64_t ws_channels,
                                    imp_plan** out) {
  return plan_create_impl(ctx, filter, M, n_filters, filter_ld, L, mode, ws_channels, false, out);
}

extern "C" int imp_conv_plan_create_paired(imp_ctx* ctx, const double* filter, int64_t M, int64_t L, int mode,
                                           int64_t ws_channels, imp_plan** out) {
  return plan_create_impl(ctx, filter, M, 1, M, L, mode, ws_channels, true, out);
}

extern "C" int imp_conv_plan_create_ex(imp_ctx* ctx, const double* filter, int64_t M, int64_t n_filters, int64_t filter_ld,
                                       int64_t L, int mode, int64_t ws_channels, int flags, imp_plan** out) {
  if (flags & ~(IMP_PLAN_PAIRED | IMP_PLAN_NO_FUSED)) return fail(IMP_ERR_INVALID, "imp_conv_plan_create_ex: unknown flag bits %d", flags);
  const bool paired = (flags & IMP_PLAN_PAIRED) != 0;
  if (paired && n_filters != 1) return fail(IMP_ERR_INVALID, "pair mode needs ONE filter shared by both channels of a pair");
  if (!filter) return plan_create_empty_impl(ctx, M, n_filters, L, mode, ws_channels, paired, out, !(flags & IMP_PLAN_NO_FUSED));
  return plan_create_impl(ctx, filter, M, n_filters, filter_ld, L, mode, ws_channels, paired, out, !(flags & IMP_PLAN_NO_FUSED));
}

extern "C" int imp_plan_kind(const imp_plan* p, int* kind) {
  if (!p || !kind) return fail(IMP_ERR_INVALID, "imp_plan_kind: null argument");
  *kind = p->fused ? 2 : p->paired ? 1 : 0;
  return IMP_OK;
}

extern "C" int imp_plan_is_paired(const imp_plan* p, int* paired) {
  if (!p || !paired) return fail(IMP_ERR_INVALID, "imp_plan_is_paired: null argument");
  *paired = p->paired ? 1 : 0;
  return IMP_OK;
}

extern "C" int imp_plan_set_filters(imp_plan* p, const double* filter, int64_t filter_ld) {
  if (!p || !filter) return fail(IMP_ERR_INVALID, "imp_plan_set_filters: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (p->n_filters > 1 && filter_ld < p->M) return fail(IMP_ERR_INVALID, "filter_ld < M");
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  if ((rc = plan_sync_lanes(p))) return rc;
  return plan_fill_spectrum(p, filter, filter_ld);
}

extern "C" int imp_plan_set_filters_device(imp_plan* p, const double* d_filter, int64_t filter_ld) {
  if (!p || !d_filter) return fail(IMP_ERR_INVALID, "imp_plan_set_filters_device: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (p->n_filters > 1 && filter_ld < p->M) return fail(IMP_ERR_INVALID, "filter_ld < M");
  if (p->lanes != 1) return fail(IMP_ERR_INVALID, "imp_plan_set_filters_device: the plan must run in stream order (lanes = 1)");
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  // stream order does the rest: launches queued earlier still read the old planes, later ones the new
  return plan_fill_spectrum(p, d_filter, filter_ld, true);
}

extern "C" int imp_debug_plan_geometry(int64_t M, int64_t L, int mode, int64_t* nfft, int64_t* out_start,
                                       int64_t* out_len) {
  imp_plan tmp;
  int rc = plan_geometry(&tmp, M, 1, L, mode, 1, false, false);      // the three-launch geometry
  if (rc) return rc;
  if (nfft) *nfft = tmp.nfft;
  if (out_start) *out_start = tmp.out_start;
  if (out_len) *out_len = tmp.out_len;
  return IMP_OK;
}

extern "C" int imp_debug_plan_geometry_fused(int64_t M, int64_t L, int mode, int64_t* history, int64_t* valid,
                                             int64_t* first_block, int64_t* blocks) {
  if (M > kFusedMaxTaps) return fail(IMP_ERR_UNSUPPORTED, "a fused FIR plan takes at most %lld taps (got %lld)",
                                     (long long)kFusedMaxTaps, (long long)M);
  imp_plan tmp;
  int rc = plan_geometry(&tmp, M, 1, L, mode, 1, false, true);
  if (rc) return rc;
  if (!tmp.fused) return fail(IMP_ERR_UNSUPPORTED, "fused FIR plans are switched off (IMPULSE_HIP_NO_FUSED_FIR)");
  if (history) *history = tmp.f_kp;
  if (valid) *valid = tmp.f_valid;
  if (first_block) *first_block = tmp.f_first;
  if (blocks) *blocks = tmp.f_blocks;
  return IMP_OK;
}

extern "C" int imp_debug_plan_geometry_paired(int64_t M, int64_t L, int mode, int64_t* nfft, int64_t* out_start,
                                              int64_t* out_len, int64_t* n1_rows) {
  imp_plan tmp;
  int rc = plan_geometry(&tmp, M, 1, L, mode, 2, true);
  if (rc) return rc;
  if (nfft) *nfft = tmp.nfft;
  if (out_start) *out_start = tmp.out_start;
  if (out_len) *out_len = tmp.out_len;
  if (n1_rows) *n1_rows = tmp.N1;
  return IMP_OK;
}

extern "C" int imp_debug_host_spectrum(const double* filter, int64_t M, int n1_rows, float* ab_out) {
  if (!filter || !ab_out || M < 1 || n1_rows < 4 || n1_rows % 2) return fail(IMP_ERR_INVALID, "imp_debug_host_spectrum: bad argument");
  const int64_t Nc = (int64_t)n1_rows * imp::kN2;
  if (M > 2 * Nc) return fail(IMP_ERR_INVALID, "filter longer than the transform");
  std::vector<cd> H;
  host_rfft(filter, M, Nc, H);
  host_alpha_beta(H, Nc, n1_rows, reinterpret_cast<float4*>(ab_out));
  return IMP_OK;
}

extern "C" int imp_plan_info(const imp_plan* p, int64_t* nfft, int64_t* out_len, int64_t* ws_channels,
                             int64_t* n1_rows) {
  if (!p) return fail(IMP_ERR_INVALID, "null plan");
  if (nfft) *nfft = p->nfft;
  if (out_len) *out_len = p->out_len;
  if (ws_channels) *ws_channels = p->ws_channels;
  if (n1_rows) *n1_rows = p->N1;
  return IMP_OK;
}

extern "C" int imp_plan_spectrum(imp_plan* p, void** dptr, size_t* bytes) {
  if (!p || !dptr || !bytes) return fail(IMP_ERR_INVALID, "imp_plan_spectrum: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (p->paired) {
    *dptr = p->hs;
    *bytes = (size_t)p->N1 * imp::kN2 * sizeof(cf);
    return IMP_OK;
  }
  *dptr = p->ab;
  // every plane the row pass can read: per-channel filters x overlap-add filter partitions
  *bytes = (size_t)p->N1 * imp::kN2 * (size_t)(p->n_filters * p->ola_parts) * sizeof(float4);
  return IMP_OK;
}

// The prepared spectrum of `src` into `dst` (a plan of the same geometry, made empty) - the one datum GPUs share: device 0
// prepares it, the other devices of a single process receive a peer copy (between processes: imp_plan_broadcast_spectrum).
extern "C" int imp_plan_copy_spectrum(imp_plan* dst, imp_plan* src) {
  if (!dst || !src) return fail(IMP_ERR_INVALID, "imp_plan_copy_spectrum: null argument");
  if (dst == src) return IMP_OK;
  if (dst->paired != src->paired || dst->fused != src->fused || dst->N1 != src->N1 || dst->M != src->M || dst->L != src->L ||
      dst->mode != src->mode || dst->n_filters != src->n_filters || dst->ola_parts != src->ola_parts)
    return fail(IMP_ERR_INVALID, "imp_plan_copy_spectrum: the two plans differ in geometry");
  void *d = nullptr, *s_ = nullptr;
  size_t nd = 0, ns = 0;
  int rc;
  if ((rc = imp_plan_spectrum(src, &s_, &ns)) || (rc = imp_plan_spectrum(dst, &d, &nd))) return rc;
  if (nd != ns) return fail(IMP_ERR_INVALID, "imp_plan_copy_spectrum: spectrum sizes differ");
  {
    IMP_CTX_LOCK(src->ctx);                              // the source must be complete
    if ((rc = ctx_bind(src->ctx))) return rc;
    HIP_TRY(hipStreamSynchronize(src->ctx->stream));
  }
  IMP_CTX_LOCK(dst->ctx);
  if ((rc = ctx_bind(dst->ctx))) return rc;
  if ((rc = plan_sync_lanes(dst))) return rc;
  HIP_TRY(hipMemcpyPeerAsync(d, dst->ctx->device, s_, src->ctx->device, ns, dst->ctx->stream));
  HIP_TRY(hipStreamSynchronize(dst->ctx->stream));
  return IMP_OK;
}

// bytes from a buffer of one context's device to a buffer of another's (or the same device's): waits for the source
// context's stream, then copies in the order of the destination context's stream (asynchronous there)
extern "C" int imp_memcpy_peer(imp_ctx* dst_ctx, void* dst, imp_ctx* src_ctx, const void* src, size_t bytes) {
  if (!dst_ctx || !src_ctx || (bytes && (!dst || !src))) return fail(IMP_ERR_INVALID, "imp_memcpy_peer: null argument");
  if (!bytes) return IMP_OK;
  int rc;
  {
    IMP_CTX_LOCK(src_ctx);
    if ((rc = ctx_bind(src_ctx))) return rc;
    HIP_TRY(hipStreamSynchronize(src_ctx->stream));
  }
  IMP_CTX_LOCK(dst_ctx);
  if ((rc = ctx_bind(dst_ctx))) return rc;
  HIP_TRY(hipMemcpyPeerAsync(dst, dst_ctx->device, src, src_ctx->device, bytes, dst_ctx->stream));
  return IMP_OK;
}

extern "C" int imp_plan_set_timing(imp_plan* p, int enable) {
  if (!p) return fail(IMP_ERR_INVALID, "null plan");
  IMP_CTX_LOCK(p->ctx);
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  p->timing = enable < 0 ? 0 : enable;
  p->group_counter = 0;
  return IMP_OK;
}

static int plan_sync_lanes(imp_plan* p) {
  HIP_TRY(hipStreamSynchronize(p->ctx->stream));
  for (int l = 1; l < p->lanes && l - 1 < (int)p->ctx->side_streams.size(); ++l)
    HIP_TRY(hipStreamSynchronize(p->ctx->side_streams[(size_t)(l - 1)]));
  return IMP_OK;
}

static int plan_collect_timing(imp_plan* p) {
  if (p->timed == 0) return IMP_OK;
  int rcs = plan_sync_lanes(p);
  if (rcs) return rcs;
  for (int64_t i = 0; i < p->timed; ++i) {
    for (int k = 0; k < 3; ++k) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, p->events[(size_t)(4 * i + k)], p->events[(size_t)(4 * i + k + 1)]));
      p->acc_ms[k] += ms;
    }
  }
  p->acc_launches += p->timed;
  p->timed = 0;
  return IMP_OK;
}

extern "C" int imp_plan_get_timing(imp_plan* p, double ms[3], int64_t* launches, int reset) {
  if (!p || !ms || !launches) return fail(IMP_ERR_INVALID, "imp_plan_get_timing: null argument");
  IMP_CTX_LOCK(p->ctx);
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  if ((rc = plan_collect_timing(p))) return rc;
  for (int k = 0; k < 3; ++k) ms[k] = p->acc_ms[k];
  *launches = p->acc_launches;
  if (reset) {
    p->acc_ms[0] = p->acc_ms[1] = p->acc_ms[2] = 0;
    p->acc_launches = 0;
  }
  return IMP_OK;
}

static int timing_event(imp_plan* p, int slot) {
  if (!p->timing) return IMP_OK;
  if (slot == 0) p->group_timed = (p->group_counter++ % p->timing) == 0;
  if (!p->group_timed) return IMP_OK;
  if (p->timed >= kMaxTimed) {
    int rc = plan_collect_timing(p);   // drains the stream; only every kMaxTimed launch groups
    if (rc) return rc;
  }
  const size_t need = (size_t)(4 * (p->timed + 1));
  while (p->events.size() < need) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    p->events.push_back(ev);
  }
  HIP_TRY(hipEventRecord(p->events[(size_t)(4 * p->timed + slot)], p->cur_stream));
  if (slot == 3) ++p->timed;
  return IMP_OK;
}

// one launch group: nchan <= ws_channels channels, device pointers
template <class Load>
static int run_group_with(imp_plan* p, Load ld, int64_t nchan, float* d_y, int64_t chan_stride_out,
                          int64_t first_chan, int last_stage);

template <class Load>
static int launch_fir_block(imp_plan* p, Load ld, int64_t nchan, float* d_y, int64_t chan_stride_out, int64_t first_chan) {
  auto kern = imp::fir_block_kernel<Load>;
  int rc = ctx_kernel_lds(p->ctx, reinterpret_cast<const void*>(kern), imp::kFirBlockLds);
  if (rc) return rc;
  const int64_t plane = (int64_t)4 * imp::kN2;
  imp::FirBlockArgs a;
  a.ab = p->ab + (p->n_filters > 1 ? first_chan * plane : 0);
  a.ab_chan_stride = p->n_filters > 1 ? plane : 0;
  a.out = d_y;
  a.out_stride = chan_stride_out;
  a.out_start = p->out_start;
  a.first_block = (int)p->f_first;
  a.out_len = p->out_len;
  a.kp = p->f_kp;
  a.valid = p->f_valid;
  a.blocks = (int)p->f_blocks;
  a.nchan = (int)nchan;
  imp::Twiddles tw{p->tw.full, p->tw.hi, p->ctx->tw_t1, p->ctx->tw_t2, p->ctx->tw_t4};
  dim3 grid((unsigned)((nchan + 7) / 8 * 8 * p->f_blocks)), block(1024);
  hipLaunchKernelGGL(kern, grid, block, imp::kFirBlockLds, p->cur_stream, ld, a, tw);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

// one launch group in pair mode: channels (2q, 2q + 1) of the group share a transform; ld.nchan = nchan
template <class Load>
static int run_group_pair(imp_plan* p, Load ld, int64_t nchan, float* d_y, int64_t chan_stride_out, int last_stage) {
  int rc;
  const int64_t lane_channels = p->ws_channels / p->lanes;
  const int lane = (p->lanes > 1) ? (int)(p->group_counter_lane++ % p->lanes) : 0;
  p->cur_stream = lane ? p->ctx->side_streams[(size_t)(lane - 1)] : p->ctx->stream;
  p->cur_ws = p->ws + (int64_t)lane * (lane_channels / 2) * p->N1 * imp::kN2;
  if (nchan < 1 || nchan > lane_channels || (lane_channels & 1))
    return fail(IMP_ERR_INVALID, "launch group of %lld channels exceeds the workspace lane (%lld, whole pairs)",
                (long long)nchan, (long long)lane_channels);
  const int64_t npairs = (nchan + 1) / 2;
  // the two rows of a pair are written through ONE 32-bit buffer range (StorePairCrop)
  if ((double)(chan_stride_out + p->out_len) * 4.0 >= 4294967296.0)
    return fail(IMP_ERR_INVALID, "chan_stride_out %lld: a pair's two output rows must lie within 4 GiB", (long long)chan_stride_out);
  imp::StoreWorkspace stw{p->cur_ws, p->N1};
  if ((rc = timing_event(p, 0))) return rc;
  if ((rc = launch_cols_any<-1>(p, npairs, ld, stw))) return rc;
  if ((rc = timing_event(p, 1))) return rc;
  if (last_stage < 1) return IMP_OK;
  if ((rc = launch_rows_single(p, npairs))) return rc;
  if ((rc = timing_event(p, 2))) return rc;
  if (last_stage < 2) return IMP_OK;
  imp::LoadWorkspace ldw{p->cur_ws, p->N1};
  imp::StorePairCrop stc{d_y, chan_stride_out, p->out_start, p->out_len, (int)nchan};
  if (p->tile_max) rc = launch_cols_any<+1>(p, npairs, ldw, imp::StorePairCropMax{stc, p->tile_max});
  else rc = launch_cols_any<+1>(p, npairs, ldw, stc);
  if (rc) return rc;
  return timing_event(p, 3);
}

static int run_group(imp_plan* p, const float* d_x, int64_t nchan, int64_t chan_stride_in,
                     int64_t elem_stride_in, float* d_y, int64_t chan_stride_out, int64_t first_chan,
                     int last_stage) {
  if (p->paired)
    return run_group_pair(p, imp::LoadPair<float>{d_x, 2 * chan_stride_in, chan_stride_in, elem_stride_in, p->L, (int)nchan, 0.f},
                          nchan, d_y, chan_stride_out, last_stage);
  imp::LoadRealPacked ld{d_x, chan_stride_in, elem_stride_in, p->L};
  return run_group_with(p, ld, nchan, d_y, chan_stride_out, first_chan, last_stage);
}

template <class Load>
static int run_group_with(imp_plan* p, Load ld, int64_t nchan, float* d_y, int64_t chan_stride_out,
                          int64_t first_chan, int last_stage) {
  int rc;
  if (p->paired) return fail(IMP_ERR_INVALID, "this loader packs one channel per transform: not a pair-mode plan");
  // lane = stream + private slice of the workspace; successive launch groups go round robin
  const int64_t lane_channels = p->ws_channels / p->lanes;
  const int lane = (p->lanes > 1) ? (int)(p->group_counter_lane++ % p->lanes) : 0;
  p->cur_stream = lane ? p->ctx->side_streams[(size_t)(lane - 1)] : p->ctx->stream;
  p->cur_ws = p->ws + (int64_t)lane * lane_channels * p->N1 * imp::kN2;
  // every kernel indexes ws[] by the group-local channel and ab[] by the global one: check both
  if (nchan < 1 || nchan > lane_channels)
    return fail(IMP_ERR_INVALID, "launch group of %lld channels exceeds the workspace lane (%lld)", (long long)nchan,
                (long long)lane_channels);
  if (p->n_filters > 1 && first_chan + nchan > p->n_filters)
    return fail(IMP_ERR_INVALID, "channel %lld has no filter: the plan holds %lld per-channel filters",
                (long long)(first_chan + nchan - 1), (long long)p->n_filters);
  if (p->fused) {
    if (last_stage < 2) return fail(IMP_ERR_UNSUPPORTED, "debug stages are not available on fused FIR plans");
    if (p->tile_max) return fail(IMP_ERR_UNSUPPORTED, "a fused FIR plan leaves no chunk maxima");
    if ((rc = timing_event(p, 0))) return rc;
    if ((rc = launch_fir_block(p, ld, nchan, d_y, chan_stride_out, first_chan))) return rc;
    for (int slot = 1; slot <= 3; ++slot)
      if ((rc = timing_event(p, slot))) return rc;              // the one launch is reported as "pass A"
    return IMP_OK;
  }
  imp::StoreWorkspace stw{p->cur_ws, p->N1};
  if (p->ola) {
    if (last_stage < 2) return fail(IMP_ERR_UNSUPPORTED, "debug stages are not available on overlap-add plans");
    HIP_TRY(hipMemset2DAsync(d_y, (size_t)chan_stride_out * sizeof(float), 0, (size_t)p->out_len * sizeof(float),
                             (size_t)nchan, p->cur_stream));
    imp::LoadWorkspace ldw{p->cur_ws, p->N1};
    for (int64_t i = 0; i < p->ola_blocks; ++i) {
      const Load ldi = ld.shifted(i * p->ola_lb, p->ola_lb);
      for (int64_t j = 0; j < p->ola_parts; ++j) {
        // pass B is in place, so every partition starts from a fresh forward transform of the block
        if ((rc = launch_cols_any<-1>(p, nchan, ldi, stw))) return rc;
        if ((rc = launch_rows(p, nchan, first_chan, j))) return rc;
        imp::StoreRealCropAdd sta{d_y, chan_stride_out, p->out_start - i * p->ola_lb - j * p->ola_mp, p->out_len};
        if ((rc = launch_cols_any<+1>(p, nchan, ldw, sta))) return rc;
      }
    }
    return IMP_OK;
  }
  if ((rc = timing_event(p, 0))) return rc;
  if ((rc = launch_cols_any<-1>(p, nchan, ld, stw))) return rc;
  if ((rc = timing_event(p, 1))) return rc;
  if (last_stage < 1) return IMP_OK;
  if ((rc = launch_rows(p, nchan, first_chan))) return rc;
  if ((rc = timing_event(p, 2))) return rc;
  if (last_stage < 2) return IMP_OK;
  imp::LoadWorkspace ldw{p->cur_ws, p->N1};
  imp::StoreRealCrop stc{d_y, chan_stride_out, p->out_start, p->out_len};
  if (p->tile_max) rc = launch_cols_any<+1>(p, nchan, ldw, imp::StoreRealCropMax{stc, p->tile_max});
  else rc = launch_cols_any<+1>(p, nchan, ldw, stc);
  if (rc) return rc;
  if ((rc = timing_event(p, 3))) return rc;
  return IMP_OK;
}

// One channel is addressed through a 32-bit buffer range, and the loaders form byte offsets up to the END OF THE
// TRANSFORM (nfft samples: the zero padding is the hardware range check), not just up to L: the whole span
// nfft * elem_stride * sizeof(sample) must stay below 4 GiB or a padded offset would wrap back into range.
static int check_input_span(const imp_plan* p, int64_t elem_stride, size_t sample_bytes) {
  // (a fused FIR plan forms sample offsets up to one block past the input's end)
  const int64_t reach = p->fused ? p->L + p->nfft : p->ola ? ((int64_t)1 << 21) : p->nfft;
  if ((double)reach * (double)elem_stride * (double)sample_bytes >= 4294967296.0)
    return fail(IMP_ERR_INVALID,
                "transform length %lld x elem_stride %lld x %zu B exceeds the 4 GiB buffer range of one channel",
                (long long)reach, (long long)elem_stride, sample_bytes);
  return IMP_OK;
}

extern "C" int imp_conv_execute_device(imp_plan* p, const float* d_x, int64_t B, int64_t chan_stride_in,
                                       int64_t elem_stride_in, float* d_y, int64_t chan_stride_out) {
  if (!p || !d_x || !d_y) return fail(IMP_ERR_INVALID, "imp_conv_execute_device: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (elem_stride_in < 1) return fail(IMP_ERR_INVALID, "elem_stride_in must be >= 1");
  int rc_span = check_input_span(p, elem_stride_in, sizeof(float));
  if (rc_span) return rc_span;
  if (chan_stride_out < p->out_len) return fail(IMP_ERR_INVALID, "chan_stride_out %lld < out_len %lld",
                                                (long long)chan_stride_out, (long long)p->out_len);
  if (p->n_filters > 1 && B > p->n_filters)
    return fail(IMP_ERR_INVALID, "B = %lld exceeds the plan's %lld per-channel filters", (long long)B,
                (long long)p->n_filters);
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  const int64_t grp = p->ws_channels / p->lanes;
  for (int64_t c0 = 0; c0 < B; c0 += grp) {
    const int64_t n = std::min(grp, B - c0);
    rc = run_group(p, d_x + c0 * chan_stride_in, n, chan_stride_in, elem_stride_in,
                   d_y + c0 * chan_stride_out, chan_stride_out, c0, 2);
    if (rc) return rc;
  }
  return IMP_OK;
}

extern "C" int imp_conv_execute_device_pcm(imp_plan* p, const void* d_pcm, int bits, int64_t B, int64_t chan_stride_in,
                                           int64_t elem_stride_in, float* d_y, int64_t chan_stride_out) {
  if (!p || !d_pcm || !d_y) return fail(IMP_ERR_INVALID, "imp_conv_execute_device_pcm: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (bits != 16 && bits != 32) return fail(IMP_ERR_INVALID, "PCM width must be 16 or 32 bits");
  if (B < 0 || elem_stride_in < 1) return fail(IMP_ERR_INVALID, "bad B or elem_stride_in");
  int rc_span = check_input_span(p, elem_stride_in, (size_t)(bits / 8));
  if (rc_span) return rc_span;
  if (chan_stride_out < p->out_len) return fail(IMP_ERR_INVALID, "chan_stride_out < out_len");
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  const int64_t grp = p->ws_channels / p->lanes;
  for (int64_t c0 = 0; c0 < B && p->paired; c0 += grp) {
    const int64_t n = std::min(grp, B - c0);
    if (bits == 32)
      rc = run_group_pair(p, imp::LoadPair<int>{(const int*)d_pcm + c0 * chan_stride_in, 2 * chan_stride_in, chan_stride_in,
                                                elem_stride_in, p->L, (int)n, 1.0f / 2147483648.0f},
                          n, d_y + c0 * chan_stride_out, chan_stride_out, 2);
    else
      rc = run_group_pair(p, imp::LoadPair<short>{(const short*)d_pcm + c0 * chan_stride_in, 2 * chan_stride_in, chan_stride_in,
                                                  elem_stride_in, p->L, (int)n, 1.0f / 32768.0f},
                          n, d_y + c0 * chan_stride_out, chan_stride_out, 2);
    if (rc) return rc;
  }
  if (p->paired) return IMP_OK;
  for (int64_t c0 = 0; c0 < B; c0 += grp) {
    const int64_t n = std::min(grp, B - c0);
    if (bits == 32) {
      imp::LoadPcmPacked<int32_t> ld{(const int32_t*)d_pcm + c0 * chan_stride_in, chan_stride_in, elem_stride_in, p->L,
                                     1.0f / 2147483648.0f};
      rc = run_group_with(p, ld, n, d_y + c0 * chan_stride_out, chan_stride_out, c0, 2);
    } else {
      imp::LoadPcmPacked<int16_t> ld{(const int16_t*)d_pcm + c0 * chan_stride_in, chan_stride_in, elem_stride_in, p->L,
                                     1.0f / 32768.0f};
      rc = run_group_with(p, ld, n, d_y + c0 * chan_stride_out, chan_stride_out, c0, 2);
    }
    if (rc) return rc;
  }
  return IMP_OK;
}

extern "C" int imp_conv_execute_device_pairs(imp_plan* p, const void* d_x, int bits, int64_t n_pairs, int64_t pair_stride,
                                             int64_t right_offset, int64_t elem_stride, float* d_y, int64_t chan_stride_out) {
  if (!p || !d_x || !d_y) return fail(IMP_ERR_INVALID, "imp_conv_execute_device_pairs: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (!p->paired) return fail(IMP_ERR_INVALID, "imp_conv_execute_device_pairs: not a pair-mode plan (imp_conv_plan_create_paired)");
  if (bits != 0 && bits != 16 && bits != 32) return fail(IMP_ERR_INVALID, "bits must be 0 (float32), 16 or 32 (PCM)");
  if (n_pairs < 0 || elem_stride < 1 || right_offset == 0 || (n_pairs > 1 && pair_stride == 0))
    return fail(IMP_ERR_INVALID, "imp_conv_execute_device_pairs: bad pair geometry");
  int rc = check_input_span(p, elem_stride, bits == 16 ? 2 : 4);
  if (rc) return rc;
  if (chan_stride_out < p->out_len) return fail(IMP_ERR_INVALID, "chan_stride_out < out_len");
  if ((rc = ctx_bind(p->ctx))) return rc;
  const int64_t grp = p->ws_channels / p->lanes / 2;                  // pairs per launch group
  for (int64_t q0 = 0; q0 < n_pairs; q0 += grp) {
    const int64_t nq = std::min(grp, n_pairs - q0);
    float* y = d_y + 2 * q0 * chan_stride_out;
    if (bits == 0)
      rc = run_group_pair(p, imp::LoadPair<float>{(const float*)d_x + q0 * pair_stride, pair_stride, right_offset, elem_stride,
                                                  p->L, (int)(2 * nq), 0.f}, 2 * nq, y, chan_stride_out, 2);
    else if (bits == 32)
      rc = run_group_pair(p, imp::LoadPair<int>{(const int*)d_x + q0 * pair_stride, pair_stride, right_offset, elem_stride, p->L,
                                                (int)(2 * nq), 1.0f / 2147483648.0f}, 2 * nq, y, chan_stride_out, 2);
    else
      rc = run_group_pair(p, imp::LoadPair<short>{(const short*)d_x + q0 * pair_stride, pair_stride, right_offset, elem_stride,
                                                  p->L, (int)(2 * nq), 1.0f / 32768.0f}, 2 * nq, y, chan_stride_out, 2);
    if (rc) return rc;
  }
  return IMP_OK;
}

extern "C" int imp_plan_set_overlap(imp_plan* p, int lanes) {
  if (!p) return fail(IMP_ERR_INVALID, "null plan");
  IMP_CTX_LOCK(p->ctx);
  if (lanes < 1 || lanes > 4) return fail(IMP_ERR_INVALID, "lanes must be in [1, 4]");
  if (p->ws_channels / lanes < 1 || (p->paired && ((p->ws_channels / lanes) & 1)))
    return fail(IMP_ERR_INVALID, "workspace of %lld channels cannot be split %d ways%s", (long long)p->ws_channels, lanes,
                p->paired ? " into whole pairs" : "");
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  if ((rc = plan_sync_lanes(p))) return rc;
  while ((int)p->ctx->side_streams.size() < lanes - 1) {
    hipStream_t st;
    if ((rc = ctx_new_stream(p->ctx, &st))) return rc;
    p->ctx->side_streams.push_back(st);
  }
  p->lanes = lanes;
  p->group_counter_lane = 0;
  return IMP_OK;
}

static int plan_staging(imp_plan* p, size_t in_bytes, size_t out_bytes) {
  if (p->d_in_bytes < in_bytes) {
    if (p->d_in) HIP_TRY(hipFree(p->d_in));
    p->d_in = nullptr;
    p->d_in_bytes = 0;
    hipError_t e = hipMalloc((void**)&p->d_in, in_bytes);
    if (e != hipSuccess) return fail(IMP_ERR_ALLOC, "hipMalloc(input staging %zu): %s", in_bytes, hipGetErrorString(e));
    p->d_in_bytes = in_bytes;
  }
  if (p->d_out_bytes < out_bytes) {
    if (p->d_out) HIP_TRY(hipFree(p->d_out));
    p->d_out = nullptr;
    p->d_out_bytes = 0;
    hipError_t e = hipMalloc((void**)&p->d_out, out_bytes);
    if (e != hipSuccess) return fail(IMP_ERR_ALLOC, "hipMalloc(output staging %zu): %s", out_bytes, hipGetErrorString(e));
    p->d_out_bytes = out_bytes;
  }
  return IMP_OK;
}

extern "C" int imp_conv_execute(imp_plan* p, const float* x, int64_t B, int64_t ld_in, float* y, int64_t ld_out) {
  if (!p || (!x && B) || (!y && B)) return fail(IMP_ERR_INVALID, "imp_conv_execute: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (ld_in < p->L) return fail(IMP_ERR_INVALID, "ld_in %lld < L %lld", (long long)ld_in, (long long)p->L);
  if (ld_out < p->out_len) return fail(IMP_ERR_INVALID, "ld_out %lld < out_len %lld", (long long)ld_out, (long long)p->out_len);
  if (B == 0) return IMP_OK;
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  // even row pitches keep the float2 fast paths aligned
  const int64_t pin = (p->L + 1) & ~(int64_t)1, pout = (p->out_len + 1) & ~(int64_t)1;
  if (p->lanes > 1) return fail(IMP_ERR_INVALID, "host-buffer execution needs imp_plan_set_overlap(plan, 1)");
  const int64_t grp = std::min(B, p->ws_channels);
  if ((rc = plan_staging(p, (size_t)(grp * pin) * sizeof(float), (size_t)(grp * pout) * sizeof(float)))) return rc;
  hipStream_t s = p->ctx->stream;
  for (int64_t c0 = 0; c0 < B; c0 += grp) {
    const int64_t n = std::min(grp, B - c0);
    HIP_TRY(hipMemcpy2DAsync(p->d_in, (size_t)pin * sizeof(float), x + c0 * ld_in, (size_t)ld_in * sizeof(float),
                             (size_t)p->L * sizeof(float), (size_t)n, hipMemcpyHostToDevice, s));
    if ((rc = run_group(p, p->d_in, n, pin, 1, p->d_out, pout, c0, 2))) return rc;
    HIP_TRY(hipMemcpy2DAsync(y + c0 * ld_out, (size_t)ld_out * sizeof(float), p->d_out, (size_t)pout * sizeof(float),
                             (size_t)p->out_len * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return IMP_OK;
}

extern "C" int imp_conv_execute_interleaved(imp_plan* p, const float* frames, int64_t C, float* y, int64_t ld_out) {
  if (!p || !frames || !y) return fail(IMP_ERR_INVALID, "imp_conv_execute_interleaved: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (C < 1) return fail(IMP_ERR_INVALID, "C < 1");
  if (ld_out < p->out_len) return fail(IMP_ERR_INVALID, "ld_out < out_len");
  int rc = check_input_span(p, C, sizeof(float));
  if (rc) return rc;
  rc = ctx_bind(p->ctx);
  if (rc) return rc;
  if (p->lanes > 1) return fail(IMP_ERR_INVALID, "host-buffer execution needs imp_plan_set_overlap(plan, 1)");
  const int64_t pout = (p->out_len + 1) & ~(int64_t)1;
  const int64_t grp = std::min(C, p->ws_channels);
  if ((rc = plan_staging(p, (size_t)(p->L * C) * sizeof(float), (size_t)(grp * pout) * sizeof(float)))) return rc;
  hipStream_t s = p->ctx->stream;
  // the whole frame block goes up once, in wire order; channels are picked apart by the loader
  HIP_TRY(hipMemcpyAsync(p->d_in, frames, (size_t)(p->L * C) * sizeof(float), hipMemcpyHostToDevice, s));
  for (int64_t c0 = 0; c0 < C; c0 += grp) {
    const int64_t n = std::min(grp, C - c0);
    if ((rc = run_group(p, p->d_in + c0, n, 1, C, p->d_out, pout, c0, 2))) return rc;
    HIP_TRY(hipMemcpy2DAsync(y + c0 * ld_out, (size_t)ld_out * sizeof(float), p->d_out, (size_t)pout * sizeof(float),
                             (size_t)p->out_len * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return IMP_OK;
}

extern "C" int imp_plan_debug_run_stage(imp_plan* p, const float* x, int64_t B, int64_t ld_in, int stage,
                                        float* ws_out_host) {
  if (!p || !x || !ws_out_host) return fail(IMP_ERR_INVALID, "imp_plan_debug_run_stage: null argument");
  IMP_CTX_LOCK(p->ctx);
  if (B < 1 || B > p->ws_channels) return fail(IMP_ERR_INVALID, "debug stage: B must be in [1, ws_channels]");
  if (stage < 0 || stage > 1) return fail(IMP_ERR_INVALID, "stage must be 0 (after pass A) or 1 (after pass B)");
  int rc = ctx_bind(p->ctx);
  if (rc) return rc;
  const int64_t pin = (p->L + 1) & ~(int64_t)1, pout = (p->out_len + 1) & ~(int64_t)1;
  if ((rc = plan_staging(p, (size_t)(B * pin) * sizeof(float), (size_t)(B * pout) * sizeof(float)))) return rc;
  hipStream_t s = p->ctx->stream;
  HIP_TRY(hipMemcpy2DAsync(p->d_in, (size_t)pin * sizeof(float), x, (size_t)ld_in * sizeof(float),
                           (size_t)p->L * sizeof(float), (size_t)B, hipMemcpyHostToDevice, s));
  if ((rc = run_group(p, p->d_in, B, pin, 1, p->d_out, pout, 0, stage))) return rc;
  HIP_TRY(hipMemcpyAsync(ws_out_host, p->ws, (size_t)(p->paired ? (B + 1) / 2 : B) * p->N1 * imp::kN2 * sizeof(cf),
                         hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

// ------------------------------------------------------------------------------------------------
// K3 peak index, K4/K8 windows (ragged batches)
// ------------------------------------------------------------------------------------------------
static int peak_index_impl(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len, int64_t B,
                           double peak_height, int64_t* idx_out, float* maxabs_out) {
  // the search works on |x| against one positive threshold (the reference's callers pass 0.12589 = -18 dB)
  if (!(peak_height > 0.0)) return fail(IMP_ERR_INVALID, "peak_height must be positive (got %g)", peak_height);
  if (B == 0) return IMP_OK;
  int64_t maxlen = 0;
  for (int64_t b = 0; b < B; ++b) maxlen = std::max(maxlen, len[b]);
  const int64_t chunks = std::max<int64_t>(1, (maxlen + imp::kPeakChunk - 1) / imp::kPeakChunk);
  // scratch: off[B], len[B], res[B] (RowPeak), chunk maxima [B][chunks]
  const size_t meta = (size_t)B * sizeof(int64_t);
  const size_t res_bytes = (size_t)B * sizeof(imp::RowPeak);
  void* scr = nullptr;
  int rc = ctx_scratch(ctx, res_bytes + (size_t)(B * chunks) * sizeof(unsigned), &scr);
  if (rc) return rc;
  imp::RowPeak* d_res = (imp::RowPeak*)scr;
  unsigned* d_chunk = (unsigned*)(d_res + B);
  hipStream_t s = ctx->stream;
  int64_t *h_tab = nullptr, *d_off = nullptr;
  if ((rc = ctx_stage(ctx, 2 * meta, (void**)&h_tab, (void**)&d_off))) return rc;
  int64_t* d_len = d_off + B;
  std::memcpy(h_tab, off, meta);
  std::memcpy(h_tab + B, len, meta);
  if ((rc = ctx_stage_push(ctx, h_tab, d_off, 2 * meta))) return rc;
  hipLaunchKernelGGL(imp::row_chunk_max_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(256), 0, s, d_x, d_off, d_len,
                     (int64_t)0, d_chunk, chunks);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(imp::row_first_peak_chunked_kernel, dim3((unsigned)B), dim3(imp::kPeakThreads), 0, s, d_x, d_off, d_len,
                     (int64_t)0, (const unsigned*)nullptr, 0, (const unsigned*)d_chunk, chunks, d_res, peak_height, (long long*)nullptr);
  HIP_TRY(hipGetLastError());
  std::vector<imp::RowPeak> h((size_t)B);
  HIP_TRY(hipMemcpyAsync(h.data(), d_res, res_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int64_t b = 0; b < B; ++b) {
    float m;
    std::memcpy(&m, &h[(size_t)b].maxabs_bits, sizeof(float));
    if (maxabs_out) maxabs_out[b] = m;
    int64_t idx;
    if (len[b] == 0 || !(m >= 1e-20f)) idx = 0;                         // EPSILON rule, impulse_response.py:56-58
    else if (h[(size_t)b].first_peak != ~0ull) idx = (int64_t)h[(size_t)b].first_peak;
    else idx = (int64_t)h[(size_t)b].first_max;                          // argmax fallback, :66-67
    idx_out[b] = idx;
  }
  return IMP_OK;
}

extern "C" int imp_peak_index_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len,
                                     int64_t B, double peak_height, int64_t* idx_out, float* maxabs_out) {
  if (!ctx || (B && (!d_x || !off || !len || !idx_out))) return fail(IMP_ERR_INVALID, "imp_peak_index_device: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  for (int64_t b = 0; b < B; ++b)
    if (len[b] < 0 || off[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)b);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  return peak_index_impl(ctx, d_x, off, len, B, peak_height, idx_out, maxabs_out);
}

extern "C" int imp_peak_index(imp_ctx* ctx, const float* x, const int64_t* off, const int64_t* len, int64_t B,
                              double peak_height, int64_t* idx_out, float* maxabs_out) {
  if (!ctx || (B && (!x || !off || !len || !idx_out))) return fail(IMP_ERR_INVALID, "imp_peak_index: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  int64_t total = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (len[b] < 0 || off[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)b);
    total = std::max(total, off[b] + len[b]);
  }
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  float* d_x = nullptr;
  if (total > 0) {
    if ((rc = ctx_block_get(ctx, (size_t)total * sizeof(float), (void**)&d_x))) return rc;
    hipError_t e = hipMemcpyAsync(d_x, x, (size_t)total * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(ctx->stream);
      (void)ctx_block_put(ctx, d_x);
      return fail(IMP_ERR_HIP, "h2d: %s", hipGetErrorString(e));
    }
  }
  rc = peak_index_impl(ctx, d_x, off, len, B, peak_height, idx_out, maxabs_out);
  (void)hipStreamSynchronize(ctx->stream);
  (void)ctx_block_put(ctx, d_x);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Chain K1 -> K3 -> K4 -> K5 on one stream without a host round trip ("deconv + FIR" of the metric)
// ------------------------------------------------------------------------------------------------
struct imp_chain {
  imp_ctx* ctx = nullptr;           // the deconvolution plan's context
  imp_ctx* tail_ctx = nullptr;      // the FIR plan's context: its stream carries the peak search and K5 of every call
  imp_plan* deconv = nullptr;       // K1: 'same' plan of the recording length, one or more lanes
  imp_plan* fir = nullptr;          // K5: 'full' plan of length n with per-channel (or one shared) filters
  int64_t B = 0, n = 0, head = 0, fade_in = 0, fade_out = 0, pitch_ir = 0;
  double peak_height = 0.12589;
  int lanes = 1;
  int sets = 1;                     // buffer sets in rotation: the lanes, and one more when the tail has a stream of its own
  int64_t calls = 0;
  int tiles = 0;
  int64_t chunks = 0;               // chunks per channel: N1 of the deconvolution (pair mode: N1 / 2)
  // buffer sets, used round robin: call i may deconvolve into one set while the tail of call i - 1 still reads another
  std::vector<float*> d_ir;         // [B][pitch_ir]: the deconvolved columns
  std::vector<unsigned*> d_tile;    // [B][column tiles][chunks]: max|y| per tile and 8 192-sample chunk, left by pass C
  std::vector<imp::RowPeak*> d_res;
  std::vector<hipEvent_t> k1_done;  // recorded on the lane's stream after pass C
  std::vector<hipEvent_t> ir_free;  // recorded on the tail stream after K5 has read the lane's buffers
  std::vector<char> ir_busy;
  int64_t* d_meta = nullptr;        // off[B], len[B]
  double* d_win = nullptr;          // the two Hann fades as tables (LoadCropAtPeak)
};

// both contexts, in address order
struct ChainLock {
  std::unique_lock<std::recursive_mutex> a, b;
  ChainLock(imp_ctx* x, imp_ctx* y) {
    if (x == y) {
      a = std::unique_lock<std::recursive_mutex>(x->mu);
    } else {
      imp_ctx* lo = x < y ? x : y;
      imp_ctx* hi = x < y ? y : x;
      a = std::unique_lock<std::recursive_mutex>(lo->mu);
      b = std::unique_lock<std::recursive_mutex>(hi->mu);
    }
  }
};

extern "C" void imp_chain_destroy(imp_chain* c) {
  if (!c) return;
  ChainLock lk(c->ctx, c->tail_ctx);
  (void)hipSetDevice(c->ctx->device);
  (void)plan_sync_lanes(c->deconv);
  (void)hipStreamSynchronize(c->tail_ctx->stream);
  for (auto p : c->d_ir) (void)hipFree(p);
  for (auto p : c->d_tile) (void)hipFree(p);
  for (auto p : c->d_res) (void)hipFree(p);
  for (auto e : c->k1_done) (void)hipEventDestroy(e);
  for (auto e : c->ir_free) (void)hipEventDestroy(e);
  (void)hipFree(c->d_meta);
  (void)hipFree(c->d_win);
  delete c;
}

extern "C" int imp_chain_create(imp_plan* deconv, imp_plan* fir, int64_t B, int64_t head, int64_t fade_in, int64_t fade_out,
                                double peak_height, imp_chain** out) {
  if (!deconv || !fir || !out) return fail(IMP_ERR_INVALID, "imp_chain_create: null argument");
  *out = nullptr;
  if (deconv->ctx->device != fir->ctx->device) return fail(IMP_ERR_INVALID, "imp_chain_create: the two plans live on different devices");
  ChainLock lk(deconv->ctx, fir->ctx);
  if (deconv->mode != IMP_MODE_SAME || fir->mode != IMP_MODE_FULL)
    return fail(IMP_ERR_INVALID, "imp_chain_create: needs a 'same' deconvolution plan and a 'full' FIR plan");
  if (!(peak_height > 0.0)) return fail(IMP_ERR_INVALID, "imp_chain_create: peak_height must be positive (got %g)", peak_height);
  if (fir->lanes != 1) return fail(IMP_ERR_INVALID, "imp_chain_create: the FIR plan must run in stream order (lanes = 1)");
  if (deconv->lanes > 1 && deconv->ctx == fir->ctx)
    return fail(IMP_ERR_INVALID, "imp_chain_create: a deconvolution plan with several lanes needs the FIR plan on a context of its own "
                                 "(its stream carries the peak search and K5 beside the lanes)");
  if (deconv->ola || fir->ola) return fail(IMP_ERR_UNSUPPORTED, "imp_chain_create: overlap-add plans cannot be chained");
  if (fir->paired) return fail(IMP_ERR_UNSUPPORTED, "imp_chain_create: the FIR stage reads one response per transform (mono plan)");
  if (deconv->fused) return fail(IMP_ERR_UNSUPPORTED, "imp_chain_create: a fused FIR plan cannot be the deconvolution stage");
  const int64_t n = fir->L;
  if (B < 1 || B > deconv->ws_channels / deconv->lanes || B > fir->ws_channels)
    return fail(IMP_ERR_INVALID, "imp_chain_create: B exceeds a plan's workspace (per lane)");
  if (fir->n_filters > 1 && B > fir->n_filters) return fail(IMP_ERR_INVALID, "imp_chain_create: fewer FIRs than channels");
  if (n > deconv->out_len || head < 0 || fade_in < 0 || fade_out < 0 || fade_in > n || fade_out > n)
    return fail(IMP_ERR_INVALID, "imp_chain_create: crop of %lld samples with fades %lld / %lld does not fit", (long long)n,
                (long long)fade_in, (long long)fade_out);
  if ((deconv->paired ? deconv->N1 / 2 : deconv->N1) > imp::kMaxPlanRows)
    return fail(IMP_ERR_UNSUPPORTED, "imp_chain_create: deconvolution plan of %d rows", deconv->N1);
  int rc = ctx_bind(deconv->ctx);
  if (rc) return rc;
  imp_chain* c = new (std::nothrow) imp_chain();
  if (!c) return fail(IMP_ERR_ALLOC, "out of host memory");
  c->ctx = deconv->ctx;
  c->tail_ctx = fir->ctx;
  c->deconv = deconv;
  c->fir = fir;
  c->B = B;
  c->n = n;
  c->head = head;
  c->fade_in = fade_in;
  c->fade_out = fade_out;
  c->peak_height = peak_height;
  c->lanes = deconv->lanes;
  c->sets = deconv->ctx == fir->ctx ? 1 : c->lanes + 1;
  c->pitch_ir = (deconv->out_len + 63) / 64 * 64;
  std::vector<int64_t> meta((size_t)(2 * B));
  for (int64_t b = 0; b < B; ++b) {
    meta[(size_t)b] = b * c->pitch_ir;
    meta[(size_t)(B + b)] = deconv->out_len;
  }
  c->tiles = plan_col_tiles(deconv);
  c->chunks = deconv->paired ? deconv->N1 / 2 : deconv->N1;
  const size_t chunk_bytes = (size_t)(B * c->chunks) * sizeof(unsigned);
  bool ok = hipMalloc((void**)&c->d_meta, (size_t)(2 * B) * sizeof(int64_t)) == hipSuccess &&
            hipMemcpy(c->d_meta, meta.data(), meta.size() * sizeof(int64_t), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc((void**)&c->d_win, (size_t)std::max<int64_t>(fade_in + fade_out, 1) * sizeof(double)) == hipSuccess;
  if (ok && fade_in + fade_out > 0) {
    hipLaunchKernelGGL(imp::fade_table_kernel, dim3((unsigned)((fade_in + fade_out + 255) / 256)), dim3(256), 0, deconv->ctx->stream,
                       c->d_win, (long long)fade_in, (long long)fade_out);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(deconv->ctx->stream) == hipSuccess;
  }
  for (int l = 0; l < c->sets && ok; ++l) {
    float* ir = nullptr;
    unsigned* tile = nullptr;
    imp::RowPeak* res = nullptr;
    hipEvent_t e1 = nullptr, e2 = nullptr;
    ok = hipMalloc((void**)&ir, (size_t)(B * c->pitch_ir) * sizeof(float)) == hipSuccess;
    if (ir) c->d_ir.push_back(ir);
    ok = ok && hipMalloc((void**)&tile, chunk_bytes * (size_t)c->tiles) == hipSuccess;
    if (tile) c->d_tile.push_back(tile);
    ok = ok && hipMalloc((void**)&res, (size_t)B * sizeof(imp::RowPeak)) == hipSuccess;
    if (res) c->d_res.push_back(res);
    ok = ok && hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess;
    if (e1) c->k1_done.push_back(e1);
    ok = ok && hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
    if (e2) c->ir_free.push_back(e2);
  }
  c->ir_busy.assign((size_t)c->sets, 0);
  if (!ok) {
    (void)hipGetLastError();
    imp_chain_destroy(c);
    return fail(IMP_ERR_ALLOC, "imp_chain_create: device allocation failed");
  }
  *out = c;
  return IMP_OK;
}

// Seven launches per call: pass A, pass B, pass C (+ chunk maxima) on the deconvolution plan's next lane, then - on the FIR
// plan's stream, ordered behind that lane by an event - the peak search and K5, whose pass A reads the cropped, faded
// responses straight out of the deconvolved columns (LoadCropAtPeak), pass B, pass C.  The lane's buffers are handed back
// by a second event, so a later call on the same lane waits for this call's K5 and nothing else does.
extern "C" int imp_chain_execute_device(imp_chain* c, const float* d_x, int64_t chan_stride_in, int64_t elem_stride_in,
                                        float* d_out, int64_t chan_stride_out, long long* d_peaks_out) {
  if (!c || !d_x || !d_out) return fail(IMP_ERR_INVALID, "imp_chain_execute_device: null argument");
  ChainLock lk(c->ctx, c->tail_ctx);
  if (chan_stride_out < c->fir->out_len)
    return fail(IMP_ERR_INVALID, "chan_stride_out %lld < out_len %lld", (long long)chan_stride_out, (long long)c->fir->out_len);
  if (c->deconv->lanes != c->lanes || c->fir->lanes != 1)
    return fail(IMP_ERR_INVALID, "imp_chain_execute_device: a plan's overlap setting changed since the chain was made");
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  const int lane = c->lanes > 1 ? (int)(c->deconv->group_counter_lane % c->lanes) : 0;   // the lane run_group will pick
  const int l = (int)(c->calls++ % c->sets);                                              // this call's buffer set
  hipStream_t lane_stream = lane ? c->ctx->side_streams[(size_t)(lane - 1)] : c->ctx->stream;
  hipStream_t tail = c->tail_ctx->stream;
  if (c->ir_busy[(size_t)l] && tail != lane_stream) HIP_TRY(hipStreamWaitEvent(lane_stream, c->ir_free[(size_t)l], 0));
  c->deconv->tile_max = c->d_tile[(size_t)l];
  rc = imp_conv_execute_device(c->deconv, d_x, c->B, chan_stride_in, elem_stride_in, c->d_ir[(size_t)l], c->pitch_ir);
  c->deconv->tile_max = nullptr;
  if (rc) return rc;
  if (c->deconv->cur_stream != lane_stream) return fail(IMP_ERR_HIP, "imp_chain_execute_device: lane bookkeeping out of step");
  if (tail != lane_stream) {
    HIP_TRY(hipEventRecord(c->k1_done[(size_t)l], lane_stream));
    HIP_TRY(hipStreamWaitEvent(tail, c->k1_done[(size_t)l], 0));
  }
  hipLaunchKernelGGL(imp::row_first_peak_chunked_kernel, dim3((unsigned)c->B), dim3(imp::kPeakThreads), 0, tail, c->d_ir[(size_t)l],
                     c->d_meta, c->d_meta + c->B, c->deconv->out_start, (const unsigned*)c->d_tile[(size_t)l], c->tiles,
                     (const unsigned*)nullptr, c->chunks, c->d_res[(size_t)l], c->peak_height, d_peaks_out);
  HIP_TRY(hipGetLastError());
  imp::LoadCropAtPeak ld{c->d_ir[(size_t)l], c->pitch_ir, c->deconv->out_len, c->d_res[(size_t)l], c->n, c->head, c->fade_in, c->fade_out,
                         c->d_win};
  if ((rc = run_group_with(c->fir, ld, c->B, d_out, chan_stride_out, 0, 2))) return rc;
  if (tail != lane_stream) {
    HIP_TRY(hipEventRecord(c->ir_free[(size_t)l], tail));
    c->ir_busy[(size_t)l] = 1;
  }
  return IMP_OK;
}

// ------------------------------------------------------------------------------------------------
// K7 segment sets
// ------------------------------------------------------------------------------------------------
struct imp_segset {
  imp_ctx* ctx = nullptr;
  int64_t B = 0;
  double* e = nullptr;               // squared, normalised segments, concatenated
  int64_t* off = nullptr;            // device copies of the row table
  int64_t* len = nullptr;
  std::vector<int64_t> h_len;
  void* qbuf = nullptr;              // query staging: 3 x int64 + 1 x double per query
  size_t qcap = 0;
  // imp_segset_create_device without maxabs_out does not drain the stream: its staging lives as long as the set
  void* keep_a = nullptr;
  void* keep_b = nullptr;
  std::vector<int64_t> keep_host;
};

extern "C" void imp_segset_destroy(imp_segset* s) {
  if (!s) return;
  IMP_CTX_LOCK(s->ctx);
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  (void)ctx_block_put(s->ctx, s->e);
  (void)ctx_block_put(s->ctx, s->off);
  (void)ctx_block_put(s->ctx, s->qbuf);
  (void)ctx_block_put(s->ctx, s->keep_a);
  (void)ctx_block_put(s->ctx, s->keep_b);
  delete s;
}

extern "C" int imp_segset_create(imp_ctx* ctx, const double* x, const int64_t* off, const int64_t* len, int64_t B,
                                 imp_segset** out, double* maxabs_out) {
  if (!ctx || !out || (B && (!x || !off || !len))) return fail(IMP_ERR_INVALID, "imp_segset_create: null argument");
  IMP_CTX_LOCK(ctx);
  *out = nullptr;
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  int64_t total = 0, maxlen = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (off[b] < 0 || len[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in segment %lld", (long long)b);
    total = std::max(total, off[b] + len[b]);
    maxlen = std::max(maxlen, len[b]);
  }
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  imp_segset* s = new (std::nothrow) imp_segset();
  if (!s) return fail(IMP_ERR_ALLOC, "out of host memory");
  s->ctx = ctx;
  s->B = B;
  s->h_len.assign(len, len + B);
  hipStream_t st = ctx->stream;
  unsigned long long* d_max = nullptr;
  auto bail = [&](int code) {
    (void)hipStreamSynchronize(st);
    (void)ctx_block_put(ctx, d_max);
    imp_segset_destroy(s);
    return code;
  };
  if (ctx_block_get(ctx, (size_t)std::max<int64_t>(total, 1) * sizeof(double), (void**)&s->e) ||
      ctx_block_get(ctx, (size_t)std::max<int64_t>(2 * B, 1) * sizeof(int64_t), (void**)&s->off) ||
      ctx_block_get(ctx, (size_t)std::max<int64_t>(B, 1) * sizeof(unsigned long long), (void**)&d_max))
    return bail(fail(IMP_ERR_ALLOC, "imp_segset_create: device allocation of %lld samples failed", (long long)total));
  s->len = s->off + B;
  if (B == 0 || total == 0) {
    (void)ctx_block_put(ctx, d_max);
    *out = s;
    if (maxabs_out)
      for (int64_t b = 0; b < B; ++b) maxabs_out[b] = 0.0;
    return IMP_OK;
  }
  if (hipMemcpyAsync(s->e, x, (size_t)total * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->off, off, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->len, len, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemsetAsync(d_max, 0, (size_t)B * sizeof(unsigned long long), st) != hipSuccess)
    return bail(fail(IMP_ERR_HIP, "imp_segset_create: upload failed"));
  const int bpr = (int)std::max<int64_t>(1, std::min<int64_t>(256, (maxlen + 4095) / 4096));
  dim3 grid((unsigned)bpr, (unsigned)B), block(256);
  hipLaunchKernelGGL(imp::seg_maxabs_kernel, grid, block, 0, st, s->e, s->off, s->len, d_max);
  hipLaunchKernelGGL(imp::seg_square_kernel, grid, block, 0, st, s->e, s->off, s->len, d_max);
  if (hipGetLastError() != hipSuccess) return bail(fail(IMP_ERR_HIP, "imp_segset_create: launch failed"));
  std::vector<unsigned long long> h((size_t)B);
  if (hipMemcpyAsync(h.data(), d_max, (size_t)B * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return bail(fail(IMP_ERR_HIP, "imp_segset_create: readback failed"));
  (void)ctx_block_put(ctx, d_max);
  d_max = nullptr;
  if (maxabs_out)
    for (int64_t b = 0; b < B; ++b) std::memcpy(&maxabs_out[b], &h[(size_t)b], sizeof(double));
  *out = s;
  return IMP_OK;
}

extern "C" int imp_segset_create_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len, int64_t B,
                                        imp_segset** out, double* maxabs_out) {
  if (!ctx || !out || (B && (!d_x || !off || !len))) return fail(IMP_ERR_INVALID, "imp_segset_create_device: null argument");
  IMP_CTX_LOCK(ctx);
  *out = nullptr;
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  int64_t total = 0, maxlen = 0;
  std::vector<int64_t> packed((size_t)std::max<int64_t>(B, 1));
  for (int64_t b = 0; b < B; ++b) {
    if (off[b] < 0 || len[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in segment %lld", (long long)b);
    packed[(size_t)b] = total;
    total += len[b];
    maxlen = std::max(maxlen, len[b]);
  }
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  imp_segset* s = new (std::nothrow) imp_segset();
  if (!s) return fail(IMP_ERR_ALLOC, "out of host memory");
  s->ctx = ctx;
  s->B = B;
  s->h_len.assign(len, len + B);
  hipStream_t st = ctx->stream;
  unsigned long long* d_max = nullptr;
  int64_t* d_src_off = nullptr;
  auto bail = [&](int code) {
    (void)hipStreamSynchronize(st);
    (void)ctx_block_put(ctx, d_max);
    (void)ctx_block_put(ctx, d_src_off);
    imp_segset_destroy(s);
    return code;
  };
  if (ctx_block_get(ctx, (size_t)std::max<int64_t>(total, 1) * sizeof(double), (void**)&s->e) ||
      ctx_block_get(ctx, (size_t)std::max<int64_t>(2 * B, 1) * sizeof(int64_t), (void**)&s->off) ||
      ctx_block_get(ctx, (size_t)std::max<int64_t>(B, 1) * sizeof(int64_t), (void**)&d_src_off) ||
      ctx_block_get(ctx, (size_t)std::max<int64_t>(B, 1) * sizeof(unsigned long long), (void**)&d_max))
    return bail(fail(IMP_ERR_ALLOC, "imp_segset_create_device: device allocation of %lld samples failed", (long long)total));
  s->len = s->off + B;
  if (B == 0 || total == 0) {
    (void)ctx_block_put(ctx, d_max);
    (void)ctx_block_put(ctx, d_src_off);
    *out = s;
    if (maxabs_out)
      for (int64_t b = 0; b < B; ++b) maxabs_out[b] = 0.0;
    return IMP_OK;
  }
  // the three small tables travel as ONE block (host copy kept by the set: without maxabs_out nothing below waits)
  s->keep_host.resize((size_t)(3 * B));
  std::copy(packed.begin(), packed.begin() + B, s->keep_host.begin());
  std::copy(len, len + B, s->keep_host.begin() + B);
  std::copy(off, off + B, s->keep_host.begin() + 2 * B);
  if (hipMemcpyAsync(s->off, s->keep_host.data(), (size_t)(2 * B) * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_src_off, s->keep_host.data() + 2 * B, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemsetAsync(d_max, 0, (size_t)B * sizeof(unsigned long long), st) != hipSuccess)
    return bail(fail(IMP_ERR_HIP, "imp_segset_create_device: upload failed"));
  const int bpr = (int)std::max<int64_t>(1, std::min<int64_t>(256, (maxlen + 4095) / 4096));
  dim3 grid((unsigned)bpr, (unsigned)B), block(256);
  hipLaunchKernelGGL(imp::seg_from_float_kernel, grid, block, 0, st, d_x, d_src_off, s->e, s->off, s->len);
  hipLaunchKernelGGL(imp::seg_maxabs_kernel, grid, block, 0, st, s->e, s->off, s->len, d_max);
  hipLaunchKernelGGL(imp::seg_square_kernel, grid, block, 0, st, s->e, s->off, s->len, d_max);
  if (hipGetLastError() != hipSuccess) return bail(fail(IMP_ERR_HIP, "imp_segset_create_device: launch failed"));
  if (!maxabs_out) {
    // stream-ordered: the first imp_segset_range_means call waits for all of it; the staging goes with the set
    s->keep_a = d_max;
    s->keep_b = d_src_off;
    *out = s;
    return IMP_OK;
  }
  std::vector<unsigned long long> h((size_t)B);
  if (hipMemcpyAsync(h.data(), d_max, (size_t)B * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return bail(fail(IMP_ERR_HIP, "imp_segset_create_device: readback failed"));
  (void)ctx_block_put(ctx, d_max);
  (void)ctx_block_put(ctx, d_src_off);
  for (int64_t b = 0; b < B; ++b) std::memcpy(&maxabs_out[b], &h[(size_t)b], sizeof(double));
  *out = s;
  return IMP_OK;
}

extern "C" int imp_segset_range_means(imp_segset* s, const int64_t* q_seg, const int64_t* q_a, const int64_t* q_b,
                                      int64_t Q, double* mean_out) {
  if (!s || (Q && (!q_seg || !q_a || !q_b || !mean_out))) return fail(IMP_ERR_INVALID, "imp_segset_range_means: null argument");
  IMP_CTX_LOCK(s->ctx);
  if (Q < 0) return fail(IMP_ERR_INVALID, "Q < 0");
  if (Q == 0) return IMP_OK;
  for (int64_t q = 0; q < Q; ++q) {
    if (q_seg[q] < 0 || q_seg[q] >= s->B) return fail(IMP_ERR_INVALID, "query %lld: segment %lld out of range", (long long)q, (long long)q_seg[q]);
    // NumPy slicing clips; the caller passes clipped bounds, anything else is a bug on the host side
    if (q_a[q] < 0 || q_b[q] > s->h_len[(size_t)q_seg[q]] || q_a[q] > q_b[q])
      return fail(IMP_ERR_INVALID, "query %lld: range [%lld, %lld) outside segment of %lld samples", (long long)q,
                  (long long)q_a[q], (long long)q_b[q], (long long)s->h_len[(size_t)q_seg[q]]);
  }
  int rc = ctx_bind(s->ctx);
  if (rc) return rc;
  hipStream_t st = s->ctx->stream;
  const size_t need = (size_t)Q * (3 * sizeof(int64_t) + sizeof(double));
  if (s->qcap < need) {
    (void)hipStreamSynchronize(st);
    (void)ctx_block_put(s->ctx, s->qbuf);
    s->qbuf = nullptr;
    s->qcap = 0;
    const size_t want = std::max(need, (size_t)1 << 16);
    if (ctx_block_get(s->ctx, want, &s->qbuf)) return IMP_ERR_ALLOC;
    s->qcap = want;
  }
  int64_t* d_seg = (int64_t*)s->qbuf;
  int64_t* d_a = d_seg + Q;
  int64_t* d_b = d_a + Q;
  double* d_m = (double*)(d_b + Q);
  HIP_TRY(hipMemcpyAsync(d_seg, q_seg, (size_t)Q * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_a, q_a, (size_t)Q * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_b, q_b, (size_t)Q * sizeof(int64_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(imp::seg_range_mean_kernel, dim3((unsigned)Q), dim3(256), 0, st, s->e, s->off, d_seg, d_a,
                     d_b, (long long)Q, d_m);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(mean_out, d_m, (size_t)Q * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return IMP_OK;
}

// K7c: peak search + Lundeby knee search of B device-resident responses without a host round trip in between
// (decay_kernels.hip.h).  flags_out[b] != 0: the device search left row b to the host search (a decision inside its
// guard band, or a shape outside the device path's limits); peak_out[b] is valid either way.
extern "C" int imp_decay_knees_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len, int64_t B,
                                      double fs, double peak_height, int64_t* peak_out, int64_t* knee_out,
                                      double* floor_out, int64_t* window_out, int32_t* flags_out) {
  if (!ctx || (B && (!d_x || !off || !len || !peak_out || !knee_out || !floor_out || !window_out || !flags_out)))
    return fail(IMP_ERR_INVALID, "imp_decay_knees_device: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (!(fs > 0.0) || !(fs < 1e9)) return fail(IMP_ERR_INVALID, "imp_decay_knees_device: fs must be positive (got %g)", fs);
  if (!(peak_height > 0.0)) return fail(IMP_ERR_INVALID, "peak_height must be positive (got %g)", peak_height);
  if (B == 0) return IMP_OK;
  int64_t maxlen = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (len[b] < 0 || off[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)b);
    if (len[b] >= ((int64_t)1 << 31)) return fail(IMP_ERR_INVALID, "row %lld is too long (%lld samples)", (long long)b, (long long)len[b]);
    maxlen = std::max(maxlen, len[b]);
  }
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  const int64_t two_fs = (int64_t)(2 * fs);               // int(2 * fs), core/decay.py:84
  const int64_t span_max = std::max<int64_t>(1, std::min(two_fs, maxlen));
  const int64_t chunks = std::max<int64_t>(1, (maxlen + imp::kPeakChunk - 1) / imp::kPeakChunk);
  const int mean_pitch = imp::kKneeMaxWindows + 1;
  // scratch: span maxima bits [B] | peak results [B] | search state [B] | window means [B][mean_pitch] | chunk maxima
  // [B][chunks].  The spans are read where they are (fp32 rows): no fp64 copy of them is made.
  const size_t meta = (size_t)B * sizeof(int64_t);
  size_t bytes = (size_t)B * sizeof(unsigned long long) + (size_t)B * sizeof(imp::RowPeak) + (size_t)B * sizeof(imp::KneeRow) +
                 (size_t)B * mean_pitch * sizeof(double) + (size_t)(B * chunks) * sizeof(unsigned);
  void* scr = nullptr;
  if ((rc = ctx_scratch(ctx, bytes, &scr))) return rc;
  int64_t *h_tab = nullptr, *d_off = nullptr;
  if ((rc = ctx_stage(ctx, 2 * meta, (void**)&h_tab, (void**)&d_off))) return rc;
  int64_t* d_len = d_off + B;
  std::memcpy(h_tab, off, meta);
  std::memcpy(h_tab + B, len, meta);
  unsigned long long* d_max = (unsigned long long*)scr;
  imp::RowPeak* d_res = (imp::RowPeak*)(d_max + B);
  imp::KneeRow* d_rows = (imp::KneeRow*)(d_res + B);
  double* d_means = (double*)(d_rows + B);
  unsigned* d_chunk = (unsigned*)(d_means + (size_t)B * mean_pitch);
  hipStream_t s = ctx->stream;
  std::vector<imp::KneeRow> h((size_t)B);
  if ((rc = ctx_stage_push(ctx, h_tab, d_off, 2 * meta))) return rc;
  hipLaunchKernelGGL(imp::row_chunk_max_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(256), 0, s, d_x, d_off, d_len,
                     (int64_t)0, d_chunk, chunks);
  hipLaunchKernelGGL(imp::row_first_peak_chunked_kernel, dim3((unsigned)B), dim3(imp::kPeakThreads), 0, s, d_x, d_off, d_len,
                     (int64_t)0, (const unsigned*)nullptr, 0, (const unsigned*)d_chunk, chunks, d_res, peak_height,
                     (long long*)nullptr);
  hipLaunchKernelGGL(imp::knee_span_kernel, dim3((unsigned)B), dim3(64), 0, s, (const imp::RowPeak*)d_res, d_off, d_len,
                     (long long)two_fs, fs, d_rows, d_max);
  const int bpr = (int)std::max<int64_t>(1, std::min<int64_t>(64, (span_max + 8191) / 8192));
  dim3 block(256);
  hipLaunchKernelGGL(imp::knee_maxabs_kernel, dim3((unsigned)bpr, (unsigned)B), block, 0, s, d_x, (const imp::KneeRow*)d_rows, d_max);
  hipLaunchKernelGGL(imp::knee_windows_kernel, dim3(imp::kKneeRound1, (unsigned)B), block, 0, s, (const imp::KneeRow*)d_rows,
                     d_x, (const unsigned long long*)d_max, d_means, mean_pitch, 1);
  hipLaunchKernelGGL(imp::knee_stage1_kernel, dim3((unsigned)B), dim3(64), 0, s, d_rows, (const double*)d_means, mean_pitch, fs);
  hipLaunchKernelGGL(imp::knee_windows_kernel, dim3(64, (unsigned)B), block, 0, s, (const imp::KneeRow*)d_rows, d_x,
                     (const unsigned long long*)d_max, d_means, mean_pitch, 0);
  hipLaunchKernelGGL(imp::knee_stage2_kernel, dim3((unsigned)B), block, 0, s, d_rows, (const double*)d_means, mean_pitch, d_x,
                     (const unsigned long long*)d_max, fs);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h.data(), d_rows, (size_t)B * sizeof(imp::KneeRow), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int64_t b = 0; b < B; ++b) {
    const imp::KneeRow& r = h[(size_t)b];
    peak_out[b] = r.peak;
    flags_out[b] = r.done ? r.flags : (r.flags ? r.flags : imp::KNEE_GUARD);
    knee_out[b] = r.knee;
    floor_out[b] = r.floor;
    window_out[b] = r.window;
  }
  return IMP_OK;
}

extern "C" int imp_decay_times(imp_ctx* ctx, const double* x, const int64_t* off, const int64_t* len, int64_t B,
                               const int64_t* peak, const int64_t* knee, const double* noise_floor,
                               const int64_t* window, double fs, double* out) {
  if (!ctx || (B && (!x || !off || !len || !peak || !knee || !noise_floor || !window || !out)))
    return fail(IMP_ERR_INVALID, "imp_decay_times: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0 || !(fs > 0)) return fail(IMP_ERR_INVALID, "imp_decay_times: bad B or fs");
  if (B == 0) return IMP_OK;
  std::vector<imp::DecayJob> jobs((size_t)B);
  int64_t total = 0, scr = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (off[b] < 0 || len[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in response %lld", (long long)b);
    if (window[b] < 1) return fail(IMP_ERR_INVALID, "response %lld: window_size must be >= 1", (long long)b);
    jobs[(size_t)b] = imp::DecayJob{off[b], len[b], peak[b], knee[b] - peak[b], window[b], noise_floor[b], scr};
    scr += 2 * (len[b] + 2);
    total = std::max(total, off[b] + len[b]);
  }
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t bytes = (size_t)(total + scr + 4 * B) * sizeof(double) + (size_t)B * sizeof(imp::DecayJob);
  void* buf = nullptr;
  if ((rc = ctx_scratch(ctx, bytes, &buf))) return rc;
  double* d_x = (double*)buf;
  double* d_scr = d_x + total;
  double* d_out = d_scr + scr;
  imp::DecayJob* d_jobs = (imp::DecayJob*)(d_out + 4 * B);
  if (total) HIP_TRY(hipMemcpyAsync(d_x, x, (size_t)total * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), (size_t)B * sizeof(imp::DecayJob), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(imp::decay_times_kernel<double>, dim3((unsigned)B), dim3(imp::kDecayThreads), 0, s, (const double*)d_x, (const imp::DecayJob*)d_jobs, d_scr, fs, d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)(4 * B) * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

extern "C" int imp_decay_times_device(imp_ctx* ctx, const float* d_x, const int64_t* off, const int64_t* len, int64_t B,
                                      const int64_t* peak, const int64_t* knee, const double* noise_floor,
                                      const int64_t* window, double fs, double* out) {
  if (!ctx || (B && (!d_x || !off || !len || !peak || !knee || !noise_floor || !window || !out)))
    return fail(IMP_ERR_INVALID, "imp_decay_times_device: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0 || !(fs > 0)) return fail(IMP_ERR_INVALID, "imp_decay_times_device: bad B or fs");
  if (B == 0) return IMP_OK;
  std::vector<imp::DecayJob> jobs((size_t)B);
  int64_t scr = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (off[b] < 0 || len[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in response %lld", (long long)b);
    if (window[b] < 1) return fail(IMP_ERR_INVALID, "response %lld: window_size must be >= 1", (long long)b);
    jobs[(size_t)b] = imp::DecayJob{off[b], len[b], peak[b], knee[b] - peak[b], window[b], noise_floor[b], scr};
    scr += 2 * (len[b] + 2);
  }
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t bytes = (size_t)(scr + 4 * B) * sizeof(double) + (size_t)B * sizeof(imp::DecayJob);
  void* buf = nullptr;
  if ((rc = ctx_scratch(ctx, bytes, &buf))) return rc;
  double* d_scr = (double*)buf;
  double* d_out = d_scr + scr;
  imp::DecayJob* d_jobs = (imp::DecayJob*)(d_out + 4 * B);
  HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), (size_t)B * sizeof(imp::DecayJob), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(imp::decay_times_kernel<float>, dim3((unsigned)B), dim3(imp::kDecayThreads), 0, s, d_x, (const imp::DecayJob*)d_jobs, d_scr, fs, d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)(4 * B) * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return IMP_OK;
}

extern "C" int imp_sosfilt(imp_ctx* ctx, const double* sos, int64_t n_sections, const double* x, const int64_t* off,
                           const int64_t* len, int64_t B, double* y) {
  if (!ctx || !sos || (B && (!x || !off || !len || !y))) return fail(IMP_ERR_INVALID, "imp_sosfilt: null argument");
  IMP_CTX_LOCK(ctx);
  if (n_sections < 1 || n_sections > 4096) return fail(IMP_ERR_INVALID, "imp_sosfilt: n_sections must be in [1, 4096]");
  for (int64_t s = 0; s < n_sections; ++s)
    if (sos[6 * s + 3] != 1.0) return fail(IMP_ERR_INVALID, "imp_sosfilt: section %lld is not normalised (a0 != 1)", (long long)s);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  int64_t total = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (off[b] < 0 || len[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)b);
    total = std::max(total, off[b] + len[b]);
  }
  if (total == 0) return IMP_OK;
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  const size_t bytes = (size_t)(2 * total + 6 * n_sections) * sizeof(double) + (size_t)(2 * B) * sizeof(int64_t);
  void* buf = nullptr;
  if ((rc = ctx_scratch(ctx, bytes, &buf))) return rc;
  double* d_x = (double*)buf;
  double* d_y = d_x + total;
  double* d_sos = d_y + total;
  int64_t* d_off = (int64_t*)(d_sos + 6 * n_sections);
  int64_t* d_len = d_off + B;
  HIP_TRY(hipMemcpyAsync(d_x, x, (size_t)total * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_sos, sos, (size_t)(6 * n_sections) * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_off, off, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_len, len, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, st));
  // a cascade is its sections applied one after the other: chunks of kMaxSections ping-pong between the buffers
  for (int64_t s0 = 0; s0 < n_sections; s0 += imp::kMaxSections) {
    const int ns = (int)std::min<int64_t>(imp::kMaxSections, n_sections - s0);
    hipLaunchKernelGGL(imp::sosfilt_kernel, dim3((unsigned)B), dim3(64), 0, st, d_sos + 6 * s0, ns, d_x, d_y, d_off, d_len);
    HIP_TRY(hipGetLastError());
    std::swap(d_x, d_y);
  }
  std::swap(d_x, d_y);                                     // d_y = output of the last chunk
  HIP_TRY(hipMemcpyAsync(y, d_y, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return IMP_OK;
}

// K10 on the stream: the lag slices of every pair, then the first maximum over the slices.  lds_doubles = longest a_len + b_len,
// nk_max = longest a_len + b_len - 1; part_k / part_val: [B * xcorr_slices(nk_max)]
static int64_t xcorr_slices(int64_t nk_max) {
  const int64_t per = (int64_t)imp::kXcorrThreads * imp::kXcorrLags;
  return std::max<int64_t>(1, (nk_max + per - 1) / per);
}
template <class Sample>
static int launch_xcorr(imp_ctx* ctx, hipStream_t s, const Sample* a, const int64_t* a_off, const int64_t* a_len, const Sample* b,
                        const int64_t* b_off, const int64_t* b_len, int64_t B, int64_t lds_doubles, int64_t nk_max, long long* part_k,
                        double* part_val, long long* d_arg, double* d_val) {
  int rc;
  if ((rc = ctx_kernel_lds(ctx, reinterpret_cast<const void*>(imp::xcorr_argmax_kernel<Sample>), (size_t)imp::xcorr_lds_doubles(16384, 0) * sizeof(double))))
    return rc;
  const int64_t S = xcorr_slices(nk_max);
  hipLaunchKernelGGL(imp::xcorr_argmax_kernel<Sample>, dim3((unsigned)B, (unsigned)S), dim3(imp::kXcorrThreads),
                     (size_t)imp::xcorr_lds_doubles(lds_doubles, 0) * sizeof(double), s, a, a_off, a_len, b, b_off, b_len, part_k, part_val);
  hipLaunchKernelGGL(imp::xcorr_reduce_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, (const long long*)part_k, (const double*)part_val, (int)S,
                     (int)B, d_arg, d_val);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

extern "C" int imp_xcorr_argmax(imp_ctx* ctx, const double* a, const int64_t* a_off, const int64_t* a_len,
                                const double* b, const int64_t* b_off, const int64_t* b_len, int64_t B,
                                int64_t* arg_out, double* val_out) {
  if (!ctx || (B && (!a || !a_off || !a_len || !b || !b_off || !b_len || !arg_out)))
    return fail(IMP_ERR_INVALID, "imp_xcorr_argmax: null argument");
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  int64_t ta = 0, tb = 0, lds = 0;
  for (int64_t p = 0; p < B; ++p) {
    if (a_len[p] < 1 || b_len[p] < 1 || a_off[p] < 0 || b_off[p] < 0)
      return fail(IMP_ERR_INVALID, "pair %lld: empty segment or negative offset", (long long)p);   // scipy raises on empty input
    if (a_len[p] + b_len[p] > 16384)
      return fail(IMP_ERR_UNSUPPORTED, "pair %lld: %lld + %lld samples exceed the 16384 the lag search holds in LDS",
                  (long long)p, (long long)a_len[p], (long long)b_len[p]);
    ta = std::max(ta, a_off[p] + a_len[p]);
    tb = std::max(tb, b_off[p] + b_len[p]);
    lds = std::max(lds, a_len[p] + b_len[p]);
  }
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  IMP_CTX_LOCK(ctx);
  hipStream_t s = ctx->stream;
  // scratch: a, b (fp64), 4 x int64 meta, arg (int64), val (fp64)
  const size_t meta = (size_t)B * sizeof(int64_t);
  const int64_t S = xcorr_slices(lds - 1);
  const size_t bytes = (size_t)(ta + tb) * sizeof(double) + 6 * meta + 2 * (size_t)(B * S) * 8;
  void* scr = nullptr;
  if ((rc = ctx_scratch(ctx, bytes, &scr))) return rc;
  double* d_a = (double*)scr;
  double* d_b = d_a + ta;
  int64_t* d_meta = (int64_t*)(d_b + tb);
  long long* d_arg = (long long*)(d_meta + 4 * B);
  double* d_val = (double*)(d_arg + B);
  long long* d_pk = (long long*)(d_val + B);
  double* d_pv = (double*)(d_pk + B * S);
  HIP_TRY(hipMemcpyAsync(d_a, a, (size_t)ta * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_b, b, (size_t)tb * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_meta, a_off, meta, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_meta + B, a_len, meta, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_meta + 2 * B, b_off, meta, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_meta + 3 * B, b_len, meta, hipMemcpyHostToDevice, s));
  if ((rc = launch_xcorr<double>(ctx, s, d_a, d_meta, d_meta + B, d_b, d_meta + 2 * B, d_meta + 3 * B, B, lds, lds - 1, d_pk, d_pv, d_arg, d_val)))
    return rc;
  std::vector<long long> h_arg((size_t)B);
  std::vector<double> h_val((size_t)B);
  HIP_TRY(hipMemcpyAsync(h_arg.data(), d_arg, (size_t)B * sizeof(long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_val.data(), d_val, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int64_t p = 0; p < B; ++p) {
    arg_out[p] = (int64_t)h_arg[(size_t)p];
    if (val_out) val_out[p] = h_val[(size_t)p];
  }
  return IMP_OK;
}

extern "C" int imp_xcorr_argmax_device(imp_ctx* ctx, const float* d_x, const int64_t* a_off, const int64_t* a_len,
                                       const int64_t* b_off, const int64_t* b_len, int64_t B, int64_t* arg_out, double* val_out) {
  if (!ctx || (B && (!d_x || !a_off || !a_len || !b_off || !b_len || !arg_out)))
    return fail(IMP_ERR_INVALID, "imp_xcorr_argmax_device: null argument");
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  int64_t lds = 0;
  for (int64_t p = 0; p < B; ++p) {
    if (a_len[p] < 1 || b_len[p] < 1 || a_off[p] < 0 || b_off[p] < 0)
      return fail(IMP_ERR_INVALID, "pair %lld: empty segment or negative offset", (long long)p);
    if (a_len[p] + b_len[p] > 16384)
      return fail(IMP_ERR_UNSUPPORTED, "pair %lld: %lld + %lld samples exceed the 16384 the lag search holds in LDS",
                  (long long)p, (long long)a_len[p], (long long)b_len[p]);
    lds = std::max(lds, a_len[p] + b_len[p]);
  }
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t meta = (size_t)B * sizeof(int64_t);
  const int64_t S = xcorr_slices(lds - 1);
  void* scr = nullptr;
  if ((rc = ctx_scratch(ctx, 6 * meta + 2 * (size_t)(B * S) * 8, &scr))) return rc;
  int64_t* d_meta = (int64_t*)scr;
  long long* d_arg = (long long*)(d_meta + 4 * B);
  double* d_val = (double*)(d_arg + B);
  long long* d_pk = (long long*)(d_val + B);
  double* d_pv = (double*)(d_pk + B * S);
  HIP_TRY(hipMemcpyAsync(d_meta, a_off, meta, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_meta + B, a_len, meta, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_meta + 2 * B, b_off, meta, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_meta + 3 * B, b_len, meta, hipMemcpyHostToDevice, s));
  if ((rc = launch_xcorr<float>(ctx, s, d_x, d_meta, d_meta + B, d_x, d_meta + 2 * B, d_meta + 3 * B, B, lds, lds - 1, d_pk, d_pv, d_arg, d_val)))
    return rc;
  std::vector<long long> h_arg((size_t)B);
  std::vector<double> h_val((size_t)B);
  HIP_TRY(hipMemcpyAsync(h_arg.data(), d_arg, (size_t)B * sizeof(long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_val.data(), d_val, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int64_t p = 0; p < B; ++p) {
    arg_out[p] = (int64_t)h_arg[(size_t)p];
    if (val_out) val_out[p] = h_val[(size_t)p];
  }
  return IMP_OK;
}

extern "C" int imp_shift_rows_device(imp_ctx* ctx, const float* d_src, const int64_t* src_off, const int64_t* len,
                                     const int64_t* shift, int64_t B, float* d_dst, const int64_t* dst_off) {
  if (!ctx || (B && (!d_src || !src_off || !len || !shift || !d_dst || !dst_off)))
    return fail(IMP_ERR_INVALID, "imp_shift_rows_device: null argument");
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  int64_t longest = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (src_off[b] < 0 || dst_off[b] < 0 || len[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)b);
    longest = std::max(longest, len[b]);
  }
  if (longest == 0) return IMP_OK;
  IMP_CTX_LOCK(ctx);
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  void *h = nullptr, *d = nullptr;
  const size_t meta = (size_t)B * sizeof(int64_t);
  if ((rc = ctx_stage(ctx, 4 * meta, &h, &d))) return rc;
  int64_t* hm = (int64_t*)h;
  std::memcpy(hm, src_off, meta);
  std::memcpy(hm + B, len, meta);
  std::memcpy(hm + 2 * B, shift, meta);
  std::memcpy(hm + 3 * B, dst_off, meta);
  if ((rc = ctx_stage_push(ctx, h, d, 4 * meta))) return rc;
  const int64_t* dm = (const int64_t*)d;
  hipLaunchKernelGGL(imp::shift_rows_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (longest + 1023) / 1024)), (unsigned)B), dim3(256), 0,
                     ctx->stream, d_src, dm, dm + B, (const long long*)nullptr, (const long long*)(dm + 2 * B), d_dst, dm + 3 * B);
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

extern "C" int imp_apply_window(imp_ctx* ctx, float* x, const int64_t* off, const int64_t* len, int64_t B,
                                const imp_window_params* params) {
  if (!ctx || (B && (!x || !off || !len || !params))) return fail(IMP_ERR_INVALID, "imp_apply_window: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  int64_t total = 0, maxlen = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (len[b] < 0 || off[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)b);
    if (params[b].fade_in < 0 || params[b].fade_out < 0 || params[b].fade_in > len[b] || params[b].fade_out > len[b])
      return fail(IMP_ERR_INVALID, "fade longer than row %lld", (long long)b);
    if (params[b].decay_half >= 0) {
      // numpy would raise on the concatenate/multiply length mismatch (core/decay.py:391-402)
      if (params[b].decay_start < 0 || params[b].decay_knee > len[b] ||
          params[b].decay_start + params[b].decay_half != params[b].decay_knee)
        return fail(IMP_ERR_INVALID, "decay window of row %lld does not tile the row", (long long)b);
    }
    total = std::max(total, off[b] + len[b]);
    maxlen = std::max(maxlen, len[b]);
  }
  if (total == 0) return IMP_OK;
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  float* d_x = nullptr;
  if ((rc = ctx_block_get(ctx, (size_t)total * sizeof(float), (void**)&d_x))) return rc;
  auto cleanup = [&](int code) {
    (void)hipStreamSynchronize(s);
    (void)ctx_block_put(ctx, d_x);
    return code;
  };
  const size_t meta = (size_t)B * sizeof(int64_t);
  void* scr = nullptr;
  if ((rc = ctx_scratch(ctx, 2 * meta + (size_t)B * sizeof(imp_window_params), &scr))) return cleanup(rc);
  int64_t* d_off = (int64_t*)scr;
  int64_t* d_len = d_off + B;
  imp_window_params* d_par = (imp_window_params*)(d_len + B);
  if (hipMemcpyAsync(d_x, x, (size_t)total * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_off, off, meta, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_len, len, meta, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_par, params, (size_t)B * sizeof(imp_window_params), hipMemcpyHostToDevice, s) != hipSuccess)
    return cleanup(fail(IMP_ERR_HIP, "imp_apply_window: h2d copy failed"));
  const int bpr = (int)std::max<int64_t>(1, std::min<int64_t>(256, (maxlen + 1023) / 1024));
  dim3 grid((unsigned)bpr, (unsigned)B), block(256);
  hipLaunchKernelGGL(imp::apply_window_kernel, grid, block, 0, s, d_x, d_off, d_len,
                     reinterpret_cast<const imp::WindowParams*>(d_par));
  if (hipGetLastError() != hipSuccess) return cleanup(fail(IMP_ERR_HIP, "apply_window launch failed"));
  if (hipMemcpyAsync(x, d_x, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess)
    return cleanup(fail(IMP_ERR_HIP, "imp_apply_window: d2h copy failed"));
  return cleanup(IMP_OK);
}

extern "C" int imp_apply_window_device(imp_ctx* ctx, const float* d_src, const int64_t* src_off, float* d_dst,
                                       const int64_t* dst_off, const int64_t* len, int64_t B,
                                       const imp_window_params* params) {
  if (!ctx || (B && (!d_src || !d_dst || !src_off || !dst_off || !len || !params)))
    return fail(IMP_ERR_INVALID, "imp_apply_window_device: null argument");
  IMP_CTX_LOCK(ctx);
  if (B < 0) return fail(IMP_ERR_INVALID, "B < 0");
  if (B == 0) return IMP_OK;
  int64_t maxlen = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (len[b] < 0 || src_off[b] < 0 || dst_off[b] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)b);
    if (params[b].fade_in < 0 || params[b].fade_out < 0 || params[b].fade_in > len[b] || params[b].fade_out > len[b])
      return fail(IMP_ERR_INVALID, "fade longer than row %lld", (long long)b);
    if (params[b].decay_half >= 0 && (params[b].decay_start < 0 || params[b].decay_knee > len[b] ||
                                      params[b].decay_start + params[b].decay_half != params[b].decay_knee))
      return fail(IMP_ERR_INVALID, "decay window of row %lld does not tile the row", (long long)b);
    maxlen = std::max(maxlen, len[b]);
  }
  if (maxlen == 0) return IMP_OK;
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t meta = (size_t)B * sizeof(int64_t);
  // the tables travel through the staging ring: one copy, and the call returns without waiting for the device
  const size_t tab_bytes = 3 * meta + (size_t)B * sizeof(imp_window_params);
  int64_t *h_tab = nullptr, *d_so = nullptr;
  if ((rc = ctx_stage(ctx, tab_bytes, (void**)&h_tab, (void**)&d_so))) return rc;
  int64_t* d_do = d_so + B;
  int64_t* d_len = d_do + B;
  imp_window_params* d_par = (imp_window_params*)(d_len + B);
  std::memcpy(h_tab, src_off, meta);
  std::memcpy(h_tab + B, dst_off, meta);
  std::memcpy(h_tab + 2 * B, len, meta);
  std::memcpy(h_tab + 3 * B, params, (size_t)B * sizeof(imp_window_params));
  if ((rc = ctx_stage_push(ctx, h_tab, d_so, tab_bytes))) return rc;
  const int bpr = (int)std::max<int64_t>(1, std::min<int64_t>(256, (maxlen + 1023) / 1024));
  hipLaunchKernelGGL(imp::apply_window_copy_kernel, dim3((unsigned)bpr, (unsigned)B), dim3(256), 0, s, d_src, d_so, d_dst, d_do,
                     d_len, reinterpret_cast<const imp::WindowParams*>(d_par));
  HIP_TRY(hipGetLastError());
  return IMP_OK;
}

extern "C" int imp_rows_to_pcm_device(imp_ctx* ctx, const float* d_rows, const int64_t* off, const int64_t* len,
                                      int64_t n_rows, const int64_t* row_of_track, int64_t n_tracks, int64_t n_frames,
                                      int bits, void* pcm_out) {
  if (!ctx || !d_rows || !off || !len || !row_of_track || !pcm_out)
    return fail(IMP_ERR_INVALID, "imp_rows_to_pcm_device: null argument");
  IMP_CTX_LOCK(ctx);
  if (bits != 16 && bits != 24 && bits != 32) return fail(IMP_ERR_INVALID, "Invalid bit depth. Accepted values are 16, 24 and 32.");
  if (n_rows < 0 || n_tracks < 1 || n_tracks > 4096 || n_frames < 0) return fail(IMP_ERR_INVALID, "imp_rows_to_pcm_device: bad sizes");
  for (int64_t r = 0; r < n_rows; ++r)
    if (off[r] < 0 || len[r] < 0) return fail(IMP_ERR_INVALID, "negative offset/length in row %lld", (long long)r);
  for (int64_t t = 0; t < n_tracks; ++t)
    if (row_of_track[t] < -1 || row_of_track[t] >= n_rows) return fail(IMP_ERR_INVALID, "track %lld names row %lld", (long long)t, (long long)row_of_track[t]);
  if (n_frames == 0) return IMP_OK;
  int rc = ctx_bind(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t sample = bits == 16 ? 2 : 4;
  const size_t out_bytes = (size_t)n_frames * (size_t)n_tracks * sample;
  const size_t meta = (size_t)(2 * std::max<int64_t>(n_rows, 1) + n_tracks) * sizeof(int64_t);
  char* d_buf = nullptr;
  if ((rc = ctx_block_get(ctx, meta + out_bytes, (void**)&d_buf))) return rc;
  int64_t* d_off = (int64_t*)d_buf;
  int64_t* d_len = d_off + std::max<int64_t>(n_rows, 1);
  int64_t* d_map = d_len + std::max<int64_t>(n_rows, 1);
  void* d_out = d_buf + meta;
  auto done = [&](int code) {
    (void)hipStreamSynchronize(s);
    (void)ctx_block_put(ctx, d_buf);
    return code;
  };
  if ((n_rows && (hipMemcpyAsync(d_off, off, (size_t)n_rows * sizeof(int64_t), hipMemcpyHostToDevice, s) != hipSuccess ||
                  hipMemcpyAsync(d_len, len, (size_t)n_rows * sizeof(int64_t), hipMemcpyHostToDevice, s) != hipSuccess)) ||
      hipMemcpyAsync(d_map, row_of_track, (size_t)n_tracks * sizeof(int64_t), hipMemcpyHostToDevice, s) != hipSuccess)
    return done(fail(IMP_ERR_HIP, "imp_rows_to_pcm_device: upload failed"));
  const int64_t total = n_frames * n_tracks;
  hipLaunchKernelGGL(imp::rows_to_pcm_kernel, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, s, d_rows,
                     d_off, d_len, d_map, (int)n_tracks, n_frames, bits, d_out);
  if (hipGetLastError() != hipSuccess) return done(fail(IMP_ERR_HIP, "imp_rows_to_pcm_device: launch failed"));
  if (hipMemcpyAsync(pcm_out, d_out, out_bytes, hipMemcpyDeviceToHost, s) != hipSuccess)
    return done(fail(IMP_ERR_HIP, "imp_rows_to_pcm_device: download failed"));
  return done(IMP_OK);
}

#include "slice_host.hip.inc"

#ifdef IMP_PHASE_TRACE
// Diagnostic build only (not in include/impulse_hip.h): copies the rows-kernel phase marks to the host.
extern "C" int imp_debug_phase_trace(unsigned long long* out, int64_t n_words) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(imp::g_phase_trace), (size_t)n_words * sizeof(unsigned long long), 0,
                          hipMemcpyDeviceToHost) != hipSuccess)
    return IMP_ERR_HIP;
  return IMP_OK;
}
#endif
