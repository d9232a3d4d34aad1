// Internal declarations shared by the translation units of libimpulse_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/impulse_hip.h"
#include "fft_regs.hip.h"

using imp::cf;

int imp_fail(int code, const char* fmt, ...);
#define fail imp_fail

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(IMP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)

struct TwSet {
  cf* full = nullptr;     // exp(-2 pi i (k1 n2 mod Nc) / Nc) at [k1*4096 + n2]
  cf* hi = nullptr;       // exp(-2 pi i 1024 m / Nc), m < Nc/1024 = 4 N1   (w_N1^j = hi[4 j])
};

struct MinPhasePlan;      // minphase.hip

struct imp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  cf* tw_t1 = nullptr;                  // row-pass stage tables, see conv_kernels.hip.h
  cf* tw_t2 = nullptr;
  cf* tw_t4 = nullptr;
  std::map<int, TwSet> tw_by_n1;        // keyed by N1
  std::recursive_mutex mu;          // serialises the entry points that touch this context (IMP_CTX_LOCK)
  // scratch for the small ragged kernels
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // staging ring for small host tables: pinned host memory mirrored in device memory (impulse_hip.hip ctx_stage)
  char* stage_host = nullptr;
  char* stage_dev = nullptr;
  size_t stage_cap = 0, stage_pos = 0;
  // extra streams for overlapped launch groups (imp_plan_set_overlap); lane 0 is `stream`
  std::vector<hipStream_t> side_streams;
  // K6 plans keyed by (taps n, fs)
  std::map<std::pair<long long, long long>, MinPhasePlan*> minphase_plans;
  // K2 plans keyed by row length n; k2_bluestein_only (IMPULSE_HIP_K2_BLUESTEIN when the context was made): every length
  // through the chirp-z transform, the cross-check of the direct transform that smooth lengths take
  std::map<long long, struct MagPlan*> magnitude_plans;
  bool k2_bluestein_only = false;
  // fp64 roots of unity on the device, keyed by transform length (filter-spectrum preparation)
  std::map<long long, void*> fft_roots;
  // kernels whose dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) has been made ON THIS DEVICE:
  // the attribute is per device, so it is tracked per context, under the context lock
  std::set<const void*> lds_opt_in;
  // imp_malloc / imp_free: blocks handed back are kept for the next request of about that size (hipFree costs ~0.4 ms
  // and drains the device; the per-measurement row blocks of the slice come and go at fixed sizes)
  std::map<void*, size_t> live_blocks;          // size of every block imp_malloc handed out
  std::multimap<size_t, void*> free_blocks;     // by size
  size_t free_bytes = 0;
  size_t free_cap = (size_t)2 << 30;            // IMPULSE_HIP_POOL_MB
};

// pooled device blocks (impulse_hip.hip): get = IMP_OK or IMP_ERR_ALLOC with the message set; put = hand back a block
// nothing in flight uses any more (false: not a block of this context's pool)
int ctx_block_get(imp_ctx* ctx, size_t bytes, void** dptr);
bool ctx_block_put(imp_ctx* ctx, void* dptr);

// staging ring (impulse_hip.hip): `bytes` at the same offset of a pinned host ring and its device mirror; fill the host
// side, then ctx_stage_push sends it in stream order.  Valid until the ring wraps (which waits for the stream).
int ctx_stage(imp_ctx* ctx, size_t bytes, void** host, void** dev);
int ctx_stage_push(imp_ctx* ctx, const void* host, void* dev, size_t bytes);

// opt a kernel into `bytes` of dynamic LDS on the context's device, once per context
int ctx_kernel_lds(imp_ctx* ctx, const void* kernel, size_t bytes);

// Every compute entry point holds the context's lock from argument check to return: the Python host
// shares one context between threads (reference: ThreadPoolExecutor workers, core/hrir.py:529-537,
// core/parallel_utils.py:53-55) and ctypes drops the GIL during calls.  GPU work is stream ordered
// per context anyway, so the lock costs no overlap.
#define IMP_CTX_LOCK(ctx) std::lock_guard<std::recursive_mutex> imp_ctx_lock_((ctx)->mu)

int ctx_bind(imp_ctx* ctx);
// a new non-blocking stream of the context's device
int ctx_new_stream(imp_ctx* ctx, hipStream_t* out);
void minphase_plans_destroy(imp_ctx* ctx);
void magnitude_plans_destroy(imp_ctx* ctx);
void fft_roots_destroy(imp_ctx* ctx);
// alpha/beta planes of `n_filters` real filters (host fp64, row pitch filter_ld) for circular length
// 2 Nc, computed in fp64 ON THE DEVICE and rounded once to fp32 into d_ab[n_filters][N1*4096]
// (register order of the row pass).  minphase.hip, next to the fp64 Stockham FFT it uses.
int spectrum_alpha_beta_device(imp_ctx* ctx, const double* filters, int64_t M, int64_t n_filters, int64_t filter_ld,
                               int64_t Nc, int N1, float4* d_ab, bool filters_on_device = false);
// Pair-mode spectrum (conv_kernels.hip.h rows_single_kernel): H[k] / Nc of ONE real filter zero-padded to the circular
// length Nc = N1 * 4096 samples, all Nc bins, fp64 on the device, rounded once to fp32 into d_hs[N1][4096] in the row
// pass's register order.
int spectrum_pair_device(imp_ctx* ctx, const double* filter, int64_t M, int64_t Nc, int N1, cf* d_hs);

// K2 of imp_slice (minphase.hip): maxima of the ear sums' magnitude responses for row lengths that live on the device
struct SliceNorm;
int slice_norm_create(imp_ctx* ctx, int64_t n_max, int64_t m_cap, SliceNorm** out);
void slice_norm_destroy(SliceNorm* p);
int64_t slice_norm_mfft(const SliceNorm* p);
int slice_norm_run(imp_ctx* ctx, SliceNorm* p, const float* d_rows, int64_t pitch, int rows_per_meas, const long long* d_n,
                   int64_t M, double* d_peak_db);
