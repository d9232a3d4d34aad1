// XCD-resident variant of the batched FFT convolution (K1/K5) for transforms whose workspace fits one XCD's L2.
//
// The three-launch path (conv_kernels.hip.h) moves every workspace byte across the L2<->fabric boundary four times
// (write A, read+write B, read C): 4.2x the algorithmic bytes at the 7.1 / 6.15 s configuration.  Here ONE launch
// keeps a channel's workspace (N1 x 4096 complex64 = 2.25 MiB at N1 = 72) inside the 4 MiB L2 of the XCD that works
// on it:
//   * every workgroup reads the XCD it actually runs on (HW_REG_XCC_ID) and serves that XCD's work list;
//   * channel c belongs to XCD c mod nx; an XCD's list is, round by round (round r = its r-th channel),
//       CA(r):  64 column tiles, each = pass C of round r-1 (tile t: ws columns -> cropped output) followed by
//               pass A of round r (input -> the SAME ws columns): the workspace is one channel large, replaced in place;
//       B(r):   N1/2 row pairs, in place (rows_pair, shared with the three-launch path);
//   * workgroups take work by ticket (one returning atomic on the XCD's counter); an item of phase p waits until the
//     XCD's done-counter says every item of phase p-1 is complete.  A ticket only ever waits for LOWER tickets, which
//     are held by workgroups that are already running, so the scheme needs no co-residency and cannot deadlock;
//   * hand-off between CUs of ONE XCD goes through that XCD's L2: producers drain their stores (s_waitcnt vmcnt(0)),
//     meet at the workgroup barrier and one lane adds to the done-counter; consumers poll the counter (sc1 load), then
//     read the workspace with sc1 loads (agent scope: never served from the CU's own, possibly stale, L1).  No L2
//     write-back is needed because reader and writer share the L2 - that is the one thing XCC_ID is trusted for.
//   * once-touched streams (input, output, alpha/beta) use nt so that they do not displace the workspace.
// Every spin is bounded by wall time (s_memrealtime); on expiry the launch sets an abort word and drains.
#pragma once
#include "conv_kernels.hip.h"

namespace imp {

constexpr int kXcdMax = 8;

struct alignas(128) XcdLine {
  unsigned v;
  unsigned pad[31];
};

// control block of one launch (the next launch uses the other of two blocks; each launch zeroes the other one)
struct XcdCtl {
  XcdLine ticket[kXcdMax];
  XcdLine done_ca[kXcdMax];   // monotonic: CA(r) complete <=> done_ca >= tiles * (r + 1)
  XcdLine done_b[kXcdMax];    // monotonic: B(r)  complete <=> done_b  >= pairs * (r + 1)
  XcdLine abort;              // set when a wait expired: outputs are invalid
  XcdLine xcc_seen;           // OR of 1 << XCC_ID over the workgroups that ran (census for the host)
  XcdLine wait_ticks;         // diagnostics: total s_memrealtime ticks lane 0 of every workgroup spent polling
  XcdLine clk[2];             // IMP_XCD_DIAG builds: [0] s_memtime ticks / 1024, [1] s_memrealtime ticks / 1024 over workgroup lifetimes
  XcdLine diag[8];            // IMP_XCD_DIAG builds: ticks in [0] ticket fetch [1] CA prefetch+wait [2] C part [3] A part
                              // [4] CA drain [5] B wait [6] B work [7] B drain, summed over workgroups
};

// four-step twiddle w_Nc^(k1 n2) as the product of two small tables (the Nc-entry table of the three-launch path is as
// large as the workspace and would compete with it for the L2): k1 = d + 8 e,
//   P[d][n2] = w_Nc^(d n2), d < 8 ;  Q[e][n2] = w_Nc^(8 e n2), e < N1/8
struct DigitTwiddles {
  const cf* __restrict__ P;
  const cf* __restrict__ Q;
};

template <class Load>
struct XcdArgs {
  Load ld;
  StoreRealCrop st;
  cf* ws;                        // [nx][N1][4096]: one slot per XCD
  const float4* ab;
  long long ab_chan_stride;      // 0: shared spectrum
  int nchan;
  int n1_total;
  int nx;                        // XCDs that share the channels (8)
  XcdCtl* ctl;
  XcdCtl* ctl_next;              // zeroed by this launch for the next one
  unsigned* sticky;              // [0] set on abort and never cleared by the device
  unsigned timeout_ticks;        // s_memrealtime ticks (100 MHz) a single wait may take
};

__device__ __forceinline__ unsigned xcd_poll(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane 0 of wave 0 polls until *counter >= want (or abort); returns false on abort/timeout.  Result is broadcast
// to the workgroup through `flag` (LDS) by the caller.
__device__ __forceinline__ bool xcd_wait(XcdCtl* ctl, unsigned* sticky, const unsigned* counter, unsigned want,
                                         unsigned timeout_ticks) {
  if (xcd_poll(counter) >= want) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  bool ok = true;
  for (;;) {
    // ~1 us between polls: every poll is a fabric read of one line, and a few hundred workgroups polling the same
    // eight lines back to back slow down whatever else lives in those memory channels
    __builtin_amdgcn_s_sleep(32);
    if (xcd_poll(counter) >= want) break;
    if ((++spins & 15u) == 0u) {
      if (xcd_poll(&ctl->abort.v) != 0u) { ok = false; break; }
      if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)timeout_ticks) {
        __hip_atomic_store(&ctl->abort.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
  }
  atomicAdd(&ctl->wait_ticks.v, (unsigned)(__builtin_amdgcn_s_memrealtime() - t0));
  return ok;
}

// ---------------------------------------------------------------------------------------------
// Column tile with F rows per thread and R2 thread groups (T = 64 R2 threads), digit twiddles.
//   inverse: ws columns (sc1) x conj four-step twiddle -> F-point IDFT -> exchange -> R2-point IDFT -> cropped output
//   forward: prefetched input -> F-point DFT -> exchange -> R2-point DFT -> x four-step twiddle -> ws columns
// Same index algebra as cols_mixed_kernel: input rows i = g + R2 j, output rows k = ka + F kb.
// ---------------------------------------------------------------------------------------------
template <int F, int R2>
struct XcdCfg {
  static constexpr int TC = 64;
  static constexpr int T = TC * R2;
  static constexpr int G = (F + R2 - 1) / R2;
  static constexpr size_t cols_lds = sizeof(cf) * F * T;
  static constexpr size_t rows_lds = sizeof(cf) * 2 * 16 * kRowPad;
  static constexpr size_t lds_bytes = cols_lds > rows_lds ? cols_lds : rows_lds;
  static_assert(T <= 512, "the persistent workgroup is 512 threads");
};

template <int DIR, int F>
__device__ __forceinline__ void fft_first_any(cf (&v)[F]) {
  if constexpr (F == 16 || F == 8) fft_first<DIR, F>(v);
  else fft_small<DIR, F>(v);
}

template <int DIR, int R>
__device__ __forceinline__ void fft_second_any(cf (&y)[R]) {
  if constexpr (R == 8) fft8<DIR>(y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7]);
  else if constexpr (R == 4) bfly4<DIR>(y[0], y[1], y[2], y[3]);
  else if constexpr (R == 2) bfly2<DIR>(y[0], y[1]);
  else if constexpr (R == 1) {}
  else fft_small<DIR, R>(y);
}

template <int F, int R2, int DIR>
__device__ __forceinline__ void xcd_first_twiddle(const Twiddles& tw, int g, cf (&v)[F]) {
  if constexpr (R2 > 1) {
    const int gu = __builtin_amdgcn_readfirstlane(g);      // 64-column tiles: g is wave-uniform
#pragma unroll
    for (int a = 1; a < F; ++a) v[a] = ctw_uniform<DIR>(v[a], tw.hi[4 * gu * a]);
  }
}

// pass C of one tile: ws slot -> output channel `chan`
template <int F, int R2>
__device__ __forceinline__ void xcd_tile_inverse(const StoreRealCrop& st, const cf* ws_slot, const Twiddles& tw,
                                                 const DigitTwiddles& dt, int n1_total, int chan, int tile, cf* buf,
                                                 const int tid) {
  using Cfg = XcdCfg<F, R2>;
  constexpr int TC = Cfg::TC, T = Cfg::T, G = Cfg::G;
  const int c = tid % TC, g = tid / TC;
  const unsigned n2 = (unsigned)(tile * TC + c);
  const __amdgpu_buffer_rsrc_t r_ws = make_rsrc(ws_slot, (unsigned)n1_total * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_p = make_rsrc(dt.P, 8u * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_q = make_rsrc(dt.Q, (unsigned)(n1_total / 8) * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_out = st.bind(chan);
  static_assert(R2 == 8 || R2 == 4 || R2 == 2 || R2 == 1, "digit twiddles address rows as d + 8 e with d = g");
  cf v[F];
  // rows i = g + R2 j.  With R2 = 8: d = g, e = j.  (R2 < 8: d = (g + R2 j) & 7, e = (g + R2 j) >> 3.)
  const unsigned e0 = (unsigned)g * kN2 + n2;
#pragma unroll
  for (int j = 0; j < F; ++j) v[j] = bload_cf<16>(r_ws, e0 * 8u, (unsigned)(j * R2 * kN2) * 8u);
  if constexpr (R2 == 8) {
    const cf pg = bload_cf(r_p, e0 * 8u, 0u);
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const cf qj = bload_cf(r_q, n2 * 8u, (unsigned)(j * kN2) * 8u);
      v[j] = cmulc(cmulc(v[j], pg), qj);
    }
  } else {
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const unsigned row = (unsigned)g + (unsigned)(R2 * j);
      const cf pd = bload_cf(r_p, ((row & 7u) * kN2 + n2) * 8u, 0u);
      const cf qe = bload_cf(r_q, ((row >> 3) * kN2 + n2) * 8u, 0u);
      v[j] = cmulc(cmulc(v[j], pd), qe);
    }
  }
  fft_first_any<+1, F>(v);
  xcd_first_twiddle<F, R2, +1>(tw, g, v);
  if constexpr (R2 > 1) {
#pragma unroll
    for (int a = 0; a < F; ++a) buf[a * T + tid] = v[a];
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int ka = g + R2 * i;
    if (ka < F) {
      cf y[R2];
      if constexpr (R2 > 1) {
#pragma unroll
        for (int gp = 0; gp < R2; ++gp) y[gp] = buf[ka * T + gp * TC + c];
        fft_second_any<+1, R2>(y);
      }
      const unsigned e = (unsigned)ka * kN2 + n2;
#pragma unroll
      for (int kb = 0; kb < R2; ++kb) {
        if constexpr (R2 > 1) st.put(r_out, chan, e, (unsigned)(kb * F * kN2), y[kb]);
      }
    }
  }
  if constexpr (R2 == 1) {
#pragma unroll
    for (int a = 0; a < F; ++a) st.put(r_out, chan, n2, (unsigned)(a * kN2), v[a]);
  }
}

// pass A of one tile: prefetched input rows (v) -> ws slot
template <int F, int R2>
__device__ __forceinline__ void xcd_tile_forward(cf (&v)[F], cf* ws_slot, const Twiddles& tw, const DigitTwiddles& dt,
                                                 int n1_total, int tile, cf* buf, const int tid) {
  using Cfg = XcdCfg<F, R2>;
  constexpr int TC = Cfg::TC, T = Cfg::T, G = Cfg::G;
  const int c = tid % TC, g = tid / TC;
  const unsigned n2 = (unsigned)(tile * TC + c);
  const __amdgpu_buffer_rsrc_t r_ws = make_rsrc(ws_slot, (unsigned)n1_total * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_p = make_rsrc(dt.P, 8u * kN2 * 8u);
  const __amdgpu_buffer_rsrc_t r_q = make_rsrc(dt.Q, (unsigned)(n1_total / 8) * kN2 * 8u);
  // four-step twiddles of this thread's outputs k1 = ka + F kb (first batch: ka = g), fetched before the exchange
  auto fetch_twd = [&](int ka, cf (&twd)[R2]) {
#pragma unroll
    for (int kb = 0; kb < R2; ++kb) {
      const unsigned k1 = (unsigned)ka + (unsigned)(F * kb);
      const cf pd = bload_cf(r_p, ((k1 & 7u) * kN2 + n2) * 8u, 0u);
      const cf qe = bload_cf(r_q, ((k1 >> 3) * kN2 + n2) * 8u, 0u);
      twd[kb] = cmul(pd, qe);
    }
  };
  cf twd[R2];
  fetch_twd(g, twd);
  fft_first_any<-1, F>(v);
  xcd_first_twiddle<F, R2, -1>(tw, g, v);
  if constexpr (R2 > 1) {
#pragma unroll
    for (int a = 0; a < F; ++a) buf[a * T + tid] = v[a];
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int ka = g + R2 * i;
    if (ka < F) {                                            // i > 0: a few leftover rows, one wave (g is wave-uniform)
      if (i > 0) fetch_twd(ka, twd);
      cf y[R2];
      if constexpr (R2 > 1) {
#pragma unroll
        for (int gp = 0; gp < R2; ++gp) y[gp] = buf[ka * T + gp * TC + c];
        fft_second_any<-1, R2>(y);
      } else {
        y[0] = v[ka];
      }
      const unsigned e = (unsigned)ka * kN2 + n2;
#pragma unroll
      for (int kb = 0; kb < R2; ++kb) bstore_cf(cmul(y[kb], twd[kb]), r_ws, e * 8u, (unsigned)(kb * F * kN2) * 8u);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The persistent kernel.  512 threads, two workgroups per CU (68 KiB LDS each), any grid size.
// ---------------------------------------------------------------------------------------------
#ifndef IMP_XCD_DIAG
#define IMP_XCD_DIAG 0
#endif
#if IMP_XCD_DIAG
#define XCD_STAMP(k)                                                             \
  do {                                                                           \
    if (tid == 0) {                                                              \
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();          \
      atomicAdd(&ctl->diag[k].v, (unsigned)(now_ - stamp_));                     \
      stamp_ = now_;                                                             \
    }                                                                            \
  } while (0)
#else
#define XCD_STAMP(k)
#endif

template <int F, int R2, class Load>
__global__ __launch_bounds__(512, 4) void xcd_conv_kernel(XcdArgs<Load> args, Twiddles tw, DigitTwiddles dt) {
  using Cfg = XcdCfg<F, R2>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cf* lds = reinterpret_cast<cf*>(smem_raw);
  __shared__ int s_word[2];                                  // [0] ticket, [1] wait verdict

  const int tid0 = threadIdx.x;
  const int xcc = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u);   // HW_REG_XCC_ID[3:0]
  XcdCtl* ctl = args.ctl;
  if (blockIdx.x == 0) {                                     // zero the NEXT launch's control block
    unsigned* z = reinterpret_cast<unsigned*>(args.ctl_next);
    for (unsigned i = tid0; i < sizeof(XcdCtl) / 4; i += 512) z[i] = 0u;
  }
  if (tid0 == 0) atomicOr(&ctl->xcc_seen.v, 1u << xcc);
  if (xcc >= args.nx) return;                                // not a team of this launch (never on an SPX device)

  constexpr int tiles = kN2 / Cfg::TC;                       // 64
  const int pairs = args.n1_total / 2;
  const int per_round = tiles + pairs;
  const int rounds = (args.nchan - xcc + args.nx - 1) / args.nx;      // channels xcc, xcc + nx, ...
  if (rounds <= 0) return;
  cf* ws_slot = args.ws + (long long)xcc * args.n1_total * kN2;
  RowsArgs ra;
  ra.ws = ws_slot;
  ra.ab = args.ab;
  ra.ab_chan_stride = args.ab_chan_stride;
  ra.n1_total = args.n1_total;
  ra.npairs = pairs;
  ra.nchan = args.nchan;

#if IMP_XCD_DIAG
  unsigned long long stamp_ = __builtin_amdgcn_s_memrealtime();
  const unsigned long long clk0_ = __builtin_amdgcn_s_memtime(), rt0_ = stamp_;
#endif
  for (;;) {
    // the thread index is made opaque once per item: everything derived from it (row / column offsets of either
    // phase) is then recomputed per item instead of being hoisted out of the loop and kept live - and spilled -
    // across both phases
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (tid == 0) s_word[0] = (int)atomicAdd(&ctl->ticket[xcc].v, 1u);
    __syncthreads();
    XCD_STAMP(0);
    const int ticket = s_word[0];
    const int r = ticket / per_round;
    const int i = ticket - r * per_round;
    if (r > rounds || (r == rounds && i >= tiles)) break;
    const bool is_ca = i < tiles;                            // workgroup-uniform
    if (is_ca) {
      const bool has_a = r < rounds, has_c = r > 0;
      const int chan_a = xcc + r * args.nx, chan_c = xcc + (r - 1) * args.nx;
      cf v[F];
      if (has_a && tid < Cfg::T) {
        const unsigned e0 = (unsigned)(tid / Cfg::TC) * kN2 + (unsigned)(i * Cfg::TC + tid % Cfg::TC);
        args.ld.template column<R2 * kN2, F>(chan_a, e0, v);          // in flight while we wait for B(r-1)
      }
      if (has_c) {
        if (tid == 0) s_word[1] = xcd_wait(ctl, args.sticky, &ctl->done_b[xcc].v, (unsigned)(pairs * r), args.timeout_ticks) ? 1 : 0;
        __syncthreads();
        XCD_STAMP(1);
        if (!s_word[1]) break;
        if (tid < Cfg::T) xcd_tile_inverse<F, R2>(args.st, ws_slot, tw, dt, args.n1_total, chan_c, i, lds, tid);
        else if constexpr (R2 > 1) __syncthreads();
        __syncthreads();                                     // C's LDS reads done before A writes the plane
        XCD_STAMP(2);
      }
      if (has_a) {
        if (tid < Cfg::T) xcd_tile_forward<F, R2>(v, ws_slot, tw, dt, args.n1_total, i, lds, tid);
        else if constexpr (R2 > 1) __syncthreads();
      }
      XCD_STAMP(3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its stores have reached the L2
      __syncthreads();
      XCD_STAMP(4);
      if (tid == 0) __hip_atomic_fetch_add(&ctl->done_ca[xcc].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const int pair = i - tiles;
      const int chan = xcc + r * args.nx;
      if (tid == 0) s_word[1] = xcd_wait(ctl, args.sticky, &ctl->done_ca[xcc].v, (unsigned)(tiles * (r + 1)), args.timeout_ticks) ? 1 : 0;
      __syncthreads();
      XCD_STAMP(5);
      if (!s_word[1]) break;
      rows_pair<16, 2>(ra, tw, 0, chan, pair, lds, tid);
      XCD_STAMP(6);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      XCD_STAMP(7);
      if (tid == 0) __hip_atomic_fetch_add(&ctl->done_b[xcc].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#if IMP_XCD_DIAG
  if (threadIdx.x == 0) {
    atomicAdd(&ctl->clk[0].v, (unsigned)((__builtin_amdgcn_s_memtime() - clk0_) >> 10));
    atomicAdd(&ctl->clk[1].v, (unsigned)((__builtin_amdgcn_s_memrealtime() - rt0_) >> 10));
  }
#endif
}

}  // namespace imp
