// Batched complex fp64 FFT whose sub-transforms live in LDS: one or two launches per transform instead of one launch per
// radix pass through global memory.
//
//   N = P1 x P2 (each a product of the radices 8 / 4 / 2 / 3 / 5 / 11, each <= 1024):
//     pass 1   vectors = the P2 columns n2 (stride-P2 samples), P1-point transform over n1, x w_N^(n2 k1) -> Y[k1][n2]
//     pass 2   vectors = the P1 rows k1 of Y (contiguous), P2-point transform over n2 -> X[k1 + P1 k2]
//   N <= 1024: one pass, vectors = the transforms of the batch.
//
// A workgroup of 256 threads owns a TILE of T vectors (16, 8 or 4: tile_vectors below): the tile is loaded with
// the lanes along whichever direction is contiguous in memory, lives in LDS as [point][vector] with the vector index
// fastest and a pitch of T + 1 complex numbers (16-byte elements: a wave's 64 accesses then spread over all banks for every
// stride the stages use), is transformed IN PLACE by decimation-in-frequency stages (a barrier between stages, nothing
// else; the stages' twiddles are the P-th roots of unity, copied once per workgroup from the ONE table of N-th roots in
// global memory into LDS), and leaves in the
// mixed-radix digit-reversed order the in-place stages produce - the store puts every point where it belongs (a small
// index table in LDS), again with the lanes along the contiguous direction of the destination.
// HBM traffic: one read and one write of the batch per pass.  fp64 throughout; no MFMA (butterflies, not a contraction).
//
// Users: the minimum-phase FIR design (K6: 2 x 32 768-point and 4 x 19 200-point transforms per channel at 48 kHz), the
// magnitude responses (K2) and the filter-spectrum preparation of the convolution plans.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace fft64 {

typedef double2 cplx;
constexpr int kMaxStages = 10;
constexpr int kMaxPoints = 1024;          // sub-transform length a tile can hold

struct Args {
  const cplx* in;
  cplx* out;
  const cplx* roots;              // exp(-2 pi i k / n_roots), k < n_roots
  long long n_groups;             // vectors in the launch = batch x vectors per transform
  int nvec;                       // vectors per transform
  long long in_batch, out_batch;  // elements between the transforms of the batch
  long long in_vec, in_elem;      // input of (vector v, point i) at v * in_vec + i * in_elem within its transform
  long long out_vec, out_elem;    // output of (vector v, bin k)
  int P;                          // points per vector
  int n_roots;
  int twiddle;                    // 1: bin k of vector v is multiplied by roots[v * k] (the four-step twiddle, v k < n_roots)
  int dir;                        // -1: forward (roots as stored), +1: inverse (conjugates); no scaling
  int nstages;
  int radix[kMaxStages];
  // division by multiplication (the kernel divides small indices by the block sizes of the stages and by nvec for every
  // point; an integer division costs the ALU dozens of instructions): q = mulhi(n, magic) is exact while n x divisor < 2^32
  unsigned m_blk[kMaxStages];     // for P / (radix[0] ... radix[s])
  unsigned m_nvec;
};

__host__ __device__ inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / d) + 1u; }
__device__ __forceinline__ unsigned div_magic(unsigned n, unsigned d, unsigned magic) { return d <= 1 ? n : __umulhi(n, magic); }

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// a * (-i) for DIR < 0 (forward), a * (+i) for DIR > 0
template <int DIR>
__device__ __forceinline__ cplx rot90(cplx a) { return DIR < 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x); }

template <int DIR>
__device__ __forceinline__ void bf2(cplx& a, cplx& b) {
  const cplx t = csub(a, b);
  a = cadd(a, b);
  b = t;
}

// x0..x3 -> X0..X3 (natural order)
template <int DIR>
__device__ __forceinline__ void bf4(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
  const cplx a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = rot90<DIR>(csub(x1, x3));
  x0 = cadd(a, c);
  x2 = csub(a, c);
  x1 = cadd(b, d);
  x3 = csub(b, d);
}

template <int DIR>
__device__ __forceinline__ void bf8(cplx* x) {
  // n = 2 n1 + n2 ; k = k1 + 4 k2: two 4-point transforms over n1, twiddle w8^(n2 k1), four 2-point ones
  constexpr double S = 0.70710678118654752440;
  bf4<DIR>(x[0], x[2], x[4], x[6]);
  bf4<DIR>(x[1], x[3], x[5], x[7]);
  // w8^1 = (1 -+ i) / sqrt2, w8^2 = -+ i, w8^3 = (-1 -+ i) / sqrt2   (upper sign: forward)
  {
    const cplx t = x[3];
    x[3] = DIR < 0 ? make_double2((t.x + t.y) * S, (t.y - t.x) * S) : make_double2((t.x - t.y) * S, (t.y + t.x) * S);
    x[5] = rot90<DIR>(x[5]);
    const cplx u = x[7];
    x[7] = DIR < 0 ? make_double2((u.y - u.x) * S, -(u.x + u.y) * S) : make_double2(-(u.x + u.y) * S, (u.x - u.y) * S);
  }
  bf2<DIR>(x[0], x[1]);     // X0, X4
  bf2<DIR>(x[2], x[3]);     // X1, X5
  bf2<DIR>(x[4], x[5]);     // X2, X6
  bf2<DIR>(x[6], x[7]);     // X3, X7
  const cplx t1 = x[1], t2 = x[2], t3 = x[3], t4 = x[4], t5 = x[5], t6 = x[6];
  x[1] = t2; x[2] = t4; x[3] = t6; x[4] = t1; x[5] = t3; x[6] = t5;
}

template <int DIR>
__device__ __forceinline__ void bf3(cplx& a, cplx& b, cplx& c) {
  constexpr double S = 0.86602540378443864676;
  const cplx t = cadd(b, c);
  const cplx m = make_double2(a.x - 0.5 * t.x, a.y - 0.5 * t.y);
  const cplx d = rot90<DIR>(make_double2(S * (b.x - c.x), S * (b.y - c.y)));
  a = cadd(a, t);
  b = cadd(m, d);
  c = csub(m, d);
}

template <int DIR>
__device__ __forceinline__ void bf5(cplx* x) {
  constexpr double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;
  constexpr double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;
  const cplx t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]), t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
  const cplx m1 = make_double2(x[0].x + C1 * t1.x + C2 * t2.x, x[0].y + C1 * t1.y + C2 * t2.y);
  const cplx m2 = make_double2(x[0].x + C2 * t1.x + C1 * t2.x, x[0].y + C2 * t1.y + C1 * t2.y);
  const cplx n1 = rot90<DIR>(make_double2(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y));
  const cplx n2 = rot90<DIR>(make_double2(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y));
  x[0] = cadd(x[0], cadd(t1, t2));
  x[1] = cadd(m1, n1);
  x[4] = csub(m1, n1);
  x[2] = cadd(m2, n2);
  x[3] = csub(m2, n2);
}

// generic R-point DFT with the R-th roots taken from the table (prime radices beyond 5: 11)
template <int R, int DIR>
__device__ __forceinline__ void bf_generic(cplx* x, const cplx* __restrict__ roots, int n_roots) {
  cplx w[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    w[k] = roots[(long long)k * (n_roots / R)];
    if (DIR > 0) w[k].y = -w[k].y;
  }
  cplx y[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    cplx acc = x[0];
#pragma unroll
    for (int k = 1; k < R; ++k) acc = cadd(acc, cmul(x[k], w[(j * k) % R]));
    y[j] = acc;
  }
#pragma unroll
  for (int j = 0; j < R; ++j) x[j] = y[j];
}

template <int R, int DIR>
__device__ __forceinline__ void butterfly(cplx* x, const cplx* __restrict__ roots, int n_roots) {
  if constexpr (R == 2) bf2<DIR>(x[0], x[1]);
  else if constexpr (R == 3) bf3<DIR>(x[0], x[1], x[2]);
  else if constexpr (R == 4) bf4<DIR>(x[0], x[1], x[2], x[3]);
  else if constexpr (R == 5) bf5<DIR>(x);
  else if constexpr (R == 8) bf8<DIR>(x);
  else bf_generic<R, DIR>(x, roots, n_roots);
}

// one decimation-in-frequency stage on the tile: blocks of `blk` points, radix R, in place; tw = the P-th roots of unity
// (LDS copy made when the tile was loaded)
template <int R, int DIR, int T>
__device__ __forceinline__ void stage(cplx* buf, int P, int blk, int tw_step, unsigned m_sub, const cplx* __restrict__ tw,
                                      const cplx* __restrict__ roots, int n_roots, int tid) {
  constexpr int pitch = T + 1;
  const int sub = blk / R, nb = P / R;                     // (R is a compile-time constant: no division)
  const int v = tid % T;
  for (int u = tid / T; u < nb; u += 256 / T) {
    const int block = (int)div_magic((unsigned)u, (unsigned)sub, m_sub), j = u - block * sub;
    cplx* p = buf + (size_t)(block * blk + j) * pitch + v;
    cplx x[R];
#pragma unroll
    for (int k = 0; k < R; ++k) x[k] = p[(size_t)k * sub * pitch];
    butterfly<R, DIR>(x, roots, n_roots);
    if (sub > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) {
        cplx w = tw[(j * k) * tw_step];                       // j k < blk: inside the table
        if (DIR > 0) w.y = -w.y;
        x[k] = cmul(x[k], w);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) p[(size_t)k * sub * pitch] = x[k];
  }
}

template <int DIR, int T>
__device__ __forceinline__ void run_stages(const Args& a, cplx* buf, const cplx* tw, int tid) {
  int blk = a.P, tw_step = 1;                              // w_blk = tw[tw_step], tw_step = P / blk
  for (int s = 0; s < a.nstages; ++s) {
    const int R = a.radix[s];
    const unsigned m = a.m_blk[s];
    switch (R) {
      case 8: stage<8, DIR, T>(buf, a.P, blk, tw_step, m, tw, a.roots, a.n_roots, tid); blk >>= 3; break;
      case 4: stage<4, DIR, T>(buf, a.P, blk, tw_step, m, tw, a.roots, a.n_roots, tid); blk >>= 2; break;
      case 2: stage<2, DIR, T>(buf, a.P, blk, tw_step, m, tw, a.roots, a.n_roots, tid); blk >>= 1; break;
      case 3: stage<3, DIR, T>(buf, a.P, blk, tw_step, m, tw, a.roots, a.n_roots, tid); blk /= 3; break;
      case 5: stage<5, DIR, T>(buf, a.P, blk, tw_step, m, tw, a.roots, a.n_roots, tid); blk /= 5; break;
      case 11: stage<11, DIR, T>(buf, a.P, blk, tw_step, m, tw, a.roots, a.n_roots, tid); blk /= 11; break;
      default: break;
    }
    tw_step *= R;
    __syncthreads();
  }
}

// optional elementwise hooks: In(value read, transform b, point index within the transform) and Out(value, b, bin index)
struct NoOp {
  __device__ __forceinline__ cplx operator()(cplx v, long long, long long) const { return v; }
};

template <int T, class InOp, class OutOp>
__global__ __launch_bounds__(256) void tile_kernel(Args a, InOp in_op, OutOp out_op) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int pitch = T + 1;
  cplx* buf = reinterpret_cast<cplx*>(smem_raw);
  cplx* tw = buf + (size_t)a.P * pitch;                    // the P-th roots of unity, for the stages
  unsigned short* rev = reinterpret_cast<unsigned short*>(tw + a.P);
  const int tid = threadIdx.x;
  const long long g0 = (long long)blockIdx.x * T;
  {
    const long long step = a.n_roots / a.P;
    for (int k = tid; k < a.P; k += 256) tw[k] = a.roots[(long long)k * step];
  }
  // where the point at position p of the in-place result belongs: its digits, most significant first, are the bin's
  // digits least significant first
  for (int p = tid; p < a.P; p += 256) {
    int rem = p, k = 0, w = 1, blk = a.P;
    for (int s = 0; s < a.nstages; ++s) {
      const int R = a.radix[s];
      blk = R == 8 ? blk >> 3 : R == 4 ? blk >> 2 : R == 2 ? blk >> 1 : blk / R;
      const int d = (int)div_magic((unsigned)rem, (unsigned)blk, a.m_blk[s]);
      rem -= d * blk;
      k += d * w;
      w *= R;
    }
    rev[p] = (unsigned short)k;
  }
  // vector g of the launch = vector (g mod nvec) of transform g / nvec
  auto split = [&](long long g, long long& b, long long& vv) {
    const unsigned q = div_magic((unsigned)g, (unsigned)a.nvec, a.m_nvec);
    b = q;
    vv = (long long)((unsigned)g - q * (unsigned)a.nvec);
  };
  // ---- load (lanes along the contiguous direction)
  if (a.in_elem == 1) {
    for (int v = 0; v < T; ++v) {
      const long long g = g0 + v;
      const bool live = g < a.n_groups;
      long long b = 0, vv = 0;
      if (live) split(g, b, vv);
      const cplx* src = a.in + b * a.in_batch + vv * a.in_vec;
      const long long e0 = vv * a.in_vec;
      for (int i = tid; i < a.P; i += 256)
        buf[(size_t)i * pitch + v] = live ? in_op(src[i], b, e0 + i) : make_double2(0.0, 0.0);
    }
  } else {
    const int v = tid % T;
    const long long g = g0 + v;
    const bool live = g < a.n_groups;
    long long b = 0, vv = 0;
    if (live) split(g, b, vv);
    const cplx* src = a.in + b * a.in_batch + vv * a.in_vec;
    const long long e0 = vv * a.in_vec;
    for (int i = tid / T; i < a.P; i += 256 / T)
      buf[(size_t)i * pitch + v] = live ? in_op(src[(long long)i * a.in_elem], b, e0 + (long long)i * a.in_elem) : make_double2(0.0, 0.0);
  }
  __syncthreads();
  if (a.dir < 0) run_stages<-1, T>(a, buf, tw, tid);
  else run_stages<+1, T>(a, buf, tw, tid);
  // ---- store
  auto emit = [&](int v, int p, long long b, long long vv, cplx* dst) {
    const int k = rev[p];
    cplx val = buf[(size_t)p * pitch + v];
    if (a.twiddle) {
      cplx w = a.roots[vv * k];
      if (a.dir > 0) w.y = -w.y;
      val = cmul(val, w);
    }
    const long long e = vv * a.out_vec + (long long)k * a.out_elem;
    dst[(long long)k * a.out_elem] = out_op(val, b, e);
  };
  if (a.out_vec == 1 || a.out_elem != 1) {
    const int v = tid % T;
    const long long g = g0 + v;
    if (g < a.n_groups) {
      long long b, vv;
      split(g, b, vv);
      cplx* dst = a.out + b * a.out_batch + vv * a.out_vec;
      for (int p = tid / T; p < a.P; p += 256 / T) emit(v, p, b, vv, dst);
    }
  } else {
    for (int v = 0; v < T; ++v) {
      const long long g = g0 + v;
      if (g >= a.n_groups) break;
      long long b, vv;
      split(g, b, vv);
      cplx* dst = a.out + b * a.out_batch + vv * a.out_vec;
      for (int p = tid; p < a.P; p += 256) emit(v, p, b, vv, dst);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
struct Plan {
  int N = 0, P1 = 0, P2 = 0;       // P2 == 1: one pass
  std::vector<int> r1, r2;
  bool ok = false;
};

inline bool factor_points(int P, std::vector<int>& radix) {
  radix.clear();
  int n = P;
  for (int r : {8, 4, 2, 3, 5, 11})
    while (n % r == 0) {
      radix.push_back(r);
      n /= r;
    }
  return n == 1 && (int)radix.size() <= kMaxStages;
}

// N = P1 x P2 with both factors <= kMaxPoints, as balanced as the factors of N allow; P1 gets the larger one (the strided
// pass then has more, shorter rows); every factor a product of the supported radices
inline Plan make_plan(int N) {
  Plan p;
  p.N = N;
  std::vector<int> tmp;
  if (N < 2 || !factor_points(N, tmp) && N <= kMaxPoints) return p;
  if (N <= kMaxPoints) {
    p.P1 = N;
    p.P2 = 1;
    p.ok = factor_points(N, p.r1);
    return p;
  }
  int best = 0;
  for (int a = 2; a <= kMaxPoints; ++a) {
    if (N % a) continue;
    const int b = N / a;
    if (b > kMaxPoints || b > a) continue;               // a >= b
    std::vector<int> ra, rb;
    if (!factor_points(a, ra) || !factor_points(b, rb)) continue;
    if (best == 0 || a < best) best = a;                 // the most balanced split: smallest a with a >= b
  }
  if (!best) return p;
  p.P1 = best;
  p.P2 = N / best;
  p.ok = factor_points(p.P1, p.r1) && factor_points(p.P2, p.r2);
  return p;
}

inline size_t tile_lds(int P, int T) {
  return (size_t)P * (T + 1) * sizeof(cplx) + (size_t)P * sizeof(cplx) + (((size_t)P * 2 + 15) & ~(size_t)15);
}
// vectors per tile: the most that fit the CU's 160 KiB of LDS (16 / 8 / 4), fewer when the launch would otherwise have
// fewer workgroups than two per CU (the small batches of the FIR design: 16 transforms of 19 200 points are 128 tiles of
// 16 vectors - half the chip idle and every workgroup a long serial chain - but 512 tiles of 4)
inline int tile_vectors(int P, long long n_groups) {
  int T = 16;
  while (T > 4 && tile_lds(P, T) > (size_t)156 * 1024) T /= 2;
  while (T > 4 && (n_groups + T - 1) / T < 512) T /= 2;
  return T;
}

}  // namespace fft64
