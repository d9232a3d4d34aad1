// In-register radix-2/3/4/5/8/16 butterflies for gfx950 (wave64, fp32).
// Every array index below is a compile-time constant after unrolling, so the
// 16 complex values of a thread stay in VGPRs (32 regs) - no scratch.
//
// The row and column passes are VALU-issue bound (tools/phase_trace.py: ~75 % of a row workgroup's
// life is butterflies), so on the device every complex primitive is ONE or TWO packed-fp32 VOP3P
// instructions (v_pk_add/mul/fma_f32) whose op_sel / neg modifiers do the re<->im swaps and sign
// flips that a multiply by +-i or by a twiddle needs: complex add = 1, a +- i b = 1, complex
// multiply = 2.  Left to itself the compiler spends 4 instructions per complex multiply plus
// v_mov/v_xor shuffles.  The host versions (same rounding sequence) serve the CPU unit test.
#pragma once
#include <hip/hip_runtime.h>

namespace imp {

typedef float2 cf;  // complex fp32: .x = re, .y = im

// Target overloads: the __device__ versions are packed VOP3P, the __host__ ones plain C++.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f to_v(cf a) { return __builtin_bit_cast(v2f, a); }
__device__ __forceinline__ cf to_c(v2f a) { return __builtin_bit_cast(cf, a); }
// op_sel[i] / op_sel_hi[i]: which half of source i feeds the low / high result lane (0 = .x, 1 = .y)
#define IMP_PK2(op, mods)                                                     \
  v2f r;                                                                      \
  asm(op " %0, %1, %2 " mods : "=v"(r) : "v"(to_v(a)), "v"(to_v(b)));         \
  return to_c(r)
__device__ __forceinline__ cf cadd(cf a, cf b) { IMP_PK2("v_pk_add_f32", ""); }
__device__ __forceinline__ cf csub(cf a, cf b) { IMP_PK2("v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]"); }
// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf cadd_mi(cf a, cf b) { IMP_PK2("v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"); }
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf cadd_pi(cf a, cf b) { IMP_PK2("v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"); }
#undef IMP_PK2
// a * b: t = (a.y b.y, a.y b.x); r = (a.x b.x - t.x, a.x b.y + t.y)
__device__ __forceinline__ cf cmul(cf a, cf b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(to_v(a)), "v"(to_v(b)));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
      : "=v"(r) : "v"(to_v(a)), "v"(to_v(b)), "v"(t));
  return to_c(r);
}
// a * conj(b): t = (a.y b.y, a.x b.y); r = (a.x b.x + t.x, a.y b.x - t.y)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(to_v(a)), "v"(to_v(b)));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]"
      : "=v"(r) : "v"(to_v(a)), "v"(to_v(b)), "v"(t));
  return to_c(r);
}
// a + s * b with a real pair s = (s0, s1) applied component-wise
__device__ __forceinline__ cf cfma_real(cf b, cf s, cf a) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(to_v(b)), "v"(to_v(s)), "v"(to_v(a)));
  return to_c(r);
}
__device__ __forceinline__ cf cmul_real(cf b, cf s) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(to_v(b)), "v"(to_v(s)));
  return to_c(r);
}
__host__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__host__ __forceinline__ cf cadd_mi(cf a, cf b) { return make_float2(a.x + b.y, a.y - b.x); }
__host__ __forceinline__ cf cadd_pi(cf a, cf b) { return make_float2(a.x - b.y, a.y + b.x); }
__host__ __forceinline__ cf cmul(cf a, cf b) {
  return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__host__ __forceinline__ cf cmulc(cf a, cf b) {
  return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}
__host__ __forceinline__ cf cfma_real(cf b, cf s, cf a) { return make_float2(fmaf(b.x, s.x, a.x), fmaf(b.y, s.y, a.y)); }
__host__ __forceinline__ cf cmul_real(cf b, cf s) { return make_float2(b.x * s.x, b.y * s.y); }
__host__ __device__ __forceinline__ cf cconj(cf a) { return make_float2(a.x, -a.y); }
// a -+ i b: forward (DIR < 0) rotates by -i, inverse by +i
template <int DIR>
__host__ __device__ __forceinline__ cf cadd_rot(cf a, cf b) { return DIR < 0 ? cadd_mi(a, b) : cadd_pi(a, b); }
template <int DIR>
__host__ __device__ __forceinline__ cf csub_rot(cf a, cf b) { return DIR < 0 ? cadd_pi(a, b) : cadd_mi(a, b); }
// multiply by twiddle w (forward table value); DIR<0 -> a*w, DIR>0 -> a*conj(w)
template <int DIR>
__host__ __device__ __forceinline__ cf ctw(cf a, cf w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }

// DIR = -1: forward kernel exp(-2 pi i nk/N); DIR = +1: inverse (unnormalised).
template <int DIR>
__host__ __device__ __forceinline__ void bfly2(cf& a, cf& b) {
  cf t = a;
  a = cadd(t, b);
  b = csub(t, b);
}
// bfly2 on (a, w b) with w = -i (forward) / +i (inverse) folded into the adds
template <int DIR>
__host__ __device__ __forceinline__ void bfly2_rot(cf& a, cf& b) {
  cf t = a;
  a = cadd_rot<DIR>(t, b);
  b = csub_rot<DIR>(t, b);
}

template <int DIR>
__host__ __device__ __forceinline__ void bfly4(cf& a, cf& b, cf& c, cf& d) {
  const cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
  a = cadd(t0, t2);
  b = cadd_rot<DIR>(t1, t3);   // t1 -+ i t3
  c = csub(t0, t2);
  d = csub_rot<DIR>(t1, t3);
}
// bfly4 on (a, b, w c, d) with w = -i (forward) / +i (inverse) folded into the first adds
template <int DIR>
__host__ __device__ __forceinline__ void bfly4_crot(cf& a, cf& b, cf& c, cf& d) {
  const cf t0 = cadd_rot<DIR>(a, c), t1 = csub_rot<DIR>(a, c), t2 = cadd(b, d), t3 = csub(b, d);
  a = cadd(t0, t2);
  b = cadd_rot<DIR>(t1, t3);
  c = csub(t0, t2);
  d = csub_rot<DIR>(t1, t3);
}

// multiply by exp(DIR * 2 pi i * m / 16) with compile-time m (m = 4, 12 are folded into butterflies
// by the callers; the values below are the forward twiddle cr - i sf, conjugated for the inverse)
template <int DIR, int M>
__host__ __device__ __forceinline__ cf mul_w16(cf a) {
  constexpr float C1 = 0.92387953251128675613f;  // cos(pi/8)
  constexpr float S1 = 0.38268343236508977173f;  // sin(pi/8)
  constexpr float R = 0.70710678118654752440f;   // sqrt(1/2)
  constexpr int m = M & 15;
  static_assert(m != 4 && m != 8 && m != 12, "rotations by +-i and -1 belong in the butterfly");
  constexpr float cr = (m == 0) ? 1.f : (m == 1) ? C1 : (m == 2) ? R : (m == 3) ? S1
                     : (m == 5) ? -S1 : (m == 6) ? -R : (m == 7) ? -C1
                     : (m == 9) ? -C1 : (m == 10) ? -R : (m == 11) ? -S1
                     : (m == 13) ? S1 : (m == 14) ? R : C1;
  constexpr float sf = (m == 0) ? 0.f : (m == 1) ? S1 : (m == 2) ? R : (m == 3) ? C1
                     : (m == 5) ? C1 : (m == 6) ? R : (m == 7) ? S1
                     : (m == 9) ? -S1 : (m == 10) ? -R : (m == 11) ? -C1
                     : (m == 13) ? -C1 : (m == 14) ? -R : -S1;
  if constexpr (m == 0) return a;
  else return ctw<DIR>(a, make_float2(cr, -sf));
}

// 16-point DFT, natural order in / natural order out.
template <int DIR>
__host__ __device__ __forceinline__ void fft16(cf (&v)[16]) {
  // n = 4*n1 + n2 ; k = k1 + 4*k2
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) bfly4<DIR>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  // now v[4*k1 + n2] = A[n2][k1]; twiddle w16^(n2*k1); w16^4 = -+i goes into the butterfly of k1 = 2
  v[4 * 1 + 1] = mul_w16<DIR, 1>(v[4 * 1 + 1]);
  v[4 * 1 + 2] = mul_w16<DIR, 2>(v[4 * 1 + 2]);
  v[4 * 1 + 3] = mul_w16<DIR, 3>(v[4 * 1 + 3]);
  v[4 * 2 + 1] = mul_w16<DIR, 2>(v[4 * 2 + 1]);
  v[4 * 2 + 3] = mul_w16<DIR, 6>(v[4 * 2 + 3]);
  v[4 * 3 + 1] = mul_w16<DIR, 3>(v[4 * 3 + 1]);
  v[4 * 3 + 2] = mul_w16<DIR, 6>(v[4 * 3 + 2]);
  v[4 * 3 + 3] = mul_w16<DIR, 9>(v[4 * 3 + 3]);
  bfly4<DIR>(v[0], v[1], v[2], v[3]);
  bfly4<DIR>(v[4], v[5], v[6], v[7]);
  bfly4_crot<DIR>(v[8], v[9], v[10], v[11]);
  bfly4<DIR>(v[12], v[13], v[14], v[15]);
  // v[4*k1 + k2] = X[k1 + 4*k2] -> transpose the 4x4 index grid
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      cf t = v[4 * a + b];
      v[4 * a + b] = v[4 * b + a];
      v[4 * b + a] = t;
    }
}

// 8-point DFT on v[0..7] (natural in / natural out).
template <int DIR>
__host__ __device__ __forceinline__ void fft8(cf& x0, cf& x1, cf& x2, cf& x3, cf& x4, cf& x5, cf& x6, cf& x7) {
  // n = 2*n1 + n2 ; k = k1 + 4*k2
  bfly4<DIR>(x0, x2, x4, x6);   // n2 = 0 -> A[0][k1] in x0,x2,x4,x6
  bfly4<DIR>(x1, x3, x5, x7);   // n2 = 1 -> A[1][k1] in x1,x3,x5,x7
  x3 = mul_w16<DIR, 2>(x3);     // w8^1
  x7 = mul_w16<DIR, 6>(x7);     // w8^3
  bfly2<DIR>(x0, x1);           // k1=0: X[0], X[4]
  bfly2<DIR>(x2, x3);           // k1=1: X[1], X[5]
  bfly2_rot<DIR>(x4, x5);       // k1=2: X[2], X[6] with w8^2 = -+i folded in
  bfly2<DIR>(x6, x7);           // k1=3: X[3], X[7]
  // currently: x0=X0 x1=X4 x2=X1 x3=X5 x4=X2 x5=X6 x6=X3 x7=X7
  cf t1 = x1, t2 = x2, t3 = x3, t4 = x4, t5 = x5, t6 = x6;
  x1 = t2; x2 = t4; x3 = t6; x4 = t1; x5 = t3; x6 = t5;
}

// 11-point DFT (prime): pairs s_j = x_j + x_{11-j}, d_j = x_j - x_{11-j}; X_k / X_{11-k} = a_k -+ i b_k with
// a_k = x_0 + sum_j cos(2 pi j k / 11) s_j and b_k = sum_j sin(2 pi j k / 11) d_j (forward; the inverse swaps the
// signs).  Serves the 66-row column pass (66 = 11 x 6): the 7.1 / 6.15 s configuration needs 65.8 rows.
template <int DIR>
__host__ __device__ __forceinline__ void fft11(cf* x) {
  constexpr float C[11] = {1.00000000000000000000f, 0.84125353283118120551f, 0.41541501300188643508f, -0.14231483827328500480f, -0.65486073394528498959f, -0.95949297361449736865f, -0.95949297361449747967f, -0.65486073394528521163f, -0.14231483827328522684f, 0.41541501300188604651f, 0.84125353283118120551f};
  constexpr float S[11] = {0.00000000000000000000f, 0.54064081745559755543f, 0.90963199535451833011f, 0.98982144188093279524f, 0.75574957435425826890f, 0.28173255684142967104f, -0.28173255684142939348f, -0.75574957435425815788f, -0.98982144188093268422f, -0.90963199535451855215f, -0.54064081745559744441f};
  cf s[5], d[5];
#pragma unroll
  for (int j = 1; j <= 5; ++j) {
    s[j - 1] = cadd(x[j], x[11 - j]);
    d[j - 1] = csub(x[j], x[11 - j]);
  }
  const cf x0 = x[0];
  cf sum = x0;
#pragma unroll
  for (int j = 0; j < 5; ++j) sum = cadd(sum, s[j]);
  x[0] = sum;
#pragma unroll
  for (int k = 1; k <= 5; ++k) {
    cf a = x0, b = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 1; j <= 5; ++j) {
      const int m = (j * k) % 11;
      a = cfma_real(s[j - 1], make_float2(C[m], C[m]), a);
      b = (j == 1) ? cmul_real(d[0], make_float2(S[m], S[m])) : cfma_real(d[j - 1], make_float2(S[m], S[m]), b);
    }
    x[k] = cadd_rot<DIR>(a, b);          // a -+ i b
    x[11 - k] = csub_rot<DIR>(a, b);
  }
}

// first stage of the column passes: F-point DFT of the thread's F rows
template <int DIR, int F>
__host__ __device__ __forceinline__ void fft_first(cf (&v)[F]) {
  static_assert(F == 16 || F == 8 || F == 11, "first stage holds 16, 11 or 8 rows per thread");
  if constexpr (F == 16) fft16<DIR>(v);
  else if constexpr (F == 11) fft11<DIR>(v);
  else fft8<DIR>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// R-point DFT applied to the G = 16/R independent groups v[i*R .. i*R+R-1].
template <int DIR, int R>
__host__ __device__ __forceinline__ void fft_groups(cf (&v)[16]) {
  if constexpr (R == 16) {
    fft16<DIR>(v);
  } else if constexpr (R == 8) {
    fft8<DIR>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    fft8<DIR>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
  } else if constexpr (R == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bfly4<DIR>(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else if constexpr (R == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bfly2<DIR>(v[2 * i], v[2 * i + 1]);
  }
}


// ---------------------------------------------------------------------------------------------
// Odd radices for the mixed column lengths N1 = F*R2, R2 in {3, 5, 6, 9, 10, 12}.
// 6, 10 and 12 use the prime-factor (Good-Thomas) mapping: no twiddles between the two factors.
// ---------------------------------------------------------------------------------------------
template <int DIR>
__host__ __device__ __forceinline__ void bfly3(cf& a, cf& b, cf& c) {
  constexpr float S = 0.86602540378443864676f;      // sin(2 pi / 3)
  const cf t1 = cadd(b, c);
  const cf m1 = cfma_real(t1, make_float2(-0.5f, -0.5f), a);
  const cf d = cmul_real(csub(b, c), make_float2(S, S));
  a = cadd(a, t1);
  b = cadd_rot<DIR>(m1, d);    // m1 -+ i S d
  c = csub_rot<DIR>(m1, d);
}

template <int DIR>
__host__ __device__ __forceinline__ void bfly5(cf& a, cf& b, cf& c, cf& d, cf& e) {
  constexpr float C1 = 0.30901699437494742410f;     // cos(2 pi / 5)
  constexpr float C2 = -0.80901699437494742410f;    // cos(4 pi / 5)
  constexpr float S1 = 0.95105651629515357212f;     // sin(2 pi / 5)
  constexpr float S2 = 0.58778525229247312917f;     // sin(4 pi / 5)
  const cf c1 = make_float2(C1, C1), c2 = make_float2(C2, C2), s1 = make_float2(S1, S1), s2 = make_float2(S2, S2);
  const cf t1 = cadd(b, e), t2 = cadd(c, d), t3 = csub(b, e), t4 = csub(c, d);
  const cf m1 = cfma_real(t2, c2, cfma_real(t1, c1, a));
  const cf m2 = cfma_real(t2, c1, cfma_real(t1, c2, a));
  const cf n1 = cfma_real(t4, s2, cmul_real(t3, s1));
  const cf n2 = cfma_real(t4, make_float2(-S1, -S1), cmul_real(t3, s2));
  // forward: X1 = m1 - i n1, X4 = m1 + i n1, X2 = m2 - i n2, X3 = m2 + i n2 ; inverse: signs swapped
  a = cadd(a, cadd(t1, t2));
  b = cadd_rot<DIR>(m1, n1);
  e = csub_rot<DIR>(m1, n1);
  c = cadd_rot<DIR>(m2, n2);
  d = csub_rot<DIR>(m2, n2);
}


// R-point DFT of x[0..R-1] in place, natural order in and out, R in {3, 5, 6, 9, 10, 11, 12, 18, 24}.
template <int DIR, int R>
__host__ __device__ __forceinline__ void fft_small(cf* x) {
  if constexpr (R == 24) {
    // N1 = 8, N2 = 3 (prime-factor mapping): n = (3 n1 + 8 n2) mod 24 ; k = (9 k1 + 16 k2) mod 24
    cf a[8][3];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
#pragma unroll
      for (int n2 = 0; n2 < 3; ++n2) a[n1][n2] = x[(3 * n1 + 8 * n2) % 24];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) bfly3<DIR>(a[n1][0], a[n1][1], a[n1][2]);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) fft8<DIR>(a[0][k2], a[1][k2], a[2][k2], a[3][k2], a[4][k2], a[5][k2], a[6][k2], a[7][k2]);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) x[(9 * k1 + 16 * k2) % 24] = a[k1][k2];
  } else if constexpr (R == 18) {
    // N1 = 2, N2 = 9 (prime-factor mapping, no twiddles between the factors): n = (9 n1 + 2 n2) mod 18 ; k = (9 k1 + 10 k2) mod 18
    cf a[2][9];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1)
#pragma unroll
      for (int n2 = 0; n2 < 9; ++n2) a[n1][n2] = x[(9 * n1 + 2 * n2) % 18];
    fft_small<DIR, 9>(a[0]);
    fft_small<DIR, 9>(a[1]);
#pragma unroll
    for (int k2 = 0; k2 < 9; ++k2) bfly2<DIR>(a[0][k2], a[1][k2]);
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 9; ++k2) x[(9 * k1 + 10 * k2) % 18] = a[k1][k2];
  } else if constexpr (R == 3) {
    bfly3<DIR>(x[0], x[1], x[2]);
  } else if constexpr (R == 11) {
    fft11<DIR>(x);
  } else if constexpr (R == 9) {
    // n = 3 n1 + n2 ; k = k1 + 3 k2 : three radix-3 over n1, twiddle w9^(n2 k1), three radix-3 over n2
    constexpr float C1 = 0.76604444311897803520f, S1 = 0.64278760968653932632f;    // 2 pi / 9
    constexpr float C2 = 0.17364817766693034885f, S2 = 0.98480775301220805937f;    // 4 pi / 9
    constexpr float C4 = -0.93969262078590838405f, S4 = 0.34202014332566873304f;   // 8 pi / 9
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) bfly3<DIR>(x[n2], x[3 + n2], x[6 + n2]);         // x[3 k1 + n2] = A[n2][k1]
    x[3 * 1 + 1] = ctw<DIR>(x[3 * 1 + 1], make_float2(C1, -S1));
    x[3 * 1 + 2] = ctw<DIR>(x[3 * 1 + 2], make_float2(C2, -S2));
    x[3 * 2 + 1] = ctw<DIR>(x[3 * 2 + 1], make_float2(C2, -S2));
    x[3 * 2 + 2] = ctw<DIR>(x[3 * 2 + 2], make_float2(C4, -S4));
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) bfly3<DIR>(x[3 * k1], x[3 * k1 + 1], x[3 * k1 + 2]);   // x[3 k1 + k2] = X[k1 + 3 k2]
    cf t;
    t = x[1]; x[1] = x[3]; x[3] = t;
    t = x[2]; x[2] = x[6]; x[6] = t;
    t = x[5]; x[5] = x[7]; x[7] = t;
  } else if constexpr (R == 5) {
    bfly5<DIR>(x[0], x[1], x[2], x[3], x[4]);
  } else if constexpr (R == 6) {
    // N1 = 2, N2 = 3: n = (3 n1 + 2 n2) mod 6 ; k = (3 k1 + 4 k2) mod 6
    cf a[2][3];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1)
#pragma unroll
      for (int n2 = 0; n2 < 3; ++n2) a[n1][n2] = x[(3 * n1 + 2 * n2) % 6];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1) bfly3<DIR>(a[n1][0], a[n1][1], a[n1][2]);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) bfly2<DIR>(a[0][k2], a[1][k2]);
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) x[(3 * k1 + 4 * k2) % 6] = a[k1][k2];
  } else if constexpr (R == 10) {
    // N1 = 2, N2 = 5: n = (5 n1 + 2 n2) mod 10 ; k = (5 k1 + 6 k2) mod 10
    cf a[2][5];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1)
#pragma unroll
      for (int n2 = 0; n2 < 5; ++n2) a[n1][n2] = x[(5 * n1 + 2 * n2) % 10];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1) bfly5<DIR>(a[n1][0], a[n1][1], a[n1][2], a[n1][3], a[n1][4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) bfly2<DIR>(a[0][k2], a[1][k2]);
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) x[(5 * k1 + 6 * k2) % 10] = a[k1][k2];
  } else if constexpr (R == 12) {
    // N1 = 4, N2 = 3: n = (3 n1 + 4 n2) mod 12 ; k = (9 k1 + 4 k2) mod 12
    cf a[4][3];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1)
#pragma unroll
      for (int n2 = 0; n2 < 3; ++n2) a[n1][n2] = x[(3 * n1 + 4 * n2) % 12];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) bfly3<DIR>(a[n1][0], a[n1][1], a[n1][2]);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) bfly4<DIR>(a[0][k2], a[1][k2], a[2][k2], a[3][k2]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) x[(9 * k1 + 4 * k2) % 12] = a[k1][k2];
  }
}

}  // namespace imp
