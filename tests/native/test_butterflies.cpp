// Host-side unit test of the in-register butterflies (csrc/fft_regs.hip.h) against a naive DFT.
// Built and run on the CPU by tests/test_host_cpu.py (hipcc host compilation; no GPU involved).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "fft_regs.hip.h"

using imp::cf;

template <int DIR>
static double check(int n, void (*run)(cf*)) {
  double worst = 0;
  for (int trial = 0; trial < 4; ++trial) {
    cf x[16], y[16];
    for (int i = 0; i < n; ++i) {
      x[i] = make_float2((float)rand() / RAND_MAX - 0.5f, (float)rand() / RAND_MAX - 0.5f);
      y[i] = x[i];
    }
    run(y);
    for (int k = 0; k < n; ++k) {
      double re = 0, im = 0;
      for (int i = 0; i < n; ++i) {
        double ang = DIR * 2.0 * M_PI * (double)((i * k) % n) / n;
        re += x[i].x * cos(ang) - x[i].y * sin(ang);
        im += x[i].x * sin(ang) + x[i].y * cos(ang);
      }
      worst = fmax(worst, fmax(fabs(re - y[k].x), fabs(im - y[k].y)));
    }
  }
  return worst;
}

template <int DIR> static void r16(cf* v) { cf (&a)[16] = *reinterpret_cast<cf(*)[16]>(v); imp::fft16<DIR>(a); }
template <int DIR> static void r8(cf* v) { imp::fft8<DIR>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]); }
template <int DIR> static void r4(cf* v) { imp::bfly4<DIR>(v[0], v[1], v[2], v[3]); }
template <int DIR> static void r2(cf* v) { imp::bfly2<DIR>(v[0], v[1]); }
template <int DIR, int R> static void rs(cf* v) { imp::fft_small<DIR, R>(v); }

int main() {
  int bad = 0;
  auto report = [&](const char* name, double e) {
    printf("%-10s max abs err %.3e\n", name, e);
    if (!(e < 5e-6)) ++bad;
  };
  report("fft16 fwd", check<-1>(16, r16<-1>));
  report("fft16 inv", check<+1>(16, r16<+1>));
  report("fft8 fwd", check<-1>(8, r8<-1>));
  report("fft8 inv", check<+1>(8, r8<+1>));
  report("fft4 fwd", check<-1>(4, r4<-1>));
  report("fft4 inv", check<+1>(4, r4<+1>));
  report("fft2", check<-1>(2, r2<-1>));
  report("fft3 fwd", check<-1>(3, rs<-1, 3>));
  report("fft3 inv", check<+1>(3, rs<+1, 3>));
  report("fft5 fwd", check<-1>(5, rs<-1, 5>));
  report("fft5 inv", check<+1>(5, rs<+1, 5>));
  report("fft6 fwd", check<-1>(6, rs<-1, 6>));
  report("fft6 inv", check<+1>(6, rs<+1, 6>));
  report("fft10 fwd", check<-1>(10, rs<-1, 10>));
  report("fft10 inv", check<+1>(10, rs<+1, 10>));
  report("fft12 fwd", check<-1>(12, rs<-1, 12>));
  report("fft12 inv", check<+1>(12, rs<+1, 12>));
  printf(bad ? "BUTTERFLIES FAIL\n" : "BUTTERFLIES OK\n");
  return bad;
}
