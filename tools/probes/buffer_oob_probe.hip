// Probe of raw-buffer (SRSRC) range checking on gfx950: which of voffset / soffset take part in the
// bounds check, what a partially out-of-range dwordx2 returns, and whether out-of-range stores are dropped.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_oob_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* src, unsigned n_floats, float* dst, unsigned dst_floats, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_floats * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dst_floats * 4, 0x00020000);
  const unsigned t = threadIdx.x;
  // case 0: voffset in range, soffset pushes the address past num_records
  out[0 * 64 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, t * 4, (n_floats) * 4, 0));
  // case 1: voffset out of range by itself
  out[1 * 64 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (n_floats + t) * 4, 0, 0));
  // case 2: dwordx2 straddling the end: first dword valid, second not (element n_floats-1)
  u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, (n_floats - 1) * 4, 0, 0);
  out[2 * 64 + 0] = __uint_as_float(x.x);
  out[2 * 64 + 1] = __uint_as_float(x.y);
  // case 3: "negative" voffset (wraps to a huge unsigned)
  out[3 * 64 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)(-(int)(t + 1) * 4), 0, 0));
  // case 4: negative voffset + positive soffset that brings the sum back in range
  out[4 * 64 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)(-(int)(t + 1) * 4), 256 * 4, 0));
  // case 5: in-range voffset + in-range soffset (sanity)
  out[5 * 64 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, t * 4, 64 * 4, 0));
  // stores: in range, out of range by voffset, out of range through soffset, negative voffset
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(100.f + t), w, t * 4, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(200.f + t), w, (dst_floats + t) * 4, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(300.f + t), w, t * 4, dst_floats * 4, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(400.f + t), w, (unsigned)(-(int)(t + 1) * 4), 0, 0);
  // dwordx2 store straddling the end of the range
  if (t == 0) {
    u32x2 y;
    y.x = __float_as_uint(500.f);
    y.y = __float_as_uint(501.f);
    __builtin_amdgcn_raw_buffer_store_b64(y, w, (dst_floats - 1) * 4, 0, 0);
  }
}

int main() {
  const unsigned n = 1024, guard = 1024;
  std::vector<float> h(n + guard);
  for (unsigned i = 0; i < n + guard; ++i) h[i] = 1000.f + i;
  float *src, *dst, *out;
  hipMalloc(&src, (n + guard) * 4);
  hipMalloc(&dst, 3 * guard * 4);
  hipMalloc(&out, 6 * 64 * 4);
  hipMemcpy(src, h.data(), (n + guard) * 4, hipMemcpyHostToDevice);
  hipMemset(dst, 0, 3 * guard * 4);
  hipMemset(out, 0, 6 * 64 * 4);
  // the store window sits in the middle third of dst so that "negative" stores would land in the first third
  hipLaunchKernelGGL(probe, 1, 64, 0, 0, src, n, dst + guard, guard, out);
  hipDeviceSynchronize();
  std::vector<float> o(6 * 64), d(3 * guard);
  hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(d.data(), dst, d.size() * 4, hipMemcpyDeviceToHost);
  printf("case0 voff in range + soffset past end : lane0 -> %.0f (0 = soffset IS range checked, %.0f = not)\n", o[0], h[n]);
  printf("case1 voffset past end                 : lane0 -> %.0f (expect 0)\n", o[64]);
  printf("case2 dwordx2 straddling end           : %.0f %.0f (valid first = %.0f)\n", o[128], o[129], h[n - 1]);
  printf("case3 negative voffset                 : lane0 -> %.0f (expect 0)\n", o[192]);
  printf("case4 negative voffset + soffset 1024B : lane0 -> %.0f (in-range sum would be %.0f)\n", o[256], h[255]);
  printf("case5 sanity                           : lane0 -> %.0f (expect %.0f)\n", o[320], h[64]);
  int in_ok = 0, below = 0, above = 0;
  for (unsigned i = 0; i < guard; ++i) {
    below += d[i] != 0.f;
    above += d[2 * guard + i] != 0.f;
  }
  for (unsigned t = 0; t < 64; ++t) in_ok += d[guard + t] == 100.f + t;
  printf("stores: %d/64 in-range landed; %d words written below the window; %d above; last word of window = %.0f, first word above = %.0f\n",
         in_ok, below, above, d[2 * guard - 1], d[2 * guard]);
  return 0;
}
