// Fabric-rate probe, second take: how fast can ANY kernel move bytes across the L2<->fabric boundary for the row pass's
// traffic (in-place read-modify-write of an Infinity-Cache-resident workspace) when it keeps U 16-byte loads per thread
// in flight?  Read-only, write-only and in-place r+w, footprints inside and outside the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe2.hip -o tools/probes/stream_probe2
#include <hip/hip_runtime.h>
#include <cstdio>

template <int U>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ p, size_t n, float* sink) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
  }
  if (acc == 12345.678f) *sink = acc;
}
template <int U>
__global__ __launch_bounds__(256) void wr(float4* __restrict__ p, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) p[i + u * stride] = make_float4(1.f, 2.f, 3.f, (float)i);
  }
}
template <int U>
__global__ __launch_bounds__(256) void rmw(float4* __restrict__ p, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u].x += 1.f;
      p[i + u * stride] = v[u];
    }
  }
}
// read A, write B (the column passes: different buffers)
template <int U>
__global__ __launch_bounds__(256) void cp(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
  }
}

template <class F>
static double run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) launch();
  hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms / reps * 1e-3;
}

template <int U>
static void sweep(float4* a, float4* b, float* sink) {
  const size_t MB = 1 << 20;
  for (int grid : {1024, 2048, 4096, 8192}) {
    for (size_t mb : {70, 140, 1024}) {
      const size_t n = mb * MB / 16;
      const double tr = run([&] { hipLaunchKernelGGL(rd<U>, grid, 256, 0, 0, a, n, sink); });
      const double tw = run([&] { hipLaunchKernelGGL(wr<U>, grid, 256, 0, 0, a, n); });
      const double tm = run([&] { hipLaunchKernelGGL(rmw<U>, grid, 256, 0, 0, a, n); });
      const double tc = run([&] { hipLaunchKernelGGL(cp<U>, grid, 256, 0, 0, a, b, n); });
      const double by = (double)mb * MB / 1e12;
      printf("U=%d grid %5d  %5zu MB: read %5.2f  write %5.2f  in-place r+w %5.2f  copy a->b %5.2f  TB/s (r+w and copy count both directions)\n", U,
             grid, mb, by / tr, by / tw, 2 * by / tm, 2 * by / tc);
      fflush(stdout);
    }
  }
}

int main() {
  const size_t MB = 1 << 20;
  float4 *a, *b;
  float* sink;
  hipMalloc(&a, 1024 * MB);
  hipMalloc(&b, 1024 * MB);
  hipMalloc(&sink, 4);
  hipMemset(a, 0, 1024 * MB);
  hipMemset(b, 0, 1024 * MB);
  sweep<1>(a, b, sink);
  sweep<4>(a, b, sink);
  sweep<8>(a, b, sink);
  return 0;
}
