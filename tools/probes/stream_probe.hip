// Fabric-rate probe: what a plain streaming kernel reaches on this chip for the traffic mixes of the three
// convolution passes (in-place read-modify-write = row pass; read A / write B of different sizes = column
// passes), at footprints inside and outside the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void rmw(float4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    v.x += 1.f;
    p[i] = v;
  }
}
__global__ __launch_bounds__(256) void copy(const float4* a, float4* b, size_t na, size_t nb) {
  // reads na, writes nb (na <= nb or nb <= na): models "read 25 MB, write 42 MB" and the reverse
  const size_t n = na > nb ? na : nb;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (i < na) v = a[i];
    if (i < nb) b[i] = v;
  }
}

static double time_ms(hipEvent_t e0, hipEvent_t e1) {
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t MB = 1 << 20;
  float4 *a, *b;
  hipMalloc(&a, 1024 * MB);
  hipMalloc(&b, 1024 * MB);
  hipMemset(a, 0, 1024 * MB);
  hipMemset(b, 0, 1024 * MB);
  for (int grid : {512, 1024, 2048, 4096}) {
    printf("grid %d x 256 threads\n", grid);
    for (size_t mb : {42, 84, 126, 252, 512, 1024}) {
      const size_t n = mb * MB / 16;
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rmw, grid, 256, 0, 0, a, n);
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rmw, grid, 256, 0, 0, a, n);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      const double ms = time_ms(e0, e1) / reps;
      printf("  in-place r+w %5zu MB footprint: %7.1f us  -> %5.2f TB/s (read+write bytes)\n", mb, ms * 1e3, 2.0 * mb * MB / (ms * 1e-3) / 1e12);
    }
    for (auto pr : std::vector<std::pair<size_t, size_t>>{{25, 42}, {42, 25}, {200, 336}, {336, 200}}) {
      const size_t na = pr.first * MB / 16, nb = pr.second * MB / 16;
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(copy, grid, 256, 0, 0, a, b, na, nb);
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy, grid, 256, 0, 0, a, b, na, nb);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      const double ms = time_ms(e0, e1) / reps;
      printf("  read %3zu MB + write %3zu MB: %7.1f us  -> %5.2f TB/s\n", pr.first, pr.second, ms * 1e3,
             (pr.first + pr.second) * (double)MB / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
