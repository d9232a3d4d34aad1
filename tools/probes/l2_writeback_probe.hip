// Does the XCD L2 of gfx950 hold dirty lines (write-back) when a kernel rewrites a small footprint, or do all store
// bytes go out to the fabric (write-through)?  256 workgroups, each rewrites its own 64 KiB region R times
// (footprint 16 MiB over 8 XCDs = 2 MiB per 4 MiB L2).  Run under rocprofv3 --pmc WRITE_SIZE for R = 1 and R = 16:
// write-back -> WRITE_SIZE stays near 16 MiB; write-through -> R x 16 MiB.
//   hipcc --offload-arch=gfx950 -O3 l2_writeback_probe.hip -o l2_writeback_probe && ./l2_writeback_probe R mode
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int AUX>
__global__ __launch_bounds__(256) void rewrite(float4* buf, int reps, int read_back) {
  float4* mine = buf + (size_t)blockIdx.x * 4096;            // 64 KiB
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 65536, 0x00020000);
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  float acc = 0.f;
  for (int k = 0; k < reps; ++k) {
    for (int i = 0; i < 16; ++i) {
      u4 v = {(unsigned)k, (unsigned)i, threadIdx.x, blockIdx.x};
      __builtin_amdgcn_raw_buffer_store_b128(v, r, (threadIdx.x + 256 * i) * 16, 0, AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (read_back) {
      for (int i = 0; i < 16; ++i) {
        u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, ((threadIdx.x + 64) % 256 + 256 * i) * 16, 0, 16);   // sc1: from L2
        acc += (float)v.x;
      }
      __syncthreads();
    }
  }
  if (acc == 12345.678f) mine[0].x = acc;
}

int main(int argc, char** argv) {
  int reps = argc > 1 ? atoi(argv[1]) : 16;
  int mode = argc > 2 ? atoi(argv[2]) : 0;       // 0 plain, 1 nt, 2 sc0
  int read_back = argc > 3 ? atoi(argv[3]) : 0;
  float4* d;
  hipMalloc(&d, 256 * 65536);
  hipMemset(d, 0, 256 * 65536);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(a);
    if (mode == 0) rewrite<0><<<256, 256>>>(d, reps, read_back);
    else if (mode == 1) rewrite<2><<<256, 256>>>(d, reps, read_back);
    else rewrite<1><<<256, 256>>>(d, reps, read_back);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("reps %d mode %d read_back %d: %.1f us, %.1f GB/s of stores\n", reps, mode, read_back, ms * 1e3,
           256.0 * 65536 * reps / ms / 1e6);
  }
  hipFree(d);
  return 0;
}
