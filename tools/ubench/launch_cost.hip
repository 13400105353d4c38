// microbenchmark: what does a launch cost outside its waves?  Kernels that do (almost) nothing, in the launch shapes of the step's kernels;
// durations from `rocprofv3 --kernel-trace --stats` (begin -> end of the dispatch), back to back on one stream.
//   k_trivial           256 x 512 threads, no LDS
//   k_lds150            256 x 512 threads, 150 KB of dynamic LDS (k_mlp_train_r32's shape)
//   k_regs256           256 x 512 threads, 150 KB LDS, 256 VGPRs per wave
//   k_dirty<MB>         256 x 512 threads writing MB megabytes with plain stores just before they end (what the successor's boundary pays)
//   k_blocks960         960 x 512 threads, 66 KB LDS (k_grid_scatter_lists' shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k_trivial(uint32_t* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void __launch_bounds__(512) k_lds150(uint32_t* p) { extern __shared__ uint32_t s[]; if (p && threadIdx.x == 9999) p[0] = s[threadIdx.x]; }
__global__ void __launch_bounds__(512) k_blocks960(uint32_t* p) { extern __shared__ uint32_t s[]; if (p && threadIdx.x == 9999) p[0] = s[threadIdx.x]; }
__global__ void __launch_bounds__(512, 2) k_regs256(uint32_t* p, int never) {
  extern __shared__ uint32_t s[];
  float v[200];
#pragma unroll
  for (int i = 0; i < 200; ++i) v[i] = (float)(threadIdx.x + i);
  if (never) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 200; ++i) v[i] = v[(i + 7) % 200] * 1.0001f + v[(i + 13) % 200];
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 200; ++i) acc += v[i];
    p[threadIdx.x] = (uint32_t)acc + s[threadIdx.x];
  }
}
template <int MB> __global__ void __launch_bounds__(512) k_dirty(u4* out) {
  const size_t per_block = (size_t)MB * (1 << 20) / 16 / 256; // 16-byte elements per block
  for (size_t i = threadIdx.x; i < per_block; i += 512) out[(size_t)blockIdx.x * per_block + i] = u4{1, 2, 3, 4};
}
template <int MB> __global__ void __launch_bounds__(512) k_dirty_wt(u4* out) { // the same bytes write-through (sc1)
  const size_t per_block = (size_t)MB * (1 << 20) / 16 / 256;
  for (size_t i = threadIdx.x; i < per_block; i += 512) {
    u4* q = out + (size_t)blockIdx.x * per_block + i;
    const u4 v = u4{1, 2, 3, 4};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(q), "v"(v) : "memory");
  }
}

int main() {
  uint32_t* p; CHECK(hipMalloc(&p, 1 << 20));
  u4* big; CHECK(hipMalloc(&big, (size_t)64 << 20));
  CHECK(hipFuncSetAttribute((const void*)k_lds150, hipFuncAttributeMaxDynamicSharedMemorySize, 153600));
  CHECK(hipFuncSetAttribute((const void*)k_regs256, hipFuncAttributeMaxDynamicSharedMemorySize, 153600));
  CHECK(hipFuncSetAttribute((const void*)k_blocks960, hipFuncAttributeMaxDynamicSharedMemorySize, 67840));
  for (int rep = 0; rep < 30; ++rep) {
    hipLaunchKernelGGL(k_trivial, dim3(256), dim3(512), 0, 0, p);
    hipLaunchKernelGGL(k_lds150, dim3(256), dim3(512), 153600, 0, p);
    hipLaunchKernelGGL(k_regs256, dim3(256), dim3(512), 153600, 0, p, 0);
    hipLaunchKernelGGL(k_blocks960, dim3(960), dim3(512), 67840, 0, p);
    hipLaunchKernelGGL(k_dirty<8>, dim3(256), dim3(512), 0, 0, big);
    hipLaunchKernelGGL(k_trivial, dim3(256), dim3(512), 0, 0, p);
    hipLaunchKernelGGL(k_dirty<32>, dim3(256), dim3(512), 0, 0, big);
    hipLaunchKernelGGL(k_trivial, dim3(256), dim3(512), 0, 0, p);
    hipLaunchKernelGGL(k_dirty_wt<32>, dim3(256), dim3(512), 0, 0, big);
    hipLaunchKernelGGL(k_trivial, dim3(256), dim3(512), 0, 0, p);
  }
  CHECK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
