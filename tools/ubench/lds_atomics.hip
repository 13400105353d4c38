// LDS atomic throughput by flavour on gfx950: 1024-thread WG, 128 KB table, random addresses
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(1024) k(uint32_t* out, uint32_t iters, uint32_t mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* t32 = (uint32_t*)smem;
  for (uint32_t i = threadIdx.x; i < 32768; i += 1024) t32[i] = 0;
  __syncthreads();
  uint32_t tid = blockIdx.x * 1024 + threadIdx.x;
  for (uint32_t e = 0; e < iters; ++e) {
    uint32_t h = hash32(tid * 977 + e * 0x9e3779b9u);
    if (MODE == 0) { __half2 v; v.x = __float2half(1.0f); v.y = __float2half(0.5f); unsafeAtomicAdd((__half2*)smem + (h & mask), v); }
    else if (MODE == 1) unsafeAtomicAdd((float*)smem + (h & mask), 1.0f);
    else if (MODE == 2) atomicAdd(t32 + (h & mask), 1u);
    else if (MODE == 3) atomicAdd((unsigned long long*)smem + (h & (mask >> 1)), 0x100000001ull);
    else if (MODE == 4) { uint32_t i = h & mask; t32[i] = t32[i] + 1; }   // non-atomic RMW (wrong under conflicts; rate reference)
  }
  __syncthreads();
  uint32_t s = 0;
  for (uint32_t i = threadIdx.x; i < 32768; i += 1024) s += t32[i];
  if (s == 0xdeadbeef) out[0] = s;
}

template <int MODE> void run(const char* name, uint32_t mask) {
  CHECK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  uint32_t* out; CHECK(hipMalloc(&out, 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const uint32_t blocks = 256, iters = 256;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 131072, 0, out, iters, mask);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  double ops = (double)blocks * 1024 * iters;
  printf("%-22s mask %6x: %8.1f us  %7.2f G lane-ops/s chip  = %.3f lane-ops/clk/CU @2.4GHz\n", name, mask, best * 1e3, ops / best / 1e6, ops / (best * 1e-3) / 256 / 2.4e9);
  CHECK(hipFree(out));
}

int main() {
  for (uint32_t mask : {0x7fffu, 0xffu}) {
    run<0>("ds_pk_add_f16", mask);
    run<1>("ds_add_f32", mask);
    run<2>("ds_add_u32", mask);
    run<3>("ds_add_u64", mask);
    run<4>("ds read+write (racy)", mask);
  }
}
