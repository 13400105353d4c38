// microbenchmark: random 16-byte gathers through the SCALAR cache (s_load_dwordx4, address from v_readlane) -- a second path beside
// the vector memory pipeline (260 G lane-gathers/s chip-wide from a 2 MB table per XCD, tools/ubench/gather_policy.hip, whatever the policy bits and the width).
//   MODE 0: vector gathers only (16 per lane in flight), MODE 1: scalar gathers only (one s_load per lane of the wave's 64 addresses),
//   MODE 2: both in the same wave (16 vector gathers per lane + 16 scalar gathers per wave, interleaved)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const v4u* __restrict__ table, uint32_t quads_per_xcd, uint32_t iters, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const v4u* __restrict__ t = table + (size_t)(blockIdx.x & 7) * quads_per_xcd;
  const uint32_t m = quads_per_xcd - 1;
  uint32_t acc = 0, sacc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    if (MODE != 1) {
      v4u v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = t[hash32(tid * 977u + (it * 16 + e) * 0x9e3779b9u) & m];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc += v[e].x + v[e].w;
    }
    if (MODE != 0) {
      const uint32_t h = hash32(tid * 31u + it * 0x85ebca6bu) & m;
      constexpr int NS = MODE == 1 ? 64 : 16;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const uint32_t off = __builtin_amdgcn_readlane(h, i);
        const v4u s = t[off]; // uniform address, read-only memory: a scalar load
        sacc += s.x + s.w;
      }
    }
  }
  if (acc + sacc == 0x12345678u) out[tid] = acc;
}

template <int MODE>
void run(const char* name, uint32_t log2_quads_per_xcd, uint32_t blocks_per_cu) {
  const uint32_t per = 1u << log2_quads_per_xcd;
  v4u* table; uint32_t* out; CHECK(hipMalloc(&table, (size_t)per * 8 * 16)); CHECK(hipMalloc(&out, 1 << 24));
  CHECK(hipMemset(table, 1, (size_t)per * 8 * 16));
  const uint32_t blocks = 256 * blocks_per_cu, threads = 256, iters = 16;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(threads), 0, 0, table, per, iters, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  const double vec = MODE == 1 ? 0.0 : (double)blocks * threads * iters * 16;
  const double sca = MODE == 0 ? 0.0 : (double)blocks * (threads / 64) * iters * (MODE == 1 ? 64 : 16);
  printf("%-8s table 2^%-2u x16B per XCD, blocks/CU=%u: %8.1f us  vector %7.1f G/s  scalar %7.1f G/s\n", name, log2_quads_per_xcd, blocks_per_cu, best * 1e3, vec / best / 1e6, sca / best / 1e6);
  CHECK(hipFree(table)); CHECK(hipFree(out));
}

int main() {
  for (uint32_t lg : {17u, 14u, 10u}) { // 2 MB, 256 KB, 16 KB per XCD
    for (uint32_t bpc : {4u, 8u}) {
      run<0>("vector", lg, bpc);
      run<1>("scalar", lg, bpc);
      run<2>("both", lg, bpc);
    }
  }
  return 0;
}
