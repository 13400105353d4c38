// microbenchmark: random 4-byte gather throughput on gfx950 by table size / placement / instruction form
//   MODE 0: global_load_dword, 32 independent gathers in flight per lane
//   MODE 1: pairs -- the second gather reads the neighbouring dword (same line): models the (x, x+1) corners
//   MODE 2: global_load_dwordx2 of an aligned pair (one lane-gather instead of two)
//   MODE 3: table staged in LDS once per workgroup (table <= 64 KB), ds_read_b32 gathers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const uint32_t* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t xcd_partition, uint32_t* __restrict__ out) {
  extern __shared__ uint32_t lds[];
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t base = 0, m = mask;
  if (xcd_partition) { m = mask >> 3; base = (blockIdx.x & 7) * (m + 1); }
  if (MODE == 3) {
    for (uint32_t i = threadIdx.x; i <= mask; i += blockDim.x) lds[i] = table[i];
    __syncthreads();
  }
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t v[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      uint32_t h = hash32(tid * 977u + (it * 32 + e) * 0x9e3779b9u);
      if (MODE == 0) v[e] = table[base + (h & m)];
      else if (MODE == 1) { uint32_t hh = hash32(tid * 977u + (it * 32 + (e & ~1)) * 0x9e3779b9u); v[e] = table[base + (((hh & m) & ~1u) | (e & 1))]; }
      else if (MODE == 2) { if (e & 1) { v[e] = 0; } else { uint2 p = *(const uint2*)&table[base + ((h & m) & ~1u)]; v[e] = p.x + p.y; } }
      else v[e] = lds[h & mask];
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) acc += v[e];
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

template <int MODE>
void run(const char* name, uint32_t log2_entries, uint32_t part, uint32_t blocks_per_cu) {
  const uint32_t n_entries = 1u << log2_entries;
  uint32_t *table, *out; CHECK(hipMalloc(&table, n_entries * 4)); CHECK(hipMalloc(&out, 1 << 24));
  CHECK(hipMemset(table, 1, n_entries * 4));
  const uint32_t blocks = 256 * blocks_per_cu, threads = 256, iters = 8;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9;
  const size_t lds_bytes = MODE == 3 ? n_entries * 4 : 0;
  if (lds_bytes > 65536) CHECK(hipFuncSetAttribute((const void*)k_gather<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(threads), lds_bytes, 0, table, n_entries - 1, iters, part, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  const double events = (double)blocks * threads * iters * 32;
  printf("%-28s table 2^%-2u x4B part=%u blocks/CU=%u: %8.1f us  %7.1f G lane-gathers/s  (%.2f clk/lane/CU at 2.4 GHz)\n", name, log2_entries, part, blocks_per_cu, best * 1e3,
         events / best / 1e6, 2.4e9 / (events / (best * 1e-3) / 256));
  CHECK(hipFree(table)); CHECK(hipFree(out));
}

int main() {
  for (uint32_t bpc : {4u, 8u}) {
    run<0>("dword", 12, 0, bpc);
    run<0>("dword", 14, 0, bpc);
    run<0>("dword", 16, 0, bpc);
    run<0>("dword", 19, 0, bpc);
    run<0>("dword", 19, 1, bpc);
    run<0>("dword", 22, 0, bpc);
    run<0>("dword", 22, 1, bpc);
    run<0>("dword", 23, 0, bpc);
    run<1>("dword neighbour pairs", 14, 0, bpc);
    run<1>("dword neighbour pairs", 19, 0, bpc);
    run<1>("dword neighbour pairs", 22, 1, bpc);
    run<2>("dwordx2 aligned pair", 14, 0, bpc);
    run<2>("dwordx2 aligned pair", 19, 0, bpc);
    run<2>("dwordx2 aligned pair", 22, 1, bpc);
    run<3>("LDS-staged table", 12, 0, bpc);
    run<3>("LDS-staged table", 14, 0, bpc);
  }
  return 0;
}
