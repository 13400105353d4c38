// microbenchmark: does a cache-policy bit change what a random gather costs on gfx950?
// k_grid_fwd_planes and k_grid_scatter both sit at ~2 clocks per gathered lane and CU from an L2-resident table (tools/ubench/gather.hip),
// which is what a full 128-byte line fill per lane over the 64 B/clk L2 -> L1 path would cost.  If a policy bit (sc0 / sc1 / nt) makes the
// vector cache fetch less than a line for a missing lane, the rate moves; if not, it is the address path itself.
//   table: 2^19 x 4 B per XCD (the C3a level table), every workgroup gathers from its XCD's table (blockIdx & 7)
//   forms: raw buffer loads with aux = 0 (default), 1 (sc0), 2 (nt), 3 (sc0 nt), 16 (sc1), 17 (sc0 sc1), 18 (sc1 nt), 19 (sc0 sc1 nt);
//          widths 4 / 8 / 16 bytes per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));


template <int AUX, int WIDTH>
__global__ void __launch_bounds__(256) k_gather(const uint32_t* __restrict__ table, uint32_t entries_per_xcd, uint32_t iters, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(table + (size_t)(blockIdx.x & 7) * entries_per_xcd), 0, (int)(entries_per_xcd * 4), 0x00020000);
  const uint32_t m = entries_per_xcd - 1;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const uint32_t h = hash32(tid * 977u + (it * 16 + e) * 0x9e3779b9u) & m;
      if (WIDTH == 4) v[e] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, h * 4, 0, AUX);
      else if (WIDTH == 8) { const v2u p = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (h & ~1u) * 4, 0, AUX); v[e] = p.x + p.y; }
      else { const v4u p = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (h & ~3u) * 4, 0, AUX); v[e] = p.x + p.y + p.z + p.w; }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += v[e];
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

template <int AUX, int WIDTH>
void run(const char* name, uint32_t log2_entries_per_xcd, uint32_t blocks_per_cu) {
  const uint32_t per = 1u << log2_entries_per_xcd;
  uint32_t *table, *out; CHECK(hipMalloc(&table, (size_t)per * 8 * 4)); CHECK(hipMalloc(&out, 1 << 24));
  CHECK(hipMemset(table, 1, (size_t)per * 8 * 4));
  const uint32_t blocks = 256 * blocks_per_cu, threads = 256, iters = 16;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<AUX, WIDTH>), dim3(blocks), dim3(threads), 0, 0, table, per, iters, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  const double events = (double)blocks * threads * iters * 16;
  printf("%-14s %2d B/lane  table 2^%-2u x4B per XCD, blocks/CU=%u: %8.1f us  %7.1f G lane-gathers/s\n", name, WIDTH, log2_entries_per_xcd, blocks_per_cu, best * 1e3, events / best / 1e6);
  CHECK(hipFree(table)); CHECK(hipFree(out));
}

template <int WIDTH>
void sweep(uint32_t lg, uint32_t bpc) {
  run<0, WIDTH>("default", lg, bpc);
  run<1, WIDTH>("sc0", lg, bpc);
  run<2, WIDTH>("nt", lg, bpc);
  run<3, WIDTH>("sc0 nt", lg, bpc);
  run<16, WIDTH>("sc1", lg, bpc);
  run<17, WIDTH>("sc0 sc1", lg, bpc);
  run<18, WIDTH>("sc1 nt", lg, bpc);
  run<19, WIDTH>("sc0 sc1 nt", lg, bpc);
}

int main() {
  for (uint32_t lg : {19u, 13u, 22u}) {
    sweep<4>(lg, 8);
    sweep<8>(lg, 8);
    sweep<16>(lg, 8);
  }
  return 0;
}
