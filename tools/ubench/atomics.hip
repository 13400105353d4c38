// microbenchmark: random scatter-add throughput by atomic flavour / scope on gfx950
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: pk_f16 agent (unsafeAtomicAdd)   1: f32 agent   2: u32 agent   3: u32 workgroup scope   4: pk_f16 workgroup scope (inline asm, no sc bits)
template <int MODE>
__global__ void k_scatter(uint32_t* table, uint32_t mask, uint32_t n_events_per_thread, uint32_t xcd_partition) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t base = 0;
  uint32_t m = mask;
  if (xcd_partition) { // every XCD class (blockIdx % 8) gets its own 1/8 of the table
    m = mask >> 3;
    base = (blockIdx.x & 7) * (m + 1);
  }
  for (uint32_t e = 0; e < n_events_per_thread; ++e) {
    uint32_t idx = base + (hash32(tid * 977 + e * 0x9e3779b9u) & m);
    if (MODE == 0) { __half2 v; v.x = __float2half(1.0f); v.y = __float2half(2.0f); unsafeAtomicAdd((__half2*)table + idx, v); }
    else if (MODE == 1) unsafeAtomicAdd((float*)table + idx, 1.0f);
    else if (MODE == 2) atomicAdd(table + idx, 1u);
    else if (MODE == 3) __hip_atomic_fetch_add(table + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 4) __hip_atomic_fetch_add((float*)table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <int MODE>
void run(const char* name, uint32_t log2_entries, uint32_t part) {
  const uint32_t n_entries = 1u << log2_entries;
  uint32_t* table; CHECK(hipMalloc(&table, n_entries * 4));
  const uint32_t blocks = 2048, threads = 256, ept = 32;   // 16.8M events
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipMemset(table, 0, n_entries * 4));
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_scatter<MODE>, dim3(blocks), dim3(threads), 0, 0, table, n_entries - 1, ept, part);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  std::vector<uint32_t> h(n_entries); CHECK(hipMemcpy(h.data(), table, n_entries * 4, hipMemcpyDeviceToHost));
  double sum = 0;
  for (uint32_t i = 0; i < n_entries; ++i) {
    if (MODE == 0) { __half2 v = *(__half2*)&h[i]; sum += __half2float(v.x); }
    else if (MODE == 1 || MODE == 4) sum += *(float*)&h[i];
    else sum += h[i];
  }
  const double events = (double)blocks * threads * ept;
  printf("%-34s table 2^%u x4B part=%u: %8.1f us  %6.2f G events/s  sum/expected = %.4f\n", name, log2_entries, part, best * 1e3, events / best / 1e6, sum / events);
  CHECK(hipFree(table));
}

int main() {
  for (uint32_t lg : {19u, 23u}) {
    for (uint32_t part : {0u, 1u}) {
      run<0>("pk_add_f16 agent", lg, part);
      run<1>("add_f32 agent", lg, part);
      run<2>("add_u32 agent", lg, part);
      run<3>("add_u32 workgroup-scope", lg, part);
      run<4>("add_f32 workgroup-scope", lg, part);
    }
  }
  return 0;
}
