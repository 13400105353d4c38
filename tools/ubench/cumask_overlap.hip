// microbenchmark: do a gather-bound kernel and an Adam-shaped streaming kernel gain by sharing the chip on disjoint
// CU sets?  Round 4's form compared a 1 ms gather launch with an 82 us stream launch and could not say.  Here BOTH kernels run for the same
// fixed wall time (every workgroup loops until the device-wide 100 MHz clock says stop) and count what they got done, so the two rates are
// measured over the same interval:
//     combined = gather rate together / gather rate alone on the whole chip  +  stream rate together / stream rate alone on the whole chip
// for 24/8, 28/4 and 30/2 CUs per XCD, the stream's loads default and nontemporal (does the stream push the gathers' 4 MB table out of the L2?).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/cumask_overlap.hip -o /tmp/cumask_overlap && /tmp/cumask_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// random 16-byte gathers from the 4 MB slice of the XCD the workgroup runs on, 16 in flight per lane, until `ticks` of the 100 MHz clock have passed
__global__ void __launch_bounds__(512, 2) k_gather(const u4* __restrict__ table, uint32_t slice_mask, unsigned long long ticks, unsigned long long* __restrict__ done, uint32_t* __restrict__ out) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg(63508) & 7;
  const u4* t = table + (size_t)xcc * (slice_mask + 1);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0, it = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    u4 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = t[hash32(tid * 977u + (it * 16 + e) * 0x9e3779b9u) & slice_mask];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += v[e][0] + v[e][3];
    ++it;
  }
  if (threadIdx.x == 0) atomicAdd(done, (unsigned long long)it * blockDim.x * 16); // lane-gathers of this workgroup (every wave makes the same number of turns, give or take one)
  if (acc == 0x12345678u) out[tid] = acc;
}

// the optimizer's traffic shape (fp32 master, two fp32 moments read and written, half gradient read, half weight written: 34 bytes per parameter)
template <bool NT>
__global__ void __launch_bounds__(256) k_stream(size_t n, float* __restrict__ w_fp, _Float16* __restrict__ w, const _Float16* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                                                uint32_t blocks_total, unsigned long long ticks, unsigned long long* __restrict__ done) {
  constexpr int Q = 2;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long turns = 0;
  for (size_t blk = blockIdx.x; __builtin_amdgcn_s_memrealtime() - t0 < ticks; blk = (blk + gridDim.x) % blocks_total, ++turns) {
    const size_t base = blk * 256 * 8;
    h4 gv[Q]; f4 wf[Q], a1[Q], a2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const size_t i = base + ((size_t)q * 256 + threadIdx.x) * 4;
      if constexpr (NT) { gv[q] = __builtin_nontemporal_load((const h4*)(g + i)); wf[q] = __builtin_nontemporal_load((const f4*)(w_fp + i)); a1[q] = __builtin_nontemporal_load((const f4*)(m1 + i)); a2[q] = __builtin_nontemporal_load((const f4*)(m2 + i)); }
      else { gv[q] = *(const h4*)(g + i); wf[q] = *(const f4*)(w_fp + i); a1[q] = *(const f4*)(m1 + i); a2[q] = *(const f4*)(m2 + i); }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const size_t i = base + ((size_t)q * 256 + threadIdx.x) * 4;
      f4 gg = {(float)gv[q][0], (float)gv[q][1], (float)gv[q][2], (float)gv[q][3]};
      a1[q] = a1[q] * 0.9f + gg * 0.1f; a2[q] = a2[q] * 0.99f + gg * gg * 0.01f; wf[q] -= a1[q] * 0.01f;
      const h4 wh = h4{(_Float16)wf[q][0], (_Float16)wf[q][1], (_Float16)wf[q][2], (_Float16)wf[q][3]};
      if constexpr (NT) { __builtin_nontemporal_store(wf[q], (f4*)(w_fp + i)); __builtin_nontemporal_store(a1[q], (f4*)(m1 + i)); __builtin_nontemporal_store(a2[q], (f4*)(m2 + i)); __builtin_nontemporal_store(wh, (h4*)(w + i)); }
      else { *(f4*)(w_fp + i) = wf[q]; *(f4*)(m1 + i) = a1[q]; *(f4*)(m2 + i) = a2[q]; *(h4*)(w + i) = wh; }
    }
  }
  if (threadIdx.x == 0) atomicAdd(done, turns * 256ull * 8ull); // parameters
}

struct Bufs { float *w_fp, *m1, *m2; _Float16 *w, *g; u4* table; uint32_t* out; unsigned long long* done; size_t n; };

static hipStream_t masked_stream(const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (mask.empty()) { CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); return s; }
  CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  return s;
}

// runs the gather kernel on sg (or not) and the stream kernel on sa (or not) for `ms` milliseconds each; returns G lane-gathers/s and TB/s
static void run(const Bufs& b, hipStream_t sg, uint32_t g_blocks, hipStream_t sa, uint32_t a_blocks, bool nt, double ms, double* g_rate, double* a_rate) {
  const unsigned long long ticks = (unsigned long long)(ms * 1e5);
  CHECK(hipMemset(b.done, 0, 16));
  CHECK(hipDeviceSynchronize());
  const uint32_t blocks_total = (uint32_t)(b.n / (256 * 8));
  if (sg) hipLaunchKernelGGL(k_gather, dim3(g_blocks), dim3(512), 0, sg, b.table, (1u << 18) - 1, ticks, b.done, b.out);
  if (sa) {
    if (nt) hipLaunchKernelGGL(k_stream<true>, dim3(a_blocks), dim3(256), 0, sa, b.n, b.w_fp, b.w, b.g, b.m1, b.m2, blocks_total, ticks, b.done + 1);
    else hipLaunchKernelGGL(k_stream<false>, dim3(a_blocks), dim3(256), 0, sa, b.n, b.w_fp, b.w, b.g, b.m1, b.m2, blocks_total, ticks, b.done + 1);
  }
  CHECK(hipDeviceSynchronize());
  unsigned long long h[2];
  CHECK(hipMemcpy(h, b.done, 16, hipMemcpyDeviceToHost));
  *g_rate = (double)h[0] / (ms * 1e-3) / 1e9;
  *a_rate = (double)h[1] * 34.0 / (ms * 1e-3) / 1e12;
}

int main() {
  Bufs b; b.n = getenv("CUMASK_N") ? (size_t)atoll(getenv("CUMASK_N")) : 11190272; b.n = b.n / 2048 * 2048; printf("stream working set: %.0f MB (34 bytes x %zu parameters)\n", 34.0 * b.n / 1e6, b.n);
  CHECK(hipMalloc(&b.w_fp, b.n * 4)); CHECK(hipMalloc(&b.m1, b.n * 4)); CHECK(hipMalloc(&b.m2, b.n * 4)); CHECK(hipMalloc(&b.w, b.n * 2)); CHECK(hipMalloc(&b.g, b.n * 2));
  CHECK(hipMemset(b.w_fp, 0, b.n * 4)); CHECK(hipMemset(b.m1, 0, b.n * 4)); CHECK(hipMemset(b.m2, 0, b.n * 4)); CHECK(hipMemset(b.g, 0x3c, b.n * 2));
  CHECK(hipMalloc(&b.table, (size_t)8 << 22)); CHECK(hipMemset(b.table, 1, (size_t)8 << 22)); CHECK(hipMalloc(&b.out, 1 << 26)); CHECK(hipMalloc(&b.done, 16));
  // bit i of the mask = CU i / 8 of XCD i % 8 (round 4, tools/ubench/cumask.hip): the first k CUs of every XCD = bits 0 .. 8 k - 1
  auto first_k = [](int k) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 8 * k; ++i) m[i >> 5] |= 1u << (i & 31); return m; };
  auto rest_k = [](int k) { std::vector<uint32_t> m(8, 0); for (int i = 8 * k; i < 256; ++i) m[i >> 5] |= 1u << (i & 31); return m; };
  const double MS = 2.0; // every launch runs for 2 ms
  double g_full, a_full, a_full_nt, x, y;
  hipStream_t whole1 = masked_stream({}), whole2 = masked_stream({});
  run(b, whole1, 512, nullptr, 0, false, MS, &g_full, &x);           // two resident workgroups of 512 per CU
  run(b, nullptr, 0, whole2, 2048, false, MS, &x, &a_full);
  run(b, nullptr, 0, whole2, 2048, true, MS, &x, &a_full_nt);
  printf("alone on the whole chip: gather %6.1f G lane-gathers/s | stream %5.2f TB/s | stream, nontemporal %5.2f TB/s\n", g_full, a_full, a_full_nt);
  run(b, whole1, 512, whole2, 2048, false, MS, &x, &y);
  printf("both on the whole chip (two unmasked streams): gather %6.1f G/s (%.2f) + stream %5.2f TB/s (%.2f) = %.2f combined\n", x, x / g_full, y, y / a_full, x / g_full + y / a_full);
  for (int k : {16, 24, 28, 30}) {
    hipStream_t sg = masked_stream(first_k(k)), sa = masked_stream(rest_k(k));
    double g_alone, a_alone, g_t, a_t, g_tn, a_tn;
    run(b, sg, 16 * k, nullptr, 0, false, MS, &g_alone, &x);
    run(b, nullptr, 0, sa, 64 * (32 - k), false, MS, &x, &a_alone);
    run(b, sg, 16 * k, sa, 64 * (32 - k), false, MS, &g_t, &a_t);
    run(b, sg, 16 * k, sa, 64 * (32 - k), true, MS, &g_tn, &a_tn);
    printf("gather on %2d CUs/XCD, stream on %2d: alone %6.1f G/s, %5.2f TB/s | together %6.1f G/s (%.2f of the whole chip's) + %5.2f TB/s (%.2f) = %.2f combined | stream nontemporal: %6.1f G/s (%.2f) + %5.2f TB/s (%.2f) = %.2f\n",
           k, 32 - k, g_alone, a_alone, g_t, g_t / g_full, a_t, a_t / a_full, g_t / g_full + a_t / a_full, g_tn, g_tn / g_full, a_tn, a_tn / a_full_nt, g_tn / g_full + a_tn / a_full_nt);
    CHECK(hipStreamDestroy(sg)); CHECK(hipStreamDestroy(sa));
  }
  return 0;
}
