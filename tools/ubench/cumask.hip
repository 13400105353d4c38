// microbenchmark: can a gather-bound kernel and a streaming kernel share the chip on disjoint CU sets (hipExtStreamCreateWithCUMask)?
//   1. which (XCC, CU) a mask bit selects (HW_REG_XCC_ID / HW_REG_HW_ID of every workgroup);
//   2. random 16-byte gathers from a 4 MB table per XCD (the scatter's / the forward kernel's shape) on k CUs per XCD;
//   3. the Adam traffic shape (stream_adam.hip, VPT 8, one pass) on k CUs per XCD;
//   4. both at once on complementary sets, and both at once on the whole chip (two unmasked streams).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void k_where(uint32_t* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (__builtin_amdgcn_s_getreg(63508) << 16) | (__builtin_amdgcn_s_getreg(63492) & 0xffff); // XCC_ID, HW_ID
  __builtin_amdgcn_s_sleep(100);
}

// every workgroup gathers from the 4 MB slice of its XCD (slice picked by the XCC id it runs on): 16 gathers of 16 bytes in flight per lane
__global__ void __launch_bounds__(512, 2) k_gather(const u4* __restrict__ table, uint32_t slice_mask, uint32_t iters, uint32_t* __restrict__ out) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg(63508) & 7;
  const u4* t = table + (size_t)xcc * (slice_mask + 1);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    u4 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = t[hash32(tid * 977u + (it * 16 + e) * 0x9e3779b9u) & slice_mask];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += v[e][0] + v[e][3];
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

__global__ void __launch_bounds__(256) k_stream(size_t n, float* __restrict__ w_fp, _Float16* __restrict__ w, const _Float16* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2, uint32_t blocks_total) {
  constexpr int Q = 2;
  for (size_t blk = blockIdx.x; blk < blocks_total; blk += gridDim.x) {
    const size_t base = blk * 256 * 8;
    h4 gv[Q]; f4 wf[Q], a1[Q], a2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const size_t i = base + ((size_t)q * 256 + threadIdx.x) * 4;
      gv[q] = *(const h4*)(g + i); wf[q] = *(const f4*)(w_fp + i); a1[q] = *(const f4*)(m1 + i); a2[q] = *(const f4*)(m2 + i);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const size_t i = base + ((size_t)q * 256 + threadIdx.x) * 4;
      f4 gg = {(float)gv[q][0], (float)gv[q][1], (float)gv[q][2], (float)gv[q][3]};
      a1[q] = a1[q] * 0.9f + gg * 0.1f; a2[q] = a2[q] * 0.99f + gg * gg * 0.01f; wf[q] -= a1[q] * 0.01f;
      *(f4*)(w_fp + i) = wf[q]; *(f4*)(m1 + i) = a1[q]; *(f4*)(m2 + i) = a2[q];
      *(h4*)(w + i) = h4{(_Float16)wf[q][0], (_Float16)wf[q][1], (_Float16)wf[q][2], (_Float16)wf[q][3]};
    }
  }
}

struct Bufs { float *w_fp, *m1, *m2; _Float16 *w, *g; u4* table; uint32_t* out; size_t n; };

static hipStream_t masked_stream(const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (mask.empty()) { CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); return s; }
  CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  return s;
}

static void where(const char* name, hipStream_t s, uint32_t* dev, int verbose) {
  const int blocks = 4096;
  std::vector<uint32_t> h(blocks);
  hipLaunchKernelGGL(k_where, dim3(blocks), dim3(64), 0, s, dev);
  CHECK(hipStreamSynchronize(s));
  CHECK(hipMemcpy(h.data(), dev, blocks * 4, hipMemcpyDeviceToHost));
  int seen[8][128] = {};
  for (uint32_t v : h) { const uint32_t xcc = (v >> 16) & 7, hw = v & 0xffff; const uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; seen[xcc][(se * 2 + sh) * 16 + cu]++; }
  printf("%-34s CUs used per XCC:", name);
  int total = 0;
  for (int x = 0; x < 8; ++x) { int c = 0; for (int i = 0; i < 128; ++i) c += seen[x][i] > 0; printf(" %2d", c); total += c; }
  printf("  (total %d)\n", total);
  if (verbose) for (int x = 0; x < 8; ++x) { printf("   xcc %d:", x); for (int i = 0; i < 128; ++i) if (seen[x][i]) printf(" se%d.sh%d.cu%d", i / 32, (i / 16) & 1, i & 15); printf("\n"); }
}

static float time_pair(const Bufs& b, hipStream_t sg, hipStream_t sa, uint32_t gather_blocks, uint32_t gather_iters, uint32_t adam_blocks, float* ms_g, float* ms_a) {
  hipEvent_t g0, g1, a0, a1; CHECK(hipEventCreate(&g0)); CHECK(hipEventCreate(&g1)); CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1));
  const uint32_t blocks_total = (uint32_t)(b.n / (256 * 8));
  float best = 1e9; *ms_g = *ms_a = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipDeviceSynchronize());
    if (sg) { CHECK(hipEventRecord(g0, sg)); hipLaunchKernelGGL(k_gather, dim3(gather_blocks), dim3(512), 0, sg, b.table, (1u << 18) - 1, gather_iters, b.out); CHECK(hipEventRecord(g1, sg)); }
    if (sa) { CHECK(hipEventRecord(a0, sa)); hipLaunchKernelGGL(k_stream, dim3(adam_blocks ? adam_blocks : blocks_total), dim3(256), 0, sa, b.n, b.w_fp, b.w, b.g, b.m1, b.m2, blocks_total); CHECK(hipEventRecord(a1, sa)); }
    CHECK(hipDeviceSynchronize());
    float tg = 0, ta = 0, span = 0;
    if (sg) CHECK(hipEventElapsedTime(&tg, g0, g1));
    if (sa) CHECK(hipEventElapsedTime(&ta, a0, a1));
    if (sg && sa) { float x; CHECK(hipEventElapsedTime(&x, g0, a1)); span = x > tg ? x : tg; } else span = tg + ta;
    if (span < best) { best = span; *ms_g = tg; *ms_a = ta; }
  }
  return best;
}

__global__ void k_tiny(uint32_t* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

// price of leaving the caller's stream and coming back: k1 (main) -> k2 (side) -> k3 (main) against all three on main
static void hops(uint32_t* dev, hipStream_t main_s, hipStream_t side, const char* name) {
  hipEvent_t t0, t1, e1, e2; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9, sum = 0; const int reps = 20, chain = 10;
    for (int rep = 0; rep < reps; ++rep) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(t0, main_s));
      for (int c = 0; c < chain; ++c) {
        hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, main_s, dev);
        if (mode == 0) hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, main_s, dev);
        else { CHECK(hipEventRecord(e1, main_s)); CHECK(hipStreamWaitEvent(side, e1, 0)); hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, side, dev); CHECK(hipEventRecord(e2, side)); CHECK(hipStreamWaitEvent(main_s, e2, 0)); }
        hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, main_s, dev);
      }
      CHECK(hipEventRecord(t1, main_s)); CHECK(hipEventSynchronize(t1));
      float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms; if (rep >= 5) sum += ms;
    }
    printf("%-40s %s: best %6.2f us, mean %6.2f us per k1-k2-k3 triple\n", name, mode ? "k2 on the side stream" : "all on one stream    ", best * 1e3 / chain, sum / (reps - 5) * 1e3 / chain);
  }
}

int main() {
  Bufs b; b.n = 11190272;
  CHECK(hipMalloc(&b.w_fp, b.n * 4)); CHECK(hipMalloc(&b.m1, b.n * 4)); CHECK(hipMalloc(&b.m2, b.n * 4)); CHECK(hipMalloc(&b.w, b.n * 2)); CHECK(hipMalloc(&b.g, b.n * 2));
  CHECK(hipMemset(b.w_fp, 0, b.n * 4)); CHECK(hipMemset(b.m1, 0, b.n * 4)); CHECK(hipMemset(b.m2, 0, b.n * 4)); CHECK(hipMemset(b.g, 0x3c, b.n * 2));
  CHECK(hipMalloc(&b.table, (size_t)8 << 22)); CHECK(hipMemset(b.table, 1, (size_t)8 << 22)); CHECK(hipMalloc(&b.out, 1 << 26));
  uint32_t* dev; CHECK(hipMalloc(&dev, 4096 * 4));

  // 1. bit -> CU
  where("no mask", masked_stream({}), dev, 0);
  where("bits 0..31", masked_stream({0xffffffffu, 0, 0, 0, 0, 0, 0, 0}), dev, 1);
  where("bits 0..127", masked_stream({~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}), dev, 0);
  where("even bits", masked_stream({0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}), dev, 0);
  where("bits with (bit>>3)&3 == 0", masked_stream({0x0f0f0f0fu & 0x00ff00ffu, 0x00ff00ffu & 0x0f0f0f0fu, 0x000f000fu, 0x000f000fu, 0x000f000fu, 0x000f000fu, 0x000f000fu, 0x000f000fu}), dev, 0);

  // masks by "CU index within XCD" under the hypothesis bit i -> XCC i % 8, CU i / 8: the first k CUs of every XCD = bits 0 .. 8 k - 1
  auto first_k = [](int k) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 8 * k; ++i) m[i >> 5] |= 1u << (i & 31); return m; };
  auto rest_k = [](int k) { std::vector<uint32_t> m(8, 0); for (int i = 8 * k; i < 256; ++i) m[i >> 5] |= 1u << (i & 31); return m; };
  for (int k : {8, 16, 24}) { char nm[64]; snprintf(nm, 64, "first %d per XCD (hypothesis)", k); where(nm, masked_stream(first_k(k)), dev, 0); snprintf(nm, 64, "rest after %d", k); where(nm, masked_stream(rest_k(k)), dev, 0); }

  { hipStream_t m = masked_stream({}), sd = masked_stream({}), sm = masked_stream(first_k(28));
    hops(dev, m, sd, "side stream unmasked"); hops(dev, m, sm, "side stream masked (28 CUs per XCD)"); hops(dev, nullptr, sm, "null stream main, masked side"); }
  const uint32_t GB = 4096, GI = 8; // 4096 x 512 lanes x 8 x 16 gathers = 268 M lane-gathers of 16 bytes
  const double gathers = (double)GB * 512 * GI * 16, bytes = 34.0 * b.n;
  float tg, ta, t;
  hipStream_t whole1 = masked_stream({}), whole2 = masked_stream({});
  t = time_pair(b, whole1, nullptr, GB, GI, 0, &tg, &ta); printf("gather alone, whole chip        : %7.1f us  %6.1f G lane-gathers/s\n", tg * 1e3, gathers / tg / 1e6);
  t = time_pair(b, nullptr, whole2, GB, GI, 0, &tg, &ta); printf("stream alone, whole chip        : %7.1f us  %5.2f TB/s\n", ta * 1e3, bytes / ta / 1e9);
  t = time_pair(b, whole1, whole2, GB, GI, 0, &tg, &ta);  printf("both, two unmasked streams      : span %7.1f us (gather %7.1f, stream %7.1f)\n", t * 1e3, tg * 1e3, ta * 1e3);
  for (int k : {8, 12, 16, 20, 24, 28}) {
    hipStream_t sg = masked_stream(first_k(k)), sa = masked_stream(rest_k(k));
    float g_alone, a_alone, x;
    time_pair(b, sg, nullptr, GB, GI, 0, &g_alone, &x);
    time_pair(b, nullptr, sa, GB, GI, 0, &x, &a_alone);
    t = time_pair(b, sg, sa, GB, GI, 0, &tg, &ta);
    printf("gather on %2d CUs/XCD, stream on %2d: gather alone %7.1f us (%6.1f G/s), stream alone %7.1f us (%5.2f TB/s), together span %7.1f us (gather %7.1f, stream %7.1f)\n", k, 32 - k,
           g_alone * 1e3, gathers / g_alone / 1e6, a_alone * 1e3, bytes / a_alone / 1e9, t * 1e3, tg * 1e3, ta * 1e3);
    CHECK(hipStreamDestroy(sg)); CHECK(hipStreamDestroy(sa));
  }
  return 0;
}
