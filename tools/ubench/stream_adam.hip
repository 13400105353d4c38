// microbenchmark: the HBM traffic shape of adam_step (read 18 B + write 18 B per parameter, in place) on gfx950,
// to find the launch shape / load-store flavour that gets closest to the achievable ~6.3 TB/s.
//   VPT   : parameters per thread (4, 8, 16)
//   NT    : 0 plain, 1 nontemporal stores, 2 nontemporal loads + stores
//   grid  : one pass (n / (256 * VPT) blocks) or persistent grid-stride (blocks = 256 CUs * k)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int NT, typename T> __device__ inline T ld(const T* p) { if (NT >= 2) return __builtin_nontemporal_load(p); return *p; }
template <int NT, typename T> __device__ inline void st(T* p, T v) { if (NT >= 1) __builtin_nontemporal_store(v, p); else *p = v; }

template <int VPT, int NT>
__global__ void __launch_bounds__(256) k_stream(size_t n, float* __restrict__ w_fp, _Float16* __restrict__ w, const _Float16* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                                                uint32_t* __restrict__ steps, int persistent) {
  constexpr int Q = VPT / 4;
  const size_t stride = persistent ? (size_t)gridDim.x * 256 * 4 : 0;
  // each thread handles Q quads spaced one block-width apart so that every wave instruction is a dense 1-KB (fp32) run
  for (size_t base = (size_t)blockIdx.x * 256 * VPT; base < n; base += persistent ? (size_t)gridDim.x * 256 * VPT : n) {
    h4 gv[Q]; f4 wf[Q], a1[Q], a2[Q]; u4 s[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const size_t i = base + ((size_t)q * 256 + threadIdx.x) * 4;
      gv[q] = ld<NT>((const h4*)(g + i)); wf[q] = ld<NT>((const f4*)(w_fp + i)); a1[q] = ld<NT>((const f4*)(m1 + i)); a2[q] = ld<NT>((const f4*)(m2 + i)); s[q] = ld<NT>((const u4*)(steps + i));
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const size_t i = base + ((size_t)q * 256 + threadIdx.x) * 4;
      f4 gg = {(float)gv[q][0], (float)gv[q][1], (float)gv[q][2], (float)gv[q][3]};
      a1[q] = a1[q] * 0.9f + gg * 0.1f; a2[q] = a2[q] * 0.99f + gg * gg * 0.01f; s[q] += 1; wf[q] -= a1[q] * 0.01f;
      st<NT>((f4*)(w_fp + i), wf[q]); st<NT>((f4*)(m1 + i), a1[q]); st<NT>((f4*)(m2 + i), a2[q]); st<NT>((u4*)(steps + i), s[q]);
      st<NT>((h4*)(w + i), h4{(_Float16)wf[q][0], (_Float16)wf[q][1], (_Float16)wf[q][2], (_Float16)wf[q][3]});
    }
    (void)stride;
  }
}

template <int VPT, int NT>
void run(size_t n, int blocks_per_cu) {
  float *w_fp, *m1, *m2; uint32_t* steps; _Float16 *w, *g;
  CHECK(hipMalloc(&w_fp, n * 4)); CHECK(hipMalloc(&m1, n * 4)); CHECK(hipMalloc(&m2, n * 4)); CHECK(hipMalloc(&steps, n * 4)); CHECK(hipMalloc(&w, n * 2)); CHECK(hipMalloc(&g, n * 2));
  CHECK(hipMemset(w_fp, 0, n * 4)); CHECK(hipMemset(m1, 0, n * 4)); CHECK(hipMemset(m2, 0, n * 4)); CHECK(hipMemset(steps, 0, n * 4)); CHECK(hipMemset(g, 0x3c, n * 2));
  const int persistent = blocks_per_cu > 0;
  const uint32_t blocks = persistent ? 256 * blocks_per_cu : (uint32_t)(n / (256 * VPT));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_stream<VPT, NT>), dim3(blocks), dim3(256), 0, 0, n, w_fp, w, g, m1, m2, steps, persistent);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  printf("VPT=%2d NT=%d %-22s blocks=%6u: %7.1f us  %5.2f TB/s\n", VPT, NT, persistent ? "persistent grid-stride" : "one pass", blocks, best * 1e3, 36.0 * n / (best * 1e-3) / 1e12);
  CHECK(hipFree(w_fp)); CHECK(hipFree(m1)); CHECK(hipFree(m2)); CHECK(hipFree(steps)); CHECK(hipFree(w)); CHECK(hipFree(g));
}

int main() {
  const size_t n = 11190272; // ~ C3a parameter count, multiple of 256 * 16
  for (int bpc : {0, 4, 8, 16}) {
    run<4, 0>(n, bpc); run<4, 1>(n, bpc); run<4, 2>(n, bpc);
    run<8, 0>(n, bpc); run<8, 1>(n, bpc); run<8, 2>(n, bpc);
    run<16, 0>(n, bpc); run<16, 1>(n, bpc); run<16, 2>(n, bpc);
  }
  return 0;
}
