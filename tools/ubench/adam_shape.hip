// microbenchmark: which access shape of k_adam's ten streams (w_fp, m1, m2 fp32 r+w; g half r; w half w; steps u16 r+w = 32 B per parameter)
// comes closest to the HBM rate?
//   MODE 0: quad per lane everywhere (16-B fp32, 8-B half / u16 accesses)             -- k_adam today
//   MODE 1: 8 parameters per lane, every access 16 B per lane: halves / counts contiguous (dense 1 KiB per wave instruction), fp32 as two
//           16-B accesses 32 B apart (each instruction touches half of every line)
//   MODE 2: 8 parameters per lane, halves / counts 16 B contiguous, fp32 dense (lane j: quads j and j + 64) with lane shuffles in between
//   MODE 3: as 0 without the counts (the shape of tools/ubench/cumask.hip's stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint16_t s4 __attribute__((ext_vector_type(4)));
typedef uint16_t s8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ inline void upd(f4& wf, f4& a1, f4& a2, const f4 gg) { a1 = a1 * 0.9f + gg * 0.1f; a2 = a2 * 0.99f + gg * gg * 0.01f; wf -= a1 * 0.01f; }

template <int MODE>
__global__ void __launch_bounds__(256) k_shape(size_t n, float* __restrict__ w_fp, _Float16* __restrict__ w, const _Float16* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2, uint16_t* __restrict__ st) {
  if (MODE == 0 || MODE == 3) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const h4 gv = *(const h4*)(g + i); f4 wf = *(const f4*)(w_fp + i), a1 = *(const f4*)(m1 + i), a2 = *(const f4*)(m2 + i);
    s4 s = {0, 0, 0, 0}; if (MODE == 0) s = *(const s4*)(st + i);
    upd(wf, a1, a2, f4{(float)gv[0], (float)gv[1], (float)gv[2], (float)gv[3]}); s += 1;
    *(f4*)(w_fp + i) = wf; *(f4*)(m1 + i) = a1; *(f4*)(m2 + i) = a2; if (MODE == 0) *(s4*)(st + i) = s;
    *(h4*)(w + i) = h4{(_Float16)wf[0], (_Float16)wf[1], (_Float16)wf[2], (_Float16)wf[3]};
  } else if (MODE == 1) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const h8 gv = *(const h8*)(g + i); s8 s = *(const s8*)(st + i);
    f4 wf0 = *(const f4*)(w_fp + i), wf1 = *(const f4*)(w_fp + i + 4), a10 = *(const f4*)(m1 + i), a11 = *(const f4*)(m1 + i + 4), a20 = *(const f4*)(m2 + i), a21 = *(const f4*)(m2 + i + 4);
    upd(wf0, a10, a20, f4{(float)gv[0], (float)gv[1], (float)gv[2], (float)gv[3]}); upd(wf1, a11, a21, f4{(float)gv[4], (float)gv[5], (float)gv[6], (float)gv[7]}); s += 1;
    *(f4*)(w_fp + i) = wf0; *(f4*)(w_fp + i + 4) = wf1; *(f4*)(m1 + i) = a10; *(f4*)(m1 + i + 4) = a11; *(f4*)(m2 + i) = a20; *(f4*)(m2 + i + 4) = a21; *(s8*)(st + i) = s;
    *(h8*)(w + i) = h8{(_Float16)wf0[0], (_Float16)wf0[1], (_Float16)wf0[2], (_Float16)wf0[3], (_Float16)wf1[0], (_Float16)wf1[1], (_Float16)wf1[2], (_Float16)wf1[3]};
  } else {
    // a wave covers 512 consecutive parameters; halves / counts: lane l holds parameters 8 l .. 8 l + 7; fp32: lane l holds quads l and l + 64
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave_base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 512;
    if (wave_base >= n) return;
    const u4 gv = *(const u4*)(g + wave_base + lane * 8); const u4 sv = *(const u4*)(st + wave_base + lane * 8);
    f4 wf0 = *(const f4*)(w_fp + wave_base + lane * 4), wf1 = *(const f4*)(w_fp + wave_base + 256 + lane * 4);
    f4 a10 = *(const f4*)(m1 + wave_base + lane * 4), a11 = *(const f4*)(m1 + wave_base + 256 + lane * 4);
    f4 a20 = *(const f4*)(m2 + wave_base + lane * 4), a21 = *(const f4*)(m2 + wave_base + 256 + lane * 4);
    // quad q (= lane for the first fp32 access, lane + 64 for the second) sits in lane q >> 1 of the half arrays, dwords 2 (q & 1) .. 2 (q & 1) + 1
    const uint32_t src0 = lane >> 1, src1 = 32 + (lane >> 1), odd = lane & 1;
    const uint32_t g0a = __shfl(odd ? gv[2] : gv[0], 0, 64); (void)g0a;
    uint32_t ga[2], gb[2], sa[2], sb[2];
    { const uint32_t x0 = __shfl(gv[0], src0, 64), x1 = __shfl(gv[1], src0, 64), x2 = __shfl(gv[2], src0, 64), x3 = __shfl(gv[3], src0, 64); ga[0] = odd ? x2 : x0; ga[1] = odd ? x3 : x1; }
    { const uint32_t x0 = __shfl(gv[0], src1, 64), x1 = __shfl(gv[1], src1, 64), x2 = __shfl(gv[2], src1, 64), x3 = __shfl(gv[3], src1, 64); gb[0] = odd ? x2 : x0; gb[1] = odd ? x3 : x1; }
    { const uint32_t x0 = __shfl(sv[0], src0, 64), x1 = __shfl(sv[1], src0, 64), x2 = __shfl(sv[2], src0, 64), x3 = __shfl(sv[3], src0, 64); sa[0] = odd ? x2 : x0; sa[1] = odd ? x3 : x1; }
    { const uint32_t x0 = __shfl(sv[0], src1, 64), x1 = __shfl(sv[1], src1, 64), x2 = __shfl(sv[2], src1, 64), x3 = __shfl(sv[3], src1, 64); sb[0] = odd ? x2 : x0; sb[1] = odd ? x3 : x1; }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 g00 = __builtin_bit_cast(h2, ga[0]), g01 = __builtin_bit_cast(h2, ga[1]), g10 = __builtin_bit_cast(h2, gb[0]), g11 = __builtin_bit_cast(h2, gb[1]);
    upd(wf0, a10, a20, f4{(float)g00[0], (float)g00[1], (float)g01[0], (float)g01[1]}); upd(wf1, a11, a21, f4{(float)g10[0], (float)g10[1], (float)g11[0], (float)g11[1]});
    sa[0] += 0x00010001u; sa[1] += 0x00010001u; sb[0] += 0x00010001u; sb[1] += 0x00010001u;
    *(f4*)(w_fp + wave_base + lane * 4) = wf0; *(f4*)(w_fp + wave_base + 256 + lane * 4) = wf1;
    *(f4*)(m1 + wave_base + lane * 4) = a10; *(f4*)(m1 + wave_base + 256 + lane * 4) = a11;
    *(f4*)(m2 + wave_base + lane * 4) = a20; *(f4*)(m2 + wave_base + 256 + lane * 4) = a21;
    // back: lane l of the half arrays needs quads 2 l and 2 l + 1: from lanes 2 l, 2 l + 1 of the first access (l < 32) or 2 l - 64, 2 l - 63 of the second
    const uint32_t h0 = __builtin_bit_cast(uint32_t, h2{(_Float16)wf0[0], (_Float16)wf0[1]}), h1 = __builtin_bit_cast(uint32_t, h2{(_Float16)wf0[2], (_Float16)wf0[3]});
    const uint32_t h2_ = __builtin_bit_cast(uint32_t, h2{(_Float16)wf1[0], (_Float16)wf1[1]}), h3 = __builtin_bit_cast(uint32_t, h2{(_Float16)wf1[2], (_Float16)wf1[3]});
    const uint32_t e = (2 * lane) & 63, hi = lane >= 32;
    u4 wv, so;
    { const uint32_t a = __shfl(h0, e, 64), b = __shfl(h2_, e, 64); wv[0] = hi ? b : a; } { const uint32_t a = __shfl(h1, e, 64), b = __shfl(h3, e, 64); wv[1] = hi ? b : a; }
    { const uint32_t a = __shfl(h0, e + 1, 64), b = __shfl(h2_, e + 1, 64); wv[2] = hi ? b : a; } { const uint32_t a = __shfl(h1, e + 1, 64), b = __shfl(h3, e + 1, 64); wv[3] = hi ? b : a; }
    { const uint32_t a = __shfl(sa[0], e, 64), b = __shfl(sb[0], e, 64); so[0] = hi ? b : a; } { const uint32_t a = __shfl(sa[1], e, 64), b = __shfl(sb[1], e, 64); so[1] = hi ? b : a; }
    { const uint32_t a = __shfl(sa[0], e + 1, 64), b = __shfl(sb[0], e + 1, 64); so[2] = hi ? b : a; } { const uint32_t a = __shfl(sa[1], e + 1, 64), b = __shfl(sb[1], e + 1, 64); so[3] = hi ? b : a; }
    *(u4*)(w + wave_base + lane * 8) = wv; *(u4*)(st + wave_base + lane * 8) = so;
  }
}

template <int MODE> void run(const char* name, size_t n, float* w_fp, _Float16* w, _Float16* g, float* m1, float* m2, uint16_t* st) {
  const int per_block = (MODE == 0 || MODE == 3) ? 1024 : 2048;
  const uint32_t blocks = (uint32_t)((n + per_block - 1) / per_block);
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9, sum = 0;
  for (int rep = 0; rep < 12; ++rep) {
    CHECK(hipEventRecord(a)); hipLaunchKernelGGL(k_shape<MODE>, dim3(blocks), dim3(256), 0, 0, n, w_fp, w, g, m1, m2, st); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; if (rep >= 2) sum += ms;
  }
  const double bytes = (MODE == 3 ? 28.0 : 32.0) * n;
  printf("%-52s best %6.1f us (%5.2f TB/s)  mean %6.1f us\n", name, best * 1e3, bytes / best / 1e9, sum / 10 * 1e3);
}

int main() {
  const size_t n = 11190272;
  float *w_fp, *m1, *m2; _Float16 *w, *g; uint16_t* st;
  CHECK(hipMalloc(&w_fp, n * 4)); CHECK(hipMalloc(&m1, n * 4)); CHECK(hipMalloc(&m2, n * 4)); CHECK(hipMalloc(&w, n * 2)); CHECK(hipMalloc(&g, n * 2)); CHECK(hipMalloc(&st, n * 2));
  CHECK(hipMemset(w_fp, 0, n * 4)); CHECK(hipMemset(m1, 0, n * 4)); CHECK(hipMemset(m2, 0, n * 4)); CHECK(hipMemset(g, 0x3c, n * 2)); CHECK(hipMemset(st, 0, n * 2));
  for (int pass = 0; pass < 2; ++pass) {
    run<3>("quad per lane, no counts (28 B)", n, w_fp, w, g, m1, m2, st);
    run<0>("quad per lane, 8-B half / count accesses (k_adam)", n, w_fp, w, g, m1, m2, st);
    run<1>("8 per lane, all 16 B, fp32 32 B apart", n, w_fp, w, g, m1, m2, st);
    run<2>("8 per lane, all 16 B dense, lane shuffles", n, w_fp, w, g, m1, m2, st);
  }
  return 0;
}
