// mfma_valu.hip -- how much vector work hides under v_mfma_f32_32x32x16_f16 on one SIMD (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
// One workgroup per CU, W waves per SIMD (W = 1, 2); every wave runs `iters` iterations of a body made of
//   M matrix instructions (independent accumulators or one dependent chain) and V vector instructions per matrix instruction,
// in one of these mixes:
//   mode 0: every wave runs the interleaved body  M (V x valu)
//   mode 1: waves 0..3 run matrix instructions only, waves 4..7 vector instructions only (needs W = 2)
// The vector instruction is v_cvt_pk_f16_f32 / v_pk_max_i16 / v_fma_f32 (kind 0 / 1 / 2).  Prints clocks per iteration (wave 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int V, int KIND>
__device__ inline void valu_block(uint32_t (&r)[8], float (&f)[8]) {
#pragma unroll
	for (int i = 0; i < V; ++i) {
		if constexpr (KIND == 0) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[i & 7]) : "v"(f[i & 7]), "v"(f[(i + 1) & 7]));
		else if constexpr (KIND == 1) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(r[i & 7]) : "v"(r[(i + 3) & 7]));
		else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i & 7]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
	}
}

template <int V, int KIND, int MODE, bool CHAIN>
__global__ void __launch_bounds__(512, 2) k(unsigned long long* out, int iters, float seed) {
	const uint32_t wave = threadIdx.x >> 6;
	h8 a, b;
	for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 0.001f + i); b[i] = (_Float16)(seed * 0.5f + i); }
	f16v acc[4];
	for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
	uint32_t r[8];
	float f[8];
	for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x + i; f[i] = seed + i; }
	const bool do_m = MODE == 0 || wave < 4, do_v = MODE == 0 || wave >= 4;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	if (MODE == 0) {
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[CHAIN ? 0 : (m & 3)]) : "v"(a), "v"(b));
				valu_block<V, KIND>(r, f);
			}
		}
	} else if (do_m) {
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[CHAIN ? 0 : (m & 3)]) : "v"(a), "v"(b));
		}
	} else if (do_v) {
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int m = 0; m < 8; ++m) valu_block<V, KIND>(r, f);
		}
	}
	asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
	const unsigned long long t1 = __builtin_readcyclecounter();
	float s = 0;
	for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
	for (int i = 0; i < 8; ++i) s += f[i] + (float)r[i];
	if (s == 12345.678f) out[1000] = 1; // keep everything alive
	if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V, int KIND, int MODE, bool CHAIN>
void run(const char* name, int waves_per_simd, unsigned long long* d_out) {
	const int iters = 2000;
	const int threads = waves_per_simd * 256;
	hipLaunchKernelGGL((k<V, KIND, MODE, CHAIN>), dim3(256), dim3(threads), 0, 0, d_out, iters, 1.0f);
	hipLaunchKernelGGL((k<V, KIND, MODE, CHAIN>), dim3(256), dim3(threads), 0, 0, d_out, iters, 1.0f);
	hipDeviceSynchronize();
	std::vector<unsigned long long> h(256 * 8);
	hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
	double w0 = 0, w4 = 0;
	for (int b = 0; b < 256; ++b) { w0 += (double)h[b * 8]; w4 += (double)h[b * 8 + (waves_per_simd == 2 ? 4 : 0)]; }
	printf("%-58s waves/SIMD %d  V/M %d : %.1f clocks per matrix slot (wave 0), %.1f (wave 4)\n", name, waves_per_simd, V, w0 / 256 / iters / 8, w4 / 256 / iters / 8);
}

int main() {
	unsigned long long* d_out;
	hipMalloc(&d_out, 4096 * 8);
	hipMemset(d_out, 0, 4096 * 8);
#define ROW(V, KIND, NAME) \
	run<V, KIND, 0, false>("interleaved, 4 accumulators, " NAME, 1, d_out); \
	run<V, KIND, 0, false>("interleaved, 4 accumulators, " NAME, 2, d_out); \
	run<V, KIND, 0, true>("interleaved, 1 accumulator chain, " NAME, 2, d_out); \
	run<V, KIND, 1, false>("matrix waves + vector waves, " NAME, 2, d_out);
	ROW(0, 0, "no vector work")
	ROW(2, 0, "cvt_pk")
	ROW(4, 0, "cvt_pk")
	ROW(6, 0, "cvt_pk")
	ROW(8, 0, "cvt_pk")
	ROW(4, 1, "pk_max_i16")
	ROW(6, 1, "pk_max_i16")
	ROW(8, 1, "pk_max_i16")
	ROW(4, 2, "fma_f32")
	ROW(6, 2, "fma_f32")
	ROW(8, 2, "fma_f32")
	return 0;
}
