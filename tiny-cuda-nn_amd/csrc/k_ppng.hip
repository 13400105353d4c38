// k_ppng.hip -- the PPNG1 and PPNG2 encodings of this fork (PPNG2: further down).  PPNG1 (encodings/ppng.h:30-119, encodings/ppng_1.h:13-213): per frequency f and phase
// s (sine / cosine) the three coordinates are mapped to sc_i = sin(freq_f (x_i - 0.5) + s pi / 2), each sc_i looks up a 1-D table
// of Q bins per (feature c, rank r) with linear interpolation, and output (f, s, c) = sum_r prod_i table_i[c][.][r](sc_i) -- a
// rank-R factorisation of a 3-D feature volume per frequency.  Parameters: half [F][2][D][C][Q][R].
//
// Forward: one thread per (sample, f, s), as there.  Backward (parameters only, as there: PPNG1 has no input gradient): the
// reference adds (half)(dL/dy * prod_{j != i} interp_j * weight) into the table with packed-fp16 global atomics; here a workgroup
// owns one (f, s) slice of the table (D C Q R values, 3072 with the defaults) as 64-bit fixed-point accumulators in LDS (LSB =
// 2^-24: every fp16 product converts exactly), walks a block of samples, and merges its exact partial sums into a scratch table
// with 64-bit integer atomics; k_ppng_finalize rounds once.  Same products, exact sum, deterministic (grid_fixed.h, as the grid
// scatter does it).  Slices that do not fit the LDS take the same route through global atomics.
#include "grid_fixed.h"

#include <cmath>

namespace tcnn_amd {
namespace {

constexpr uint32_t PPNG_D = 3;
constexpr uint32_t PPNG_MAX_R = 16;
constexpr uint32_t PPNG_LDS_ENTRIES = 16384; // 128 KB of accumulators
constexpr uint32_t PPNG_BWD_SAMPLES = 8192;  // samples per workgroup of the backward pass

// ppng_1.h:176-190: the frequency of level f, then sc_i.  The reference evaluates these with double literals: powf(2.0, b) * 3.1415926535
// and freq * (p - 0.5) + s * M_HI are double products / sums, rounded to float where it stores them or calls sinf.
__device__ inline float ppng_freq(const uint32_t f, const uint32_t F, const int32_t log2_min, const int32_t log2_max) {
	const float freq_base = ((float)(int32_t)(f * (uint32_t)(log2_max - log2_min))) / ((float)(F - 1)) + (float)log2_min;
	return (float)((double)powf(2.0f, freq_base) * 3.1415926535);
}
__device__ inline float ppng_sc(const float freq, const float p, const uint32_t s) { return sinf((float)((double)freq * ((double)p - 0.5) + (double)s * 1.57079632679489661923)); }

// ppng_1.h:25-35: bin pair and weight of one coordinate
__device__ inline void ppng_bins(const float sc, const uint32_t Q, uint32_t& p0, uint32_t& p1, float& w) {
	const float p = (float)(((double)(sc + 1.0f) * 0.5) * (double)(Q - 1)); // (sc + 1) is a float sum there, the two products are double
	p0 = min(max((uint32_t)floorf(p), 0u), Q - 1);
	p1 = max(min((uint32_t)ceilf(p), Q - 1), 0u);
	w = p - (float)p0;
}

template <typename T>
__global__ void __launch_bounds__(256) k_ppng1_fwd(const uint32_t n, const uint32_t F, const uint32_t Q, const uint32_t C, const uint32_t R, const int32_t log2_min, const int32_t log2_max,
                                                   const MatView x, const half_t* __restrict__ features, T* __restrict__ out, const uint32_t out_stride) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	const float freq = ppng_freq(f, F, log2_min, log2_max);
	features += ((size_t)f * 2 + s) * PPNG_D * C * Q * R;
	uint32_t p0[PPNG_D], p1[PPNG_D];
	float w[PPNG_D];
#pragma unroll
	for (uint32_t i = 0; i < PPNG_D; ++i) ppng_bins(ppng_sc(freq, x.data[(size_t)b * x.stride_sample + (size_t)i * x.stride_dim], s), Q, p0[i], p1[i], w[i]);
	for (uint32_t c = 0; c < C; ++c) {
		float fs = 0;
		for (uint32_t r = 0; r < R; ++r) {
			float prod = 1;
#pragma unroll
			for (uint32_t i = 0; i < PPNG_D; ++i) {
				const float f0 = (float)features[((size_t)i * C + c) * Q * R + (size_t)p0[i] * R + r];
				const float f1 = (float)features[((size_t)i * C + c) * Q * R + (size_t)p1[i] * R + r];
				prod *= (w[i] * f1) + ((1 - w[i]) * f0);
			}
			fs += prod;
		}
		out[(size_t)b * out_stride + (size_t)f * 2 * C + s * C + c] = (T)fs;
	}
}

// padding columns (the reference's kernel leaves them unwritten; the encoding interface says ones)
template <typename T>
__global__ void __launch_bounds__(256) k_ppng_pad(const uint32_t n, const uint32_t first, const uint32_t out_stride, T* __restrict__ out) {
	const uint32_t pad = out_stride - first;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n * pad) return;
	const uint32_t b = gid / pad, j = gid - b * pad;
	out[(size_t)b * out_stride + first + j] = (T)1.0f;
}

// ppng_1.h:57-140 for a block of samples of one (f, s): exact integer sums, in LDS when the slice fits
template <typename T>
__global__ void __launch_bounds__(256) k_ppng1_bwd(const uint32_t n, const uint32_t F, const uint32_t Q, const uint32_t C, const uint32_t R, const int32_t log2_min, const int32_t log2_max,
                                                   const MatView x, const half_t* __restrict__ features, const T* __restrict__ dL_dy, const uint32_t dy_stride,
                                                   unsigned long long* __restrict__ scratch, const int in_lds) {
	extern __shared__ __attribute__((aligned(16))) char ppng_smem[];
	typedef __attribute__((address_space(3))) unsigned long long lds_u64;
	lds_u64* acc = (lds_u64*)ppng_smem;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	const uint32_t slice = PPNG_D * C * Q * R;
	const size_t slice_off = ((size_t)f * 2 + s) * slice;
	if (in_lds) {
		for (uint32_t e = threadIdx.x; e < slice; e += blockDim.x) ((unsigned long long*)ppng_smem)[e] = 0ull;
		__syncthreads();
	}
	const float freq = ppng_freq(f, F, log2_min, log2_max);
	features += slice_off;
	unsigned long long* global_acc = scratch + slice_off;
	const uint32_t begin = blockIdx.x * PPNG_BWD_SAMPLES, end = min(n, begin + PPNG_BWD_SAMPLES);
	for (uint32_t b = begin + threadIdx.x; b < end; b += blockDim.x) {
		uint32_t p0[PPNG_D], p1[PPNG_D];
		float w[PPNG_D];
#pragma unroll
		for (uint32_t i = 0; i < PPNG_D; ++i) ppng_bins(ppng_sc(freq, x.data[(size_t)b * x.stride_sample + (size_t)i * x.stride_dim], s), Q, p0[i], p1[i], w[i]);
		for (uint32_t c = 0; c < C; ++c) {
			const float go = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
			for (uint32_t r = 0; r < R; ++r) {
				float fa[PPNG_D];
#pragma unroll
				for (uint32_t i = 0; i < PPNG_D; ++i) {
					const float f0 = (float)features[((size_t)i * C + c) * Q * R + (size_t)p0[i] * R + r];
					const float f1 = (float)features[((size_t)i * C + c) * Q * R + (size_t)p1[i] * R + r];
					fa[i] = (w[i] * f1) + ((1 - w[i]) * f0);
				}
#pragma unroll
				for (uint32_t i = 0; i < PPNG_D; ++i) {
					// grad_cache[r][i] = prod_{j != i} fa[j], multiplied up in the order j = 0, 1, 2 starting from 1 (ppng_1.h:100-108)
					float cache = 1;
#pragma unroll
					for (uint32_t j = 0; j < PPNG_D; ++j) cache *= (i == j) ? 1.0f : fa[j];
					const half_t v0 = (half_t)(go * cache * (1 - w[i])), v1 = (half_t)(go * cache * w[i]);
					const uint32_t e0 = (i * C + c) * Q * R + p0[i] * R + r, e1 = (i * C + c) * Q * R + p1[i] * R + r;
					if (in_lds) {
						__hip_atomic_fetch_add(acc + e0, (unsigned long long)half_to_fixed_fast(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						__hip_atomic_fetch_add(acc + e1, (unsigned long long)half_to_fixed_fast(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					} else {
						atomicAdd(global_acc + e0, (unsigned long long)half_to_fixed_fast(v0));
						atomicAdd(global_acc + e1, (unsigned long long)half_to_fixed_fast(v1));
					}
				}
			}
		}
	}
	if (in_lds) {
		__syncthreads();
		for (uint32_t e = threadIdx.x; e < slice; e += blockDim.x) {
			const unsigned long long v = ((unsigned long long*)ppng_smem)[e];
			if (v != 0) atomicAdd(global_acc + e, v);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------- PPNG2
// encodings/ppng_2.h:12-272: per (f, s) three PLANES of Q x Q bins per (feature, rank) -- the X plane indexed by (z, y), the Y plane
// by (z, x), the Z plane by (y, x) -- and output (f, s, c) = sum_r sum_{8 corners} w_corner fx fy fz with the planes' NEAREST entries
// at that corner.  Parameters: half [F][2][3][C][Q][Q][R].
struct Ppng2Lookup {
	uint32_t p[PPNG_D][2]; // bin of axis i at corner bit 0 / 1
	float w8[8];           // corner weights, index = 4 zb + 2 yb + xb
};
__device__ inline Ppng2Lookup ppng2_lookup(const float freq, const MatView x, const uint32_t b, const uint32_t s, const uint32_t Q) {
	Ppng2Lookup L;
	float w[PPNG_D];
#pragma unroll
	for (uint32_t i = 0; i < PPNG_D; ++i) ppng_bins(ppng_sc(freq, x.data[(size_t)b * x.stride_sample + (size_t)i * x.stride_dim], s), Q, L.p[i][0], L.p[i][1], w[i]);
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) { // ppng_2.h:33-40: (x factor * y factor) * z factor
		const float a = (k & 1u) ? w[0] : 1 - w[0], bb = (k & 2u) ? w[1] : 1 - w[1], cc = (k & 4u) ? w[2] : 1 - w[2];
		L.w8[k] = a * bb * cc;
	}
	return L;
}
// entry offset inside plane `pl` (0: X plane (z, y); 1: Y plane (z, x); 2: Z plane (y, x)) of feature c at corner k, rank 0
__device__ inline uint32_t ppng2_entry(const Ppng2Lookup& L, const uint32_t pl, const uint32_t c, const uint32_t k, const uint32_t C, const uint32_t Q, const uint32_t R) {
	const uint32_t hi = pl == 2 ? 1u : 2u, lo = pl == 0 ? 1u : 0u; // the plane's two axes
	return ((pl * C + c) * Q + L.p[hi][(k >> hi) & 1u]) * Q * R + L.p[lo][(k >> lo) & 1u] * R;
}

template <typename T>
__global__ void __launch_bounds__(256) k_ppng2_fwd(const uint32_t n, const uint32_t F, const uint32_t Q, const uint32_t C, const uint32_t R, const int32_t log2_min, const int32_t log2_max,
                                                   const MatView x, const half_t* __restrict__ features, T* __restrict__ out, const uint32_t out_stride) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	features += ((size_t)f * 2 + s) * PPNG_D * C * Q * Q * R;
	const Ppng2Lookup L = ppng2_lookup(ppng_freq(f, F, log2_min, log2_max), x, b, s, Q);
	for (uint32_t c = 0; c < C; ++c) {
		uint32_t e[PPNG_D][8];
#pragma unroll
		for (uint32_t pl = 0; pl < PPNG_D; ++pl)
#pragma unroll
			for (uint32_t k = 0; k < 8; ++k) e[pl][k] = ppng2_entry(L, pl, c, k, C, Q, R);
		float fs = 0;
		for (uint32_t r = 0; r < R; ++r) {
			float sum = 0;
#pragma unroll
			for (uint32_t k = 0; k < 8; ++k) { // ppng_2.h:62-70: the corners in the order 000, 001, ..., 111 (z y x)
				const float v = (float)features[e[0][k] + r] * (float)features[e[1][k] + r] * (float)features[e[2][k] + r];
				sum = k == 0 ? L.w8[0] * v : sum + L.w8[k] * v;
			}
			fs += sum;
		}
		out[(size_t)b * out_stride + (size_t)f * 2 * C + s * C + c] = (T)fs;
	}
}

// ppng_2.h:78-272 for ONE plane of one feature of one (f, s): blockIdx = (sample block, f * 2 + s, pl * C + c).  The workgroup owns
// that plane's Q x Q x R entries in LDS when they fit.  The reference's loop over the three dimensions repeats the twelve additions
// of a sample unchanged three times (its body does not depend on the loop index, ppng_2.h:131-270): the gradient it produces is
// three times the derivative, and so is this one (3 x the fixed-point value: the same exact sum).
template <typename T>
__global__ void __launch_bounds__(256) k_ppng2_bwd(const uint32_t n, const uint32_t F, const uint32_t Q, const uint32_t C, const uint32_t R, const int32_t log2_min, const int32_t log2_max,
                                                   const MatView x, const half_t* __restrict__ features, const T* __restrict__ dL_dy, const uint32_t dy_stride,
                                                   unsigned long long* __restrict__ scratch, const int in_lds, const uint32_t samples_per_block) {
	extern __shared__ __attribute__((aligned(16))) char ppng_smem[];
	typedef __attribute__((address_space(3))) unsigned long long lds_u64;
	lds_u64* acc = (lds_u64*)ppng_smem;
	const uint32_t f = blockIdx.y >> 1, s = blockIdx.y & 1u;
	const uint32_t pl = blockIdx.z / C, c = blockIdx.z - pl * C;
	const uint32_t plane = Q * Q * R;
	const size_t slice_off = ((size_t)f * 2 + s) * PPNG_D * C * plane;
	const uint32_t plane_off = (pl * C + c) * plane;
	if (in_lds) {
		for (uint32_t e = threadIdx.x; e < plane; e += blockDim.x) ((unsigned long long*)ppng_smem)[e] = 0ull;
		__syncthreads();
	}
	const float freq = ppng_freq(f, F, log2_min, log2_max);
	features += slice_off;
	unsigned long long* global_acc = scratch + slice_off + plane_off;
	const uint32_t o1 = pl == 0 ? 1u : 0u, o2 = pl == 2 ? 1u : 2u; // the other two planes, lower index first (ppng_2.h:176-187: w * f_lower * f_higher)
	const uint32_t begin = blockIdx.x * samples_per_block, end = min(n, begin + samples_per_block);
	for (uint32_t b = begin + threadIdx.x; b < end; b += blockDim.x) {
		const Ppng2Lookup L = ppng2_lookup(freq, x, b, s, Q);
		const float go = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
		uint32_t e1[8], e2[8];
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) { e1[k] = ppng2_entry(L, o1, c, k, C, Q, R); e2[k] = ppng2_entry(L, o2, c, k, C, Q, R); }
		for (uint32_t r = 0; r < R; ++r) {
			// the plane's four entries under this sample: in-plane corner (hb, lb); the sum runs over the excluded axis' two bins
#pragma unroll
			for (uint32_t q = 0; q < 4; ++q) {
				const uint32_t hi = pl == 2 ? 1u : 2u, lo = pl == 0 ? 1u : 0u, ex = pl; // axes: plane 0 excludes x (0), 1 excludes y, 2 excludes z
				const uint32_t k0 = (((q >> 1) & 1u) << hi) | ((q & 1u) << lo), k1 = k0 | (1u << ex);
				const float g = go * ((L.w8[k0] * (float)features[e1[k0] + r] * (float)features[e2[k0] + r]) + (L.w8[k1] * (float)features[e1[k1] + r] * (float)features[e2[k1] + r]));
				const unsigned long long v = (unsigned long long)(3 * half_to_fixed_fast((half_t)g));
				const uint32_t e = ppng2_entry(L, pl, c, k0, C, Q, R) - plane_off + r;
				if (in_lds) __hip_atomic_fetch_add(acc + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else atomicAdd(global_acc + e, v);
			}
		}
	}
	if (in_lds) {
		__syncthreads();
		for (uint32_t e = threadIdx.x; e < plane; e += blockDim.x) {
			const unsigned long long v = ((unsigned long long*)ppng_smem)[e];
			if (v != 0) atomicAdd(global_acc + e, v);
		}
	}
}

// scratch (exact sums) -> gradients, rounded once; the scratch is left zeroed for the next step
template <typename T>
__global__ void __launch_bounds__(256) k_ppng_finalize(const size_t n_params, unsigned long long* __restrict__ scratch, T* __restrict__ grad, const int accumulate) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_params) return;
	long long s = (long long)scratch[i];
	scratch[i] = 0;
	if constexpr (sizeof(T) == 2) {
		if (accumulate) s += half_to_fixed(grad[i]);
		grad[i] = fixed_to_half(s);
	} else {
		const float v = (float)((double)s * (1.0 / 16777216.0));
		grad[i] = accumulate ? grad[i] + v : v;
	}
}

} // namespace

void ppng1_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min, int32_t log2_max, MatView x, const void* features, void* out,
                   uint32_t out_stride) {
	if (n == 0 || out_stride == 0) return;
	CHECK_THROW(F >= 2 && Q >= 2 && R >= 1 && R <= PPNG_MAX_R);
	const dim3 grid(div_round_up(n, 256u), F, 2);
	const uint32_t live = F * 2 * C;
	if (fp32) hipLaunchKernelGGL(k_ppng1_fwd<float>, grid, dim3(256), 0, stream, n, F, Q, C, R, log2_min, log2_max, x, (const half_t*)features, (float*)out, out_stride);
	else hipLaunchKernelGGL(k_ppng1_fwd<half_t>, grid, dim3(256), 0, stream, n, F, Q, C, R, log2_min, log2_max, x, (const half_t*)features, (half_t*)out, out_stride);
	if (out_stride > live) {
		const uint32_t total = n * (out_stride - live);
		if (fp32) hipLaunchKernelGGL(k_ppng_pad<float>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, out_stride, (float*)out);
		else hipLaunchKernelGGL(k_ppng_pad<half_t>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, out_stride, (half_t*)out);
	}
}

void ppng1_backward(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min, int32_t log2_max, MatView x, const void* features,
                    const void* dL_dy, uint32_t dy_stride, uint64_t* scratch, void* grad, bool accumulate) {
	const size_t n_params = (size_t)F * 2 * PPNG_D * C * Q * R;
	if (n_params == 0) return;
	if (n > 0) {
		const uint32_t slice = PPNG_D * C * Q * R;
		const bool in_lds = slice <= PPNG_LDS_ENTRIES;
		const uint32_t lds_bytes = in_lds ? slice * 8 : 0;
		const dim3 grid(div_round_up(n, PPNG_BWD_SAMPLES), F, 2);
		auto go = [&](auto kernel, auto* dy) {
			if (lds_bytes > 64 * 1024) HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
			hipLaunchKernelGGL(kernel, grid, dim3(256), lds_bytes, stream, n, F, Q, C, R, log2_min, log2_max, x, (const half_t*)features, dy, dy_stride, (unsigned long long*)scratch, in_lds ? 1 : 0);
			HIP_CHECK_THROW(hipGetLastError());
		};
		if (fp32) go(k_ppng1_bwd<float>, (const float*)dL_dy);
		else go(k_ppng1_bwd<half_t>, (const half_t*)dL_dy);
	}
	const uint32_t blocks = (uint32_t)((n_params + 255) / 256);
	if (fp32) hipLaunchKernelGGL(k_ppng_finalize<float>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (float*)grad, accumulate ? 1 : 0);
	else hipLaunchKernelGGL(k_ppng_finalize<half_t>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (half_t*)grad, accumulate ? 1 : 0);
}

void ppng2_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min, int32_t log2_max, MatView x, const void* features, void* out,
                   uint32_t out_stride) {
	if (n == 0 || out_stride == 0) return;
	CHECK_THROW(F >= 2 && Q >= 2 && R >= 1 && R <= PPNG_MAX_R);
	const dim3 grid(div_round_up(n, 256u), F, 2);
	const uint32_t live = F * 2 * C;
	if (fp32) hipLaunchKernelGGL(k_ppng2_fwd<float>, grid, dim3(256), 0, stream, n, F, Q, C, R, log2_min, log2_max, x, (const half_t*)features, (float*)out, out_stride);
	else hipLaunchKernelGGL(k_ppng2_fwd<half_t>, grid, dim3(256), 0, stream, n, F, Q, C, R, log2_min, log2_max, x, (const half_t*)features, (half_t*)out, out_stride);
	if (out_stride > live) {
		const uint32_t total = n * (out_stride - live);
		if (fp32) hipLaunchKernelGGL(k_ppng_pad<float>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, out_stride, (float*)out);
		else hipLaunchKernelGGL(k_ppng_pad<half_t>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, out_stride, (half_t*)out);
	}
}

void ppng2_backward(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min, int32_t log2_max, MatView x, const void* features,
                    const void* dL_dy, uint32_t dy_stride, uint64_t* scratch, void* grad, bool accumulate) {
	const size_t n_params = (size_t)F * 2 * PPNG_D * C * Q * Q * R;
	if (n_params == 0) return;
	if (n > 0) {
		const uint32_t plane = Q * Q * R;
		const bool in_lds = plane <= PPNG_LDS_ENTRIES;
		const uint32_t lds_bytes = in_lds ? plane * 8 : 0;
		// one workgroup per plane walks up to 64k samples: 6 x 2 x 3 x 4 = 144 workgroups with the defaults; more samples, more blocks per plane
		const uint32_t samples_per_block = 65536;
		const dim3 grid(div_round_up(n, samples_per_block), F * 2, PPNG_D * C);
		auto go = [&](auto kernel, auto* dy) {
			if (lds_bytes > 64 * 1024) HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
			hipLaunchKernelGGL(kernel, grid, dim3(256), lds_bytes, stream, n, F, Q, C, R, log2_min, log2_max, x, (const half_t*)features, dy, dy_stride, (unsigned long long*)scratch, in_lds ? 1 : 0,
			                   samples_per_block);
			HIP_CHECK_THROW(hipGetLastError());
		};
		if (fp32) go(k_ppng2_bwd<float>, (const float*)dL_dy);
		else go(k_ppng2_bwd<half_t>, (const half_t*)dL_dy);
	}
	const uint32_t blocks = (uint32_t)((n_params + 255) / 256);
	if (fp32) hipLaunchKernelGGL(k_ppng_finalize<float>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (float*)grad, accumulate ? 1 : 0);
	else hipLaunchKernelGGL(k_ppng_finalize<half_t>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (half_t*)grad, accumulate ? 1 : 0);
}

} // namespace tcnn_amd
