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
__global__ void __launch_bounds__(256) k_ppng_pad(const uint32_t n, const uint32_t first, const uint32_t out_stride, T* __restrict__ out, const float value = 1.0f) {
	const uint32_t pad = out_stride - first;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n * pad) return;
	const uint32_t b = gid / pad, j = gid - b * pad;
	out[(size_t)b * out_stride + first + j] = (T)value;
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

// N consecutive halves (one entry's ranks / one cell's features) as one 2 N-byte load when the table is aligned for it
template <uint32_t N>
__device__ inline void ppng_load_vec(const half_t* __restrict__ p, float (&v)[N], const bool aligned) {
	if (aligned) {
		typedef half_t hv __attribute__((ext_vector_type(N)));
		const hv h = *(const hv*)p;
#pragma unroll
		for (uint32_t i = 0; i < N; ++i) v[i] = (float)h[i];
	} else {
#pragma unroll
		for (uint32_t i = 0; i < N; ++i) v[i] = (float)p[i];
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

template <typename T, uint32_t R>
__global__ void __launch_bounds__(256) k_ppng2_fwd(const uint32_t n, const uint32_t F, const uint32_t Q, const uint32_t C, const int32_t log2_min, const int32_t log2_max,
                                                   const MatView x, const half_t* __restrict__ features, T* __restrict__ out, const uint32_t out_stride, const bool aligned) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	features += ((size_t)f * 2 + s) * PPNG_D * C * Q * Q * R;
	const Ppng2Lookup L = ppng2_lookup(ppng_freq(f, F, log2_min, log2_max), x, b, s, Q);
	for (uint32_t c = 0; c < C; ++c) {
		float sum[R]; // per rank: the corners in the order 000, 001, ..., 111 (z y x), ppng_2.h:62-70
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			float fx[R], fy[R], fz[R];
			ppng_load_vec<R>(features + ppng2_entry(L, 0, c, k, C, Q, R), fx, aligned);
			ppng_load_vec<R>(features + ppng2_entry(L, 1, c, k, C, Q, R), fy, aligned);
			ppng_load_vec<R>(features + ppng2_entry(L, 2, c, k, C, Q, R), fz, aligned);
#pragma unroll
			for (uint32_t r = 0; r < R; ++r) {
				const float v = fx[r] * fy[r] * fz[r];
				sum[r] = k == 0 ? L.w8[0] * v : sum[r] + L.w8[k] * v;
			}
		}
		float fs = 0;
#pragma unroll
		for (uint32_t r = 0; r < R; ++r) fs += sum[r];
		out[(size_t)b * out_stride + (size_t)f * 2 * C + s * C + c] = (T)fs;
	}
}

// ppng_2.h:78-272 for ONE plane of one feature of one (f, s): blockIdx = (sample block, f * 2 + s, pl * C + c).  The workgroup owns
// that plane's Q x Q x R entries in LDS when they fit.  The reference's loop over the three dimensions repeats the twelve additions
// of a sample unchanged three times (its body does not depend on the loop index, ppng_2.h:131-270): the gradient it produces is
// three times the derivative, and so is this one (3 x the fixed-point value: the same exact sum).
template <typename T, uint32_t R, uint32_t PL> // PL: the plane, a compile-time constant so that the corner tables stay in registers
__device__ inline void ppng2_bwd_plane(const uint32_t n, const uint32_t F, const uint32_t Q, const uint32_t C, const int32_t log2_min, const int32_t log2_max, const MatView x,
                                       const half_t* __restrict__ features, const T* __restrict__ dL_dy, const uint32_t dy_stride, unsigned long long* __restrict__ scratch,
                                       const int in_lds, const uint32_t samples_per_block, const bool aligned, const uint32_t c) {
	extern __shared__ __attribute__((aligned(16))) char ppng_smem[];
	typedef __attribute__((address_space(3))) unsigned long long lds_u64;
	lds_u64* acc = (lds_u64*)ppng_smem;
	const uint32_t f = blockIdx.y >> 1, s = blockIdx.y & 1u;
	const uint32_t plane = Q * Q * R;
	const size_t slice_off = ((size_t)f * 2 + s) * PPNG_D * C * plane;
	const uint32_t plane_off = (PL * C + c) * plane;
	const float freq = ppng_freq(f, F, log2_min, log2_max);
	features += slice_off;
	unsigned long long* global_acc = scratch + slice_off + plane_off;
	constexpr uint32_t o1 = PL == 0 ? 1u : 0u, o2 = PL == 2 ? 1u : 2u; // the other two planes, lower index first (ppng_2.h:176-187: w * f_lower * f_higher)
	constexpr uint32_t hi = PL == 2 ? 1u : 2u, lo = PL == 0 ? 1u : 0u, ex = PL; // the plane's axes; plane 0 excludes x (0), 1 excludes y, 2 excludes z
	const uint32_t begin = blockIdx.x * samples_per_block, end = min(n, begin + samples_per_block);
	for (uint32_t b = begin + threadIdx.x; b < end; b += blockDim.x) {
		const Ppng2Lookup L = ppng2_lookup(freq, x, b, s, Q);
		const float go = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
		// the plane's four entries under this sample: in-plane corner (hb, lb); the sum runs over the excluded axis' two bins
#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) {
			const uint32_t k0 = (((q >> 1) & 1u) << hi) | ((q & 1u) << lo), k1 = k0 | (1u << ex);
			float a0[R], b0[R], a1[R], b1[R];
			ppng_load_vec<R>(features + ppng2_entry(L, o1, c, k0, C, Q, R), a0, aligned);
			ppng_load_vec<R>(features + ppng2_entry(L, o2, c, k0, C, Q, R), b0, aligned);
			ppng_load_vec<R>(features + ppng2_entry(L, o1, c, k1, C, Q, R), a1, aligned);
			ppng_load_vec<R>(features + ppng2_entry(L, o2, c, k1, C, Q, R), b1, aligned);
			const uint32_t e = ppng2_entry(L, PL, c, k0, C, Q, R) - plane_off;
#pragma unroll
			for (uint32_t r = 0; r < R; ++r) {
				const float g = go * ((L.w8[k0] * a0[r] * b0[r]) + (L.w8[k1] * a1[r] * b1[r]));
				const unsigned long long v = (unsigned long long)(3 * half_to_fixed_fast((half_t)g));
				if (in_lds) __hip_atomic_fetch_add(acc + e + r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else atomicAdd(global_acc + e + r, v);
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

template <typename T, uint32_t R>
__global__ void __launch_bounds__(1024) k_ppng2_bwd(const uint32_t n, const uint32_t F, const uint32_t Q, const uint32_t C, const int32_t log2_min, const int32_t log2_max,
                                                    const MatView x, const half_t* __restrict__ features, const T* __restrict__ dL_dy, const uint32_t dy_stride,
                                                    unsigned long long* __restrict__ scratch, const int in_lds, const uint32_t samples_per_block, const bool aligned) {
	extern __shared__ __attribute__((aligned(16))) char ppng_smem[];
	if (in_lds) {
		for (uint32_t e = threadIdx.x; e < Q * Q * R; e += blockDim.x) ((unsigned long long*)ppng_smem)[e] = 0ull;
		__syncthreads();
	}
	const uint32_t pl = blockIdx.z / C, c = blockIdx.z - pl * C;
	if (pl == 0) ppng2_bwd_plane<T, R, 0>(n, F, Q, C, log2_min, log2_max, x, features, dL_dy, dy_stride, scratch, in_lds, samples_per_block, aligned, c);
	else if (pl == 1) ppng2_bwd_plane<T, R, 1>(n, F, Q, C, log2_min, log2_max, x, features, dL_dy, dy_stride, scratch, in_lds, samples_per_block, aligned, c);
	else ppng2_bwd_plane<T, R, 2>(n, F, Q, C, log2_min, log2_max, x, features, dL_dy, dy_stride, scratch, in_lds, samples_per_block, aligned, c);
}

// ---------------------------------------------------------------------------------------------------------------- PPNG3
// encodings/ppng_3.h:300-560 + interp.h:25-133: per (f, s) a full Q x Q x Q volume of C features, trilinear interpolation at
// (sc_0, sc_1, sc_2).  Parameters: half [F][2][Q^3][C], cell = p_0 + Q p_1 + Q^2 p_2.  The corner loop runs l = 0..7 with the bit of
// axis i at position 2 - i; the weight is ((1 * a_0) * a_1) * a_2.
struct Ppng3Lookup {
	uint32_t p[PPNG_D][2];
	float w[PPNG_D];
};
__device__ inline Ppng3Lookup ppng3_lookup(const float freq, const MatView x, const uint32_t b, const uint32_t s, const uint32_t Q) {
	Ppng3Lookup L;
#pragma unroll
	for (uint32_t i = 0; i < PPNG_D; ++i) ppng_bins(ppng_sc(freq, x.data[(size_t)b * x.stride_sample + (size_t)i * x.stride_dim], s), Q, L.p[i][0], L.p[i][1], L.w[i]);
	return L;
}
__device__ inline uint32_t ppng3_bit(const uint32_t l, const uint32_t i) { return (l >> (PPNG_D - 1 - i)) & 1u; }
__device__ inline uint32_t ppng3_cell(const Ppng3Lookup& L, const uint32_t l, const uint32_t Q) {
	return L.p[0][ppng3_bit(l, 0)] + Q * (L.p[1][ppng3_bit(l, 1)] + Q * L.p[2][ppng3_bit(l, 2)]);
}
__device__ inline float ppng3_weight(const Ppng3Lookup& L, const uint32_t l) {
	float weight = 1;
#pragma unroll
	for (uint32_t i = 0; i < PPNG_D; ++i) weight *= ppng3_bit(l, i) ? L.w[i] : 1 - L.w[i];
	return weight;
}
template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_ppng3_fwd(const uint32_t n, const uint32_t F, const uint32_t Q, const int32_t log2_min, const int32_t log2_max, const MatView x,
                                                   const half_t* __restrict__ features, T* __restrict__ out, const uint32_t out_stride, const bool aligned) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	features += ((size_t)f * 2 + s) * Q * Q * Q * C;
	const Ppng3Lookup L = ppng3_lookup(ppng_freq(f, F, log2_min, log2_max), x, b, s, Q);
	float results[C];
#pragma unroll
	for (uint32_t c = 0; c < C; ++c) results[c] = 0;
#pragma unroll
	for (uint32_t l = 0; l < 8; ++l) {
		float v[C];
		ppng_load_vec<C>(features + (size_t)ppng3_cell(L, l, Q) * C, v, aligned);
		const float weight = ppng3_weight(L, l);
#pragma unroll
		for (uint32_t c = 0; c < C; ++c) results[c] += v[c] * weight;
	}
#pragma unroll
	for (uint32_t c = 0; c < C; ++c) out[(size_t)b * out_stride + (size_t)f * 2 * C + s * C + c] = (T)results[c];
}

// interp.h:74-133 (the reference: packed fp16 atomics into the volume, 8 C per sample and (f, s)).  Here a workgroup OWNS one
// layer p_2 = blockIdx.z of the volume of one (f, s) -- Q x Q x C exact 64-bit accumulators in LDS (16384 with the defaults).
// k_ppng3_zbins writes the third coordinate's bin of every (f, s, sample) once (2 bytes); an owner reads those, each wave compacts
// the samples that touch its layer (2 of Q on average) into a queue in LDS and evaluates full batches of 64.  No global atomics
// except the merge of sample blocks; same products, exact sum, deterministic.
// Volumes whose layer does not fit the LDS: k_ppng3_bwd_atomic, 64-bit global atomics.
__host__ __device__ inline size_t ppng3_zbins_row(const uint32_t n) { return ((size_t)n + 7u) & ~(size_t)7u; } // 16-byte rows
__global__ void __launch_bounds__(256) k_ppng3_zbins(const uint32_t n, const uint32_t F, const uint32_t Q, const int32_t log2_min, const int32_t log2_max, const MatView x, uint16_t* __restrict__ zbins) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	uint32_t p0, p1;
	float w;
	ppng_bins(ppng_sc(ppng_freq(f, F, log2_min, log2_max), x.data[(size_t)b * x.stride_sample + (size_t)2 * x.stride_dim], s), Q, p0, p1, w);
	zbins[((size_t)f * 2 + s) * ppng3_zbins_row(n) + b] = (uint16_t)((p0 << 1) | (p1 != p0 ? 1u : 0u));
}

constexpr uint32_t PPNG3_OWNER_THREADS = 1024;
constexpr uint32_t PPNG3_QUEUE = 128; // per wave: up to 63 left over + 64 new

template <typename T, uint32_t C>
__global__ void __launch_bounds__(PPNG3_OWNER_THREADS) k_ppng3_bwd_owner(const uint32_t n, const uint32_t F, const uint32_t Q, const int32_t log2_min, const int32_t log2_max, const MatView x,
                                                                         const T* __restrict__ dL_dy, const uint32_t dy_stride, const uint16_t* __restrict__ zbins,
                                                                         unsigned long long* __restrict__ scratch, const uint32_t samples_per_block) {
	extern __shared__ __attribute__((aligned(16))) char ppng_smem[];
	__shared__ uint32_t queues[PPNG3_OWNER_THREADS / 64][PPNG3_QUEUE];
	typedef __attribute__((address_space(3))) unsigned long long lds_u64;
	lds_u64* acc = (lds_u64*)ppng_smem;
	const uint32_t f = blockIdx.y >> 1, s = blockIdx.y & 1u;
	// sc = sin(..) piles the samples up at the ends of the range (the two outermost layers see ~8 % of them each, the mean is 2 / Q):
	// the outermost layers are dispatched first, the middle ones last
	const uint32_t layer = (blockIdx.z & 1u) ? Q - 1 - (blockIdx.z >> 1) : (blockIdx.z >> 1);
	const uint32_t layer_entries = Q * Q * C;
	for (uint32_t e = threadIdx.x; e < layer_entries; e += blockDim.x) ((unsigned long long*)ppng_smem)[e] = 0ull;
	__syncthreads();
	const float freq = ppng_freq(f, F, log2_min, log2_max);
	zbins += ((size_t)f * 2 + s) * ppng3_zbins_row(n);
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t* queue = queues[threadIdx.x >> 6];
	auto add = [&](const uint32_t b) {
		const Ppng3Lookup L = ppng3_lookup(freq, x, b, s, Q);
		float go[C];
#pragma unroll
		for (uint32_t c = 0; c < C; ++c) go[c] = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
#pragma unroll
		for (uint32_t l = 0; l < 8; ++l) {
			if (L.p[2][ppng3_bit(l, 2)] != layer) continue;
			const float weight = ppng3_weight(L, l);
			const uint32_t e = (L.p[0][ppng3_bit(l, 0)] + Q * L.p[1][ppng3_bit(l, 1)]) * C;
#pragma unroll
			for (uint32_t c = 0; c < C; ++c)
				__hip_atomic_fetch_add(acc + e + c, (unsigned long long)half_to_fixed_fast((half_t)(go[c] * weight)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	};
	const uint32_t begin = blockIdx.x * samples_per_block, end = min(n, begin + samples_per_block);
	uint32_t queued = 0; // wave-uniform
	// a lane reads the bins of 8 consecutive samples at once (16 bytes; the rows are padded to 8 samples), the next 8 are in flight
	// while these are sorted into the queue
	constexpr uint32_t PER_WAVE = 64 * 8;
	const uint32_t stride = (blockDim.x >> 6) * PER_WAVE;
	auto load8 = [&](const uint32_t b0) { return b0 < end ? *(const uint4*)(zbins + b0) : make_uint4(0, 0, 0, 0); };
	uint32_t base = begin + (threadIdx.x >> 6) * PER_WAVE;
	uint4 z_next = load8(base + lane * 8);
	for (; base < end; base += stride) {
		const uint32_t b0 = base + lane * 8;
		const uint4 z = z_next;
		z_next = load8(b0 + stride);
		const unsigned long long lo = (unsigned long long)z.x | ((unsigned long long)z.y << 32), hi = (unsigned long long)z.z | ((unsigned long long)z.w << 32);
		for (uint32_t j = 0; j < 8; ++j) {
			const uint32_t zz = (uint32_t)((j < 4 ? lo : hi) >> (16 * (j & 3u))) & 0xffffu, p0 = zz >> 1;
			const bool hit = b0 + j < end && (p0 == layer || p0 + (zz & 1u) == layer);
			const unsigned long long ballot = __ballot(hit);
			if (ballot == 0) continue;
			const uint32_t blo = (uint32_t)ballot, bhi = (uint32_t)(ballot >> 32);
			if (hit) queue[queued + __builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0))] = b0 + j;
			queued += __builtin_popcount(blo) + __builtin_popcount(bhi);
			if (queued >= 64) {
				// the queue is wave-private and LDS serves one wave's instructions in order: compiler barriers suffice
				__atomic_signal_fence(__ATOMIC_SEQ_CST);
				__builtin_amdgcn_wave_barrier();
				const uint32_t id = queue[lane], rest = queue[64 + lane];
				__builtin_amdgcn_wave_barrier();
				queued -= 64;
				if (lane < queued) queue[lane] = rest;
				__atomic_signal_fence(__ATOMIC_SEQ_CST);
				add(id);
			}
		}
	}
	__atomic_signal_fence(__ATOMIC_SEQ_CST);
	__builtin_amdgcn_wave_barrier();
	if (lane < queued) add(queue[lane]);
	__syncthreads();
	unsigned long long* global_acc = scratch + ((size_t)f * 2 + s) * Q * Q * Q * C + (size_t)layer * layer_entries;
	if (gridDim.x == 1) { // the only workgroup of this layer: plain stores of the sums (the scratch holds zeros)
		for (uint32_t e = threadIdx.x; e < layer_entries; e += blockDim.x) {
			const unsigned long long v = ((unsigned long long*)ppng_smem)[e];
			if (v != 0) global_acc[e] = v;
		}
	} else {
		for (uint32_t e = threadIdx.x; e < layer_entries; e += blockDim.x) {
			const unsigned long long v = ((unsigned long long*)ppng_smem)[e];
			if (v != 0) atomicAdd(global_acc + e, v);
		}
	}
}

template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_ppng3_bwd_atomic(const uint32_t n, const uint32_t F, const uint32_t Q, const int32_t log2_min, const int32_t log2_max, const MatView x,
                                                          const T* __restrict__ dL_dy, const uint32_t dy_stride, unsigned long long* __restrict__ scratch) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	unsigned long long* global_acc = scratch + ((size_t)f * 2 + s) * Q * Q * Q * C;
	const Ppng3Lookup L = ppng3_lookup(ppng_freq(f, F, log2_min, log2_max), x, b, s, Q);
	float go[C];
#pragma unroll
	for (uint32_t c = 0; c < C; ++c) go[c] = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
#pragma unroll
	for (uint32_t l = 0; l < 8; ++l) {
		const float weight = ppng3_weight(L, l);
		const size_t e = (size_t)ppng3_cell(L, l, Q) * C;
#pragma unroll
		for (uint32_t c = 0; c < C; ++c) atomicAdd(global_acc + e + c, (unsigned long long)half_to_fixed_fast((half_t)(go[c] * weight)));
	}
}

// ppng_3.h:13-84 (grad_point_helper) + :350-385: dL/dx_k = sum_{f, s, c} dL/dy * sum_corners feature * prod_i (i == k ? +-dw_i : a_i), dw_i = d sc_i / dx_i * (Q - 1) / 2.
// The reference adds the F * 2 * C terms of a sample with float atomics in arbitrary order; here one thread owns the sample and adds them
// in the order f, s, c.
template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_ppng3_bwd_input(const uint32_t n, const uint32_t F, const uint32_t Q, const int32_t log2_min, const int32_t log2_max, const MatView x,
                                                         const half_t* __restrict__ features, const T* __restrict__ dL_dy, const uint32_t dy_stride, const MatViewMut dL_dx, const bool aligned) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	float px[PPNG_D], sum[PPNG_D] = {0, 0, 0};
#pragma unroll
	for (uint32_t i = 0; i < PPNG_D; ++i) px[i] = x.data[(size_t)b * x.stride_sample + (size_t)i * x.stride_dim];
	for (uint32_t f = 0; f < F; ++f) {
		const float freq = ppng_freq(f, F, log2_min, log2_max);
		for (uint32_t s = 0; s < 2; ++s) {
			const half_t* vol = features + ((size_t)f * 2 + s) * Q * Q * Q * C;
			Ppng3Lookup L;
			float dw[PPNG_D];
#pragma unroll
			for (uint32_t i = 0; i < PPNG_D; ++i) {
				const float arg = (float)((double)freq * ((double)px[i] - 0.5) + (double)s * 1.57079632679489661923);
				ppng_bins(sinf(arg), Q, L.p[i][0], L.p[i][1], L.w[i]);
				const float dsc = cosf(arg) * freq;
				dw[i] = (float)(((double)dsc * 0.5) * (double)(Q - 1));
			}
			float results[PPNG_D][C];
#pragma unroll
			for (uint32_t k = 0; k < PPNG_D; ++k)
#pragma unroll
				for (uint32_t c = 0; c < C; ++c) results[k][c] = 0;
#pragma unroll
			for (uint32_t l = 0; l < 8; ++l) {
				float weights[PPNG_D] = {1, 1, 1};
#pragma unroll
				for (uint32_t i = 0; i < PPNG_D; ++i)
#pragma unroll
					for (uint32_t k = 0; k < PPNG_D; ++k) weights[k] *= (i == k) ? (ppng3_bit(l, i) ? dw[i] : -dw[i]) : (ppng3_bit(l, i) ? L.w[i] : 1 - L.w[i]);
				float v[C];
				ppng_load_vec<C>(vol + (size_t)ppng3_cell(L, l, Q) * C, v, aligned);
#pragma unroll
				for (uint32_t c = 0; c < C; ++c)
#pragma unroll
					for (uint32_t k = 0; k < PPNG_D; ++k) results[k][c] += v[c] * weights[k];
			}
#pragma unroll
			for (uint32_t c = 0; c < C; ++c) {
				const float go = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
#pragma unroll
				for (uint32_t k = 0; k < PPNG_D; ++k) sum[k] += go * results[k][c];
			}
		}
	}
#pragma unroll
	for (uint32_t k = 0; k < PPNG_D; ++k) dL_dx.data[(size_t)b * dL_dx.stride_sample + (size_t)k * dL_dx.stride_dim] = sum[k];
}

// ppng_3.h:86-198 (grad_grad_helper) + :387-428: the second-order pass with respect to the parameters and dL/dy.  With v = dL_ddLdx
// (one 3-vector per sample): g2f(corner) = sum_k prod_i (i == k ? +-dw_i : a_i) v_k;  parameter gradient += (half)(dL/dy_c g2f) per
// corner (64-bit exact sums in the scratch, as in the first-order pass; the reference: packed fp16 atomics);  dL_ddLdy_c = sum_k
// sum_corners feature_c weights_k v_k.
template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_ppng3_bwdbwd(const uint32_t n, const uint32_t F, const uint32_t Q, const int32_t log2_min, const int32_t log2_max, const MatView x, const MatView dL_ddLdx,
                                                      const half_t* __restrict__ features, const T* __restrict__ dL_dy, const uint32_t dy_stride, unsigned long long* __restrict__ scratch,
                                                      T* __restrict__ dL_ddLdy, const bool aligned) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	const uint32_t f = blockIdx.y, s = blockIdx.z;
	const float freq = ppng_freq(f, F, log2_min, log2_max);
	const size_t vol_off = ((size_t)f * 2 + s) * Q * Q * Q * C;
	Ppng3Lookup L;
	float dw[PPNG_D], dps[PPNG_D];
#pragma unroll
	for (uint32_t i = 0; i < PPNG_D; ++i) {
		const float arg = (float)((double)freq * ((double)x.data[(size_t)b * x.stride_sample + (size_t)i * x.stride_dim] - 0.5) + (double)s * 1.57079632679489661923);
		ppng_bins(sinf(arg), Q, L.p[i][0], L.p[i][1], L.w[i]);
		dw[i] = (float)(((double)(cosf(arg) * freq) * 0.5) * (double)(Q - 1));
		dps[i] = dL_ddLdx.data[(size_t)b * dL_ddLdx.stride_sample + (size_t)i * dL_ddLdx.stride_dim];
	}
	float go[C], results[PPNG_D][C];
#pragma unroll
	for (uint32_t c = 0; c < C; ++c) {
		go[c] = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
#pragma unroll
		for (uint32_t k = 0; k < PPNG_D; ++k) results[k][c] = 0;
	}
#pragma unroll
	for (uint32_t l = 0; l < 8; ++l) {
		float weights[PPNG_D] = {1, 1, 1};
#pragma unroll
		for (uint32_t i = 0; i < PPNG_D; ++i)
#pragma unroll
			for (uint32_t k = 0; k < PPNG_D; ++k) weights[k] *= (i == k) ? (ppng3_bit(l, i) ? dw[i] : -dw[i]) : (ppng3_bit(l, i) ? L.w[i] : 1 - L.w[i]);
		const size_t cell = (size_t)ppng3_cell(L, l, Q) * C;
		if (dL_ddLdy) {
			float v[C];
			ppng_load_vec<C>(features + vol_off + cell, v, aligned);
#pragma unroll
			for (uint32_t c = 0; c < C; ++c)
#pragma unroll
				for (uint32_t k = 0; k < PPNG_D; ++k) results[k][c] += v[c] * weights[k] * dps[k];
		}
		if (scratch) {
			float g2f = 0;
#pragma unroll
			for (uint32_t k = 0; k < PPNG_D; ++k) g2f += weights[k] * dps[k];
#pragma unroll
			for (uint32_t c = 0; c < C; ++c) atomicAdd(scratch + vol_off + cell + c, (unsigned long long)half_to_fixed_fast((half_t)(go[c] * g2f)));
		}
	}
	if (dL_ddLdy) {
#pragma unroll
		for (uint32_t c = 0; c < C; ++c) {
			float ggo = 0;
#pragma unroll
			for (uint32_t k = 0; k < PPNG_D; ++k) ggo += results[k][c];
			dL_ddLdy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c] = (T)ggo;
		}
	}
}

// ppng_3.h:200-275 (grad2_points_helper) + :430-473: the second-order pass with respect to the input.  Per corner and axis i:
// weights_i = sum_j v_j prod_k (j == i ? (k == i ? +-ddw_k : a_k) : (k == i or k == j ? +-dw_k : a_k)), ddw = d2 sc / dx2 (Q - 1) / 2;
// dL/dx_i = sum_{f, s, c} dL/dy_c sum_corners feature_c weights_i.  One thread per sample, terms added in the order f, s, c (the reference:
// float atomics).
template <typename T, uint32_t C>
__global__ void __launch_bounds__(256) k_ppng3_bwdbwd_input(const uint32_t n, const uint32_t F, const uint32_t Q, const int32_t log2_min, const int32_t log2_max, const MatView x,
                                                            const MatView dL_ddLdx, const half_t* __restrict__ features, const T* __restrict__ dL_dy, const uint32_t dy_stride,
                                                            const MatViewMut dL_dx, const bool aligned) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n) return;
	float px[PPNG_D], dps[PPNG_D], sum[PPNG_D] = {0, 0, 0};
#pragma unroll
	for (uint32_t i = 0; i < PPNG_D; ++i) {
		px[i] = x.data[(size_t)b * x.stride_sample + (size_t)i * x.stride_dim];
		dps[i] = dL_ddLdx.data[(size_t)b * dL_ddLdx.stride_sample + (size_t)i * dL_ddLdx.stride_dim];
	}
	for (uint32_t f = 0; f < F; ++f) {
		const float freq = ppng_freq(f, F, log2_min, log2_max);
		for (uint32_t s = 0; s < 2; ++s) {
			const half_t* vol = features + ((size_t)f * 2 + s) * Q * Q * Q * C;
			Ppng3Lookup L;
			float dw[PPNG_D], ddw[PPNG_D];
#pragma unroll
			for (uint32_t i = 0; i < PPNG_D; ++i) {
				const float arg = (float)((double)freq * ((double)px[i] - 0.5) + (double)s * 1.57079632679489661923);
				const float sn = sinf(arg);
				ppng_bins(sn, Q, L.p[i][0], L.p[i][1], L.w[i]);
				dw[i] = (float)(((double)(cosf(arg) * freq) * 0.5) * (double)(Q - 1));
				ddw[i] = (float)((double)(-sn * freq * freq) * (0.5 * (double)(Q - 1)));
			}
			float results[PPNG_D][C];
#pragma unroll
			for (uint32_t k = 0; k < PPNG_D; ++k)
#pragma unroll
				for (uint32_t c = 0; c < C; ++c) results[k][c] = 0;
#pragma unroll
			for (uint32_t l = 0; l < 8; ++l) {
				float weights[PPNG_D] = {0, 0, 0};
#pragma unroll
				for (uint32_t i = 0; i < PPNG_D; ++i) {
#pragma unroll
					for (uint32_t j = 0; j < PPNG_D; ++j) {
						float weight = 1;
#pragma unroll
						for (uint32_t k = 0; k < PPNG_D; ++k) {
							const bool bit = ppng3_bit(l, k) != 0;
							if (j == i) weight *= (k == i) ? (bit ? ddw[k] : -ddw[k]) : (bit ? L.w[k] : 1 - L.w[k]);
							else weight *= (k == i || k == j) ? (bit ? dw[k] : -dw[k]) : (bit ? L.w[k] : 1 - L.w[k]);
						}
						weights[i] += weight * dps[j];
					}
				}
				float v[C];
				ppng_load_vec<C>(vol + (size_t)ppng3_cell(L, l, Q) * C, v, aligned);
#pragma unroll
				for (uint32_t c = 0; c < C; ++c)
#pragma unroll
					for (uint32_t k = 0; k < PPNG_D; ++k) results[k][c] += v[c] * weights[k];
			}
#pragma unroll
			for (uint32_t c = 0; c < C; ++c) {
				const float go = (float)dL_dy[(size_t)b * dy_stride + (size_t)f * 2 * C + s * C + c];
#pragma unroll
				for (uint32_t k = 0; k < PPNG_D; ++k) sum[k] += go * results[k][c];
			}
		}
	}
#pragma unroll
	for (uint32_t k = 0; k < PPNG_D; ++k) dL_dx.data[(size_t)b * dL_dx.stride_sample + (size_t)k * dL_dx.stride_dim] = sum[k];
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

namespace {
template <uint32_t R>
void ppng2_forward_r(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min, int32_t log2_max, MatView x, const void* features, void* out, uint32_t out_stride) {
	const dim3 grid(div_round_up(n, 256u), F, 2);
	const bool aligned = (uintptr_t)features % (R * sizeof(half_t)) == 0;
	if (fp32) hipLaunchKernelGGL((k_ppng2_fwd<float, R>), grid, dim3(256), 0, stream, n, F, Q, C, log2_min, log2_max, x, (const half_t*)features, (float*)out, out_stride, aligned);
	else hipLaunchKernelGGL((k_ppng2_fwd<half_t, R>), grid, dim3(256), 0, stream, n, F, Q, C, log2_min, log2_max, x, (const half_t*)features, (half_t*)out, out_stride, aligned);
}

template <uint32_t R>
void ppng2_backward_r(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min, int32_t log2_max, MatView x, const void* features, const void* dL_dy,
                      uint32_t dy_stride, uint64_t* scratch) {
	const uint32_t plane = Q * Q * R;
	const bool in_lds = plane <= PPNG_LDS_ENTRIES;
	const uint32_t lds_bytes = in_lds ? plane * 8 : 0;
	const bool aligned = (uintptr_t)features % (R * sizeof(half_t)) == 0;
	// one workgroup per plane walks up to 64k samples: 6 x 2 x 3 x 4 = 144 workgroups with the defaults; more samples, more blocks per plane
	const uint32_t samples_per_block = 65536;
	const dim3 grid(div_round_up(n, samples_per_block), F * 2, PPNG_D * C);
	auto go = [&](auto kernel, auto* dy) {
		if (lds_bytes > 48 * 1024) HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
		hipLaunchKernelGGL(kernel, grid, dim3(1024), lds_bytes, stream, n, F, Q, C, log2_min, log2_max, x, (const half_t*)features, dy, dy_stride, (unsigned long long*)scratch, in_lds ? 1 : 0,
		                   samples_per_block, aligned);
		HIP_CHECK_THROW(hipGetLastError());
	};
	if (fp32) go(k_ppng2_bwd<float, R>, (const float*)dL_dy);
	else go(k_ppng2_bwd<half_t, R>, (const half_t*)dL_dy);
}
} // namespace

#define PPNG2_BY_R(fn, ...)                                  \
	switch (R) {                                             \
		case 2: fn<2>(__VA_ARGS__); break;                   \
		case 4: fn<4>(__VA_ARGS__); break;                   \
		case 8: fn<8>(__VA_ARGS__); break;                   \
		case 16: fn<16>(__VA_ARGS__); break;                 \
		default: throw std::runtime_error{"PPNG2: rank must be 2, 4, 8 or 16"}; \
	}

void ppng2_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min, int32_t log2_max, MatView x, const void* features, void* out,
                   uint32_t out_stride) {
	if (n == 0 || out_stride == 0) return;
	CHECK_THROW(F >= 2 && Q >= 2);
	PPNG2_BY_R(ppng2_forward_r, stream, fp32, n, F, Q, C, log2_min, log2_max, x, features, out, out_stride);
	const uint32_t live = F * 2 * C;
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
	if (n > 0) { PPNG2_BY_R(ppng2_backward_r, stream, fp32, n, F, Q, C, log2_min, log2_max, x, features, dL_dy, dy_stride, scratch); }
	const uint32_t blocks = (uint32_t)((n_params + 255) / 256);
	if (fp32) hipLaunchKernelGGL(k_ppng_finalize<float>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (float*)grad, accumulate ? 1 : 0);
	else hipLaunchKernelGGL(k_ppng_finalize<half_t>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (half_t*)grad, accumulate ? 1 : 0);
}
#undef PPNG2_BY_R

namespace {
bool ppng3_owner_form(uint32_t Q, uint32_t C) { return Q * Q * C <= PPNG_LDS_ENTRIES && Q < 32768; }

template <uint32_t C>
void ppng3_forward_c(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, int32_t log2_min, int32_t log2_max, MatView x, const void* features, void* out, uint32_t out_stride) {
	const dim3 grid(div_round_up(n, 256u), F, 2);
	const bool aligned = (uintptr_t)features % (C * sizeof(half_t)) == 0;
	if (fp32) hipLaunchKernelGGL((k_ppng3_fwd<float, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, (const half_t*)features, (float*)out, out_stride, aligned);
	else hipLaunchKernelGGL((k_ppng3_fwd<half_t, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, (const half_t*)features, (half_t*)out, out_stride, aligned);
}

template <uint32_t C>
void ppng3_backward_c(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, int32_t log2_min, int32_t log2_max, MatView x, const void* dL_dy, uint32_t dy_stride, uint16_t* zbins,
                      uint64_t* scratch) {
	const uint32_t layer_entries = Q * Q * C;
	if (ppng3_owner_form(Q, C)) {
		CHECK_THROW(zbins != nullptr);
		hipLaunchKernelGGL(k_ppng3_zbins, dim3(div_round_up(n, 256u), F, 2), dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, zbins);
		// F * 2 * Q workgroups (768 with the defaults), each walking the bins of up to 2^20 samples
		const uint32_t samples_per_block = 1u << 20, lds_bytes = layer_entries * 8;
		const dim3 grid(div_round_up(n, samples_per_block), F * 2, Q);
		auto go = [&](auto kernel, auto* dy) {
			if (lds_bytes > 48 * 1024) HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
			hipLaunchKernelGGL(kernel, grid, dim3(PPNG3_OWNER_THREADS), lds_bytes, stream, n, F, Q, log2_min, log2_max, x, dy, dy_stride, (const uint16_t*)zbins, (unsigned long long*)scratch, samples_per_block);
			HIP_CHECK_THROW(hipGetLastError());
		};
		if (fp32) go(k_ppng3_bwd_owner<float, C>, (const float*)dL_dy);
		else go(k_ppng3_bwd_owner<half_t, C>, (const half_t*)dL_dy);
	} else {
		const dim3 grid(div_round_up(n, 256u), F, 2);
		if (fp32) hipLaunchKernelGGL((k_ppng3_bwd_atomic<float, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, (const float*)dL_dy, dy_stride, (unsigned long long*)scratch);
		else hipLaunchKernelGGL((k_ppng3_bwd_atomic<half_t, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, (const half_t*)dL_dy, dy_stride, (unsigned long long*)scratch);
		HIP_CHECK_THROW(hipGetLastError());
	}
}

template <uint32_t C>
void ppng3_backward_input_c(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, int32_t log2_min, int32_t log2_max, MatView x, const void* features, const void* dL_dy,
                            uint32_t dy_stride, MatViewMut dL_dx) {
	const dim3 grid(div_round_up(n, 256u));
	const bool aligned = (uintptr_t)features % (C * sizeof(half_t)) == 0;
	if (fp32) hipLaunchKernelGGL((k_ppng3_bwd_input<float, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, (const half_t*)features, (const float*)dL_dy, dy_stride, dL_dx, aligned);
	else hipLaunchKernelGGL((k_ppng3_bwd_input<half_t, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, (const half_t*)features, (const half_t*)dL_dy, dy_stride, dL_dx, aligned);
}
} // namespace

#define PPNG3_BY_C(fn, ...)                                          \
	switch (C) {                                                     \
		case 2: fn<2>(__VA_ARGS__); break;                           \
		case 4: fn<4>(__VA_ARGS__); break;                           \
		case 8: fn<8>(__VA_ARGS__); break;                           \
		default: throw std::runtime_error{"PPNG3: n_features must be 2, 4 or 8 in this build"}; \
	}

void ppng3_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min, int32_t log2_max, MatView x, const void* features, void* out,
                   uint32_t out_stride) {
	if (n == 0 || out_stride == 0) return;
	CHECK_THROW(F >= 2 && Q >= 2 && (uint64_t)Q * Q * Q * C * 2 * F < (1ull << 32));
	PPNG3_BY_C(ppng3_forward_c, stream, fp32, n, F, Q, log2_min, log2_max, x, features, out, out_stride);
	const uint32_t live = F * 2 * C;
	if (out_stride > live) {
		const uint32_t total = n * (out_stride - live);
		if (fp32) hipLaunchKernelGGL(k_ppng_pad<float>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, out_stride, (float*)out);
		else hipLaunchKernelGGL(k_ppng_pad<half_t>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, out_stride, (half_t*)out);
	}
}

size_t ppng3_backward_workspace_bytes(uint32_t n, uint32_t F, uint32_t Q, uint32_t C) { return ppng3_owner_form(Q, C) ? ppng3_zbins_row(n) * F * 2 * sizeof(uint16_t) : 0; }

void ppng3_backward(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min, int32_t log2_max, MatView x, const void* dL_dy, uint32_t dy_stride,
                    void* workspace, uint64_t* scratch, void* grad, bool accumulate) {
	const size_t n_params = (size_t)F * 2 * Q * Q * Q * C;
	if (n_params == 0) return;
	if (n > 0) { PPNG3_BY_C(ppng3_backward_c, stream, fp32, n, F, Q, log2_min, log2_max, x, dL_dy, dy_stride, (uint16_t*)workspace, scratch); }
	const uint32_t blocks = (uint32_t)((n_params + 255) / 256);
	if (fp32) hipLaunchKernelGGL(k_ppng_finalize<float>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (float*)grad, accumulate ? 1 : 0);
	else hipLaunchKernelGGL(k_ppng_finalize<half_t>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (half_t*)grad, accumulate ? 1 : 0);
}

void ppng3_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min, int32_t log2_max, MatView x, const void* features,
                          const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx) {
	if (n == 0) return;
	PPNG3_BY_C(ppng3_backward_input_c, stream, fp32, n, F, Q, log2_min, log2_max, x, features, dL_dy, dy_stride, dL_dx);
}
namespace {
template <uint32_t C>
void ppng3_bwdbwd_c(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, int32_t log2_min, int32_t log2_max, MatView x, MatView dL_ddLdx, const void* features, const void* dL_dy,
                    uint32_t dy_stride, uint64_t* scratch, void* dL_ddLdy, MatViewMut* dL_dx) {
	const bool aligned = (uintptr_t)features % (C * sizeof(half_t)) == 0;
	if (scratch || dL_ddLdy) {
		const dim3 grid(div_round_up(n, 256u), F, 2);
		if (fp32) hipLaunchKernelGGL((k_ppng3_bwdbwd<float, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, dL_ddLdx, (const half_t*)features, (const float*)dL_dy, dy_stride, (unsigned long long*)scratch, (float*)dL_ddLdy, aligned);
		else hipLaunchKernelGGL((k_ppng3_bwdbwd<half_t, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, dL_ddLdx, (const half_t*)features, (const half_t*)dL_dy, dy_stride, (unsigned long long*)scratch, (half_t*)dL_ddLdy, aligned);
	}
	if (dL_dx) {
		const dim3 grid(div_round_up(n, 256u));
		if (fp32) hipLaunchKernelGGL((k_ppng3_bwdbwd_input<float, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, dL_ddLdx, (const half_t*)features, (const float*)dL_dy, dy_stride, *dL_dx, aligned);
		else hipLaunchKernelGGL((k_ppng3_bwdbwd_input<half_t, C>), grid, dim3(256), 0, stream, n, F, Q, log2_min, log2_max, x, dL_ddLdx, (const half_t*)features, (const half_t*)dL_dy, dy_stride, *dL_dx, aligned);
	}
	HIP_CHECK_THROW(hipGetLastError());
}
} // namespace

void ppng3_backward_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min, int32_t log2_max, MatView x, MatView dL_ddLdx, const void* features,
                                   const void* dL_dy, uint32_t dy_stride, uint64_t* scratch, void* grad, bool accumulate, void* dL_ddLdy, MatViewMut* dL_dx) {
	if (n > 0) { PPNG3_BY_C(ppng3_bwdbwd_c, stream, fp32, n, F, Q, log2_min, log2_max, x, dL_ddLdx, features, dL_dy, dy_stride, grad ? scratch : nullptr, dL_ddLdy, dL_dx); }
	const uint32_t live = F * 2 * C;
	if (n > 0 && dL_ddLdy && dy_stride > live) { // padding columns: zero (the reference leaves them unwritten)
		const uint32_t total = n * (dy_stride - live);
		if (fp32) hipLaunchKernelGGL(k_ppng_pad<float>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, dy_stride, (float*)dL_ddLdy, 0.0f);
		else hipLaunchKernelGGL(k_ppng_pad<half_t>, dim3(div_round_up(total, 256u)), dim3(256), 0, stream, n, live, dy_stride, (half_t*)dL_ddLdy, 0.0f);
	}
	if (grad) {
		const size_t n_params = (size_t)F * 2 * Q * Q * Q * C;
		const uint32_t blocks = (uint32_t)((n_params + 255) / 256);
		if (fp32) hipLaunchKernelGGL(k_ppng_finalize<float>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (float*)grad, accumulate ? 1 : 0);
		else hipLaunchKernelGGL(k_ppng_finalize<half_t>, dim3(blocks), dim3(256), 0, stream, n_params, (unsigned long long*)scratch, (half_t*)grad, accumulate ? 1 : 0);
	}
}
#undef PPNG3_BY_C

} // namespace tcnn_amd
