// grid_device.h -- device helpers shared by the grid encoding kernels (k_grid.hip, k_grid_scatter.hip).
// Restates include/tiny-cuda-nn/common_device.h:631-718 (hashing, grid_index), :825-868 (pos_fract) of the reference.
#pragma once

#include "tcnn_common.h"

#include <hip/hip_fp16.h>

namespace tcnn_amd {
namespace {

typedef _Float16 half_t;

template <typename T, int N> struct VecOf { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T> struct VecOf<T, 1> { typedef T type; };

__device__ inline float smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
__device__ inline float smoothstep_derivative(float v) { return 6 * v * (1.0f - v); }

// common_device.h:856-868
__device__ inline uint32_t pos_fract(float input, float scale, uint32_t interpolation, float* pos, float* pos_derivative) {
	float p = fmaf(scale, input, 0.5f);
	const float tmp = floorf(p);
	const uint32_t cell = (uint32_t)(int)tmp;
	p -= tmp;
	if (interpolation == (uint32_t)InterpolationType::Smoothstep) {
		*pos_derivative = smoothstep_derivative(p);
		*pos = smoothstep(p);
	} else {
		*pos_derivative = 1.0f;
		*pos = p;
	}
	return cell;
}

// pcg32 pieces needed by HashType::Rng (common_device.h:663-676)
__device__ inline uint32_t rng_hash_device(const uint32_t* pos, int n_dims) {
	const uint64_t MULT = 0x5851f42d4c957f2dULL;
	const uint32_t bits_per_dim = 64 / n_dims;
	uint64_t step = 0;
	for (int i = 0; i < n_dims; ++i) step ^= (uint64_t)pos[i] << (i * bits_per_dim);
	// pcg32{1337}: seed()
	uint64_t inc = (1ull << 1u) | 1u;
	uint64_t state = 0;
	state = state * MULT + inc;
	state += 1337ull;
	state = state * MULT + inc;
	// advance(step)
	uint64_t cur_mult = MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
	uint64_t delta = step;
	while (delta > 0) {
		if (delta & 1) {
			acc_mult *= cur_mult;
			acc_plus = acc_plus * cur_mult + cur_plus;
		}
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta /= 2;
	}
	state = acc_mult * state + acc_plus;
	// next_uint()
	const uint64_t old = state;
	const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	const uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

// grid_index (common_device.h:690-707) with the stride loop folded into GridLevel::stride / ::hashed on the host
// RNG = false: the caller has excluded HashType::Rng on the host (its pcg32 advance is a loop; k_grid_scatter_lists wants loop-free code between its loads and their uses)
template <int D, bool RNG = true>
__device__ inline uint32_t level_index(const GridLevel& lv, const uint32_t* primes, uint32_t hash_type, const uint32_t* cell) {
	uint32_t index;
	if (lv.hashed) {
		if (RNG && hash_type == (uint32_t)HashType::Rng) {
			index = rng_hash_device(cell, D);
		} else {
			index = 0;
#pragma unroll
			for (int d = 0; d < D; ++d) index ^= cell[d] * primes[d];
		}
	} else {
		index = 0;
#pragma unroll
		for (int d = 0; d < D; ++d) index += cell[d] * lv.stride[d];
	}
	if (lv.size_mask) return index & lv.size_mask;
	return index >= lv.size ? index % lv.size : index;
}

// coordinates of sample i; AoS inputs (stride_dim == 1) are read with one wide load where alignment allows
template <int D>
__device__ inline void load_coords(const MatView& x, uint32_t i, float (&out)[D]) {
	if (x.stride_dim == 1 && x.stride_sample == D && (D == 2 || D == 4)) {
		typedef float vecD __attribute__((ext_vector_type(D)));
		const vecD v = *(const vecD*)(x.data + (size_t)i * D);
#pragma unroll
		for (int d = 0; d < D; ++d) out[d] = v[d];
	} else {
#pragma unroll
		for (int d = 0; d < D; ++d) out[d] = x.data[(size_t)i * x.stride_sample + (size_t)d * x.stride_dim];
	}
}

// which chunk of the LDS scatter (k_grid_bwd_lds) owns entry `index` of this level
__device__ inline uint32_t scatter_chunk(const GridLevel& lv, uint32_t index) {
	return lv.scatter_shift != 0xffffffffu ? (index >> lv.scatter_shift) : (index / lv.scatter_per_chunk);
}

} // namespace
} // namespace tcnn_amd
