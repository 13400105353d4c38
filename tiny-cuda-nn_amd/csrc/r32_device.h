// r32_device.h -- device helpers of the fused MLP training kernels built on v_mfma_f32_32x32x16_f16 (k_train_r32.hip, k_train_r32ob.hip):
// the matrix instruction, result tile -> next operand, ReLU and its derivative on packed halves.  Operand maps: mlp_side_jobs.h (R32Frags).
#pragma once

#include "mlp_device.h"

namespace tcnn_amd {
namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline f16v mfma32(const h8 a, const h8 b, const f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ inline f16v zero16() {
	f16v z;
#pragma unroll
	for (int i = 0; i < 16; ++i) z[i] = 0.0f;
	return z;
}
// result registers 8 s .. 8 s + 7 -> the B fragment of the next layer's k-step (round to nearest even, like the reference's fp16 accumulators are read)
__device__ inline h8 pack8(const f16v v, const int s) {
	const f8v t = {v[8 * s + 0], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6], v[8 * s + 7]};
	return __builtin_convertvector(t, h8);
}
// ReLU(half) = x > 0 ? x : 0 (common_device.h:92-98) as a signed integer maximum of the bit patterns (never -0: the backward pass tells
// "positive" from "zero" by the bits)
__device__ inline h8 relu8(const h8 v) { return __builtin_bit_cast(h8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), s16x8{0, 0, 0, 0, 0, 0, 0, 0})); }
// ... and its derivative from the forward output (common_device.h:241-297): the gradient where the output has any bit set, +0 elsewhere
// (min(bits, 1) = 0 / 1, times the gradient's bits as an integer product)
template <bool AND_FORM = false> __device__ inline h8 relu_bwd8(const h8 g, const h8 fwd) {
	const u32x4 f = __builtin_bit_cast(u32x4, fwd), gb = __builtin_bit_cast(u32x4, g);
	u32x4 r;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		uint32_t m, o;
		asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(m) : "v"(f[i]));
		if constexpr (AND_FORM) { // 0 - (0 / 1) = all zeros / all ones, and
			asm("v_pk_sub_u16 %0, 0, %1 op_sel_hi:[0,1]" : "=v"(o) : "v"(m));
			o &= gb[i];
		} else {
			asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o) : "v"(gb[i]), "v"(m));
		}
		r[i] = o;
	}
	return __builtin_bit_cast(h8, r);
}
} // namespace
} // namespace tcnn_amd
