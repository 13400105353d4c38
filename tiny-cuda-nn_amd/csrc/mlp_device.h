// mlp_device.h -- device helpers shared by the MLP kernels (k_mlp.hip, k_train.hip): MFMA wrappers, activations
// (common_device.h:102-160, 241-297 of the reference), fragment k-orders, transposing LDS read, dL/dx store.
#pragma once

#include "tcnn_common.h"

#include <hip/hip_fp16.h>

namespace tcnn_amd {
namespace {

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr float K_ACT = 10.0f;

__device__ inline float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }

// common_device.h:102-160, applied to the fp16-rounded accumulator like the reference's warp_activation
__device__ inline half_t activation_fwd(uint32_t act, half_t pre) {
	const float x = (float)pre;
	switch (act) {
		case (uint32_t)Activation::ReLU: return x > 0.0f ? pre : (half_t)0.0f;
		case (uint32_t)Activation::LeakyReLU: return pre * (half_t)(x > 0.0f ? 1.0f : 0.01f);
		case (uint32_t)Activation::Exponential: return (half_t)expf(x);
		case (uint32_t)Activation::Sine: return (half_t)sinf(x);
		case (uint32_t)Activation::Sigmoid: return (half_t)logistic(x);
		case (uint32_t)Activation::Squareplus: { const float y = x * K_ACT; return (half_t)(0.5f * (y + sqrtf(y * y + 4)) / K_ACT); }
		case (uint32_t)Activation::Softplus: return (half_t)(logf(expf(x * K_ACT) + 1.0f) / K_ACT);
		case (uint32_t)Activation::Tanh: return (half_t)tanhf(x);
		default: return pre;
	}
}

// common_device.h:241-297: derivative from the forward OUTPUT
__device__ inline half_t activation_bwd(uint32_t act, half_t grad, half_t fwd) {
	const float y = (float)fwd;
	switch (act) {
		case (uint32_t)Activation::ReLU: return y > 0.0f ? grad : grad * (half_t)0.0f;
		case (uint32_t)Activation::LeakyReLU: return grad * (half_t)(y > 0.0f ? 1.0f : 0.01f);
		case (uint32_t)Activation::Exponential: return grad * fwd;
		case (uint32_t)Activation::Sigmoid: return grad * (half_t)(fwd * (half_t)(1.0f - y));
		case (uint32_t)Activation::Squareplus: { const float t = y * K_ACT; return grad * (half_t)(t * t / (t * t + 1)); }
		case (uint32_t)Activation::Softplus: return grad * (half_t)(1.0f - expf(-y * K_ACT));
		case (uint32_t)Activation::Tanh: return grad * (half_t)(1.0f - (y * y));
		default: return grad; // None; Sine is unsupported from outputs (common_device.h:261-265)
	}
}

template <int ACT> __device__ inline half_t act_fwd_t(uint32_t act, half_t v) {
	if constexpr (ACT == (int)Activation::ReLU) return v > (half_t)0.0f ? v : (half_t)0.0f;
	else if constexpr (ACT == (int)Activation::None) return v;
	else return activation_fwd(act, v);
}
template <int ACT> __device__ inline half_t act_bwd_t(uint32_t act, half_t g, half_t fwd) {
	if constexpr (ACT == (int)Activation::ReLU) return fwd > (half_t)0.0f ? g : g * (half_t)0.0f;
	else if constexpr (ACT == (int)Activation::None) return g;
	else return activation_bwd(act, g, fwd);
}

__device__ inline f4 mfma(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// k = 16: A lane l holds A[row = l & 15][k = 4 (l >> 4) + j], B lane l holds B[k = 4 (l >> 4) + j][col = l & 15], j = 0..3
__device__ inline f4 mfma16(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

// k index of element j of a chained-layer fragment
__host__ __device__ inline uint32_t k_chain(uint32_t s, uint32_t q, uint32_t j) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); }
__host__ __device__ inline uint32_t k_natural(uint32_t s, uint32_t q, uint32_t j) { return 32 * s + 8 * q + j; }

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ inline h4 lds_read_tr(const half_t* p) {
	fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((fp16x4 __attribute__((address_space(3)))*)p);
	return __builtin_bit_cast(h4, v);
}

// Store 4 consecutive input-gradient features k0..k0+3 of sample s.  AoS: one 8-byte store.  Level planes (what the grid
// scatter reads with unit stride): feature k lives at ((k / F) * n + s) * F + k % F.
__device__ inline void store_dx(half_t* base, uint32_t plane_f, uint32_t n, uint32_t in_w, uint32_t s, uint32_t k0, h4 v) {
	typedef _Float16 h2 __attribute__((ext_vector_type(2)));
	if (plane_f == 0) {
		*(h4*)(base + (size_t)s * in_w + k0) = v;
	} else if (plane_f == 2) {
		*(h2*)(base + ((size_t)(k0 / 2) * n + s) * 2) = h2{v[0], v[1]};
		*(h2*)(base + ((size_t)(k0 / 2 + 1) * n + s) * 2) = h2{v[2], v[3]};
	} else if (plane_f >= 4) {
		*(h4*)(base + ((size_t)(k0 / plane_f) * n + s) * plane_f + (k0 % plane_f)) = v;
	} else {
#pragma unroll
		for (int r = 0; r < 4; ++r) base[(size_t)(k0 + r) * n + s] = v[r];
	}
}

// "Records" for the grid scatter (k_grid_scatter.hip): 16-byte records = the sample's D coordinates (floats) followed by
// gradients (halves), so that the scatter fetches everything it needs about a hit with ONE load.  Requires 4 D + 2 F <= 16.
//   D = 2, F = 2: float4 rec[level / 2][n] = {x, y, gradients of the even level, gradients of the odd level} (two levels share a record)
//   D = 3, F = 2: float4 rec[level][n]     = {x, y, z, gradients}
//   D = 2, F = 4: float4 rec[level][n]     = {x, y, gradients}
// v = features k0..k0+3 of sample s (k0 a multiple of 4).
__device__ inline void store_dx_record(half_t* base, uint32_t F, uint32_t D, uint32_t n, uint32_t s, uint32_t k0, h4 v, const float* xs) {
	typedef uint32_t u4 __attribute__((ext_vector_type(4)));
	typedef _Float16 h2 __attribute__((ext_vector_type(2)));
	u4* recs = (u4*)base;
	const uint32_t x0 = __builtin_bit_cast(uint32_t, xs[0]), x1 = __builtin_bit_cast(uint32_t, xs[1]), x2 = __builtin_bit_cast(uint32_t, xs[2]);
	const uint32_t lo = __builtin_bit_cast(uint32_t, (h2{v[0], v[1]})), hi = __builtin_bit_cast(uint32_t, (h2{v[2], v[3]}));
	if (F == 2 && D == 3) {
		recs[(size_t)(k0 / 2) * n + s] = u4{x0, x1, x2, lo};
		recs[(size_t)(k0 / 2 + 1) * n + s] = u4{x0, x1, x2, hi};
	} else { // F == 2, D == 2: levels k0 / 2 and k0 / 2 + 1 = pair k0 / 4;  F == 4, D == 2: level k0 / 4
		recs[(size_t)(k0 / 4) * n + s] = u4{x0, x1, lo, hi};
	}
}


// ---- L2 / RelativeL2 inside the fused training kernels (losses/l2.h:40-74, relative_l2.h:40-75):
//     value = d^2 / (p^2 + 0.01) / pdf / n_total,   dL/dy = (half)(loss_scale * (2 d / (p^2 + 0.01) / pdf) / n_total),   d = p - target.
// Floating point, so its bar is the north-star tolerance, not bit equality with the reference's order of operations: ONE reciprocal
// (v_rcp_f32 refined by one Newton step) serves value and gradient, and the two divisions by n_total are multiplications by constants
// computed once per kernel -- 13 dependent instructions where four IEEE divisions were ~100 (the loss was 940 of a wave's 5.9 k clocks
// per trip in k_mlp_train_r32).  Against the reference's order (the oracle; k_loss keeps it): values within 4 ulp, >= 99.9 % of the half
// gradients identical and the rest one half-ulp apart (tests/test_losses.py::test_fused_loss_against_the_exact_order).
struct LossScales { float inv_n_total, scale_over_n; };
__device__ inline LossScales loss_scales(const uint32_t n_total, const float loss_scale) { return LossScales{1.0f / (float)n_total, loss_scale / (float)n_total}; }
template <bool RELATIVE>
__device__ inline void loss_l2_fused(const float prediction, const float target, const LossScales& k, float& value, half_t& grad, const bool has_pdf = false, const float pdf = 1.0f) {
	const float d = prediction - target;
	float dr = d; // d / (p^2 + 0.01) / pdf
	if constexpr (RELATIVE) {
		const float q = __builtin_fmaf(prediction, prediction, 0.01f);
		const float r0 = __builtin_amdgcn_rcpf(q);
		dr = d * __builtin_fmaf(r0, __builtin_fmaf(-q, r0, 1.0f), r0);
	}
	if (has_pdf) dr = dr / pdf; // (uniform; relative_l2.h:66-70: an optional importance-sampling density)
	value = d * dr * k.inv_n_total;
	grad = (half_t)((dr + dr) * k.scale_over_n);
}

} // namespace
} // namespace tcnn_amd
