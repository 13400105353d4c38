// k_misc.hip -- streaming kernels around the MLP: OneBlob / Identity encodings, losses, sum reduction, Adam,
// parameter initialisation and dtype plumbing.  All HBM-bound: 16-byte accesses, wave64 reductions.
//
// Replaces (reference, /root/reference/include/tiny-cuda-nn):
//   encodings/oneblob.h:47-164   kernel_one_blob / _backward          -> k_oneblob_fwd / k_oneblob_bwd_input
//   encodings/identity.h:46-85   identity / identity_backward         -> k_identity_fwd / k_identity_bwd_input
//   losses/l2.h:40-74, losses/relative_l2.h:40-75                     -> k_loss
//   reduce_sum.h:52-157          block_reduce + atomicAdd             -> k_reduce_stage1 / k_reduce_stage2 (deterministic)
//   optimizers/adam.h:48-119     adam_step                            -> k_adam
//   random.h:40-55               generate_random_kernel               -> k_random_uniform
//   common_device.h:990-1014     trim_and_cast / cast / cast_from     -> k_trim_and_cast / k_cast_*
#include "tcnn_common.h"
#include "adam_device.h"
#include "grid_fixed.h"
#include "mlp_side_jobs.h"
#include "oneblob_device.h"

#include <hip/hip_fp16.h>

#include <cmath>
#include <limits>

namespace tcnn_amd {
namespace {

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// OneBlob, definition form of oneblob.h:47-67: bin k = C(l_{k+1}) - C(l_k), C(l) = cdf(l-x) + cdf(l-x-1) + cdf(l-x+1),
// the last bin's right edge is bin 0's left edge + 1 (wrap-around).  One thread per output element (AoS).
template <typename T>
__global__ void __launch_bounds__(256) k_oneblob_fwd(const uint32_t n, const uint32_t n_dims, const uint32_t log2_bins, const MatView x, T* __restrict__ out, const uint32_t out_stride) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / out_stride;
	if (i >= n) return;
	const uint32_t j = gid - i * out_stride;
	const uint32_t n_bins = 1u << log2_bins;
	if (j >= n_dims * n_bins) { // oneblob.h:207-209
		out[gid] = (T)1.0f;
		return;
	}
	const uint32_t dim = j >> log2_bins, bin = j & (n_bins - 1);
	const float xv = x.data[(size_t)i * x.stride_sample + (size_t)dim * x.stride_dim];
	const float nb = (float)n_bins;
	const float lb = scalbnf((float)bin, -(int)log2_bins);
	const float l = quartic_cdf(lb - xv, nb) + quartic_cdf(lb - xv - 1.0f, nb) + quartic_cdf(lb - xv + 1.0f, nb);
	const float rb = scalbnf((float)((bin + 1) & (n_bins - 1)), -(int)log2_bins);
	float r = quartic_cdf(rb - xv, nb) + quartic_cdf(rb - xv - 1.0f, nb) + quartic_cdf(rb - xv + 1.0f, nb);
	if (bin == n_bins - 1) r += 1;
	out[gid] = (T)(r - l);
}

// The same, 8 consecutive bins per thread (n_bins and out_stride multiples of 8): the 9 bin edges are evaluated once each --
// the right edge of bin k IS the left edge of bin k + 1, bit for bit -- and the 8 results leave in one store.  27 cdf evaluations
// for 8 outputs instead of 48: the element-per-thread form took 26 of C2's 84 us per step, ALU-bound.
template <typename T>
__global__ void __launch_bounds__(256) k_oneblob_fwd8(const uint32_t n, const uint32_t n_dims, const uint32_t log2_bins, const MatView x, T* __restrict__ out, const uint32_t out_stride) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t chunks = out_stride / 8;
	const uint32_t i = gid / chunks;
	if (i >= n) return;
	const uint32_t j0 = (gid - i * chunks) * 8;
	const uint32_t n_bins = 1u << log2_bins;
	typedef T vec8 __attribute__((ext_vector_type(8)));
	vec8 v;
	if (j0 >= n_dims * n_bins) { // oneblob.h:207-209
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = (T)1.0f;
	} else {
		const uint32_t dim = j0 >> log2_bins, b0 = j0 & (n_bins - 1);
		const float xv = x.data[(size_t)i * x.stride_sample + (size_t)dim * x.stride_dim];
		const float nb = (float)n_bins;
		float edge[9];
#pragma unroll
		for (int k = 0; k < 9; ++k) {
			const float lb = scalbnf((float)((b0 + k) & (n_bins - 1)), -(int)log2_bins);
			edge[k] = quartic_cdf(lb - xv, nb) + quartic_cdf(lb - xv - 1.0f, nb) + quartic_cdf(lb - xv + 1.0f, nb);
		}
		if (b0 + 8 == n_bins) edge[8] += 1; // the last bin's right edge is bin 0's left edge + 1
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = (T)(edge[k + 1] - edge[k]);
	}
	*(vec8*)(out + (size_t)i * out_stride + j0) = v;
}

// The same, 8 consecutive bins per thread, for rows that are mostly zeros.  The quartic kernel has radius 1 / n_bins, so for an
// input in [0, 1] only the bins within one bin of x (and of its wrap-around images x +- 1, which fall on the same bins modulo
// n_bins) see anything but saturated cdf values; every other bin is a difference of equal integers, exactly +0.  The n_bins / 8
// threads of a row (consecutive lanes of one wave) share the work on the five bins floor(x n_bins) - 2 .. + 2 -- evaluated in the
// definition form, same operations, same bits -- exchange them with lane shuffles, and every thread stores its 8 bins: 30 cdf
// evaluations per row instead of 27 per 8 bins (k_oneblob_fwd8 was ALU-bound: 14.5 of C2's 60 us), dense 1 KB stores per wave.
// Inputs outside [0, 1] take the general form.  The threads behind the last row write the padding columns (ones, oneblob.h:207-209).
template <typename T, int CPR> // CPR = n_bins / 8 = threads per row, a power of two <= 64
__global__ void __launch_bounds__(256) k_oneblob_fwd_sparse(const uint32_t n, const uint32_t n_dims, const uint32_t log2_bins, const MatView x, T* __restrict__ out, const uint32_t out_stride) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	typedef T vec8 __attribute__((ext_vector_type(8)));
	const uint32_t n_bins = 1u << log2_bins;
	constexpr uint32_t cpr = CPR;
	const uint32_t row_threads = n * n_dims * cpr;
	if (gid >= row_threads) {
		const uint32_t pad_chunks = (out_stride - n_dims * n_bins) >> 3;
		const uint32_t p = gid - row_threads;
		if (pad_chunks == 0 || p >= n * pad_chunks) return;
		const uint32_t i = p / pad_chunks, pc = p - i * pad_chunks;
		vec8 ones;
#pragma unroll
		for (int k = 0; k < 8; ++k) ones[k] = (T)1.0f;
		*(vec8*)(out + (size_t)i * out_stride + n_dims * n_bins + 8 * pc) = ones;
		return;
	}
	const uint32_t r = gid / cpr, j = gid - r * cpr; // row, chunk of 8 bins inside it
	const uint32_t i = r / n_dims, dim = r - i * n_dims;
	const uint32_t b0 = 8 * j;
	const float xv = x.data[(size_t)i * x.stride_sample + (size_t)dim * x.stride_dim];
	const float nb = (float)n_bins;
	auto edge = [&](const uint32_t bin) {
		const float lb = scalbnf((float)(bin & (n_bins - 1)), -(int)log2_bins);
		return quartic_cdf(lb - xv, nb) + quartic_cdf(lb - xv - 1.0f, nb) + quartic_cdf(lb - xv + 1.0f, nb);
	};
	vec8 v;
	if (!(xv >= 0.0f && xv <= 1.0f)) { // general form (k_oneblob_fwd8); the threads of a row agree about this branch
		float e[9];
#pragma unroll
		for (int k = 0; k < 9; ++k) e[k] = edge(b0 + k);
		if (b0 + 8 == n_bins) e[8] += 1;
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = (T)(e[k + 1] - e[k]);
	} else {
		const uint32_t centre = (uint32_t)(int)floorf(xv * nb) + n_bins - 2u; // window bin o is (centre + o) mod n_bins, o = 0..4: distinct since n_bins >= 8
		// thread j evaluates the window bins j, j + CPR, ... (every thread the same number of them: a surplus thread repeats bin 4)
		constexpr int PER = (5 + CPR - 1) / CPR;
		float mine[PER];
#pragma unroll
		for (int t = 0; t < PER; ++t) {
			const uint32_t o = min(j + (uint32_t)t * cpr, 4u);
			const uint32_t bin = (centre + o) & (n_bins - 1);
			const float l = edge(bin);
			float rr = edge(bin + 1);
			if (bin == n_bins - 1) rr += 1; // the last bin's right edge is bin 0's left edge + 1
			mine[t] = rr - l;
		}
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = (T)0.0f;
#pragma unroll
		for (int o = 0; o < 5; ++o) {
			const float val = CPR == 1 ? mine[o / CPR] : __shfl(mine[o / CPR], o % CPR, CPR);
			const uint32_t d = ((centre + o) & (n_bins - 1)) - b0;
#pragma unroll
			for (int k = 0; k < 8; ++k) v[k] = d == (uint32_t)k ? (T)val : v[k];
		}
	}
	*(vec8*)(out + (size_t)i * out_stride + (size_t)dim * n_bins + b0) = v;
}

template <typename T>
__global__ void __launch_bounds__(128) k_oneblob_bwd_input(const uint32_t n, const uint32_t n_dims, const uint32_t log2_bins, const MatView x, const T* __restrict__ dL_dy, const uint32_t dy_stride, const MatViewMut dL_dx) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / n_dims;
	if (i >= n) return;
	const uint32_t j = gid - i * n_dims;
	const uint32_t n_bins = 1u << log2_bins;
	const float nb = (float)n_bins;
	const float xv = x.data[(size_t)i * x.stride_sample + (size_t)j * x.stride_dim];
	float result = 0;
	float left = quartic_cdf_deriv(-xv, nb) + quartic_cdf_deriv(-xv - 1.0f, nb) + quartic_cdf_deriv(-xv + 1.0f, nb);
	for (uint32_t k = 0; k < n_bins; ++k) {
		const float rb = scalbnf((float)(k + 1), -(int)log2_bins);
		const float right = quartic_cdf_deriv(rb - xv, nb) + quartic_cdf_deriv(rb - xv - 1.0f, nb) + quartic_cdf_deriv(rb - xv + 1.0f, nb);
		const float deriv = left - right;
		left = right;
		result += (float)dL_dy[(size_t)i * dy_stride + j * n_bins + k] * deriv;
	}
	dL_dx.data[(size_t)i * dL_dx.stride_sample + (size_t)j * dL_dx.stride_dim] = result;
}

template <typename T>
__global__ void __launch_bounds__(256) k_identity_fwd(const uint32_t n, const uint32_t n_dims, const float scale, const float offset, const MatView x, T* __restrict__ out, const uint32_t out_stride) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / out_stride;
	if (i >= n) return;
	const uint32_t j = gid - i * out_stride;
	out[gid] = j >= n_dims ? (T)1.0f : (T)(x.data[(size_t)i * x.stride_sample + (size_t)j * x.stride_dim] * scale + offset);
}

template <typename T>
__global__ void __launch_bounds__(256) k_identity_bwd_input(const uint32_t n, const uint32_t n_dims, const float scale, const T* __restrict__ dL_dy, const uint32_t dy_stride, const MatViewMut dL_dx) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / n_dims;
	if (i >= n) return;
	const uint32_t j = gid - i * n_dims;
	dL_dx.data[(size_t)i * dL_dx.stride_sample + (size_t)j * dL_dx.stride_dim] = (float)(T)((float)dL_dy[(size_t)i * dy_stride + j] * scale);
}

// ---- loss: l2.h:40-74 / relative_l2.h:40-75 and the other element-wise losses.  One element: the reference's expressions, term for term.
// (row: the sample's padded prediction row -- RelativeL2Luminance reads the pixel's other channels from it)
__device__ inline void loss_element(const uint32_t type, const float prediction, const float target, const float pdf, const uint32_t n_total, const float loss_scale,
                                    const half_t* __restrict__ row, const uint32_t dims, float& value_out, half_t& grad_out) {
	const float difference = prediction - target;
	float value, gradient;
	switch ((LossType)type) {
		case LossType::RelativeL2: { // relative_l2.h:60-73
			const float prediction_sq_plus_epsilon = prediction * prediction + 0.01f;
			value = difference * difference / prediction_sq_plus_epsilon / pdf / n_total;
			gradient = 2 * difference / prediction_sq_plus_epsilon / pdf;
			break;
		}
		case LossType::RelativeL2Luminance: { // relative_l2_luminance.h:40-85: the divisor is the squared luminance of the pixel
			const half_t* px = row;
			float r = (float)px[0], g = (float)px[1], b = (float)px[2];
			if (dims >= 6) {
				r += (float)px[3];
				g += (float)px[4];
				b += (float)px[5];
			}
			const float luminance = (0.299f * r + 0.587f * g + 0.114f * b);
			const float prediction_sq_plus_epsilon = luminance * luminance + 0.01f;
			value = difference * difference / prediction_sq_plus_epsilon / pdf / n_total;
			gradient = 2 * difference / prediction_sq_plus_epsilon / pdf;
			break;
		}
		case LossType::L1: // l1.h:40-70
			value = fabsf(difference) / pdf / n_total;
			gradient = copysignf(1.0f / pdf, difference);
			break;
		case LossType::RelativeL1: { // relative_l1.h:40-72
			const float scale = 1.0f / (fabsf(prediction) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
			break;
		}
		case LossType::Mape: { // mape.h:40-73
			const float scale = 1.0f / (fabsf(target) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
			break;
		}
		case LossType::Smape: { // smape.h:40-73
			const float scale = 1.0f / (0.5f * (fabsf(target) + fabsf(prediction)) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
			break;
		}
		case LossType::CrossEntropy: { // cross_entropy.h:40-72: the gradient already carries 1 / n_total
			const float factor = -target / pdf / n_total;
			value_out = factor * logf(prediction);
			grad_out = (half_t)(loss_scale * (factor / prediction));
			return;
		}
		case LossType::Variance: { // variance_is.h:40-72
			const float factor = target * target / pdf / n_total;
			value_out = factor / prediction - factor / pdf;
			grad_out = (half_t)(loss_scale * (-factor / (prediction * prediction)));
			return;
		}
		default: // L2, l2.h:60-72
			value = difference * difference / pdf / n_total;
			gradient = 2 * difference / pdf;
			break;
	}
	value_out = value;
	grad_out = (half_t)(loss_scale * gradient / n_total);
}

// one thread per padded output element (any stride)
__global__ void __launch_bounds__(256) k_loss(
	const uint32_t type, const uint32_t n_elements, const uint32_t stride, const uint32_t dims, const float loss_scale,
	const half_t* __restrict__ predictions, const float* __restrict__ targets, float* __restrict__ values, half_t* __restrict__ gradients, const float* __restrict__ data_pdf
) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	const uint32_t intra = i % stride;
	const uint32_t inter = i / stride;
	if (intra >= dims) {
		values[i] = 0;
		gradients[i] = (half_t)0.0f;
		return;
	}
	const uint32_t target_idx = inter * dims + intra;
	const uint32_t n_total = n_elements / stride * dims;
	const float prediction = (float)predictions[i];
	const float pdf = data_pdf ? data_pdf[target_idx] : 1;
	float value;
	half_t grad;
	loss_element(type, prediction, targets[target_idx], pdf, n_total, loss_scale, predictions + (i - intra), dims, value, grad);
	values[i] = value;
	gradients[i] = grad;
}

// The same for strides that are multiples of 8 (every network's padded output): eight consecutive elements per thread -- one 16-byte load of
// predictions, one 16-byte store of gradients, two of values -- and ONE division by the stride per thread instead of three per element
// (round 5: 20 -> 13 us at 2^18 x 16, the unfused step's loss launch).  Same expressions per element, same bits.
__global__ void __launch_bounds__(256) k_loss8(
	const uint32_t type, const uint32_t n_elements, const uint32_t stride, const uint32_t dims, const uint32_t n_total, const float loss_scale,
	const half_t* __restrict__ predictions, const float* __restrict__ targets, float* __restrict__ values, half_t* __restrict__ gradients, const float* __restrict__ data_pdf
) {
	const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
	if (i0 >= n_elements) return;
	const uint32_t inter = i0 / stride, intra0 = i0 - inter * stride;
	const h8 pv = *(const h8*)(predictions + i0);
	float v[8];
	h8 g;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint32_t intra = intra0 + k;
		v[k] = 0;
		g[k] = (half_t)0.0f;
		if (intra < dims) {
			const uint32_t target_idx = inter * dims + intra;
			const float pdf = data_pdf ? data_pdf[target_idx] : 1;
			half_t gk;
			loss_element(type, (float)pv[k], targets[target_idx], pdf, n_total, loss_scale, predictions + (i0 - intra0), dims, v[k], gk);
			g[k] = gk;
		}
	}
	*(float4*)(values + i0) = float4{v[0], v[1], v[2], v[3]};
	*(float4*)(values + i0 + 4) = float4{v[4], v[5], v[6], v[7]};
	*(h8*)(gradients + i0) = g;
}

// ---- deterministic two-stage sum: stage 1 = one partial per block (fixed grid), stage 2 = one block sums the partials
__device__ inline float wave_sum(float v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

__device__ inline float block_sum_256(float v, float* smem4) {
	v = wave_sum(v);
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0) smem4[w] = v;
	__syncthreads();
	return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

__global__ void __launch_bounds__(256) k_reduce_stage1(const size_t n, const float* __restrict__ values, float* __restrict__ partials) {
	__shared__ float sm[4];
	float acc = 0.0f;
	const size_t n4 = n / 4;
	const float4* v4 = (const float4*)values;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		const float4 v = v4[i];
		acc += (v.x + v.y) + (v.z + v.w);
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc += values[n4 * 4 + threadIdx.x];
	const float s = block_sum_256(acc, sm);
	if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_reduce_stage2(const uint32_t n_partials, const float* __restrict__ partials, float* __restrict__ result) {
	__shared__ float sm[4];
	float acc = 0.0f;
	for (uint32_t i = threadIdx.x; i < n_partials; i += 256) acc += partials[i];
	const float s = block_sum_256(acc, sm);
	if (threadIdx.x == 0) *result = s;
}

// ---- Adam, adam.h:48-119.  4 parameters per thread, 16-byte accesses; 36 B/param of HBM traffic is the floor.

// debiasing factor of adam.h:97-98
__device__ inline float adam_debias(const float beta1, const float beta2, uint32_t step) { return sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step)); }

// table[t] = adam_debias(t) for t in [from, to): the factor depends on (beta1, beta2, t) only.  Evaluating it per parameter
// (two powf, ~300 instructions) made the kernel ALU-bound, and more so with every step: a grid parameter that misses one
// update (zero gradient) has a step count of its own from then on, so after a few dozen steps almost every parameter does.
__global__ void __launch_bounds__(256) k_adam_debias_table(const float beta1, const float beta2, const uint32_t from, const uint32_t to, float* __restrict__ table) {
	const uint32_t t = from + blockIdx.x * blockDim.x + threadIdx.x;
	if (t < to) table[t] = adam_debias(beta1, beta2, t);
}

constexpr int ADAM_Q = 1; // quads (of 4 parameters) per thread, one block-width apart (measured: 1 -> 82 us, 2 -> 84 us, 4 -> 91 us on C3a; occupancy wins)

// QUAD_UNIFORM: n_matrix is a multiple of 4, so the 4 parameters of a quad are all matrix weights or all not
// STEP_T: uint32_t, or uint16_t while every count fits (AdamOptimizer widens the array before one could overflow)
typedef uint16_t adam_u16x4 __attribute__((ext_vector_type(4)));
template <typename STEP_T> __device__ inline uint4 adam_load_steps(const STEP_T* p) {
	if constexpr (sizeof(STEP_T) == 2) {
		const adam_u16x4 v = *(const adam_u16x4*)p;
		return uint4{v[0], v[1], v[2], v[3]};
	} else {
		return *(const uint4*)p;
	}
}
template <typename STEP_T> __device__ inline void adam_store_steps(STEP_T* p, const uint4 v) {
	if constexpr (sizeof(STEP_T) == 2) *(adam_u16x4*)p = adam_u16x4{(uint16_t)v.x, (uint16_t)v.y, (uint16_t)v.z, (uint16_t)v.w};
	else *(uint4*)p = v;
}
template <bool QUAD_UNIFORM, typename STEP_T>
__global__ void __launch_bounds__(256) k_adam(
	const AdamArgs a, const size_t n, const size_t n_matrix,
	float* __restrict__ w_fp, half_t* __restrict__ w, const half_t* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2, STEP_T* __restrict__ steps,
	const float* __restrict__ debias_table
) {
	const size_t base = (size_t)blockIdx.x * (256 * 4 * ADAM_Q) + threadIdx.x * 4;
	const float debias = debias_table[a.common_step];
	const auto from_table = [&](const uint32_t t) { return debias_table[t]; };
	// phase 1: the gradients of all quads (8 B each).  Grid (non-matrix) quads whose 4 gradients are all zero are skipped
	// without touching the other 32 B/param (adam.h:76-79 returns before reading anything else).
	h4 gv[ADAM_Q];
	bool live[ADAM_Q];
#pragma unroll
	for (int q = 0; q < ADAM_Q; ++q) {
		const size_t i4 = base + (size_t)q * 1024;
		live[q] = i4 + 4 <= n;
		gv[q] = live[q] ? *(const h4*)(g + i4) : h4{0, 0, 0, 0};
	}
#pragma unroll
	for (int q = 0; q < ADAM_Q; ++q) {
		const size_t i4 = base + (size_t)q * 1024;
		const bool zero = gv[q][0] == (half_t)0.0f && gv[q][1] == (half_t)0.0f && gv[q][2] == (half_t)0.0f && gv[q][3] == (half_t)0.0f;
		if (live[q] && i4 >= n_matrix && zero) live[q] = false;
	}
	// phase 2: all remaining loads in flight together
	float4 wf[ADAM_Q], a1[ADAM_Q], a2[ADAM_Q];
	uint4 st[ADAM_Q];
#pragma unroll
	for (int q = 0; q < ADAM_Q; ++q) {
		const size_t i4 = base + (size_t)q * 1024;
		if (live[q]) {
			wf[q] = *(const float4*)(w_fp + i4);
			a1[q] = *(const float4*)(m1 + i4);
			a2[q] = *(const float4*)(m2 + i4);
			st[q] = adam_load_steps(steps + i4);
		}
	}
#pragma unroll
	for (int q = 0; q < ADAM_Q; ++q) {
		const size_t i4 = base + (size_t)q * 1024;
		if (live[q]) {
			half_t wh[4];
			bool up[4];
			const bool quad_matrix = i4 < n_matrix;
			adam_one(a, from_table, debias, QUAD_UNIFORM ? quad_matrix : i4 + 0 < n_matrix, gv[q][0], wf[q].x, wh[0], a1[q].x, a2[q].x, st[q].x, up[0]);
			adam_one(a, from_table, debias, QUAD_UNIFORM ? quad_matrix : i4 + 1 < n_matrix, gv[q][1], wf[q].y, wh[1], a1[q].y, a2[q].y, st[q].y, up[1]);
			adam_one(a, from_table, debias, QUAD_UNIFORM ? quad_matrix : i4 + 2 < n_matrix, gv[q][2], wf[q].z, wh[2], a1[q].z, a2[q].z, st[q].z, up[2]);
			adam_one(a, from_table, debias, QUAD_UNIFORM ? quad_matrix : i4 + 3 < n_matrix, gv[q][3], wf[q].w, wh[3], a1[q].w, a2[q].w, st[q].w, up[3]);
			*(float4*)(w_fp + i4) = wf[q];
			*(float4*)(m1 + i4) = a1[q];
			*(float4*)(m2 + i4) = a2[q];
			adam_store_steps(steps + i4, st[q]);
			if (up[0] && up[1] && up[2] && up[3]) {
				*(h4*)(w + i4) = h4{wh[0], wh[1], wh[2], wh[3]};
			} else { // skipped parameters keep their half value, whatever it is: only the updated ones are stored
#pragma unroll
				for (int e = 0; e < 4; ++e) if (up[e]) w[i4 + e] = wh[e];
			}
		} else if (i4 < n && i4 + 4 > n) { // ragged tail
			for (size_t i = i4; i < n; ++i) {
				bool up;
				half_t wh;
				uint32_t st1 = steps[i];
				adam_one(a, from_table, debias, i < n_matrix, g[i], w_fp[i], wh, m1[i], m2[i], st1, up);
				steps[i] = (STEP_T)st1;
				if (up) w[i] = wh;
			}
		}
	}
}

// One parameter per thread, no vector accesses: for parameter ranges that do not start on a 16-byte boundary (a Composite
// optimizer hands its nested optimizers slices at arbitrary offsets, optimizers/composite.h:126-135)
template <typename STEP_T>
__global__ void __launch_bounds__(256) k_adam_scalar(
	const AdamArgs a, const size_t n, const size_t n_matrix,
	float* __restrict__ w_fp, half_t* __restrict__ w, const half_t* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2, STEP_T* __restrict__ steps,
	const float* __restrict__ debias_table
) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float debias = debias_table[a.common_step];
	const auto from_table = [&](const uint32_t t) { return debias_table[t]; };
	bool up;
	half_t wh;
	uint32_t st1 = steps[i];
	adam_one(a, from_table, debias,  i < n_matrix, g[i], w_fp[i], wh, m1[i], m2[i], st1, up);
	steps[i] = (STEP_T)st1;
	if (up) w[i] = wh;
}

// ---- pcg32 strided uniform fill, random.h:40-55 (N_TO_GENERATE = 4, thread i advances a copy of the rng by 4 i)
__global__ void __launch_bounds__(128) k_random_uniform(const size_t n, const uint64_t state, const uint64_t inc, float* __restrict__ out, const float lower, const float upper) {
	const uint64_t MULT = 0x5851f42d4c957f2dULL;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t n_threads = (size_t)blockDim.x * gridDim.x;
	// advance(4 i)
	uint64_t cur_mult = MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
	uint64_t delta = (uint64_t)i * 4;
	while (delta > 0) {
		if (delta & 1) {
			acc_mult *= cur_mult;
			acc_plus = acc_plus * cur_mult + cur_plus;
		}
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta /= 2;
	}
	uint64_t st = acc_mult * state + acc_plus;
	for (size_t j = 0; j < 4; ++j) {
		const size_t idx = i + n_threads * j;
		if (idx >= n) return;
		const uint64_t old = st;
		st = old * MULT + inc;
		const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		const uint32_t rot = (uint32_t)(old >> 59u);
		const uint32_t u = ((xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31)));
		const float f = __uint_as_float((u >> 9) | 0x3f800000u) - 1.0f;
		out[idx] = f * (upper - lower) + lower;
	}
}

__global__ void __launch_bounds__(256) k_cast_f2h(const size_t n, const float* __restrict__ in, half_t* __restrict__ out) {
	const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (i + 4 <= n) {
		const float4 v = *(const float4*)(in + i);
		*(h4*)(out + i) = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
	} else {
		for (size_t k = i; k < n; ++k) out[k] = (half_t)in[k];
	}
}

__global__ void __launch_bounds__(256) k_cast_h2f(const size_t n, const half_t* __restrict__ in, float* __restrict__ out) {
	const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (i + 4 <= n) {
		const h4 v = *(const h4*)(in + i);
		*(float4*)(out + i) = float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
	} else {
		for (size_t k = i; k < n; ++k) out[k] = (float)in[k];
	}
}

template <typename T>
__global__ void __launch_bounds__(256) k_trim_and_cast(const uint32_t n, const uint32_t in_stride, const uint32_t dims, const T* __restrict__ in, const MatViewMut out) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / dims;
	if (i >= n) return;
	const uint32_t d = gid - i * dims;
	out.data[(size_t)i * out.stride_sample + (size_t)d * out.stride_dim] = (float)in[(size_t)i * in_stride + d];
}

__global__ void __launch_bounds__(256) k_fill_half(const size_t n, half_t* __restrict__ out, const half_t v) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = v;
}

inline uint32_t log2_exact(uint32_t v) {
	uint32_t l = 0;
	while ((1u << l) < v) ++l;
	return l;
}

inline uint32_t blocks_for(uint64_t total, uint32_t threads) {
	const uint64_t b = (total + threads - 1) / threads;
	CHECK_THROW(b < (1ull << 31));
	return (uint32_t)b;
}

} // namespace

void oneblob_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, uint32_t n_bins, MatView x, void* out, uint32_t out_stride) {
	const uint64_t total = (uint64_t)n * out_stride;
	if (total == 0) return;
	CHECK_THROW(total < (1ull << 32));
	const uint32_t lb = log2_exact(n_bins);
	if (n_bins % 8 == 0 && n_bins <= 512 && out_stride % 8 == 0) { // rows of at most 64 threads
		const uint64_t threads = total / 8; // one per 8 outputs, the rows first, then the padding columns
		const dim3 grid(blocks_for(threads, 256));
#define TCNN_ONEBLOB_SPARSE(CPR_) \
		case CPR_: \
			if (fp32) hipLaunchKernelGGL((k_oneblob_fwd_sparse<float, CPR_>), grid, dim3(256), 0, stream, n, n_dims, lb, x, (float*)out, out_stride); \
			else hipLaunchKernelGGL((k_oneblob_fwd_sparse<half_t, CPR_>), grid, dim3(256), 0, stream, n, n_dims, lb, x, (half_t*)out, out_stride); \
			break;
		switch (n_bins / 8) {
			TCNN_ONEBLOB_SPARSE(1) TCNN_ONEBLOB_SPARSE(2) TCNN_ONEBLOB_SPARSE(4) TCNN_ONEBLOB_SPARSE(8)
			TCNN_ONEBLOB_SPARSE(16) TCNN_ONEBLOB_SPARSE(32) TCNN_ONEBLOB_SPARSE(64)
			default: CHECK_THROW(false);
		}
#undef TCNN_ONEBLOB_SPARSE
		return;
	}
	if (n_bins % 8 == 0 && out_stride % 8 == 0) {
		const uint64_t threads = total / 8;
		if (fp32) hipLaunchKernelGGL(k_oneblob_fwd8<float>, dim3(blocks_for(threads, 256)), dim3(256), 0, stream, n, n_dims, lb, x, (float*)out, out_stride);
		else hipLaunchKernelGGL(k_oneblob_fwd8<half_t>, dim3(blocks_for(threads, 256)), dim3(256), 0, stream, n, n_dims, lb, x, (half_t*)out, out_stride);
		return;
	}
	if (fp32) hipLaunchKernelGGL(k_oneblob_fwd<float>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, n_dims, lb, x, (float*)out, out_stride);
	else hipLaunchKernelGGL(k_oneblob_fwd<half_t>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, n_dims, lb, x, (half_t*)out, out_stride);
}

void oneblob_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, uint32_t n_bins, MatView x, const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx) {
	const uint64_t total = (uint64_t)n * n_dims;
	if (total == 0) return;
	const uint32_t lb = log2_exact(n_bins);
	if (fp32) hipLaunchKernelGGL(k_oneblob_bwd_input<float>, dim3(blocks_for(total, 128)), dim3(128), 0, stream, n, n_dims, lb, x, (const float*)dL_dy, dy_stride, dL_dx);
	else hipLaunchKernelGGL(k_oneblob_bwd_input<half_t>, dim3(blocks_for(total, 128)), dim3(128), 0, stream, n, n_dims, lb, x, (const half_t*)dL_dy, dy_stride, dL_dx);
}

void identity_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, float scale, float offset, MatView x, void* out, uint32_t out_stride) {
	const uint64_t total = (uint64_t)n * out_stride;
	if (total == 0) return;
	CHECK_THROW(total < (1ull << 32));
	if (fp32) hipLaunchKernelGGL(k_identity_fwd<float>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, n_dims, scale, offset, x, (float*)out, out_stride);
	else hipLaunchKernelGGL(k_identity_fwd<half_t>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, n_dims, scale, offset, x, (half_t*)out, out_stride);
}

void identity_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, float scale, const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx) {
	const uint64_t total = (uint64_t)n * n_dims;
	if (total == 0) return;
	if (fp32) hipLaunchKernelGGL(k_identity_bwd_input<float>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, n_dims, scale, (const float*)dL_dy, dy_stride, dL_dx);
	else hipLaunchKernelGGL(k_identity_bwd_input<half_t>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, n_dims, scale, (const half_t*)dL_dy, dy_stride, dL_dx);
}

void loss_evaluate(hipStream_t stream, LossType type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale,
                   const void* pred_half, const float* target, float* values, void* grads_half, const float* data_pdf) {
	const uint64_t total = (uint64_t)n * stride;
	if (total == 0) return;
	CHECK_THROW(total < (1ull << 32));
	if (stride % 8 == 0 && ((uintptr_t)pred_half | (uintptr_t)values | (uintptr_t)grads_half) % 16 == 0) {
		hipLaunchKernelGGL(k_loss8, dim3(blocks_for(total / 8, 256)), dim3(256), 0, stream, (uint32_t)type, (uint32_t)total, stride, dims, (uint32_t)(total / stride * dims), loss_scale,
		                   (const half_t*)pred_half, target, values, (half_t*)grads_half, data_pdf);
		return;
	}
	hipLaunchKernelGGL(k_loss, dim3(blocks_for(total, 256)), dim3(256), 0, stream, (uint32_t)type, (uint32_t)total, stride, dims, loss_scale,
	                   (const half_t*)pred_half, target, values, (half_t*)grads_half, data_pdf);
}

void reduce_sum(hipStream_t stream, size_t n, const float* values, float* partials, float* result_dev) {
	uint32_t blocks = blocks_for((n + 3) / 4, 256);
	if (blocks > 1024) blocks = 1024;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL(k_reduce_stage1, dim3(blocks), dim3(256), 0, stream, n, values, partials);
	hipLaunchKernelGGL(k_reduce_stage2, dim3(1), dim3(256), 0, stream, blocks, partials, result_dev);
}

AdamArgs make_adam_args(const AdamHyper& h, float loss_scale, uint32_t current_step) {
	AdamArgs a;
	a.relative_weight_decay = h.relative_decay;
	a.absolute_weight_decay = h.absolute_decay;
	a.weight_clipping_magnitude = h.clipping_magnitude;
	a.loss_scale = loss_scale;
	a.learning_rate = h.learning_rate;
	a.non_matrix_learning_rate_factor = h.non_matrix_learning_rate_factor;
	a.beta1 = h.beta1;
	a.beta2 = h.beta2;
	a.epsilon = h.epsilon;
	a.lower_lr_bound = 0;
	a.upper_lr_bound = std::numeric_limits<float>::max();
	if (h.adabound) { // adam.h:157-160
		a.lower_lr_bound = 0.1f - 0.1f / ((1 - h.beta2) * (float)current_step + 1);
		a.upper_lr_bound = 0.1f + 0.1f / ((1 - h.beta2) * (float)current_step);
	}
	a.l2_reg = h.l2_reg;
	a.optimize_matrix_params = h.optimize_matrix_params;
	a.optimize_non_matrix_params = h.optimize_non_matrix_params;
	a.common_step = current_step;
	int exponent = 0;
	a.inv_loss_scale_exact = (std::frexp(loss_scale, &exponent) == 0.5f && loss_scale >= 1.0f / 65536 && loss_scale <= 65536.0f) ? 1 : 0;
	a.inv_loss_scale = 1.0f / loss_scale;
	return a;
}

namespace {
__global__ void __launch_bounds__(256) k_adam_widen_steps(const size_t n, const uint16_t* __restrict__ in, uint32_t* __restrict__ out) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = in[i];
}
template <typename STEP_T>
void adam_launch(hipStream_t stream, const AdamArgs& a, size_t n, size_t n_matrix, float* w_fp, void* w_half, const void* g_half, float* m1, float* m2, STEP_T* steps, const float* debias_table) {
	// the quad kernel reads float4 / 4 step counts / half4: every base pointer must allow that
	const auto aligned = [](const void* p, size_t bytes) { return ((uintptr_t)p & (bytes - 1)) == 0; };
	if (!(aligned(w_fp, 16) && aligned(m1, 16) && aligned(m2, 16) && aligned(steps, 4 * sizeof(STEP_T)) && aligned(w_half, 8) && aligned(g_half, 8))) {
		hipLaunchKernelGGL(k_adam_scalar<STEP_T>, dim3(blocks_for(n, 256)), dim3(256), 0, stream, a, n, n_matrix, w_fp, (half_t*)w_half, (const half_t*)g_half, m1, m2, steps, debias_table);
		return;
	}
	const dim3 grid(blocks_for((n + 3) / 4, 256 * ADAM_Q));
	if (n_matrix % 4 == 0) hipLaunchKernelGGL((k_adam<true, STEP_T>), grid, dim3(256), 0, stream, a, n, n_matrix, w_fp, (half_t*)w_half, (const half_t*)g_half, m1, m2, steps, debias_table);
	else hipLaunchKernelGGL((k_adam<false, STEP_T>), grid, dim3(256), 0, stream, a, n, n_matrix, w_fp, (half_t*)w_half, (const half_t*)g_half, m1, m2, steps, debias_table);
}
} // namespace

void adam_widen_steps(hipStream_t stream, size_t n, const void* steps16, void* steps32) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_adam_widen_steps, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, (const uint16_t*)steps16, (uint32_t*)steps32);
}

void adam_step(hipStream_t stream, const AdamHyper& h, size_t n, size_t n_matrix, float loss_scale, uint32_t current_step,
               float* w_fp, void* w_half, const void* g_half, float* m1, float* m2, void* steps, bool steps16, const float* debias_table) {
	if (n == 0) return;
	const AdamArgs a = make_adam_args(h, loss_scale, current_step);
	if (steps16) adam_launch<uint16_t>(stream, a, n, n_matrix, w_fp, w_half, g_half, m1, m2, (uint16_t*)steps, debias_table);
	else adam_launch<uint32_t>(stream, a, n, n_matrix, w_fp, w_half, g_half, m1, m2, (uint32_t*)steps, debias_table);
}

// ---- Adam with the backward pass's last two reductions in front (AdamPrologue, tcnn_common.h): workgroups [0, n_reduce_blocks) sum the MLP's
// weight-gradient slabs (the order of mlp_reduce_block, mlp_side_jobs.h) and step those weights; the next pro_range_blocks() per shared
// range round the scatter's exact sums into the gradient (what k_grid_scatter_finalize does) and step those parameters; the others are
// k_adam on everything else.  Whoever writes a gradient's final value applies adam_one to it at once.
namespace {
constexpr uint32_t PRO_THREADS = SLAB_REDUCE_ELEMS * SLAB_REDUCE_GROUPS; // 1024
constexpr uint32_t PRO_MAX_BLOCKS_PER_RANGE = 64;
// workgroups of a shared range: one per 1024 quads, at most 64 (k_grid_scatter_finalize's number)
__host__ __device__ inline uint32_t pro_range_blocks(const uint32_t n_elems) { return min(PRO_MAX_BLOCKS_PER_RANGE, max(1u, (n_elems / 4 + PRO_THREADS - 1) / PRO_THREADS)); }

template <typename STEP_T, typename DebiasOf>
__device__ inline void adam_quad(const AdamArgs& a, DebiasOf&& from_table, const float debias, const bool quad_matrix, const size_t i4, const h4 gv,
                                 float* __restrict__ w_fp, half_t* __restrict__ w, float* __restrict__ m1, float* __restrict__ m2, STEP_T* __restrict__ steps) {
	// adam.h:76-79: a grid parameter with a zero gradient is left alone -- nothing else of it is read
	if (!quad_matrix && gv[0] == (half_t)0.0f && gv[1] == (half_t)0.0f && gv[2] == (half_t)0.0f && gv[3] == (half_t)0.0f) return;
	float4 wf = *(const float4*)(w_fp + i4), a1 = *(const float4*)(m1 + i4), a2 = *(const float4*)(m2 + i4);
	uint4 st = adam_load_steps(steps + i4);
	half_t wh[4];
	bool up[4];
	adam_one(a, from_table, debias, quad_matrix, gv[0], wf.x, wh[0], a1.x, a2.x, st.x, up[0]);
	adam_one(a, from_table, debias, quad_matrix, gv[1], wf.y, wh[1], a1.y, a2.y, st.y, up[1]);
	adam_one(a, from_table, debias, quad_matrix, gv[2], wf.z, wh[2], a1.z, a2.z, st.z, up[2]);
	adam_one(a, from_table, debias, quad_matrix, gv[3], wf.w, wh[3], a1.w, a2.w, st.w, up[3]);
	*(float4*)(w_fp + i4) = wf;
	*(float4*)(m1 + i4) = a1;
	*(float4*)(m2 + i4) = a2;
	adam_store_steps(steps + i4, st);
	if (up[0] && up[1] && up[2] && up[3]) {
		*(h4*)(w + i4) = h4{wh[0], wh[1], wh[2], wh[3]};
	} else { // skipped parameters keep their half value, whatever it is
#pragma unroll
		for (int e = 0; e < 4; ++e) if (up[e]) w[i4 + e] = wh[e];
	}
}

struct PrologueArgs {
	uint32_t n_reduce_blocks, reduce_elems, reduce_slabs;
	int reduce_accumulate;
	const float* slabs;
	const GridScatterRange* ranges;
	uint32_t n_ranges, n_range_blocks; // (the sum of pro_range_blocks over the ranges)
	int accumulate;
	unsigned long long* scratch;
	size_t range_base; // parameter index of the gradient element the ranges' grad_begin counts from
	size_t bulk_begin; // first parameter of the k_adam part (behind the MLP's weights when their slabs are summed here)
};

// (98 scalar registers per wave admit ONE 16-wave workgroup per CU.  Limiting them to 80 -- amdgpu_num_sgpr, two workgroups per CU, the
// 8 waves per SIMD k_adam keeps -- was measured on C3a, same device, three alternations: optimizer 61.0 - 62.5 us against 58.8 - 59.8.)
#ifndef TCNN_PRO_BOUNDS
#define TCNN_PRO_BOUNDS __launch_bounds__(PRO_THREADS)
#endif
#ifndef TCNN_PRO_Q
#define TCNN_PRO_Q 1
#endif
constexpr int PRO_Q = TCNN_PRO_Q; // quads per thread of the k_adam part, one workgroup width apart
template <typename STEP_T>
__global__ void TCNN_PRO_BOUNDS k_adam_prologue(const AdamArgs a, const size_t n, const size_t n_matrix, float* __restrict__ w_fp, half_t* __restrict__ w, half_t* __restrict__ g,
                                                               float* __restrict__ m1, float* __restrict__ m2, STEP_T* __restrict__ steps, const float* __restrict__ debias_table, const PrologueArgs p) {
	__shared__ float part[SLAB_REDUCE_GROUPS * SLAB_REDUCE_ELEMS];
	const float debias = debias_table[a.common_step];
	const auto from_table = [&](const uint32_t t) { return debias_table[t]; };
	const uint32_t tid = threadIdx.x;
	if (blockIdx.x < p.n_reduce_blocks) {
		// mlp_reduce_block's sum of element i over the slabs, term for term
		const uint32_t e = tid & (SLAB_REDUCE_ELEMS - 1), grp = tid / SLAB_REDUCE_ELEMS;
		const uint32_t i = blockIdx.x * SLAB_REDUCE_ELEMS + e;
		float q[4] = {0, 0, 0, 0};
		if (i < p.reduce_elems) {
			uint32_t k = grp;
			for (; k + 3 * SLAB_REDUCE_GROUPS < p.reduce_slabs; k += 4 * SLAB_REDUCE_GROUPS) {
#pragma unroll
				for (int u = 0; u < 4; ++u) q[u] += p.slabs[(size_t)(k + u * SLAB_REDUCE_GROUPS) * p.reduce_elems + i];
			}
			for (; k < p.reduce_slabs; k += SLAB_REDUCE_GROUPS) q[0] += p.slabs[(size_t)k * p.reduce_elems + i];
		}
		part[grp * SLAB_REDUCE_ELEMS + e] = (q[0] + q[1]) + (q[2] + q[3]);
		__syncthreads();
		if (grp == 0 && i < p.reduce_elems) {
			float s = 0;
#pragma unroll
			for (int gi = 0; gi < SLAB_REDUCE_GROUPS; ++gi) s += part[gi * SLAB_REDUCE_ELEMS + e];
			if (p.reduce_accumulate) s += (float)g[i];
			const half_t gh = (half_t)s;
			g[i] = gh;
			bool up;
			half_t wh;
			uint32_t st1 = steps[i];
			adam_one(a, from_table, debias, i < n_matrix, gh, w_fp[i], wh, m1[i], m2[i], st1, up);
			steps[i] = (STEP_T)st1;
			if (up) w[i] = wh;
		}
		return;
	}
	const uint32_t b = blockIdx.x - p.n_reduce_blocks;
	if (b < p.n_range_blocks) {
		// k_grid_scatter_finalize's rounding of a shared range, four elements per thread; the scratch is left zero for the next step
		uint32_t ri = 0, first = 0; // which range this workgroup belongs to, and which of the range's workgroups it is
		for (uint32_t nb = pro_range_blocks(p.ranges[0].n_elems); b >= first + nb; nb = pro_range_blocks(p.ranges[ri].n_elems)) { first += nb; ++ri; }
		const GridScatterRange r = p.ranges[ri];
		const uint32_t my = b - first, of = pro_range_blocks(r.n_elems);
		typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
		u64x2* sc = (u64x2*)(p.scratch + r.scratch_begin);
		const size_t g0 = p.range_base + r.grad_begin;
		for (uint32_t quad = my * PRO_THREADS + tid; quad < r.n_elems / 4; quad += of * PRO_THREADS) {
			const u64x2 lo = sc[2 * quad], hi = sc[2 * quad + 1];
			sc[2 * quad] = u64x2{0, 0};
			sc[2 * quad + 1] = u64x2{0, 0};
			long long s[4] = {(long long)lo.x, (long long)lo.y, (long long)hi.x, (long long)hi.y};
			const size_t i4 = g0 + 4 * (size_t)quad;
			if (p.accumulate) {
				const h4 old = *(const h4*)(g + i4);
#pragma unroll
				for (int e = 0; e < 4; ++e) s[e] += half_to_fixed(old[e]);
			}
			const h4 gv = h4{fixed_to_half(s[0]), fixed_to_half(s[1]), fixed_to_half(s[2]), fixed_to_half(s[3])};
			*(h4*)(g + i4) = gv;
			adam_quad(a, from_table, debias, i4 < n_matrix, i4, gv, w_fp, w, m1, m2, steps);
		}
		return;
	}
	h4 gv[PRO_Q];
	bool live[PRO_Q];
#pragma unroll
	for (int k = 0; k < PRO_Q; ++k) {
		const size_t i4 = p.bulk_begin + (((size_t)(b - p.n_range_blocks) * PRO_Q + k) * PRO_THREADS + tid) * 4;
		live[k] = i4 < n; // (n - bulk_begin is a multiple of 4: adam_step_with_prologue)
		for (uint32_t r = 0; r < p.n_ranges; ++r) { // the shared ranges have been done above
			const size_t rb = p.range_base + p.ranges[r].grad_begin;
			if (i4 >= rb && i4 < rb + p.ranges[r].n_elems) live[k] = false;
		}
		if (live[k]) gv[k] = *(const h4*)(g + i4);
	}
#pragma unroll
	for (int k = 0; k < PRO_Q; ++k) {
		const size_t i4 = p.bulk_begin + (((size_t)(b - p.n_range_blocks) * PRO_Q + k) * PRO_THREADS + tid) * 4;
		if (live[k]) adam_quad(a, from_table, debias, i4 < n_matrix, i4, gv[k], w_fp, w, m1, m2, steps);
	}
}
} // namespace

bool adam_step_with_prologue(hipStream_t stream, const AdamHyper& h, size_t n, size_t n_matrix, float loss_scale, uint32_t current_step,
                             float* w_fp, void* w_half, void* g_half, float* m1, float* m2, void* steps, bool steps16, const float* debias_table, const AdamPrologue& pro) {
	const auto aligned = [](const void* ptr, size_t bytes) { return ((uintptr_t)ptr & (bytes - 1)) == 0; };
	if (n == 0 || !(aligned(w_fp, 16) && aligned(m1, 16) && aligned(m2, 16) && aligned(steps, steps16 ? 8 : 16) && aligned(w_half, 8) && aligned(g_half, 8))) return false;
	PrologueArgs p{};
	p.bulk_begin = 0;
	if (pro.has_reduce) {
		// the MLP's weights are the first parameters and all of the matrix weights: quads must not straddle the end of either
		if (pro.reduce_elems % 4 != 0 || pro.reduce_elems > n || n_matrix % 4 != 0) return false;
		p.n_reduce_blocks = div_round_up(pro.reduce_elems, (uint32_t)SLAB_REDUCE_ELEMS);
		p.reduce_elems = pro.reduce_elems;
		p.reduce_slabs = pro.reduce_slabs;
		p.reduce_accumulate = pro.reduce_accumulate;
		p.slabs = pro.slabs;
		p.bulk_begin = pro.reduce_elems;
	} else if (n_matrix % 4 != 0) return false;
	if ((n - p.bulk_begin) % 4 != 0) return false;
	if (!pro.ranges.empty()) {
		if ((const char*)pro.grad_base < (const char*)g_half || !aligned(pro.scratch, 16)) return false;
		p.range_base = (size_t)((const half_t*)pro.grad_base - (const half_t*)g_half);
		if (p.range_base % 4 != 0) return false;
		for (const GridScatterRange& r : pro.ranges) {
			if (r.grad_begin % 4 != 0 || r.n_elems % 4 != 0 || r.scratch_begin % 2 != 0 || p.range_base + r.grad_begin + r.n_elems > n || p.range_base + r.grad_begin < p.bulk_begin) return false;
		}
		// shared ranges that follow the k_adam part's first parameter without a gap (the coarse levels at the front of the table) are not its
		// business: it starts behind them (its workgroups skip whatever other range they meet)
		for (bool moved = true; moved;) {
			moved = false;
			for (const GridScatterRange& r : pro.ranges) {
				if (r.n_elems && p.range_base + r.grad_begin == p.bulk_begin) { p.bulk_begin += r.n_elems; moved = true; }
			}
		}
		p.ranges = pro.dev_ranges;
		p.n_ranges = (uint32_t)pro.ranges.size();
		for (const GridScatterRange& r : pro.ranges) p.n_range_blocks += pro_range_blocks(r.n_elems);
		p.scratch = (unsigned long long*)pro.scratch;
		p.accumulate = pro.accumulate ? 1 : 0;
	}
	const AdamArgs a = make_adam_args(h, loss_scale, current_step);
	const uint32_t bulk_blocks = div_round_up((uint32_t)((n - p.bulk_begin) / 4), PRO_THREADS * (uint32_t)PRO_Q);
	const dim3 grid(p.n_reduce_blocks + p.n_range_blocks + bulk_blocks);
	if (steps16) hipLaunchKernelGGL(k_adam_prologue<uint16_t>, grid, dim3(PRO_THREADS), 0, stream, a, n, n_matrix, w_fp, (half_t*)w_half, (half_t*)g_half, m1, m2, (uint16_t*)steps, debias_table, p);
	else hipLaunchKernelGGL(k_adam_prologue<uint32_t>, grid, dim3(PRO_THREADS), 0, stream, a, n, n_matrix, w_fp, (half_t*)w_half, (half_t*)g_half, m1, m2, (uint32_t*)steps, debias_table, p);
	HIP_CHECK_THROW(hipGetLastError());
	return true;
}

void adam_fill_debias_table(hipStream_t stream, float beta1, float beta2, uint32_t from, uint32_t to, float* table) {
	if (to <= from) return;
	hipLaunchKernelGGL(k_adam_debias_table, dim3(blocks_for(to - from, 256)), dim3(256), 0, stream, beta1, beta2, from, to, table);
}

namespace {
// optimizers/sgd.h:44-72
__global__ void __launch_bounds__(256) k_sgd(const size_t n, const float loss_scale, const float learning_rate, const float l2_reg, float* __restrict__ w_fp, half_t* __restrict__ w,
                                             const half_t* __restrict__ g) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float weight_fp = w_fp[i];
	float gradient = (float)g[i] / loss_scale;
	gradient += l2_reg * weight_fp;
	const float new_weight = weight_fp - learning_rate * gradient;
	w_fp[i] = new_weight;
	w[i] = (half_t)new_weight;
}

// optimizers/ema.h:44-78: tmp != nullptr keeps the average in fp32 (full_precision), else it lives in the half weights themselves
__global__ void __launch_bounds__(256) k_ema(const size_t n, const float decay, const float debias_old, const float debias_new, const half_t* __restrict__ weights, half_t* __restrict__ weights_ema,
                                             float* __restrict__ tmp) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float previous = tmp ? tmp[i] : (float)weights_ema[i];
	const float filtered = (previous * decay * debias_old + (float)weights[i] * (1 - decay)) * debias_new;
	if (tmp) tmp[i] = filtered;
	weights_ema[i] = (half_t)filtered;
}
// optimizers/average.h:44-60: running mean of the last n_samples weight vectors; `current` is the slot the new weights replace
__global__ void __launch_bounds__(256) k_average_step(const size_t n, const uint32_t n_samples, const half_t* __restrict__ weights, half_t* __restrict__ current, half_t* __restrict__ average) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const half_t weight = weights[i];
	average[i] = (half_t)((float)average[i] + ((float)weight - (float)current[i]) / (float)n_samples);
	current[i] = weight;
}
// optimizers/batched.h:44-61: pool (+)= gradient / batch_size_multiplier, restarted with the first gradient of a group
__global__ void __launch_bounds__(256) k_batched_accumulate(const size_t n, const int first, const uint32_t multiplier, const half_t* __restrict__ gradients, float* __restrict__ pool) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float v = first ? 0.0f : pool[i];
	v += (float)gradients[i] / (float)multiplier;
	pool[i] = v;
}
// optimizers/novograd.h:44-94.  One workgroup per layer sums the squared gradients (fixed tree: reproducible) and folds the sum
// into the layer's second moment, moment = beta2 moment + (1 - beta2) sum / loss_scale / loss_scale; then the element-wise step.
__global__ void __launch_bounds__(1024) k_novo_second_moment(const size_t n, const float loss_scale, const float beta2, const half_t* __restrict__ gradients, float* __restrict__ moment) {
	__shared__ float part[1024];
	float acc = 0.0f;
	for (size_t i = threadIdx.x; i < n; i += 1024) {
		const float g = (float)gradients[i];
		acc += g * g;
	}
	part[threadIdx.x] = acc;
	__syncthreads();
	for (uint32_t stride = 512; stride > 0; stride >>= 1) {
		if (threadIdx.x < stride) part[threadIdx.x] += part[threadIdx.x + stride];
		__syncthreads();
	}
	if (threadIdx.x == 0) *moment = beta2 * *moment + (1 - beta2) * part[0] / loss_scale / loss_scale;
}
__global__ void __launch_bounds__(256) k_novo_step(const size_t n, const float relative_weight_decay, const float absolute_weight_decay, const float loss_scale, const float learning_rate,
                                                   const float beta1, const float epsilon, float* __restrict__ weights_fp, half_t* __restrict__ weights, const half_t* __restrict__ gradients,
                                                   float* __restrict__ first_moments, const float* __restrict__ layer_second_moment) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float weight_fp = weights_fp[i];
	const float gradient = (float)gradients[i] / loss_scale;
	const float first_moment = beta1 * first_moments[i] + (1 - beta1) * gradient / (sqrtf(*layer_second_moment) + epsilon);
	first_moments[i] = first_moment;
	// weight_decay(rel * lr, abs * lr, w), common_device.h:870-873
	const float decayed_weight = (1 - relative_weight_decay * learning_rate) * weight_fp - copysignf(absolute_weight_decay * learning_rate, weight_fp);
	const float new_weight = decayed_weight - learning_rate * first_moment;
	weights_fp[i] = new_weight;
	weights[i] = (half_t)new_weight;
}
// optimizers/lookahead.h:44-59: slow weights <- slow * (1 - alpha) + fast * alpha, and the fast weights restart from them
__global__ void __launch_bounds__(256) k_lookahead_step(const size_t n, const float alpha, float* __restrict__ weights_fp, half_t* __restrict__ weights, half_t* __restrict__ lookahead) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float new_weight = (float)lookahead[i] * (1.0f - alpha) + weights_fp[i] * alpha;
	weights_fp[i] = new_weight;
	const half_t h = (half_t)new_weight;
	lookahead[i] = h;
	weights[i] = h;
}
} // namespace

namespace {
// dst[i][dst_col + j] = src[i][src_col + j], j < width (Composite encoding: nested AoS blocks <-> column ranges of the composite row)
template <typename T>
__global__ void __launch_bounds__(256) k_copy_columns(const uint32_t n_elements, const uint32_t width, const T* __restrict__ src, const uint32_t src_stride, const uint32_t src_col,
                                                      T* __restrict__ dst, const uint32_t dst_stride, const uint32_t dst_col) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_elements) return;
	const uint32_t i = e / width, j = e - i * width;
	dst[(size_t)i * dst_stride + dst_col + j] = src[(size_t)i * src_stride + src_col + j];
}
} // namespace

void copy_columns(hipStream_t stream, size_t elem_bytes, uint32_t n, const void* src, uint32_t src_stride, uint32_t src_col, void* dst, uint32_t dst_stride, uint32_t dst_col, uint32_t width) {
	const uint64_t total = (uint64_t)n * width;
	if (total == 0) return;
	CHECK_THROW(total < (1ull << 32) && (elem_bytes == 2 || elem_bytes == 4));
	const dim3 blocks((uint32_t)((total + 255) / 256));
	if (elem_bytes == 2) hipLaunchKernelGGL((k_copy_columns<uint16_t>), blocks, dim3(256), 0, stream, (uint32_t)total, width, (const uint16_t*)src, src_stride, src_col, (uint16_t*)dst, dst_stride, dst_col);
	else hipLaunchKernelGGL((k_copy_columns<uint32_t>), blocks, dim3(256), 0, stream, (uint32_t)total, width, (const uint32_t*)src, src_stride, src_col, (uint32_t*)dst, dst_stride, dst_col);
}

void sgd_step(hipStream_t stream, size_t n, float loss_scale, float learning_rate, float l2_reg, float* weights_full_precision, void* weights, const void* gradients) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_sgd, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, loss_scale, learning_rate, l2_reg, weights_full_precision, (half_t*)weights, (const half_t*)gradients);
}

void average_step(hipStream_t stream, size_t n, uint32_t n_samples, const void* weights, void* current_sample, void* average) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_average_step, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, n_samples, (const half_t*)weights, (half_t*)current_sample, (half_t*)average);
}
void batched_accumulate(hipStream_t stream, size_t n, bool first, uint32_t multiplier, const void* gradients, float* pool) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_batched_accumulate, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, first ? 1 : 0, multiplier, (const half_t*)gradients, pool);
}
void novograd_layer_step(hipStream_t stream, size_t n, float relative_decay, float absolute_decay, float loss_scale, float learning_rate, float beta1, float beta2, float epsilon,
                         float* weights_full_precision, void* weights, const void* gradients, float* first_moments, float* layer_second_moment) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_novo_second_moment, dim3(1), dim3(1024), 0, stream, n, loss_scale, beta2, (const half_t*)gradients, layer_second_moment);
	hipLaunchKernelGGL(k_novo_step, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, relative_decay, absolute_decay, loss_scale, learning_rate, beta1, epsilon, weights_full_precision,
	                   (half_t*)weights, (const half_t*)gradients, first_moments, layer_second_moment);
}
void lookahead_step(hipStream_t stream, size_t n, float alpha, float* weights_full_precision, void* weights, void* weights_lookahead) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_lookahead_step, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, alpha, weights_full_precision, (half_t*)weights, (half_t*)weights_lookahead);
}

void ema_step(hipStream_t stream, size_t n, float decay, float debias_old, float debias_new, const void* weights, void* weights_ema, float* tmp) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_ema, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, decay, debias_old, debias_new, (const half_t*)weights, (half_t*)weights_ema, tmp);
}

namespace {
void pcg32_advance_host(uint64_t* st, uint64_t delta) {
	const uint64_t MULT = 0x5851f42d4c957f2dULL;
	uint64_t cur_mult = MULT, cur_plus = st[1], acc_mult = 1u, acc_plus = 0u;
	while (delta > 0) {
		if (delta & 1) {
			acc_mult *= cur_mult;
			acc_plus = acc_plus * cur_mult + cur_plus;
		}
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta /= 2;
	}
	st[0] = acc_mult * st[0] + acc_plus;
}
} // namespace

void generate_random_uniform(hipStream_t stream, uint64_t* state_inc_host, size_t n, float* out, float lower, float upper) {
	if (n > 0) {
		const size_t n_threads = (n + 3) / 4;
		const uint32_t blocks = blocks_for(n_threads, 128); // n_blocks_linear(n_threads), N_THREADS_LINEAR = 128 (common.h:236-245)
		hipLaunchKernelGGL(k_random_uniform, dim3(blocks), dim3(128), 0, stream, n, state_inc_host[0], state_inc_host[1], out, lower, upper);
	}
	pcg32_advance_host(state_inc_host, (uint64_t)n); // random.h:64
}

void cast_float_to_half(hipStream_t stream, size_t n, const float* in, void* out) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_cast_f2h, dim3(blocks_for((n + 3) / 4, 256)), dim3(256), 0, stream, n, in, (half_t*)out);
}

void cast_half_to_float(hipStream_t stream, size_t n, const void* in, float* out) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_cast_h2f, dim3(blocks_for((n + 3) / 4, 256)), dim3(256), 0, stream, n, (const half_t*)in, out);
}

void trim_and_cast(hipStream_t stream, bool fp32, uint32_t n, uint32_t in_stride, uint32_t dims, const void* in, MatViewMut out) {
	const uint64_t total = (uint64_t)n * dims;
	if (total == 0) return;
	CHECK_THROW(total < (1ull << 32));
	if (fp32) hipLaunchKernelGGL(k_trim_and_cast<float>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, in_stride, dims, (const float*)in, out);
	else hipLaunchKernelGGL(k_trim_and_cast<half_t>, dim3(blocks_for(total, 256)), dim3(256), 0, stream, n, in_stride, dims, (const half_t*)in, out);
}

void fill_half(hipStream_t stream, size_t n, void* out, float value) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_fill_half, dim3(blocks_for(n, 256)), dim3(256), 0, stream, n, (half_t*)out, (half_t)value);
}

} // namespace tcnn_amd
