// k_encodings.hip -- parameter-free encodings beside OneBlob / Identity: Frequency, TriangleWave, SphericalHarmonics.
//
// Replaces (reference, /root/reference/include/tiny-cuda-nn):
//   encodings/frequency.h:44-101           frequency_encoding / _backward        -> k_frequency_fwd, k_periodic_bwd_input
//   encodings/triangle_wave.h:44-107       triangle_wave_encoding / _backward    -> k_trianglewave_fwd, k_periodic_bwd_input
//   encodings/spherical_harmonics.h:44-108 kernel_sh / kernel_sh_backward, common_device.h:339-700 sh_enc / sh_enc_grad
//                                                                                -> k_sh_fwd, k_sh_bwd_input
// All streaming, AoS output [n][out_stride] (padding = 1).  The reference hard-codes the 64 spherical-harmonic polynomials
// of degree <= 8 and their 192 partial derivatives (generated from the recurrences of Sloan, "Stupid Spherical Harmonics
// Tricks", appendix A1); here the same polynomials are evaluated through those recurrences (see the oracle's sh_eval): identical
// functions of (x, y, z), summed in another floating-point order (~1e-7 relative).  Frequency: sinf / cosf instead of the
// reference's __sinf / __cosf hardware approximations.
#include "tcnn_common.h"

#include <hip/hip_fp16.h>

#include <cmath>

namespace tcnn_amd {
namespace {

typedef _Float16 half_t;

struct ShNorms { float v[64]; }; // N_l^m at [l * 8 + m]

template <typename T>
__global__ void __launch_bounds__(256) k_frequency_fwd(const uint32_t n_elements, const uint32_t n_frequencies, const uint32_t n_dims, const uint32_t out_stride, const MatView x,
                                                       T* __restrict__ out, float* __restrict__ dy_dx) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_elements) return;
	const uint32_t fan_out_encoded = n_dims * n_frequencies * 2;
	const uint32_t i = e / out_stride, j = e - i * out_stride;
	if (j >= fan_out_encoded) {
		out[e] = (T)1.0f;
		return;
	}
	const float PI = 3.14159265358979323846f;
	const uint32_t feature = j / (n_frequencies * 2);
	const uint32_t log2_frequency = (j / 2) % n_frequencies;
	const float phase_shift = (j % 2) * (PI / 2);
	const float v = scalbnf(x.data[(size_t)i * x.stride_sample + (size_t)feature * x.stride_dim], (int)log2_frequency);
	const float input = v * PI + phase_shift;
	out[e] = (T)sinf(input);
	if (dy_dx) dy_dx[(size_t)i * fan_out_encoded + j] = scalbnf(1.0f, (int)log2_frequency) * PI * cosf(input);
}

template <typename T>
__global__ void __launch_bounds__(256) k_trianglewave_fwd(const uint32_t n_elements, const uint32_t n_frequencies, const uint32_t n_dims, const uint32_t out_stride, const MatView x,
                                                          T* __restrict__ out, float* __restrict__ dy_dx) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_elements) return;
	const uint32_t fan_out_encoded = n_dims * n_frequencies;
	const uint32_t i = e / out_stride, j = e - i * out_stride;
	if (j >= fan_out_encoded) {
		out[e] = (T)1.0f;
		return;
	}
	const uint32_t feature = j / n_frequencies;
	const int log2_frequency = (int)(j - feature * n_frequencies);
	const float v = scalbnf(x.data[(size_t)i * x.stride_sample + (size_t)feature * x.stride_dim], log2_frequency - 1);
	const float val = v + log2_frequency * 0.25f;
	out[e] = (T)(fabsf(val - floorf(val) - 0.5f) * 4 - 1);
	if (dy_dx) dy_dx[(size_t)i * fan_out_encoded + j] = scalbnf((int)floorf(val * 2.0f) % 2 == 0 ? -1.0f : 1.0f, log2_frequency + 1);
}

template <typename T>
__global__ void __launch_bounds__(256) k_periodic_bwd_input(const uint32_t n_elements, const uint32_t n_dims, const uint32_t outputs_per_input, const T* __restrict__ dL_dy,
                                                            const uint32_t dy_stride, const float* __restrict__ dy_dx, const MatViewMut dL_dx) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_elements) return;
	const uint32_t i = e / n_dims, j = e - i * n_dims;
	float result = 0;
	for (uint32_t k = 0; k < outputs_per_input; ++k) result += (float)dL_dy[(size_t)i * dy_stride + j * outputs_per_input + k] * dy_dx[((size_t)i * n_dims + j) * outputs_per_input + k];
	dL_dx.data[(size_t)i * dL_dx.stride_sample + (size_t)j * dL_dx.stride_dim] = result;
}

// all degree^2 values (values != nullptr) and / or the gradient against dL_dy at one point; same operation order as the oracle's sh_eval
template <typename T>
__device__ inline void sh_eval(const uint32_t degree, const ShNorms& norms, const float x, const float y, const float z, T* __restrict__ values, const T* __restrict__ dL_dy, float (&grad)[3]) {
	float gx = 0, gy = 0, gz = 0;
	float c = 1, s = 0, cp = 0, sp = 0;
	float qmm = 1;
	for (uint32_t m = 0; m < degree; ++m) {
		if (m > 0) {
			cp = c;
			sp = s;
			c = x * cp - y * sp;
			s = x * sp + y * cp;
			qmm = -qmm * (float)(2 * m - 1);
		}
		float q2 = 0, q1 = 0, d2 = 0, d1 = 0;
		for (uint32_t l = m; l < degree; ++l) {
			float q, dq;
			if (l == m) { q = qmm; dq = 0; }
			else if (l == m + 1) { q = (float)(2 * m + 1) * z * q1; dq = (float)(2 * m + 1) * q1; }
			else {
				q = ((float)(2 * l - 1) * z * q1 - (float)(l + m - 1) * q2) / (float)(l - m);
				dq = ((float)(2 * l - 1) * (q1 + z * d1) - (float)(l + m - 1) * d2) / (float)(l - m);
			}
			q2 = q1; q1 = q; d2 = d1; d1 = dq;
			const float nq = norms.v[l * 8 + m] * q, ndq = norms.v[l * 8 + m] * dq;
			const uint32_t base = l * l + l;
			if (m == 0) {
				if (values) values[base] = (T)nq;
				if (dL_dy) gz += (float)dL_dy[base] * ndq;
			} else {
				if (values) { values[base + m] = (T)(nq * c); values[base - m] = (T)(nq * s); }
				if (dL_dy) {
					const float gp = (float)dL_dy[base + m], gm = (float)dL_dy[base - m];
					gx += gp * (nq * (float)m * cp) + gm * (nq * (float)m * sp);
					gy += gp * (nq * -(float)m * sp) + gm * (nq * (float)m * cp);
					gz += gp * (ndq * c) + gm * (ndq * s);
				}
			}
		}
	}
	grad[0] = gx; grad[1] = gy; grad[2] = gz;
}

template <typename T>
__global__ void __launch_bounds__(128) k_sh_fwd(const uint32_t n, const uint32_t degree, const uint32_t n_to_pad, const ShNorms norms, const MatView x, T* __restrict__ out, const uint32_t out_stride) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	T* o = out + (size_t)i * out_stride;
	for (uint32_t j = 0; j < n_to_pad; ++j) o[j] = (T)1.0f; // the padding columns come first (spherical_harmonics.h:58-64)
	float unused[3];
	const float* p = x.data + (size_t)i * x.stride_sample;
	sh_eval<T>(degree, norms, p[0] * 2.f - 1.f, p[x.stride_dim] * 2.f - 1.f, p[2 * (size_t)x.stride_dim] * 2.f - 1.f, o + n_to_pad, nullptr, unused);
}

template <typename T>
__global__ void __launch_bounds__(128) k_sh_bwd_input(const uint32_t n, const uint32_t degree, const uint32_t n_to_pad, const ShNorms norms, const MatView x, const T* __restrict__ dL_dy,
                                                      const uint32_t dy_stride, const MatViewMut dL_dx) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float g[3];
	const float* p = x.data + (size_t)i * x.stride_sample;
	sh_eval<T>(degree, norms, p[0] * 2.f - 1.f, p[x.stride_dim] * 2.f - 1.f, p[2 * (size_t)x.stride_dim] * 2.f - 1.f, (T*)nullptr, dL_dy + (size_t)i * dy_stride + n_to_pad, g);
	// times 2: [0, 1]^3 -> [-1, 1]^3 (spherical_harmonics.h:100-104)
	for (int d = 0; d < 3; ++d) dL_dx.data[(size_t)i * dL_dx.stride_sample + (size_t)d * dL_dx.stride_dim] = 2.0f * g[d];
}

ShNorms make_norms(uint32_t degree) {
	ShNorms n{};
	for (uint32_t l = 0; l < degree; ++l) {
		for (uint32_t m = 0; m <= l; ++m) {
			double ratio = 1.0; // (l - m)! / (l + m)!
			for (uint32_t k = l - m + 1; k <= l + m; ++k) ratio /= (double)k;
			n.v[l * 8 + m] = (float)(std::sqrt((2.0 * l + 1.0) / (4.0 * 3.14159265358979323846) * ratio) * (m ? std::sqrt(2.0) : 1.0));
		}
	}
	return n;
}

} // namespace

void periodic_forward(hipStream_t stream, bool triangle, bool fp32, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, MatView x, void* out, uint32_t out_stride, float* dy_dx) {
	const uint64_t total = (uint64_t)n * out_stride;
	if (total == 0) return;
	CHECK_THROW(total < (1ull << 32));
	const dim3 blocks((uint32_t)((total + 255) / 256));
#define TCNN_PER(K, T) hipLaunchKernelGGL((K<T>), blocks, dim3(256), 0, stream, (uint32_t)total, n_frequencies, n_dims, out_stride, x, (T*)out, dy_dx)
	if (triangle) { if (fp32) TCNN_PER(k_trianglewave_fwd, float); else TCNN_PER(k_trianglewave_fwd, half_t); }
	else { if (fp32) TCNN_PER(k_frequency_fwd, float); else TCNN_PER(k_frequency_fwd, half_t); }
#undef TCNN_PER
}

void periodic_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, uint32_t outputs_per_input, const void* dL_dy, uint32_t dy_stride, const float* dy_dx, MatViewMut dL_dx) {
	const uint64_t total = (uint64_t)n * n_dims;
	if (total == 0) return;
	CHECK_THROW(total < (1ull << 32));
	const dim3 blocks((uint32_t)((total + 255) / 256));
	if (fp32) hipLaunchKernelGGL((k_periodic_bwd_input<float>), blocks, dim3(256), 0, stream, (uint32_t)total, n_dims, outputs_per_input, (const float*)dL_dy, dy_stride, dy_dx, dL_dx);
	else hipLaunchKernelGGL((k_periodic_bwd_input<half_t>), blocks, dim3(256), 0, stream, (uint32_t)total, n_dims, outputs_per_input, (const half_t*)dL_dy, dy_stride, dy_dx, dL_dx);
}

void sh_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t degree, MatView x, void* out, uint32_t out_stride) {
	if (n == 0) return;
	CHECK_THROW(degree >= 1 && degree <= 8 && out_stride >= degree * degree);
	const ShNorms norms = make_norms(degree);
	const uint32_t n_to_pad = out_stride - degree * degree;
	if (fp32) hipLaunchKernelGGL((k_sh_fwd<float>), dim3(div_round_up(n, 128)), dim3(128), 0, stream, n, degree, n_to_pad, norms, x, (float*)out, out_stride);
	else hipLaunchKernelGGL((k_sh_fwd<half_t>), dim3(div_round_up(n, 128)), dim3(128), 0, stream, n, degree, n_to_pad, norms, x, (half_t*)out, out_stride);
}

void sh_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t degree, MatView x, const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx) {
	if (n == 0) return;
	CHECK_THROW(degree >= 1 && degree <= 8 && dy_stride >= degree * degree);
	const ShNorms norms = make_norms(degree);
	const uint32_t n_to_pad = dy_stride - degree * degree;
	if (fp32) hipLaunchKernelGGL((k_sh_bwd_input<float>), dim3(div_round_up(n, 128)), dim3(128), 0, stream, n, degree, n_to_pad, norms, x, (const float*)dL_dy, dy_stride, dL_dx);
	else hipLaunchKernelGGL((k_sh_bwd_input<half_t>), dim3(div_round_up(n, 128)), dim3(128), 0, stream, n, degree, n_to_pad, norms, x, (const half_t*)dL_dy, dy_stride, dL_dx);
}

namespace {
// composite.h:47-133, one thread per element of the reduced output: the nested values are combined in fp32 in nesting order
template <typename T, bool PRODUCT>
__global__ void __launch_bounds__(256) k_composite_reduce_fwd(const size_t n_elems, const uint32_t n_nested, const T* __restrict__ in, T* __restrict__ out) {
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_elems) return;
	float result = PRODUCT ? 1.0f : 0.0f;
	for (uint32_t k = 0; k < n_nested; ++k) {
		const float v = (float)in[(size_t)k * n_elems + e];
		if (PRODUCT) result *= v; else result += v;
	}
	out[e] = (T)result;
}

template <typename T, bool PRODUCT>
__global__ void __launch_bounds__(256) k_composite_reduce_bwd(const size_t n_elems, const uint32_t n_nested, const T* __restrict__ in, const T* __restrict__ dL_dout, T* __restrict__ dL_din) {
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_elems) return;
	const T passed = dL_dout[e];
	for (uint32_t k = 0; k < n_nested; ++k) {
		if (!PRODUCT) {
			dL_din[(size_t)k * n_elems + e] = passed; // :83-86
		} else { // :118-131: the product of the OTHER factors, multiplied up in nesting order (no division)
			float result = (float)passed;
			for (uint32_t l = 0; l + 1 < n_nested; ++l) result *= (float)in[(size_t)(l < k ? l : l + 1) * n_elems + e];
			dL_din[(size_t)k * n_elems + e] = (T)result;
		}
	}
}
} // namespace

void composite_reduce_forward(hipStream_t stream, bool fp32, bool product, size_t n_elems, uint32_t n_nested, const void* in, void* out) {
	if (n_elems == 0) return;
	const dim3 grid((uint32_t)((n_elems + 255) / 256));
	if (fp32) {
		if (product) hipLaunchKernelGGL((k_composite_reduce_fwd<float, true>), grid, dim3(256), 0, stream, n_elems, n_nested, (const float*)in, (float*)out);
		else hipLaunchKernelGGL((k_composite_reduce_fwd<float, false>), grid, dim3(256), 0, stream, n_elems, n_nested, (const float*)in, (float*)out);
	} else {
		if (product) hipLaunchKernelGGL((k_composite_reduce_fwd<_Float16, true>), grid, dim3(256), 0, stream, n_elems, n_nested, (const _Float16*)in, (_Float16*)out);
		else hipLaunchKernelGGL((k_composite_reduce_fwd<_Float16, false>), grid, dim3(256), 0, stream, n_elems, n_nested, (const _Float16*)in, (_Float16*)out);
	}
}

void composite_reduce_backward(hipStream_t stream, bool fp32, bool product, size_t n_elems, uint32_t n_nested, const void* in, const void* dL_dout, void* dL_din) {
	if (n_elems == 0) return;
	const dim3 grid((uint32_t)((n_elems + 255) / 256));
	if (fp32) {
		if (product) hipLaunchKernelGGL((k_composite_reduce_bwd<float, true>), grid, dim3(256), 0, stream, n_elems, n_nested, (const float*)in, (const float*)dL_dout, (float*)dL_din);
		else hipLaunchKernelGGL((k_composite_reduce_bwd<float, false>), grid, dim3(256), 0, stream, n_elems, n_nested, (const float*)in, (const float*)dL_dout, (float*)dL_din);
	} else {
		if (product) hipLaunchKernelGGL((k_composite_reduce_bwd<_Float16, true>), grid, dim3(256), 0, stream, n_elems, n_nested, (const _Float16*)in, (const _Float16*)dL_dout, (_Float16*)dL_din);
		else hipLaunchKernelGGL((k_composite_reduce_bwd<_Float16, false>), grid, dim3(256), 0, stream, n_elems, n_nested, (const _Float16*)in, (const _Float16*)dL_dout, (_Float16*)dL_din);
	}
}

} // namespace tcnn_amd
