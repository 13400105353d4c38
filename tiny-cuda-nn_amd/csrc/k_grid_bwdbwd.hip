// k_grid_bwdbwd.hip -- second-order input gradients of the grid encoding: backward_backward_input.
//
// Replaces (reference, /root/reference/include/tiny-cuda-nn/encodings/grid.h):
//   :352-454  kernel_grid_backward_input_backward_grid       d/dgrid   of  sum_i dL_ddLdx_i . dL_dx_i     -> k_grid_bwdbwd_grid
//   :456-624  kernel_grid_backward_input_backward_input      d/dx      (Hessian terms)                    -> k_grid_bwdbwd_input
//   :626-650  kernel_grid_backward_input_backward_dLdoutput  d/d(dL_dy)                                   -> k_grid_bwdbwd_dLdoutput
//   :902-1026 backward_backward_input_impl (host)                                                        -> grid_backward_backward_input
// Used by eikonal / SDF style callers through tcnn.Encoding's double backward (modules.py:120-160); not on the training
// hot path, so the shapes are the plain ones: the grid term scatters with global packed-fp16 atomics like the reference
// (order-dependent in the last fp16 bits), the input term is ONE thread per sample that walks levels and feature pairs in
// order -- deterministic, and bit-identical to the oracle -- where the reference adds per-thread partial sums with fp32 atomics.
#include "grid_device.h"

namespace tcnn_amd {
namespace {

__device__ inline float smoothstep_2nd_derivative(float v) { return 6.0f - 12.0f * v; } // common_device.h:809-811

// pos_fract with first and second derivative (common_device.h:825-838)
template <int D>
__device__ inline void fractions(const MatView& x, const uint32_t i, const float scale, const uint32_t interpolation, float (&pos)[D], float (&d1)[D], float (&d2)[D], uint32_t (&cell)[D]) {
#pragma unroll
	for (int d = 0; d < D; ++d) {
		float p = fmaf(scale, x.data[(size_t)i * x.stride_sample + (size_t)d * x.stride_dim], 0.5f);
		const float tmp = floorf(p);
		cell[d] = (uint32_t)(int)tmp;
		p -= tmp;
		if (interpolation == (uint32_t)InterpolationType::Smoothstep) {
			d2[d] = smoothstep_2nd_derivative(p);
			d1[d] = smoothstep_derivative(p);
			pos[d] = smoothstep(p);
		} else {
			d2[d] = 0.0f;
			d1[d] = 1.0f;
			pos[d] = p;
		}
	}
}

template <typename GT> __device__ inline void atomic_add_pair(GT* p, GT a, GT b);
template <> __device__ inline void atomic_add_pair<half_t>(half_t* p, half_t a, half_t b) {
	__half2 v;
	v.x = *(const __half*)&a;
	v.y = *(const __half*)&b;
	unsafeAtomicAdd((__half2*)p, v);
}

// one thread per (sample, level); grad is accumulated in place (the caller zeroes it for GradientMode::Overwrite, grid.h:944)
template <typename T, typename GT, int D, int F>
__global__ void __launch_bounds__(256) k_grid_bwdbwd_grid(const GridMeta* __restrict__ meta, const uint32_t n, const MatView x, const MatView dL_ddLdx, const T* __restrict__ dL_dy,
                                                          const uint32_t dy_stride, GT* __restrict__ grad) {
	typedef typename VecOf<T, F>::type vecF;
	const uint32_t n_levels = meta->n_levels;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / n_levels;
	if (i >= n) return;
	const uint32_t level = gid - i * n_levels;
	const uint32_t interpolation = meta->interpolation;
	if (interpolation == (uint32_t)InterpolationType::Nearest) return; // d(dy_dx)/dgrid is zero without interpolation (grid.h:420-423)
	const uint32_t hash_type = meta->hash_type;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = meta->primes[d];
	const GridLevel lv = meta->levels[level];
	GT* __restrict__ lgrad = grad + (size_t)lv.offset * F;

	float pos[D], d1[D], d2[D];
	uint32_t cell[D];
	fractions<D>(x, i, lv.scale, interpolation, pos, d1, d2, cell);
	const vecF gv = *(const vecF*)&dL_dy[(size_t)i * dy_stride + level * F];
	T g[F];
#pragma unroll
	for (int f = 0; f < F; ++f) { if constexpr (F == 1) g[f] = gv; else g[f] = gv[f]; }

	auto add = [&](const uint32_t* local, float weight) { // grid.h:393-396
		GT* p = lgrad + (size_t)level_index<D>(lv, primes, hash_type, local) * F;
		if constexpr (sizeof(GT) == 2) {
			asm volatile("" : "+v"(weight)); // round the fp32 product before the cast (hipcc would fold mul + cvt, see k_grid.hip)
			const GT w = (GT)weight;
#pragma unroll
			for (int f = 0; f < F; f += 2) atomic_add_pair<GT>(p + f, (GT)(w * (GT)g[f]), (GT)(w * (GT)g[f + 1]));
		} else {
#pragma unroll
			for (int f = 0; f < F; ++f) unsafeAtomicAdd((float*)p + f, weight * (float)g[f]);
		}
	};

#pragma unroll
	for (int grad_dim = 0; grad_dim < D; ++grad_dim) {
		const float grad_in = lv.scale * dL_ddLdx.data[(size_t)i * dL_ddLdx.stride_sample + (size_t)grad_dim * dL_ddLdx.stride_dim] * d1[grad_dim];
#pragma unroll
		for (int idx = 0; idx < (1 << (D - 1)); ++idx) {
			float weight = grad_in;
			uint32_t local[D];
#pragma unroll
			for (int ngd = 0; ngd < D - 1; ++ngd) {
				const int d = ngd >= grad_dim ? (ngd + 1) : ngd;
				if ((idx & (1 << ngd)) == 0) {
					weight *= 1 - pos[d];
					local[d] = cell[d];
				} else {
					weight *= pos[d];
					local[d] = cell[d] + 1;
				}
			}
			local[grad_dim] = cell[grad_dim];
			add(local, -weight);
			local[grad_dim] = cell[grad_dim] + 1;
			add(local, weight);
		}
	}
}

// one thread per sample: levels, then feature pairs, in order
template <typename T, int D, int F>
__global__ void __launch_bounds__(128) k_grid_bwdbwd_input(const GridMeta* __restrict__ meta, const uint32_t n, const MatView x, const MatView dL_ddLdx, const T* __restrict__ dL_dy,
                                                           const uint32_t dy_stride, const T* __restrict__ grid, const MatViewMut dL_dx) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	constexpr int FT = F < 2 ? F : 2; // N_FEATURES_PER_THREAD (grid.h:998)
	const uint32_t interpolation = meta->interpolation;
	const uint32_t hash_type = meta->hash_type;
	const bool smooth = interpolation == (uint32_t)InterpolationType::Smoothstep;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = meta->primes[d];
	float v[D], out[D];
#pragma unroll
	for (int d = 0; d < D; ++d) {
		v[d] = dL_ddLdx.data[(size_t)i * dL_ddLdx.stride_sample + (size_t)d * dL_ddLdx.stride_dim];
		out[d] = 0.0f;
	}
	const uint32_t n_levels = interpolation == (uint32_t)InterpolationType::Nearest ? 0u : meta->n_levels; // grid.h:525-528
	for (uint32_t level = 0; level < n_levels; ++level) {
		const GridLevel lv = meta->levels[level];
		const T* __restrict__ lgrid = grid + (size_t)lv.offset * F;
		float pos[D], d1[D], d2[D];
		uint32_t cell[D];
		fractions<D>(x, i, lv.scale, interpolation, pos, d1, d2, cell);
		float diag[D], other[D];
#pragma unroll
		for (int gd = 0; gd < D; ++gd) {
			diag[gd] = lv.scale * lv.scale * v[gd] * d2[gd];
			other[gd] = lv.scale * lv.scale * v[gd] * d1[gd];
		}
		for (int feature = 0; feature < F; feature += FT) {
			const T* gy = dL_dy + (size_t)i * dy_stride + level * F + feature;
			auto calc = [&](const uint32_t* local, const float weight) { // grid.h:531-541
				const size_t index = (size_t)level_index<D>(lv, primes, hash_type, local) * F + feature;
				float r = 0;
#pragma unroll
				for (int f = 0; f < FT; ++f) r += (float)lgrid[index + f] * (float)gy[f] * weight;
				return r;
			};
#pragma unroll
			for (int grad_dim = 0; grad_dim < D; ++grad_dim) {
				float grad_out = 0;
#pragma unroll
				for (int idx = 0; idx < (1 << (D - 1)); ++idx) {
					if (smooth) { // diagonal of the Hessian; zero for linear interpolation
						float w = diag[grad_dim];
						uint32_t local[D];
#pragma unroll
						for (int ngd = 0; ngd < D - 1; ++ngd) {
							const int d = ngd >= grad_dim ? (ngd + 1) : ngd;
							if ((idx & (1 << ngd)) == 0) {
								w *= 1 - pos[d];
								local[d] = cell[d];
							} else {
								w *= pos[d];
								local[d] = cell[d] + 1;
							}
						}
						local[grad_dim] = cell[grad_dim];
						grad_out += calc(local, -w);
						local[grad_dim] = cell[grad_dim] + 1;
						grad_out += calc(local, w);
					}
					if constexpr (D > 1) { // mixed part: d(dy/d[other])/d[grad_dim]
#pragma unroll
						for (int og = 0; og < D - 1; ++og) {
							const int rog = og >= grad_dim ? (og + 1) : og;
							float w = other[rog] * d1[grad_dim];
							uint32_t local[D];
#pragma unroll
							for (int ngd = 0; ngd < D - 1; ++ngd) {
								const int d = ngd >= rog ? (ngd + 1) : ngd;
								if ((idx & (1 << ngd)) == 0) {
									if (d != grad_dim) w *= 1 - pos[d];
									else w *= -1;
									local[d] = cell[d];
								} else {
									if (d != grad_dim) w *= pos[d];
									local[d] = cell[d] + 1;
								}
							}
							local[rog] = cell[rog];
							grad_out += calc(local, -w);
							local[rog] = cell[rog] + 1;
							grad_out += calc(local, w);
						}
					}
				}
				out[grad_dim] += grad_out;
			}
		}
	}
#pragma unroll
	for (int d = 0; d < D; ++d) dL_dx.data[(size_t)i * dL_dx.stride_sample + (size_t)d * dL_dx.stride_dim] = out[d];
}

// one thread per sample; padded columns are written 0
template <typename T, int D>
__global__ void __launch_bounds__(128) k_grid_bwdbwd_dLdoutput(const uint32_t n, const uint32_t n_features, const MatView dL_ddLdx, const float* __restrict__ dy_dx, T* __restrict__ dL_ddLdy,
                                                               const uint32_t dy_stride) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float v[D];
#pragma unroll
	for (int d = 0; d < D; ++d) v[d] = dL_ddLdx.data[(size_t)i * dL_ddLdx.stride_sample + (size_t)d * dL_ddLdx.stride_dim];
	for (uint32_t k = 0; k < dy_stride; ++k) {
		float result = 0;
		if (k < n_features) {
#pragma unroll
			for (int d = 0; d < D; ++d) result += dy_dx[((size_t)i * n_features + k) * D + d] * v[d];
		}
		dL_ddLdy[(size_t)i * dy_stride + k] = (T)result;
	}
}

template <typename T, typename GT, int D>
void launch_all(hipStream_t s, const GridMeta& meta, const GridMeta* dm, uint32_t n, MatView x, MatView v, const void* dy, uint32_t ds, const void* grid, const float* dy_dx, void* grad, void* ddy,
                MatViewMut* dx) {
	const uint32_t F = meta.n_features_per_level;
	if (grad) {
		const uint64_t total = (uint64_t)n * meta.n_levels;
		CHECK_THROW(total < (1ull << 32));
		const dim3 blocks((uint32_t)((total + 255) / 256));
#define TCNN_BB_GRID(F_) hipLaunchKernelGGL((k_grid_bwdbwd_grid<T, GT, D, F_>), blocks, dim3(256), 0, s, dm, n, x, v, (const T*)dy, ds, (GT*)grad)
		switch (F) {
			case 1:
				if constexpr (sizeof(GT) == 4) TCNN_BB_GRID(1);
				else throw std::runtime_error{"GridEncoding: F == 1 accumulates gradients in fp32"};
				break;
			case 2: TCNN_BB_GRID(2); break;
			case 4: TCNN_BB_GRID(4); break;
			case 8: TCNN_BB_GRID(8); break;
			default: throw std::runtime_error{"GridEncoding: n_features_per_level must be 1, 2, 4, or 8."};
		}
#undef TCNN_BB_GRID
	}
	if (ddy) {
		CHECK_THROW(dy_dx != nullptr);
		hipLaunchKernelGGL((k_grid_bwdbwd_dLdoutput<T, D>), dim3(div_round_up(n, 128)), dim3(128), 0, s, n, meta.n_levels * F, v, dy_dx, (T*)ddy, ds);
	}
	if (dx) {
		const dim3 blocks(div_round_up(n, 128));
#define TCNN_BB_IN(F_) hipLaunchKernelGGL((k_grid_bwdbwd_input<T, D, F_>), blocks, dim3(128), 0, s, dm, n, x, v, (const T*)dy, ds, (const T*)grid, *dx)
		switch (F) {
			case 1: TCNN_BB_IN(1); break;
			case 2: TCNN_BB_IN(2); break;
			case 4: TCNN_BB_IN(4); break;
			case 8: TCNN_BB_IN(8); break;
			default: throw std::runtime_error{"GridEncoding: n_features_per_level must be 1, 2, 4, or 8."};
		}
#undef TCNN_BB_IN
	}
	HIP_CHECK_THROW(hipGetLastError());
}

} // namespace

void grid_backward_backward_input(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, bool fp32, bool fp32_grad, uint32_t n, MatView x, MatView dL_ddLdx, const void* dL_dy,
                                  uint32_t dy_stride, const void* grid, const float* dy_dx, void* grad, void* dL_ddLdy, MatViewMut* dL_dx) {
	if (n == 0) return;
#define TCNN_BB(T, GT) \
	switch (meta.n_pos_dims) { \
		case 2: return launch_all<T, GT, 2>(stream, meta, dev_meta, n, x, dL_ddLdx, dL_dy, dy_stride, grid, dy_dx, grad, dL_ddLdy, dL_dx); \
		case 3: return launch_all<T, GT, 3>(stream, meta, dev_meta, n, x, dL_ddLdx, dL_dy, dy_stride, grid, dy_dx, grad, dL_ddLdy, dL_dx); \
		case 4: return launch_all<T, GT, 4>(stream, meta, dev_meta, n, x, dL_ddLdx, dL_dy, dy_stride, grid, dy_dx, grad, dL_ddLdy, dL_dx); \
		default: throw std::runtime_error{"GridEncoding: number of input dims must be 2 or 3."}; \
	}
	if (fp32) {
		CHECK_THROW(fp32_grad);
		TCNN_BB(float, float)
	} else if (fp32_grad) {
		TCNN_BB(half_t, float)
	} else {
		TCNN_BB(half_t, half_t)
	}
#undef TCNN_BB
}

} // namespace tcnn_amd
