// k_grid.hip -- multiresolution hash / dense / tiled grid encoding for gfx950.
//
// Replaces (reference, /root/reference):
//   include/tiny-cuda-nn/encodings/grid.h:49-212   kernel_grid            -> k_grid_fwd
//   include/tiny-cuda-nn/encodings/grid.h:215-320  kernel_grid_backward   -> k_grid_bwd
//   include/tiny-cuda-nn/encodings/grid.h:323-349  kernel_grid_backward_input -> k_grid_bwd_input
//   include/tiny-cuda-nn/common_device.h:631-718, 825-868   hashing / indexing / pos_fract
//
// Differences in SHAPE (not in arithmetic):
//   * one thread produces 8 consecutive output features (8/F levels) of one sample and stores them with one 16-byte
//     store into an AoS [n][stride] matrix -- the layout the MFMA MLP consumes directly; the reference writes SoA and
//     transposes.  Per-feature arithmetic (fp32 weights, fp16 hfma chain in corner order) is bit-identical.
//   * the per-level scale / dense strides / hash-or-dense decision are computed once on the host (GridLevel), including
//     the uint32 wrap-around quirk of grid_index's stride loop, so device and host can never disagree.
//
// Compiled with -ffp-contract=off: every fused multiply-add is explicit.
#include "grid_device.h"

namespace tcnn_amd {
namespace {

template <typename T> __device__ inline T fma_t(T a, T b, T c);
template <> __device__ inline float fma_t<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ inline half_t fma_t<half_t>(half_t a, half_t b, half_t c) { return __builtin_fmaf16(a, b, c); }

template <typename T, int D, int F>
__global__ void __launch_bounds__(256) k_grid_fwd(
	const GridMeta* __restrict__ meta, const uint32_t n, const MatView x, const T* __restrict__ grid,
	T* __restrict__ out, const uint32_t out_stride, const uint32_t n_chunks, float* __restrict__ dy_dx, unsigned long long* __restrict__ chunk_mask
) {
	constexpr int LPT = 8 / F; // levels per thread: 8 output features
	typedef typename VecOf<T, F>::type vecF;

	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / n_chunks;
	if (i >= n) return;
	const uint32_t chunk = gid - i * n_chunks;

	const uint32_t n_levels = meta->n_levels;
	const uint32_t interpolation = meta->interpolation;
	const uint32_t hash_type = meta->hash_type;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = meta->primes[d];

	float xin[D];
	load_coords<D>(x, i, xin);

	T res[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) res[k] = (T)0.0f;

#pragma unroll
	for (int ll = 0; ll < LPT; ++ll) {
		const uint32_t level = chunk * LPT + ll;
		if (level >= n_levels) continue;
		const GridLevel lv = meta->levels[level];
		const T* __restrict__ lgrid = grid + (size_t)lv.offset * F;

		float pos[D], pos_derivative[D];
		uint32_t cell[D];
#pragma unroll
		for (int d = 0; d < D; ++d) cell[d] = pos_fract(xin[d], lv.scale, interpolation, &pos[d], &pos_derivative[d]);

		if (interpolation == (uint32_t)InterpolationType::Nearest) { // grid.h:121-140
			const uint32_t index = level_index<D>(lv, primes, hash_type, cell);
			const vecF v = *(const vecF*)&lgrid[(size_t)index * F];
			if (chunk_mask) {
				const uint32_t ch = scatter_chunk(lv, index);
#pragma unroll
				for (uint32_t half = 0; half < GRID_FILTER_MAX_CHUNKS / 64; ++half) {
					chunk_mask[((size_t)level * n + i) * (GRID_FILTER_MAX_CHUNKS / 64) + half] = (ch >> 6) == half ? 1ull << (ch & 63u) : 0ull;
				}
			}
#pragma unroll
			for (int f = 0; f < F; ++f) { if constexpr (F == 1) res[ll * F + f] = v; else res[ll * F + f] = v[f]; }
			if (dy_dx) {
#pragma unroll
				for (int f = 0; f < F; ++f)
#pragma unroll
					for (int d = 0; d < D; ++d) dy_dx[((size_t)i * n_levels * F + level * F + f) * D + d] = 0.0f;
			}
			continue;
		}

		// N-linear interpolation (grid.h:142-169): weight product in fp32 (dim order), cast to T, fma chain in corner order
		T acc[F];
		unsigned long long touched[GRID_FILTER_MAX_CHUNKS / 64]; // scatter chunks of this level the sample's corners fall into
#pragma unroll
		for (uint32_t half = 0; half < GRID_FILTER_MAX_CHUNKS / 64; ++half) touched[half] = 0;
#pragma unroll
		for (int f = 0; f < F; ++f) acc[f] = (T)0.0f;
#pragma unroll
		for (int idx = 0; idx < (1 << D); ++idx) {
			float weight = 1;
			uint32_t local[D];
#pragma unroll
			for (int d = 0; d < D; ++d) {
				if ((idx & (1 << d)) == 0) {
					weight *= 1 - pos[d];
					local[d] = cell[d];
				} else {
					weight *= pos[d];
					local[d] = cell[d] + 1;
				}
			}
			const uint32_t index = level_index<D>(lv, primes, hash_type, local);
			const vecF v = *(const vecF*)&lgrid[(size_t)index * F];
			{
				const uint32_t ch = scatter_chunk(lv, index);
#pragma unroll
				for (uint32_t half = 0; half < GRID_FILTER_MAX_CHUNKS / 64; ++half) touched[half] |= (ch >> 6) == half ? 1ull << (ch & 63u) : 0ull;
			}
			// The reference rounds the fp32 weight product to fp32 FIRST and to T afterwards.  Without this barrier hipcc
			// folds "fp32 multiply + convert" into v_fma_mixlo_f16 (one rounding from the exact product), which differs
			// from the reference in ~1e-5 of the weights.
			asm volatile("" : "+v"(weight));
			const T w = (T)weight;
#pragma unroll
			for (int f = 0; f < F; ++f) {
				T val;
				if constexpr (F == 1) val = v; else val = v[f];
				acc[f] = fma_t<T>(w, val, acc[f]);
			}
		}
#pragma unroll
		for (int f = 0; f < F; ++f) res[ll * F + f] = acc[f];
		if (chunk_mask) {
#pragma unroll
			for (uint32_t half = 0; half < GRID_FILTER_MAX_CHUNKS / 64; ++half) chunk_mask[((size_t)level * n + i) * (GRID_FILTER_MAX_CHUNKS / 64) + half] = touched[half];
		}

		if (dy_dx) { // grid.h:172-211
			float grads[F][D];
#pragma unroll
			for (int f = 0; f < F; ++f)
#pragma unroll
				for (int d = 0; d < D; ++d) grads[f][d] = 0.0f;
#pragma unroll
			for (int grad_dim = 0; grad_dim < D; ++grad_dim) {
#pragma unroll
				for (int idx = 0; idx < (1 << (D - 1)); ++idx) {
					float weight = lv.scale;
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
					const vecF vl = *(const vecF*)&lgrid[(size_t)level_index<D>(lv, primes, hash_type, local) * F];
					local[grad_dim] = cell[grad_dim] + 1;
					const vecF vr = *(const vecF*)&lgrid[(size_t)level_index<D>(lv, primes, hash_type, local) * F];
#pragma unroll
					for (int f = 0; f < F; ++f) {
						float l, r;
						if constexpr (F == 1) { l = (float)vl; r = (float)vr; } else { l = (float)vl[f]; r = (float)vr[f]; }
						grads[f][grad_dim] += weight * (r - l) * pos_derivative[grad_dim];
					}
				}
			}
#pragma unroll
			for (int f = 0; f < F; ++f)
#pragma unroll
				for (int d = 0; d < D; ++d) dy_dx[((size_t)i * n_levels * F + level * F + f) * D + d] = grads[f][d];
		}
	}

	T* o = out + (size_t)i * out_stride + chunk * 8;
	if ((out_stride & 7u) == 0) {
		typedef T vec8 __attribute__((ext_vector_type(8)));
		vec8 v;
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = res[k];
		*(vec8*)o = v;
	} else {
#pragma unroll
		for (int k = 0; k < 8; ++k) if (chunk * 8 + k < out_stride) o[k] = res[k];
	}
}

template <typename GT> __device__ inline void atomic_add_pair(GT* p, GT a, GT b);
template <> __device__ inline void atomic_add_pair<half_t>(half_t* p, half_t a, half_t b) {
	// one global_atomic_pk_add_f16 (vec.h:326-348 uses red.global.add.noftz.f16x2)
	__half2 v;
	v.x = *(const __half*)&a;
	v.y = *(const __half*)&b;
	unsafeAtomicAdd((__half2*)p, v);
}

template <typename T, typename GT, int D, int F>
__global__ void __launch_bounds__(256) k_grid_bwd(
	const GridMeta* __restrict__ meta, const uint32_t n, const MatView x, const T* __restrict__ dL_dy, const uint32_t dy_stride, GT* __restrict__ grad
) {
	typedef typename VecOf<T, F>::type vecF;
	const uint32_t n_levels = meta->n_levels;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i = gid / n_levels;
	if (i >= n) return;
	const uint32_t level = gid - i * n_levels;

	const uint32_t interpolation = meta->interpolation;
	const uint32_t hash_type = meta->hash_type;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = meta->primes[d];
	const GridLevel lv = meta->levels[level];
	GT* __restrict__ lgrad = grad + (size_t)lv.offset * F;

	float pos[D], unused;
	uint32_t cell[D];
#pragma unroll
	for (int d = 0; d < D; ++d) cell[d] = pos_fract(x.data[(size_t)i * x.stride_sample + (size_t)d * x.stride_dim], lv.scale, interpolation, &pos[d], &unused);

	const vecF gv = *(const vecF*)&dL_dy[(size_t)i * dy_stride + level * F];
	T g[F];
#pragma unroll
	for (int f = 0; f < F; ++f) { if constexpr (F == 1) g[f] = gv; else g[f] = gv[f]; }

	auto add = [&](const uint32_t* local, float weight) { // grid.h:252-255
		const uint32_t index = level_index<D>(lv, primes, hash_type, local);
		GT* p = lgrad + (size_t)index * F;
		if constexpr (sizeof(GT) == 2) {
			asm volatile("" : "+v"(weight)); // keep the fp32 rounding of the weight product (see k_grid_fwd)
			const GT w = (GT)weight;
#pragma unroll
			for (int f = 0; f < F; f += 2) atomic_add_pair<GT>(p + f, (GT)(w * (GT)g[f]), (GT)(w * (GT)g[f + 1]));
		} else {
#pragma unroll
			for (int f = 0; f < F; ++f) unsafeAtomicAdd((float*)p + f, weight * (float)g[f]);
		}
	};

	if (interpolation == (uint32_t)InterpolationType::Nearest) {
		add(cell, 1.0f);
		return;
	}
#pragma unroll
	for (int idx = 0; idx < (1 << D); ++idx) {
		float weight = 1;
		uint32_t local[D];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if ((idx & (1 << d)) == 0) {
				weight *= 1 - pos[d];
				local[d] = cell[d];
			} else {
				weight *= pos[d];
				local[d] = cell[d] + 1;
			}
		}
		add(local, weight);
	}
}

template <typename T, int D>
__global__ void __launch_bounds__(128) k_grid_bwd_input(
	const uint32_t n, const uint32_t n_features, const T* __restrict__ dL_dy, const uint32_t dy_stride, const float* __restrict__ dy_dx, const MatViewMut dL_dx
) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float result[D];
#pragma unroll
	for (int d = 0; d < D; ++d) result[d] = 0.0f;
	for (uint32_t k = 0; k < n_features; ++k) {
		const float dl = (float)dL_dy[(size_t)i * dy_stride + k];
#pragma unroll
		for (int d = 0; d < D; ++d) result[d] += dl * dy_dx[((size_t)i * n_features + k) * D + d];
	}
#pragma unroll
	for (int d = 0; d < D; ++d) dL_dx.data[(size_t)i * dL_dx.stride_sample + (size_t)d * dL_dx.stride_dim] = result[d];
}

template <typename T, int D, int F>
void launch_fwd(hipStream_t stream, const GridMeta* dev_meta, uint32_t n, MatView x, const void* grid, void* out, uint32_t out_stride, float* dy_dx, uint64_t* chunk_mask) {
	const uint32_t n_chunks = div_round_up(out_stride, 8);
	const uint64_t total = (uint64_t)n * n_chunks;
	CHECK_THROW(total < (1ull << 32));
	const uint32_t blocks = (uint32_t)((total + 255) / 256);
	if (blocks == 0) return;
	hipLaunchKernelGGL((k_grid_fwd<T, D, F>), dim3(blocks), dim3(256), 0, stream, dev_meta, n, x, (const T*)grid, (T*)out, out_stride, n_chunks, dy_dx, (unsigned long long*)chunk_mask);
}

template <typename T, typename GT, int D, int F>
void launch_bwd(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, uint32_t n, MatView x, const void* dL_dy, uint32_t dy_stride, void* grad) {
	const uint64_t total = (uint64_t)n * meta.n_levels;
	CHECK_THROW(total < (1ull << 32));
	const uint32_t blocks = (uint32_t)((total + 255) / 256);
	if (blocks == 0) return;
	hipLaunchKernelGGL((k_grid_bwd<T, GT, D, F>), dim3(blocks), dim3(256), 0, stream, dev_meta, n, x, (const T*)dL_dy, dy_stride, (GT*)grad);
}

template <typename T, int D>
void dispatch_fwd_F(hipStream_t s, uint32_t F, const GridMeta* dm, uint32_t n, MatView x, const void* grid, void* out, uint32_t os, float* dy_dx, uint64_t* cm) {
	switch (F) {
		case 1: return launch_fwd<T, D, 1>(s, dm, n, x, grid, out, os, dy_dx, cm);
		case 2: return launch_fwd<T, D, 2>(s, dm, n, x, grid, out, os, dy_dx, cm);
		case 4: return launch_fwd<T, D, 4>(s, dm, n, x, grid, out, os, dy_dx, cm);
		case 8: return launch_fwd<T, D, 8>(s, dm, n, x, grid, out, os, dy_dx, cm);
		default: throw std::runtime_error{"GridEncoding: n_features_per_level must be 1, 2, 4, or 8."};
	}
}

template <typename T, typename GT, int D>
void dispatch_bwd_F(hipStream_t s, const GridMeta& m, const GridMeta* dm, uint32_t n, MatView x, const void* dy, uint32_t ds, void* grad) {
	switch (m.n_features_per_level) {
		case 1:
			if constexpr (sizeof(GT) == 4) return launch_bwd<T, GT, D, 1>(s, m, dm, n, x, dy, ds, grad);
			else throw std::runtime_error{"GridEncoding: F == 1 accumulates gradients in fp32"};
		case 2: return launch_bwd<T, GT, D, 2>(s, m, dm, n, x, dy, ds, grad);
		case 4: return launch_bwd<T, GT, D, 4>(s, m, dm, n, x, dy, ds, grad);
		case 8: return launch_bwd<T, GT, D, 8>(s, m, dm, n, x, dy, ds, grad);
		default: throw std::runtime_error{"GridEncoding: n_features_per_level must be 1, 2, 4, or 8."};
	}
}

} // namespace

void grid_forward(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, bool fp32, uint32_t n, MatView x, const void* grid, void* out, uint32_t out_stride, float* dy_dx,
                  uint64_t* chunk_mask) {
	const uint32_t F = meta.n_features_per_level;
#define TCNN_GRID_FWD(T) \
	switch (meta.n_pos_dims) { \
		case 2: return dispatch_fwd_F<T, 2>(stream, F, dev_meta, n, x, grid, out, out_stride, dy_dx, chunk_mask); \
		case 3: return dispatch_fwd_F<T, 3>(stream, F, dev_meta, n, x, grid, out, out_stride, dy_dx, chunk_mask); \
		case 4: return dispatch_fwd_F<T, 4>(stream, F, dev_meta, n, x, grid, out, out_stride, dy_dx, chunk_mask); \
		default: throw std::runtime_error{"GridEncoding: number of input dims must be 2 or 3."}; \
	}
	if (fp32) { TCNN_GRID_FWD(float) } else { TCNN_GRID_FWD(half_t) }
#undef TCNN_GRID_FWD
}

void grid_backward(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, bool fp32_grad, uint32_t n, MatView x, const void* dL_dy, bool dy_fp32, uint32_t dy_stride, void* grad) {
#define TCNN_GRID_BWD(T, GT) \
	switch (meta.n_pos_dims) { \
		case 2: return dispatch_bwd_F<T, GT, 2>(stream, meta, dev_meta, n, x, dL_dy, dy_stride, grad); \
		case 3: return dispatch_bwd_F<T, GT, 3>(stream, meta, dev_meta, n, x, dL_dy, dy_stride, grad); \
		case 4: return dispatch_bwd_F<T, GT, 4>(stream, meta, dev_meta, n, x, dL_dy, dy_stride, grad); \
		default: throw std::runtime_error{"GridEncoding: number of input dims must be 2 or 3."}; \
	}
	if (dy_fp32) {
		CHECK_THROW(fp32_grad);
		TCNN_GRID_BWD(float, float)
	} else if (fp32_grad) {
		TCNN_GRID_BWD(half_t, float)
	} else {
		TCNN_GRID_BWD(half_t, half_t)
	}
#undef TCNN_GRID_BWD
}

void grid_backward_input(hipStream_t stream, const GridMeta& meta, bool fp32, uint32_t n, const void* dL_dy, uint32_t dy_stride, const float* dy_dx, MatViewMut dL_dx) {
	const uint32_t nf = meta.n_levels * meta.n_features_per_level;
	const uint32_t blocks = div_round_up(n, 128);
	if (blocks == 0) return;
#define TCNN_GRID_BWDI(T) \
	switch (meta.n_pos_dims) { \
		case 2: hipLaunchKernelGGL((k_grid_bwd_input<T, 2>), dim3(blocks), dim3(128), 0, stream, n, nf, (const T*)dL_dy, dy_stride, dy_dx, dL_dx); return; \
		case 3: hipLaunchKernelGGL((k_grid_bwd_input<T, 3>), dim3(blocks), dim3(128), 0, stream, n, nf, (const T*)dL_dy, dy_stride, dy_dx, dL_dx); return; \
		case 4: hipLaunchKernelGGL((k_grid_bwd_input<T, 4>), dim3(blocks), dim3(128), 0, stream, n, nf, (const T*)dL_dy, dy_stride, dy_dx, dL_dx); return; \
		default: throw std::runtime_error{"GridEncoding: number of input dims must be 2 or 3."}; \
	}
	if (fp32) { TCNN_GRID_BWDI(float) } else { TCNN_GRID_BWDI(half_t) }
#undef TCNN_GRID_BWDI
}

} // namespace tcnn_amd
