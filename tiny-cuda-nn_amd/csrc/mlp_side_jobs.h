// mlp_side_jobs.h -- small pieces of the MLP's training step written as device functions, so that a kernel which runs anyway can
// carry one along instead of a ~5 us launch on the step's critical path:
//   * mlp_prep_element: one element of the fragment images (k_mlp.hip, k_mlp_prep) -- carried by k_grid_fwd_planes, which it does
//     not depend on (measured on C3a: the 4.6 us launch ahead of the encoding's forward kernel is gone, that kernel is no slower);
//   * mlp_reduce_block: one block of the fixed-order slab reduction of the weight gradients (k_wgrad_reduce).  Carried on the grid
//     scatter itself it did not pay (model.h, fused_mlp_and_scatter); it rides on the scatter's small finalize launch instead
//     (MlpReduceJob, k_grid_scatter_finalize): the two ~4.5 us reductions of a step are one launch.
#pragma once

#include "tcnn_common.h"

namespace tcnn_amd {

// k index of element j of a chained-layer / natural-order fragment (see k_mlp.hip)
__host__ __device__ inline uint32_t frag_k_chain(uint32_t s, uint32_t q, uint32_t j) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); }
__host__ __device__ inline uint32_t frag_k_natural(uint32_t s, uint32_t q, uint32_t j) { return 32 * s + 8 * q + j; }

// weights (row-major half) -> fragment images, element gid of n_frags_total * 512:
// forward fragment (layer l, row tile t, k-step s), lane (r = lane & 15, q = lane >> 4), element j:  W_l[16 t + r][k], 0 beyond the matrix
// backward fragment (A = W_l^T; row tile t over the COLUMNS of W_l, k over its ROWS):              W_l[k_chain(s, q, j)][16 t + r]
__device__ inline void mlp_prep_element(const MlpDesc& d, const _Float16* __restrict__ params, _Float16* __restrict__ image, const uint32_t gid) {
	uint32_t frag = gid >> 9;
	const uint32_t lane = (gid >> 3) & 63;
	const uint32_t j = gid & 7;
	const uint32_t r = lane & 15, q = lane >> 4;
	const bool bwd = frag >= d.n_frags_fwd;
	if (bwd) frag -= d.n_frags_fwd;
	uint32_t l = 0;
	for (uint32_t i = 1; i < d.n_layers; ++i) {
		if (frag >= (bwd ? d.layers[i].bwd_off : d.layers[i].fwd_off)) l = i;
	}
	const MlpLayer L = d.layers[l];
	const _Float16* W = params + L.w_off;
	_Float16 v = (_Float16)0.0f;
	if (!bwd) {
		const uint32_t local = frag - L.fwd_off;
		const uint32_t t = local / L.ks_fwd, s = local - t * L.ks_fwd;
		const uint32_t row = 16 * t + r;
		const uint32_t k = L.natural_k ? frag_k_natural(s, q, j) : frag_k_chain(s, q, j);
		if (row < L.rows && k < L.cols) v = W[(size_t)row * L.cols + k];
	} else {
		const uint32_t local = frag - L.bwd_off;
		const uint32_t t = local / L.ks_bwd, s = local - t * L.ks_bwd;
		const uint32_t col = 16 * t + r;
		const uint32_t k = frag_k_chain(s, q, j);
		if (col < L.cols && k < L.rows) v = W[(size_t)k * L.cols + col];
	}
	image[gid] = v;
}

// side job of the encoding's forward kernel: build the fragment images of `params` (all n_frags_fwd + n_frags_bwd fragments)
struct MlpPrepJob {
	MlpDesc desc;
	const void* params; // half, the MLP's matrices at the front
	void* image;
};

// grad[i] (=|+=) sum_k slabs[k][i] in a fixed order, rounded to half once: block `block` of ceil(n_elems / 64), 1024 threads,
// part = 16 x 64 floats of LDS
constexpr int SLAB_REDUCE_ELEMS = 64, SLAB_REDUCE_GROUPS = 16;
__device__ inline void mlp_reduce_block(float* part, const uint32_t block, const uint32_t tid, const uint32_t n_elems, const uint32_t cols, const uint32_t ldg, const uint32_t n_slabs,
                                        const float* __restrict__ slabs, _Float16* __restrict__ grad, const int accumulate) {
	const uint32_t e = tid & (SLAB_REDUCE_ELEMS - 1), grp = tid / SLAB_REDUCE_ELEMS;
	const uint32_t i = block * SLAB_REDUCE_ELEMS + e;
	float p[4] = {0, 0, 0, 0};
	if (i < n_elems) {
		uint32_t k = grp;
		for (; k + 3 * SLAB_REDUCE_GROUPS < n_slabs; k += 4 * SLAB_REDUCE_GROUPS) {
#pragma unroll
			for (int u = 0; u < 4; ++u) p[u] += slabs[(size_t)(k + u * SLAB_REDUCE_GROUPS) * n_elems + i];
		}
		for (; k < n_slabs; k += SLAB_REDUCE_GROUPS) p[0] += slabs[(size_t)k * n_elems + i];
	}
	part[grp * SLAB_REDUCE_ELEMS + e] = (p[0] + p[1]) + (p[2] + p[3]);
	__syncthreads();
	if (grp == 0 && i < n_elems) {
		float s = 0;
#pragma unroll
		for (int g = 0; g < SLAB_REDUCE_GROUPS; ++g) s += part[g * SLAB_REDUCE_ELEMS + e];
		const uint32_t row = i / cols, col = i - row * cols;
		_Float16* g = grad + (size_t)row * ldg + col;
		if (accumulate) s += (float)*g;
		*g = (_Float16)s;
	}
}

// side job of the grid scatter's finalize launch: grad[i] (=|+=) sum over the slabs, for the MLP's n_elems weights
struct MlpReduceJob {
	uint32_t n_elems = 0, n_slabs = 0;
	const float* slabs = nullptr;
	void* grad = nullptr; // half
	int accumulate = 0;
	mutable bool taken = false; // set by the launch that carried the job
};

} // namespace tcnn_amd
