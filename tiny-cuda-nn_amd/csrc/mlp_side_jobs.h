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

// ---- fragments for v_mfma_f32_32x32x16_f16 (k_train_r32.hip): the image's third section, 1 KiB per (32-row tile, 16-deep k-step).
// Operand maps of that instruction: lane (r = lane & 31, h = lane >> 5), element j = 0..7: A[row r][k = 8 h + j], B[k = 8 h + j][column r];
// result register g of lane (c, h): D[row (g & 3) + 8 (g >> 2) + 4 h][column c].  A layer's result registers 8 s .. 8 s + 7 of row tile t,
// converted to halves, ARE the B fragment of the next layer's k-step 2 t + s, element j being feature
//     r32_chain_k(2 t + s, h, j) = 16 (2 t + s) + 8 (j >> 2) + 4 h + (j & 3);
// the fragments absorb that order.  The output layer's rows are "positions": position rho < 16 holds output 2 g + h' where
// (g, h') are the result register and lane half that hold row rho (g = (rho & 3) + 4 (rho >> 3), h' = (rho >> 2) & 1) -- every lane then has
// the outputs 2 g + h in its registers g = 0..7, and with at most four outputs two registers per lane are live.  Positions 16..31: zero rows.
__host__ __device__ inline uint32_t r32_chain_k(uint32_t ks, uint32_t h, uint32_t j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }
struct R32Frags {
	uint32_t wt, ks0, ksw, it, nh;   // row tiles of a hidden layer, k-steps of layer 0, k-steps over the hidden width, row tiles of dL/dinput, hidden layers
	__host__ __device__ explicit R32Frags(const MlpDesc& d) : wt(d.width / 32), ks0(d.in_width / 16), ksw(d.width / 16), it((d.in_width + 31) / 32), nh(d.n_hidden) {}
	// forward: layer 0 [t][s], hidden layers l >= 1 [t][ks], output layer [ks]; backward (A = W^T): output layer [t], hidden l >= 1 [t][ks], layer 0 [ti][ks]
	__host__ __device__ uint32_t fwd0() const { return 0; }
	__host__ __device__ uint32_t fwd_hidden(uint32_t l) const { return wt * ks0 + (l - 1) * wt * ksw; }
	__host__ __device__ uint32_t fwd_out() const { return wt * ks0 + (nh - 1) * wt * ksw; }
	__host__ __device__ uint32_t bwd_out() const { return fwd_out() + ksw; }
	__host__ __device__ uint32_t bwd_hidden(uint32_t l) const { return bwd_out() + wt + (l - 1) * wt * ksw; }
	__host__ __device__ uint32_t bwd0() const { return bwd_out() + wt + (nh - 1) * wt * ksw; }
	__host__ __device__ uint32_t n_frags() const { return bwd0() + it * ksw; }
};
// networks that get the section: (16 | 32 | 128) -> 64 -> [64 ->] 16 and 64 -> 128 -> 128 -> 16
__host__ __device__ inline bool r32_shape_ok(const MlpDesc& d) {
	if (d.out_width != 16 || d.n_hidden < 1 || d.n_hidden > 2) return false;
	return (d.width == 64 && (d.in_width == 16 || d.in_width == 32 || d.in_width == 128)) || (d.width == 128 && d.in_width == 64 && d.n_hidden == 2);
}

// index into `params` of the weight that element (frag, lane, j) of the third section holds, -1: the element is zero
__host__ __device__ inline int32_t r32_prep_source(const MlpDesc& d, const uint32_t frag, const uint32_t lane, const uint32_t j) {
	const R32Frags f(d);
	const uint32_t r = lane & 31, h = lane >> 5;
	const MlpLayer L0 = d.layers[0], LO = d.layers[d.n_hidden];
	if (frag < f.fwd_hidden(1)) { // layer 0, natural k
		const uint32_t t = frag / f.ks0, s = frag - t * f.ks0;
		return (int32_t)(L0.w_off + (32 * t + r) * L0.cols + 16 * s + 8 * h + j);
	}
	if (frag < f.fwd_out()) {
		const uint32_t local = frag - f.fwd_hidden(1), per = f.wt * f.ksw;
		const uint32_t l = 1 + local / per, rem = local - (l - 1) * per, t = rem / f.ksw, ks = rem - t * f.ksw;
		const MlpLayer L = d.layers[l];
		return (int32_t)(L.w_off + (32 * t + r) * L.cols + r32_chain_k(ks, h, j));
	}
	if (frag < f.bwd_out()) { // output layer: row = position
		const uint32_t ks = frag - f.fwd_out();
		if (r >= 16) return -1;
		const uint32_t o = 2 * ((r & 3) + 4 * (r >> 3)) + ((r >> 2) & 1);
		return (int32_t)(LO.w_off + o * LO.cols + r32_chain_k(ks, h, j));
	}
	if (frag < f.bwd_hidden(1)) { // Wout^T: k = position (h, j) <-> output 2 j + h
		const uint32_t t = frag - f.bwd_out();
		return (int32_t)(LO.w_off + (2 * j + h) * LO.cols + 32 * t + r);
	}
	if (frag < f.bwd0()) {
		const uint32_t local = frag - f.bwd_hidden(1), per = f.wt * f.ksw;
		const uint32_t l = 1 + local / per, rem = local - (l - 1) * per, t = rem / f.ksw, ks = rem - t * f.ksw;
		const MlpLayer L = d.layers[l];
		return (int32_t)(L.w_off + r32_chain_k(ks, h, j) * L.cols + 32 * t + r);
	}
	const uint32_t local = frag - f.bwd0(), t = local / f.ksw, ks = local - t * f.ksw;
	const uint32_t col = 32 * t + r;
	return col < L0.cols ? (int32_t)(L0.w_off + r32_chain_k(ks, h, j) * L0.cols + col) : -1;
}

// weights (row-major half) -> fragment images, element gid of n_frags_total * 512:
// forward fragment (layer l, row tile t, k-step s), lane (r = lane & 15, q = lane >> 4), element j:  W_l[16 t + r][k], 0 beyond the matrix
// backward fragment (A = W_l^T; row tile t over the COLUMNS of W_l, k over its ROWS):              W_l[k_chain(s, q, j)][16 t + r]
// mlp_prep_source: the index into `params` of the weight element gid holds, -1 for the zeros.  (Host-callable: Network::image_inverse
// turns it round into "which image elements hold parameter i", so that an optimizer kernel can keep an image current.)
__host__ __device__ inline int32_t mlp_prep_source(const MlpDesc& d, const uint32_t gid) {
	uint32_t frag = gid >> 9;
	const uint32_t lane = (gid >> 3) & 63;
	const uint32_t j = gid & 7;
	if (frag >= d.n_frags_fwd + d.n_frags_bwd) return r32_prep_source(d, frag - (d.n_frags_fwd + d.n_frags_bwd), lane, j); // third section
	const uint32_t r = lane & 15, q = lane >> 4;
	const bool bwd = frag >= d.n_frags_fwd;
	if (bwd) frag -= d.n_frags_fwd;
	uint32_t l = 0;
	for (uint32_t i = 1; i < d.n_layers; ++i) {
		if (frag >= (bwd ? d.layers[i].bwd_off : d.layers[i].fwd_off)) l = i;
	}
	const MlpLayer L = d.layers[l];
	if (!bwd) {
		const uint32_t local = frag - L.fwd_off;
		const uint32_t t = local / L.ks_fwd, s = local - t * L.ks_fwd;
		const uint32_t row = 16 * t + r;
		const uint32_t k = L.natural_k ? frag_k_natural(s, q, j) : frag_k_chain(s, q, j);
		return (row < L.rows && k < L.cols) ? (int32_t)(L.w_off + row * L.cols + k) : -1;
	}
	const uint32_t local = frag - L.bwd_off;
	const uint32_t t = local / L.ks_bwd, s = local - t * L.ks_bwd;
	const uint32_t col = 16 * t + r;
	const uint32_t k = frag_k_chain(s, q, j);
	return (col < L.cols && k < L.rows) ? (int32_t)(L.w_off + k * L.cols + col) : -1;
}
__device__ inline void mlp_prep_element(const MlpDesc& d, const _Float16* __restrict__ params, _Float16* __restrict__ image, const uint32_t gid) {
	const int32_t src = mlp_prep_source(d, gid);
	image[gid] = src < 0 ? (_Float16)0.0f : params[src];
}

// side job of the encoding's forward kernel: build the fragment images of `params` (all n_frags_fwd + n_frags_bwd + n_frags_r32 fragments)
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
