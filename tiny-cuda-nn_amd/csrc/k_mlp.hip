// k_mlp.hip -- fully fused fp16 MLP for gfx950 (wave64, MFMA 16x16x32 f16, fp32 accumulate).
//
// Replaces (reference, /root/reference):
//   src/fully_fused_mlp.cu:500-557  kernel_mlp_fused            -> k_mlp_fwd
//   src/fully_fused_mlp.cu:151-259  kernel_mlp_fused_backward   -> k_mlp_bwd
//   src/fully_fused_mlp.cu:785,819,828 + include/tiny-cuda-nn/cutlass_matmul.h:438-479 (split-K weight gradients)
//                                                               -> k_wgrad + k_wgrad_reduce
//
// Design (NOT the reference's tiling -- see DESIGN.md "MLP"):
//   Activations are kept TRANSPOSED: H[feature][sample].  One MFMA computes a 16(feature) x 16(sample) tile
//       D[f][s] = sum_k W[f][k] * H_prev[k][s]        A = weights, B = previous activations.
//   With v_mfma_f32_16x16x32_f16 the result tile has the SAMPLE on the lane (lane & 15) and 4 consecutive features in
//   the lane's 4 accumulator registers (feature = 16*tile + 4*(lane >> 4) + reg).  That is exactly the B-operand
//   register layout of the NEXT layer's MFMA (k on registers, column on the lane) up to a fixed permutation of k --
//   so a wave chains all layers of the network in registers: no LDS round trip, no barrier, no cross-wave traffic.
//   The k permutation is absorbed into the A operand: weights are pre-permuted once per call into "fragment images"
//   (k_mlp_prep), 1 KiB per (16-row tile, 32-deep k-step), read with one 16-byte load per lane.
//
//   k order of a chained layer, k-step s, lane quarter q = lane >> 4, element j = 0..7:
//       k(s, q, j) = 32 s + 16 (j >> 2) + 4 q + (j & 3)
//   (elements 0..3 come from accumulator tile 2s, elements 4..7 from tile 2s+1.)
//   Layer 0 reads its B operand from memory ([n][in] half, 16 bytes per lane) in natural order k = 32 s + 8 q + j.
#include "mlp_device.h"
#include "mlp_side_jobs.h"
#include "adam_device.h"
#include "oneblob_device.h"

namespace tcnn_amd {
namespace {

// ------------------------------------------------------------------------------------------------------------------
// weights (row-major half) -> fragment images.  One thread per image element.
// forward fragment (layer l, row tile t, k-step s), lane (r = lane & 15, q = lane >> 4), element j:
//     W_l[16 t + r][k]           k = natural or chained order, 0 beyond the matrix
// backward fragment (A = W_l^T; row tile t over the COLUMNS of W_l, k over its ROWS):
//     W_l[k_chain(s, q, j)][16 t + r]
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mlp_prep(const MlpDesc d, const half_t* __restrict__ params, half_t* __restrict__ image, const uint32_t n_frags_total) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n_frags_total * 512) return;
	mlp_prep_element(d, params, image, gid); // mlp_side_jobs.h: the encoding's forward kernel can carry this along
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
struct FwdArgs {
	const half_t* x;      // [n][in_width], or level planes [in_width / F][n][F] when x_plane_f = F > 0; unused with x_f32
	half_t* out;          // optional [n][out_width]
	half_t* hidden;       // optional [n_hidden][n][width]
	const h8* image;      // forward fragments
	uint32_t n;
	uint32_t x_plane_f;
	// fused Identity encoding (identity.h:46-66): feature k = (half)(x[k] * scale + offset) for k < x_f32_dims, 1 for the padding
	MatView x_f32;        // data == nullptr: not used
	uint32_t x_f32_dims;
	float x_scale, x_offset;
	// fused OneBlob encoding (oneblob.h:47-67): x_f32 holds the coordinates, feature dim * n_bins + bin; 0 = off
	uint32_t oneblob_log2;
	// fused trim_and_cast (common_device.h:990-1002): float output of the first out_f32_dims outputs
	MatViewMut out_f32;   // data == nullptr: not used
	uint32_t out_f32_dims;
};

// 8 consecutive input features k0..k0+7 of one sample: the B operand of layer 0
__device__ inline h8 load_mlp_input(const FwdArgs& a, const uint32_t in_w, const uint32_t sample, const uint32_t k0) {
	if (a.x_f32.data) {
		h8 v;
		if (a.x_f32.stride_dim == 1 && (a.x_f32.stride_sample & 3u) == 0 && k0 + 8 <= a.x_f32_dims) { // two 16-byte loads
			const float4* p = (const float4*)(a.x_f32.data + (size_t)sample * a.x_f32.stride_sample + k0);
			const float4 lo = p[0], hi = p[1];
			v[0] = (half_t)(lo.x * a.x_scale + a.x_offset); v[1] = (half_t)(lo.y * a.x_scale + a.x_offset);
			v[2] = (half_t)(lo.z * a.x_scale + a.x_offset); v[3] = (half_t)(lo.w * a.x_scale + a.x_offset);
			v[4] = (half_t)(hi.x * a.x_scale + a.x_offset); v[5] = (half_t)(hi.y * a.x_scale + a.x_offset);
			v[6] = (half_t)(hi.z * a.x_scale + a.x_offset); v[7] = (half_t)(hi.w * a.x_scale + a.x_offset);
			return v;
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint32_t k = k0 + j;
			v[j] = k < a.x_f32_dims ? (half_t)(a.x_f32.data[(size_t)sample * a.x_f32.stride_sample + (size_t)k * a.x_f32.stride_dim] * a.x_scale + a.x_offset) : (half_t)1.0f;
		}
		return v;
	}
	if (a.x_plane_f == 2) {
		uint4 v;
		v.x = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 0) * a.n + sample) * 2);
		v.y = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 1) * a.n + sample) * 2);
		v.z = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 2) * a.n + sample) * 2);
		v.w = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 3) * a.n + sample) * 2);
		return __builtin_bit_cast(h8, v);
	}
	if (a.x_plane_f == 4) {
		const uint2 lo = *(const uint2*)(a.x + ((size_t)(k0 / 4) * a.n + sample) * 4);
		const uint2 hi = *(const uint2*)(a.x + ((size_t)(k0 / 4 + 1) * a.n + sample) * 4);
		uint4 v;
		v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
		return __builtin_bit_cast(h8, v);
	}
	if (a.x_plane_f == 8) return *(const h8*)(a.x + ((size_t)(k0 / 8) * a.n + sample) * 8);
	return *(const h8*)(a.x + (size_t)sample * in_w + k0);
}

// pack NB accumulator column blocks of T row tiles into chained B fragments (KS k-steps), applying the activation
template <int T, int KS, int NB, int ACT>
__device__ inline void activate_pack(const f4 (&acc)[T][NB], h8 (&hf)[KS][NB], uint32_t act) {
#pragma unroll
	for (int s = 0; s < KS; ++s) {
#pragma unroll
		for (int b = 0; b < NB; ++b) {
			h8 v;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				v[r] = act_fwd_t<ACT>(act, (half_t)acc[2 * s][b][r]);
				if (2 * s + 1 < T) v[4 + r] = act_fwd_t<ACT>(act, (half_t)acc[(2 * s + 1 < T) ? 2 * s + 1 : 0][b][r]);
				else v[4 + r] = (half_t)0.0f;
			}
			hf[s][b] = v;
		}
	}
}

// IMG_LDS: the forward fragment image is copied into LDS once per workgroup (persistent workgroups, one per CU) and read
// with ds_read_b128 -- for wide, deep networks (C4: 108 KB of fragments) whose image falls out of the 32 KB L1, where every
// MFMA otherwise waits on a 1 KB fetch from L2.  Used for inference with NB = 4 column blocks per wave (one fragment read
// feeds 4 MFMAs) and 8 waves per workgroup (2 per SIMD: one wave's activation VALU work under the other's MFMAs).
// OB: the input is the OneBlob encoding of a.x_f32, evaluated in the layer-0 loop (a.oneblob_log2; instantiated for widths 64 and 128).
#ifndef TCNN_MLP_FWD_PF
#define TCNN_MLP_FWD_PF 3
#endif
// The stored activations of the hidden layers and their gradients (k_mlp_fwd -> k_mlp_bwd -> the weight-gradient products; nobody else sees
// them) are TILED: [layer][16 samples][16 features] blocks of 512 contiguous bytes, a layer's blocks sample-tile major.  One block is exactly
// what a wave's 64 lanes x 8 bytes hold of one result tile, so every such load or store is four whole 128-byte lines; as rows [n][W] the
// same instruction touched 32 bytes in each of 16 rows (round 5: the forward kernel's stores cost 72 of its 143 us at 128 x 5, 2^18 samples).
// Returns the offset in halves of block (samples sample16 .. +15, features 16 t .. +15) of layer l.
__device__ __host__ inline size_t hidden_tile_off(const uint32_t n, const uint32_t W, const uint32_t l, const uint32_t sample16, const uint32_t t) {
	return ((size_t)l * n + sample16) * W + (size_t)t * 256;
}

template <int W, int NB, int ACT, bool IMG_LDS = false, int THREADS = 256, bool OB = false>
// (W = 128 with the fragments in the L2: the 120 + 64 registers the compiler took on its own left two waves per SIMD to hide every fragment's
// and input's latency; held to 168 -- no spills -- it is three: 158 -> 143 us for 128 x 5 at 2^18 samples, round 5)
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu((W >= 128 && !IMG_LDS) ? 3 : 1, 8))) k_mlp_fwd(const MlpDesc d, const FwdArgs a) {
	constexpr int T = W / 16;
	constexpr int KS = (T + 1) / 2;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t c = lane & 15, q = lane >> 4;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t n_iters = a.n / (16 * NB);
	const uint32_t in_w = d.in_width, out_w = d.out_width;

	extern __shared__ __attribute__((aligned(16))) char fwd_smem[];
	typedef __attribute__((address_space(3))) const h8 lds_h8;
	if constexpr (IMG_LDS) {
		h8* dst = (h8*)fwd_smem;
		const uint32_t n16 = d.n_frags_fwd * 64;
		for (uint32_t i = threadIdx.x; i < n16; i += THREADS) dst[i] = a.image[i];
		__syncthreads();
	}
	// fragment `index` of the forward image (1 KB = 64 lanes x 16 B)
	auto frag = [&](const uint32_t index) -> h8 {
		if constexpr (IMG_LDS) return *((lds_h8*)fwd_smem + index * 64 + lane);
		else return a.image[(size_t)index * 64 + lane];
	};

	for (uint32_t it = wave; it < n_iters; it += n_waves) {
		const uint32_t s0 = it * 16 * NB;

		// ---- layer 0: B operand straight from memory, natural k order
		f4 acc[T][NB];
#pragma unroll
		for (int t = 0; t < T; ++t)
#pragma unroll
			for (int b = 0; b < NB; ++b) acc[t][b] = f4{0, 0, 0, 0};
		{
			const uint32_t ks0 = d.layers[0].ks_fwd;
			const uint32_t img = d.layers[0].fwd_off;
			// fused OneBlob (n_bins >= 32: the four quarters of a k-step are four 8-bin chunks of ONE dimension's row).  Per sample
			// and dimension only the five bins around x differ from +0 (oneblob_device.h); the four lanes (c, 0..3) of a sample share
			// their evaluation (lane q: bins q and min(q + 4, 4)), exchange them with lane shuffles at the first k-step of the
			// dimension and keep them for its other k-steps.
			uint32_t ob_first[OB ? NB : 1];
			float ob_win[OB ? NB : 1][5], ob_x[OB ? NB : 1];
			for (uint32_t s = 0; s < ks0; ++s) {
				h8 bf[NB];
				const uint32_t k0 = 32 * s + 8 * q;
				if constexpr (OB) {
					const uint32_t n_bins = 1u << a.oneblob_log2;
					const uint32_t dim = (32 * s) >> a.oneblob_log2; // wave-uniform
					if (dim >= a.x_f32_dims) { // padding columns: ones
#pragma unroll
						for (int b = 0; b < NB; ++b) bf[b] = h8{1, 1, 1, 1, 1, 1, 1, 1};
					} else {
						if (((32 * s) & (n_bins - 1)) == 0) { // first k-step of this dimension
#pragma unroll
							for (int b = 0; b < NB; ++b) {
								const uint32_t sample = s0 + 16 * b + c;
								const float xv = a.x_f32.data[(size_t)sample * a.x_f32.stride_sample + (size_t)dim * a.x_f32.stride_dim];
								ob_x[b] = xv;
								ob_first[b] = oneblob_window_first(xv, a.oneblob_log2);
								const float m0 = oneblob_bin(xv, (ob_first[b] + q) & (n_bins - 1), a.oneblob_log2);
								const float m1 = oneblob_bin(xv, (ob_first[b] + 4) & (n_bins - 1), a.oneblob_log2);
#pragma unroll
								for (int o = 0; o < 4; ++o) ob_win[b][o] = __shfl(m0, (int)(c + 16 * o), 64);
								ob_win[b][4] = m1;
							}
						}
						const uint32_t b0 = k0 & (n_bins - 1);
#pragma unroll
						for (int b = 0; b < NB; ++b) {
							h8 v;
							if (oneblob_in_unit_interval(ob_x[b])) {
#pragma unroll
								for (int k = 0; k < 8; ++k) v[k] = (half_t)0.0f;
#pragma unroll
								for (int o = 0; o < 5; ++o) {
									const uint32_t dd = ((ob_first[b] + o) & (n_bins - 1)) - b0;
#pragma unroll
									for (int k = 0; k < 8; ++k) v[k] = dd == (uint32_t)k ? (half_t)ob_win[b][o] : v[k];
								}
							} else { // general form: the chunk's 9 edges
								float e[9];
#pragma unroll
								for (int k = 0; k < 9; ++k) e[k] = oneblob_edge(ob_x[b], b0 + k, a.oneblob_log2);
								if (b0 + 8 == n_bins) e[8] += 1;
#pragma unroll
								for (int k = 0; k < 8; ++k) v[k] = (half_t)(e[k + 1] - e[k]);
							}
							bf[b] = v;
						}
					}
				} else {
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					if (k0 < in_w) bf[b] = load_mlp_input(a, in_w, s0 + 16 * b + c, k0);
					else bf[b] = h8{0, 0, 0, 0, 0, 0, 0, 0};
				}
				}
#pragma unroll
				for (int t = 0; t < T; ++t) {
					const h8 af = frag(img + t * ks0 + s);
#pragma unroll
					for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af, bf[b], acc[t][b]);
				}
			}
		}
		h8 hf[KS][NB];
		activate_pack<T, KS, NB, ACT>(acc, hf, d.activation);

		auto store_hidden = [&](uint32_t l) {
			if constexpr (IMG_LDS || OB) return; // (inference only, launched with hidden == nullptr: the dead stores' addresses cost the LDS form's 254 registers eleven spills)
			if (!a.hidden) return;
#pragma unroll
			for (int t = 0; t < T; ++t)
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					h4 v;
#pragma unroll
					for (int r = 0; r < 4; ++r) v[r] = hf[t / 2][b][(t & 1) * 4 + r];
					*(h4*)(a.hidden + hidden_tile_off(a.n, W, l, s0 + 16 * b, t) + 16 * c + 4 * q) = v; // (the wave's 64 x 8 bytes: one tile, 512 contiguous bytes)
				}
		};
		store_hidden(0);

		// ---- hidden layers 1 .. n_hidden-1: chained in registers
		for (uint32_t l = 1; l < d.n_hidden; ++l) {
			const uint32_t img = d.layers[l].fwd_off;
			// Software pipeline over the KS * T weight fragments of the layer: fragment i + 1 is requested before the NB MFMAs of
			// fragment i are issued, in a second register set.  Written out in this order the compiler waits with lgkmcnt(1); as a
			// plain "load, use" loop it reused ONE register quad and exposed the whole LDS latency before every group of MFMAs
			// (measured on C4: the MFMA pipe 36 % busy).
			// (PF = 1 fragment ahead covers 4 MFMAs = 64 clocks of a ~130-clock LDS read; TCNN_MLP_FWD_PF fragments ahead cover it whole)
			constexpr int PF = IMG_LDS ? TCNN_MLP_FWD_PF : 1;
			auto frag_of = [&](const int i) -> h8 { return frag(img + (i % T) * KS + i / T); };
			h8 af[PF + 1];
#pragma unroll
			for (int i = 0; i < PF; ++i) af[i] = frag_of(i);
#pragma unroll
			for (int i = 0; i < KS * T; ++i) {
				const int s = i / T, t = i % T;
				if (i + PF < KS * T) af[(i + PF) % (PF + 1)] = frag_of(i + PF);
#pragma unroll
				for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af[i % (PF + 1)], hf[s][b], s == 0 ? f4{0, 0, 0, 0} : acc[t][b]); // first k-step: C = inline 0, no register clearing
			}
			activate_pack<T, KS, NB, ACT>(acc, hf, d.activation);
			store_hidden(l);
		}

		// ---- output layer: out_width / 16 row tiles
		{
			const MlpLayer Lo = d.layers[d.n_layers - 1];
			const uint32_t img = Lo.fwd_off;
			for (uint32_t to = 0; to < out_w / 16; ++to) {
				f4 o[NB];
#pragma unroll
				for (int s = 0; s < KS; ++s) {
					const h8 af = frag(img + to * KS + s);
#pragma unroll
					for (int b = 0; b < NB; ++b) o[b] = mfma(af, hf[s][b], s == 0 ? f4{0, 0, 0, 0} : o[b]);
				}
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					h4 v;
#pragma unroll
					for (int r = 0; r < 4; ++r) v[r] = activation_fwd(d.output_activation, (half_t)o[b][r]);
					const uint32_t sample = s0 + 16 * b + c, j0 = 16 * to + 4 * q;
					if (a.out) *(h4*)(a.out + (size_t)sample * out_w + j0) = v;
					if (a.out_f32.data) {
						if (a.out_f32.stride_dim == 1 && (a.out_f32.stride_sample & 3u) == 0 && j0 + 4 <= a.out_f32_dims) {
							*(float4*)(a.out_f32.data + (size_t)sample * a.out_f32.stride_sample + j0) = float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
						} else {
#pragma unroll
							for (int r = 0; r < 4; ++r)
								if (j0 + r < a.out_f32_dims) a.out_f32.data[(size_t)sample * a.out_f32.stride_sample + (size_t)(j0 + r) * a.out_f32.stride_dim] = (float)v[r];
						}
					}
				}
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------------------------
// backward (data gradients), reference-shaped: reads the stored forward activations, writes dL/dhidden for the
// weight-gradient kernel.  (The fused trainer path in k_train.hip never materialises either.)
// ------------------------------------------------------------------------------------------------------------------
struct BwdArgs {
	const half_t* dL_dout; // [n][out_width]
	const half_t* out;     // [n][out_width]   (only read when output_activation != None)
	const half_t* hidden;  // [n_hidden][n][width]
	half_t* dhidden;       // [n_hidden][n][width]
	half_t* dL_dx;         // optional [n][in_width], or level planes [in_width / F][n][F] when dx_plane_f = F > 0
	const h8* image;       // backward fragments
	uint32_t n;
	uint32_t dx_plane_f;
};

template <int W, int NB, int ACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W == 32 ? 3 : 1, 8))) k_mlp_bwd(const MlpDesc d, const BwdArgs a) { // (three waves per SIMD, as k_mlp_fwd has them at W = 128, cost 24 spilled registers here: 227 -> 255 us)
	constexpr int T = W / 16;
	constexpr int KS = (T + 1) / 2;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t c = lane & 15, q = lane >> 4;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t n_iters = a.n / (16 * NB);
	const uint32_t in_w = d.in_width, out_w = d.out_width;
	const uint32_t nh = d.n_hidden;

	for (uint32_t it = wave; it < n_iters; it += n_waves) {
		const uint32_t s0 = it * 16 * NB;
		f4 acc[T][NB];
#pragma unroll
		for (int t = 0; t < T; ++t)
#pragma unroll
			for (int b = 0; b < NB; ++b) acc[t][b] = f4{0, 0, 0, 0};

		// ---- dH_last (pre-activation-derivative) = Wout^T * dY ; dY loaded in chained k order (two 8-byte loads)
		{
			const MlpLayer Lo = d.layers[d.n_layers - 1];
			const uint32_t kso = Lo.ks_bwd;
			const h8* img = a.image + (size_t)Lo.bwd_off * 64 + lane;
			for (uint32_t s = 0; s < kso; ++s) {
				h8 bf[NB];
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					const size_t row = (size_t)(s0 + 16 * b + c) * out_w;
					h4 lo = h4{0, 0, 0, 0}, hi = h4{0, 0, 0, 0};
					const uint32_t k_lo = 32 * s + 4 * q, k_hi = 32 * s + 16 + 4 * q;
					if (k_lo < out_w) {
						lo = *(const h4*)(a.dL_dout + row + k_lo);
						if (d.output_activation != (uint32_t)Activation::None) {
							const h4 ov = *(const h4*)(a.out + row + k_lo);
#pragma unroll
							for (int r = 0; r < 4; ++r) lo[r] = activation_bwd(d.output_activation, lo[r], ov[r]);
						}
					}
					if (k_hi < out_w) {
						hi = *(const h4*)(a.dL_dout + row + k_hi);
						if (d.output_activation != (uint32_t)Activation::None) {
							const h4 ov = *(const h4*)(a.out + row + k_hi);
#pragma unroll
							for (int r = 0; r < 4; ++r) hi[r] = activation_bwd(d.output_activation, hi[r], ov[r]);
						}
					}
					bf[b] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
				}
#pragma unroll
				for (int t = 0; t < T; ++t) {
					const h8 af = img[(size_t)(t * kso + s) * 64];
#pragma unroll
					for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af, bf[b], acc[t][b]);
				}
			}
		}

		h8 hf[KS][NB];
		// The stored forward outputs of layer l are requested a whole layer ahead -- before the products that lead to dH_l -- and the weight
		// fragments one ahead of the matrix instructions that use them (round 5: written as "load, use" every one of a layer's 16 + 32 loads
		// exposed its latency to a wave that has only one partner on its SIMD: 270 us for 128 x 5 at 2^18 samples).  Same arithmetic, same order.
		// (W = 256: the 32 registers this takes push the kernel past 256 -- one wave per SIMD, 497 against ~300 us -- so there the values are read where they are used)
		constexpr bool AHEAD = W == 64 || W == 128; // (W = 32 with its four column blocks: three waves per SIMD without, two with)
		h4 hv[AHEAD ? T : 1][AHEAD ? NB : 1];
		auto request_hidden = [&](const int l) {
			if constexpr (!AHEAD) return;
#pragma unroll
			for (int t = 0; t < T; ++t)
#pragma unroll
				for (int b = 0; b < NB; ++b) hv[AHEAD ? t : 0][AHEAD ? b : 0] = *(const h4*)(a.hidden + hidden_tile_off(a.n, W, l, s0 + 16 * b, t) + 16 * c + 4 * q);
		};
		request_hidden((int)nh - 1);
		for (int l = (int)nh - 1; l >= 0; --l) {
			// acc = W_{l+1}^T dH_{l+1}; multiply by act'(H_l) (from the stored forward output), store, pack
#pragma unroll
			for (int t = 0; t < T; ++t)
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					const size_t off = hidden_tile_off(a.n, W, l, s0 + 16 * b, t) + 16 * c + 4 * q;
					h4 g, hvv;
					if constexpr (AHEAD) hvv = hv[t][b];
					else hvv = *(const h4*)(a.hidden + off);
#pragma unroll
					for (int r = 0; r < 4; ++r) g[r] = act_bwd_t<ACT>(d.activation, (half_t)acc[t][b][r], hvv[r]);
					*(h4*)(a.dhidden + off) = g;
#pragma unroll
					for (int r = 0; r < 4; ++r) hf[t / 2][b][(t & 1) * 4 + r] = g[r];
				}
			if (l > 0) request_hidden(l - 1);
			if constexpr (T & 1) {
#pragma unroll
				for (int b = 0; b < NB; ++b)
#pragma unroll
					for (int r = 0; r < 4; ++r) hf[KS - 1][b][4 + r] = (half_t)0.0f;
			}
			if (l > 0) {
				const h8* img = a.image + (size_t)d.layers[l].bwd_off * 64 + lane;
#pragma unroll
				for (int t = 0; t < T; ++t)
#pragma unroll
					for (int b = 0; b < NB; ++b) acc[t][b] = f4{0, 0, 0, 0};
				auto frag_of = [&](const int i) -> h8 { return img[(size_t)((i % T) * KS + i / T) * 64]; };
				if constexpr (AHEAD) {
					h8 af[2];
					af[0] = frag_of(0);
#pragma unroll
					for (int i = 0; i < KS * T; ++i) {
						const int s = i / T, t = i % T;
						if (i + 1 < KS * T) af[(i + 1) & 1] = frag_of(i + 1);
#pragma unroll
						for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af[i & 1], hf[s][b], acc[t][b]);
					}
				} else {
#pragma unroll
					for (int i = 0; i < KS * T; ++i) {
						const int s = i / T, t = i % T;
						const h8 af = frag_of(i);
#pragma unroll
						for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af, hf[s][b], acc[t][b]);
					}
				}
			}
		}

		// ---- dX = W0^T dH_0, in_width / 16 row tiles
		if (a.dL_dx) {
			const h8* img = a.image + (size_t)d.layers[0].bwd_off * 64 + lane;
			for (uint32_t ti = 0; ti < in_w / 16; ++ti) {
				f4 o[NB];
#pragma unroll
				for (int b = 0; b < NB; ++b) o[b] = f4{0, 0, 0, 0};
#pragma unroll
				for (int s = 0; s < KS; ++s) {
					const h8 af = img[(size_t)(ti * KS + s) * 64];
#pragma unroll
					for (int b = 0; b < NB; ++b) o[b] = mfma(af, hf[s][b], o[b]);
				}
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					const h4 v = h4{(half_t)o[b][0], (half_t)o[b][1], (half_t)o[b][2], (half_t)o[b][3]};
					store_dx(a.dL_dx, a.dx_plane_f, a.n, in_w, s0 + 16 * b + c, 16 * ti + 4 * q, v);
				}
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradients: dW[R x C] = sum_i dO[i][R]^T In[i][C].
// Chunks of 64 samples are staged in LDS as [sample][feature] rows; MFMA operands need [feature on lane][8 samples in
// registers], which is exactly what gfx950's transposing LDS read (ds_read_b64_tr_b16) delivers from that image.
// Each workgroup accumulates a private partial in registers over its chunks and writes one fp32 slab; k_wgrad_reduce
// sums the slabs in a fixed order (bitwise reproducible, no atomics).
// ------------------------------------------------------------------------------------------------------------------
// Piece p (8 halves) of a chunk of WG_CHUNK samples x 8 per_row features: where it lies in memory (offset in halves from the operand's
// pointer) and in the LDS image [sample][feature].  Rows [n][ld], or the tiled form above (ld: the tiled matrix's full width; a panel's first
// column is absorbed by the pointer: + (column / 16) * 256).  Consecutive p are consecutive in memory either way.
struct WgPiece { size_t src; uint32_t row, col; };
__device__ inline WgPiece wg_piece(const uint32_t p, const uint32_t per_row, const size_t base_sample, const uint32_t ld, const bool tiled) {
	WgPiece r;
	if (!tiled) {
		r.row = p / per_row;
		r.col = (p - r.row * per_row) * 8;
		r.src = (base_sample + r.row) * ld + r.col;
	} else {
		const uint32_t half = p & 1u, c = (p >> 1) & 15u, rest = p >> 5, tiles = per_row / 2, st = rest / tiles, t = rest - st * tiles;
		r.row = st * 16 + c;
		r.col = t * 16 + half * 8;
		r.src = (base_sample + st * 16) * ld + (size_t)t * 256 + c * 16 + half * 8;
	}
	return r;
}

constexpr int WG_CHUNK = 64;       // samples per staged chunk (2 k-steps)
constexpr int WG_MAX_TILES = 16;   // accumulator tiles per wave: R*C <= 128*128 with 4 waves
constexpr int WG_PAD = 16;         // halfs of row padding in LDS (keeps rows 16-byte aligned, breaks the power-of-2 stride)

__global__ void __launch_bounds__(256) k_wgrad(
	const uint32_t n, const half_t* __restrict__ dO, const uint32_t ldo, const uint32_t R,
	const half_t* __restrict__ In, const uint32_t ldi, const uint32_t C, float* __restrict__ slabs, const uint32_t tiled
) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t rp = R + WG_PAD, cp = C + WG_PAD; // padded row lengths (halfs)
	half_t* P = (half_t*)smem;                        // [WG_CHUNK][rp]
	half_t* Q = P + (size_t)WG_CHUNK * rp;            // [WG_CHUNK][cp]

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63, w = tid >> 6;
	const uint32_t grp = lane >> 4, li = lane & 15; // 16-lane group / lane within the group
	const uint32_t TR = R / 16, TC = C / 16;
	const uint32_t n_tiles = TR * TC;
	const uint32_t per = (n_tiles + 3) / 4;
	const uint32_t tile0 = w * per;

	f4 acc[WG_MAX_TILES];
#pragma unroll
	for (int i = 0; i < WG_MAX_TILES; ++i) acc[i] = f4{0, 0, 0, 0};

	const uint32_t n_chunks = n / WG_CHUNK;
	for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
		const size_t base = (size_t)ch * WG_CHUNK;
		// stage: 16-byte pieces, rows are contiguous in global memory
		const uint32_t ppr = R / 8, pqr = C / 8;
		for (uint32_t p = tid; p < WG_CHUNK * ppr; p += 256) {
			const WgPiece w = wg_piece(p, ppr, base, ldo, (tiled & 1u) != 0);
			*(h8*)(P + (size_t)w.row * rp + w.col) = *(const h8*)(dO + w.src);
		}
		for (uint32_t p = tid; p < WG_CHUNK * pqr; p += 256) {
			const WgPiece w = wg_piece(p, pqr, base, ldi, (tiled & 2u) != 0);
			*(h8*)(Q + (size_t)w.row * cp + w.col) = *(const h8*)(In + w.src);
		}
		__syncthreads();

#pragma unroll
		for (int i = 0; i < WG_MAX_TILES; ++i) {
			const uint32_t tile = tile0 + i;
			if (i < (int)per && tile < n_tiles) { // wave-uniform
				const uint32_t tr = tile / TC, tc = tile - tr * TC;
#pragma unroll
				for (int ks = 0; ks < WG_CHUNK / 32; ++ks) {
					// lane (li = 4*qq + p) supplies the address of sample row (.. + qq), feature columns 4p..4p+3;
					// lane li receives feature column li of the 4 sample rows.
					const uint32_t row_lo = 32 * ks + 8 * grp + (li >> 2);
					const uint32_t colo = 4 * (li & 3);
					const h4 a_lo = lds_read_tr(P + (size_t)row_lo * rp + 16 * tr + colo);
					const h4 a_hi = lds_read_tr(P + (size_t)(row_lo + 4) * rp + 16 * tr + colo);
					const h4 b_lo = lds_read_tr(Q + (size_t)row_lo * cp + 16 * tc + colo);
					const h4 b_hi = lds_read_tr(Q + (size_t)(row_lo + 4) * cp + 16 * tc + colo);
					const h8 af = h8{a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
					const h8 bf = h8{b_lo[0], b_lo[1], b_lo[2], b_lo[3], b_hi[0], b_hi[1], b_hi[2], b_hi[3]};
					acc[i] = mfma(af, bf, acc[i]);
				}
			}
		}
		__syncthreads();
	}

	float* slab = slabs + (size_t)blockIdx.x * R * C;
#pragma unroll
	for (int i = 0; i < WG_MAX_TILES; ++i) {
		const uint32_t tile = tile0 + i;
		if (i < (int)per && tile < n_tiles) {
			const uint32_t tr = tile / TC, tc = tile - tr * TC;
#pragma unroll
			for (int r = 0; r < 4; ++r) slab[(size_t)(16 * tr + 4 * grp + r) * C + 16 * tc + li] = acc[i][r];
		}
	}
}

// The same product for the shapes whose tile rows divide over the four waves (R = 64 or 128 output features, C <= 128 input features: every
// hidden layer of a 64- or 128-wide network).  Round 5: the kernel above re-reads both operands of every tile from LDS (four transposing reads
// per matrix instruction) and loads a chunk, waits, multiplies, waits: 91 us for 128 x 128 over 2^18 samples, a quarter of what the 134 MB it
// reads allow.  Here wave w owns TRW tile rows x all TC tile columns: a tile row's operand is read once per k-step and a tile column's once for
// all of the wave's rows (40 reads for 32 matrix instructions at 128 x 128 instead of 128), and the next chunk's global loads are in flight
// while this chunk is multiplied (through registers: one LDS image, two barriers per chunk).  Every tile still sums its chunks and k-steps
// in the same order: bit-identical slabs.
// (blockIdx.y: which product of the launch -- the hidden layers of a network are as many products of one shape, independent of each other)
constexpr uint32_t WG_MAX_JOBS = 8;
struct WgradJobs {
	const half_t* dO[WG_MAX_JOBS];
	const half_t* In[WG_MAX_JOBS];
	float* slabs[WG_MAX_JOBS];
	half_t* grad[WG_MAX_JOBS];
	uint32_t ldo[WG_MAX_JOBS], ldi[WG_MAX_JOBS], ldg[WG_MAX_JOBS];
	uint32_t tiled[WG_MAX_JOBS]; // bit 0: dO, bit 1: In in the tiled form (hidden_tile_off)
};

template <int TRW, int TC>
__global__ void __launch_bounds__(256) k_wgrad_rows(const uint32_t n, const WgradJobs jobs) {
	const half_t* __restrict__ dO = jobs.dO[blockIdx.y];
	const half_t* __restrict__ In = jobs.In[blockIdx.y];
	float* __restrict__ slabs = jobs.slabs[blockIdx.y];
	const uint32_t ldo = jobs.ldo[blockIdx.y], ldi = jobs.ldi[blockIdx.y];
	const bool p_tiled = (jobs.tiled[blockIdx.y] & 1u) != 0, q_tiled = (jobs.tiled[blockIdx.y] & 2u) != 0;
	constexpr uint32_t R = 4 * TRW * 16, C = TC * 16;
	constexpr uint32_t rp = R + WG_PAD, cp = C + WG_PAD;
	constexpr uint32_t ppr = R / 8, pqr = C / 8;                                      // 16-byte pieces per row
	constexpr uint32_t NP = WG_CHUNK * ppr / 256, NQ = WG_CHUNK * pqr / 256;           // pieces per thread
	static_assert(WG_CHUNK * ppr % 256 == 0 && WG_CHUNK * pqr % 256 == 0, "whole pieces per thread");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	half_t* P = (half_t*)smem;             // [WG_CHUNK][rp]
	half_t* Q = P + (size_t)WG_CHUNK * rp; // [WG_CHUNK][cp]

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63, w = tid >> 6;
	const uint32_t grp = lane >> 4, li = lane & 15;
	f4 acc[TRW][TC];
#pragma unroll
	for (int i = 0; i < TRW; ++i)
#pragma unroll
		for (int j = 0; j < TC; ++j) acc[i][j] = f4{0, 0, 0, 0};

	const uint32_t n_chunks = n / WG_CHUNK;
	h8 pv[NP], qv[NQ];
	auto fetch = [&](const uint32_t ch) { // (a chunk past the end: the last one again, never stored)
		const size_t base = (size_t)min(ch, n_chunks - 1) * WG_CHUNK;
#pragma unroll
		for (uint32_t k = 0; k < NP; ++k) pv[k] = *(const h8*)(dO + wg_piece(tid + k * 256, ppr, base, ldo, p_tiled).src);
#pragma unroll
		for (uint32_t k = 0; k < NQ; ++k) qv[k] = *(const h8*)(In + wg_piece(tid + k * 256, pqr, base, ldi, q_tiled).src);
	};
	if (blockIdx.x < n_chunks) fetch(blockIdx.x);
	for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
#pragma unroll
		for (uint32_t k = 0; k < NP; ++k) {
			const WgPiece w = wg_piece(tid + k * 256, ppr, 0, ldo, p_tiled);
			*(h8*)(P + (size_t)w.row * rp + w.col) = pv[k];
		}
#pragma unroll
		for (uint32_t k = 0; k < NQ; ++k) {
			const WgPiece w = wg_piece(tid + k * 256, pqr, 0, ldi, q_tiled);
			*(h8*)(Q + (size_t)w.row * cp + w.col) = qv[k];
		}
		__syncthreads();
		fetch(ch + gridDim.x); // in flight while this chunk is multiplied
#pragma unroll
		for (int ks = 0; ks < WG_CHUNK / 32; ++ks) {
			const uint32_t row_lo = 32 * ks + 8 * grp + (li >> 2), colo = 4 * (li & 3); // (the lane map of k_wgrad)
			h8 af[TRW];
#pragma unroll
			for (int i = 0; i < TRW; ++i) {
				const uint32_t tr = w * TRW + i;
				const h4 lo = lds_read_tr(P + (size_t)row_lo * rp + 16 * tr + colo), hi = lds_read_tr(P + (size_t)(row_lo + 4) * rp + 16 * tr + colo);
				af[i] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
			}
#pragma unroll
			for (int tc = 0; tc < TC; ++tc) {
				const h4 lo = lds_read_tr(Q + (size_t)row_lo * cp + 16 * tc + colo), hi = lds_read_tr(Q + (size_t)(row_lo + 4) * cp + 16 * tc + colo);
				const h8 bf = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
				for (int i = 0; i < TRW; ++i) acc[i][tc] = mfma(af[i], bf, acc[i][tc]);
			}
		}
		__syncthreads();
	}

	float* slab = slabs + (size_t)blockIdx.x * R * C;
#pragma unroll
	for (int i = 0; i < TRW; ++i)
#pragma unroll
		for (int tc = 0; tc < TC; ++tc)
#pragma unroll
			for (int r = 0; r < 4; ++r) slab[(size_t)(16 * (w * TRW + i) + 4 * grp + r) * C + 16 * tc + li] = acc[i][tc][r];
}

// ... and for the output layer's product (R = 16 padded outputs: ONE tile row): wave w owns TCW tile columns.  Same lane maps, same order
// of a tile's sums, same prefetch; 54 -> 17 us for 16 x 128 over 2^18 samples.
template <int TCW>
__global__ void __launch_bounds__(256) k_wgrad_cols(const uint32_t n, const WgradJobs jobs) {
	const half_t* __restrict__ dO = jobs.dO[blockIdx.y];
	const half_t* __restrict__ In = jobs.In[blockIdx.y];
	float* __restrict__ slabs = jobs.slabs[blockIdx.y];
	const uint32_t ldo = jobs.ldo[blockIdx.y], ldi = jobs.ldi[blockIdx.y];
	const bool p_tiled = (jobs.tiled[blockIdx.y] & 1u) != 0, q_tiled = (jobs.tiled[blockIdx.y] & 2u) != 0;
	constexpr uint32_t R = 16, C = 4 * TCW * 16;
	constexpr uint32_t rp = R + WG_PAD, cp = C + WG_PAD;
	constexpr uint32_t ppr = R / 8, pqr = C / 8;
	constexpr uint32_t NQ = WG_CHUNK * pqr / 256;
	static_assert(WG_CHUNK * ppr <= 256 && WG_CHUNK * pqr % 256 == 0, "one piece of dO per thread at most, whole pieces of In per thread");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	half_t* P = (half_t*)smem;
	half_t* Q = P + (size_t)WG_CHUNK * rp;

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63, w = tid >> 6;
	const uint32_t grp = lane >> 4, li = lane & 15;
	f4 acc[TCW];
#pragma unroll
	for (int j = 0; j < TCW; ++j) acc[j] = f4{0, 0, 0, 0};

	const uint32_t n_chunks = n / WG_CHUNK;
	const bool has_p = tid < WG_CHUNK * ppr;
	const WgPiece pw = wg_piece(has_p ? tid : 0u, ppr, 0, ldo, p_tiled);
	h8 pv = h8{0, 0, 0, 0, 0, 0, 0, 0}, qv[NQ];
	auto fetch = [&](const uint32_t ch) {
		const size_t base = (size_t)min(ch, n_chunks - 1) * WG_CHUNK;
		if (has_p) pv = *(const h8*)(dO + base * ldo + pw.src); // (rows and tiles alike: a chunk starts base * ld halves in)
#pragma unroll
		for (uint32_t k = 0; k < NQ; ++k) qv[k] = *(const h8*)(In + wg_piece(tid + k * 256, pqr, base, ldi, q_tiled).src);
	};
	if (blockIdx.x < n_chunks) fetch(blockIdx.x);
	for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
		if (has_p) *(h8*)(P + (size_t)pw.row * rp + pw.col) = pv;
#pragma unroll
		for (uint32_t k = 0; k < NQ; ++k) {
			const WgPiece w = wg_piece(tid + k * 256, pqr, 0, ldi, q_tiled);
			*(h8*)(Q + (size_t)w.row * cp + w.col) = qv[k];
		}
		__syncthreads();
		fetch(ch + gridDim.x);
#pragma unroll
		for (int ks = 0; ks < WG_CHUNK / 32; ++ks) {
			const uint32_t row_lo = 32 * ks + 8 * grp + (li >> 2), colo = 4 * (li & 3);
			const h4 alo = lds_read_tr(P + (size_t)row_lo * rp + colo), ahi = lds_read_tr(P + (size_t)(row_lo + 4) * rp + colo);
			const h8 af = h8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
			for (int j = 0; j < TCW; ++j) {
				const uint32_t tc = w * TCW + j;
				const h4 lo = lds_read_tr(Q + (size_t)row_lo * cp + 16 * tc + colo), hi = lds_read_tr(Q + (size_t)(row_lo + 4) * cp + 16 * tc + colo);
				acc[j] = mfma(af, h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, acc[j]);
			}
		}
		__syncthreads();
	}
	float* slab = slabs + (size_t)blockIdx.x * R * C;
#pragma unroll
	for (int j = 0; j < TCW; ++j)
#pragma unroll
		for (int r = 0; r < 4; ++r) slab[(size_t)(4 * grp + r) * C + 16 * (w * TCW + j) + li] = acc[j][r];
}

// Sum of the per-workgroup slabs.  64 elements x 16 slab groups per workgroup: group g adds slabs g, g + 16, ... with four
// loads in flight, the 16 group sums are combined through LDS in a fixed order (bitwise reproducible, no atomics).
constexpr int WR_ELEMS = SLAB_REDUCE_ELEMS, WR_GROUPS = SLAB_REDUCE_GROUPS;
__global__ void __launch_bounds__(WR_ELEMS * WR_GROUPS) k_wgrad_reduce(const uint32_t n_elems, const uint32_t cols, const uint32_t ldg, const uint32_t n_slabs, const float* __restrict__ slabs, half_t* __restrict__ grad, const int accumulate) {
	__shared__ float part[WR_GROUPS * WR_ELEMS];
	mlp_reduce_block(part, blockIdx.x, threadIdx.x, n_elems, cols, ldg, n_slabs, slabs, grad, accumulate); // mlp_side_jobs.h: the grid scatter can carry this along
}

// ... for the products of one launch of k_wgrad_rows / k_wgrad_cols (blockIdx.y: which)
__global__ void __launch_bounds__(WR_ELEMS * WR_GROUPS) k_wgrad_reduce_jobs(const uint32_t n_elems, const uint32_t cols, const uint32_t n_slabs, const WgradJobs jobs, const int accumulate) {
	__shared__ float part[WR_GROUPS * WR_ELEMS];
	mlp_reduce_block(part, blockIdx.x, threadIdx.x, n_elems, cols, jobs.ldg[blockIdx.y], n_slabs, jobs.slabs[blockIdx.y], jobs.grad[blockIdx.y], accumulate);
}

// The same reduction with the optimizer's update behind it (AdamInFlush, tcnn_common.h): a matrix weight's gradient is final the moment
// its slabs are summed, so adam.h:48-119 runs on it at once -- the same adam_one as k_adam, on the same half-rounded gradient: the same
// bits.  For networks without encoding parameters (BASELINE config 2) this IS the optimizer step: one ~4 us launch less.
template <typename STEP_T>
__global__ void __launch_bounds__(WR_ELEMS * WR_GROUPS) k_wgrad_reduce_adam(const uint32_t n_elems, const uint32_t n_slabs, const float* __restrict__ slabs, half_t* __restrict__ grad,
                                                                            const AdamInFlush adam) {
	__shared__ float part[WR_GROUPS * WR_ELEMS];
	mlp_reduce_block(part, blockIdx.x, threadIdx.x, n_elems, n_elems, n_elems, n_slabs, slabs, grad, 0);
	const uint32_t e = threadIdx.x & (WR_ELEMS - 1), grp = threadIdx.x / WR_ELEMS;
	const uint32_t i = blockIdx.x * WR_ELEMS + e;
	if (grp != 0 || i >= n_elems) return;
	const half_t g = grad[i]; // written by this very thread a moment ago
	float w_fp = adam.w_fp[i], m1 = adam.m1[i], m2 = adam.m2[i];
	STEP_T* steps = (STEP_T*)adam.steps;
	uint32_t step = steps[i];
	half_t w_h;
	bool updated;
	const float* __restrict__ table = adam.debias_table;
	adam_one(adam.args, [&](const uint32_t t) { return table[t]; }, table[adam.args.common_step], /*is_matrix=*/true, g, w_fp, w_h, m1, m2, step, updated);
	if (updated) {
		adam.w_fp[i] = w_fp;
		((half_t*)adam.w_half)[i] = w_h;
		adam.m1[i] = m1;
		adam.m2[i] = m2;
		steps[i] = (STEP_T)step;
		if (adam.image) { // keep the fragment images current (AdamInFlush::image): what k_mlp_prep would gather at the start of the next step
			const uint4 where = *(const uint4*)(adam.image_inv + (size_t)IMAGE_INV_WIDTH * i);
			half_t* image = (half_t*)adam.image;
			if (where.x != 0xffffffffu) image[where.x] = w_h;
			if (where.y != 0xffffffffu) image[where.y] = w_h;
			if (where.z != 0xffffffffu) image[where.z] = w_h;
			if (where.w != 0xffffffffu) image[where.w] = w_h;
		}
	}
}

// kernel_activation_backward_output (common_device.h:748): dL/d(pre-activation output) from the forward OUTPUT values
__global__ void __launch_bounds__(256) k_act_bwd_output(const uint32_t n_elems, const uint32_t act, const half_t* __restrict__ dL_dout, const half_t* __restrict__ out, half_t* __restrict__ result) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elems) return;
	result[i] = activation_bwd(act, dL_dout[i], out[i]);
}

inline uint32_t wgrad_grid(uint32_t n) {
	const uint32_t n_chunks = n / WG_CHUNK;
	return n_chunks < 256 ? (n_chunks ? n_chunks : 1) : 256;
}

template <int W, int NB>
void launch_fwd_oneblob(hipStream_t stream, const MlpDesc& d, const FwdArgs& a, uint32_t grid) {
	if (d.activation == (uint32_t)Activation::ReLU) hipLaunchKernelGGL((k_mlp_fwd<W, NB, (int)Activation::ReLU, false, 256, true>), dim3(grid), dim3(256), 0, stream, d, a);
	else hipLaunchKernelGGL((k_mlp_fwd<W, NB, -1, false, 256, true>), dim3(grid), dim3(256), 0, stream, d, a);
}

template <int W, int NB>
void launch_fwd_act(hipStream_t stream, const MlpDesc& d, const FwdArgs& a, uint32_t grid) {
	if (d.activation == (uint32_t)Activation::ReLU) hipLaunchKernelGGL((k_mlp_fwd<W, NB, (int)Activation::ReLU>), dim3(grid), dim3(256), 0, stream, d, a);
	else if (d.activation == (uint32_t)Activation::None) hipLaunchKernelGGL((k_mlp_fwd<W, NB, (int)Activation::None>), dim3(grid), dim3(256), 0, stream, d, a);
	else hipLaunchKernelGGL((k_mlp_fwd<W, NB, -1>), dim3(grid), dim3(256), 0, stream, d, a);
}

template <int W, int NB>
void launch_bwd_act(hipStream_t stream, const MlpDesc& d, const BwdArgs& a, uint32_t grid) {
	if (d.activation == (uint32_t)Activation::ReLU) hipLaunchKernelGGL((k_mlp_bwd<W, NB, (int)Activation::ReLU>), dim3(grid), dim3(256), 0, stream, d, a);
	else if (d.activation == (uint32_t)Activation::None) hipLaunchKernelGGL((k_mlp_bwd<W, NB, (int)Activation::None>), dim3(grid), dim3(256), 0, stream, d, a);
	else hipLaunchKernelGGL((k_mlp_bwd<W, NB, -1>), dim3(grid), dim3(256), 0, stream, d, a);
}

// column blocks (of 16 samples) a wave processes per iteration
constexpr int nb_for_width(int W) { return W >= 256 ? 1 : (W >= 128 ? 2 : 4); }

inline uint32_t mlp_grid(uint32_t n, int nb) {
	const uint32_t n_iters = n / (16 * nb);
	const uint32_t wgs = div_round_up(n_iters, 4);
	const uint32_t cap = 256 * 8;
	return wgs < cap ? (wgs ? wgs : 1) : cap;
}

} // namespace

size_t mlp_image_bytes(const MlpDesc& d) { return (size_t)(d.n_frags_fwd + d.n_frags_bwd + d.n_frags_r32) * 1024; }

void mlp_prepare_weights(hipStream_t stream, const MlpDesc& d, const void* params, void* image, bool want_bwd) {
	const uint32_t n_frags = d.n_frags_fwd + (want_bwd ? d.n_frags_bwd + d.n_frags_r32 : 0);
	const uint32_t total = n_frags * 512;
	hipLaunchKernelGGL(k_mlp_prep, dim3(div_round_up(total, 256)), dim3(256), 0, stream, d, (const half_t*)params, (half_t*)image, n_frags);
}

void mlp_forward(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, void* out, void* hidden) {
	MlpIo io{};
	io.x_half = x;
	io.out_half = out;
	mlp_forward_io(stream, d, image, n, io, hidden);
}

void mlp_forward_io(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const MlpIo& io, void* hidden) {
	CHECK_THROW(n % BATCH_SIZE_GRANULARITY == 0);
	CHECK_THROW(d.n_hidden >= 1);
	CHECK_THROW(io.x_half != nullptr || io.x_f32.data != nullptr);
	CHECK_THROW(io.out_half != nullptr || io.out_f32.data != nullptr);
	if (n == 0) return;
	FwdArgs a{(const half_t*)io.x_half, (half_t*)io.out_half, (half_t*)hidden, (const h8*)image, n, io.x_plane_features,
	          io.x_f32, io.x_f32_dims, io.x_scale, io.x_offset, 0u, io.out_f32, io.out_f32_dims};
	if (io.x_oneblob_bins) {
		CHECK_THROW(io.x_f32.data != nullptr && io.x_oneblob_bins >= 32 && (io.x_oneblob_bins & (io.x_oneblob_bins - 1)) == 0);
		while ((1u << a.oneblob_log2) < io.x_oneblob_bins) ++a.oneblob_log2;
	}
	// Small batches: with 4 column blocks per wave a batch of 2^14 rows is 256 waves -- 64 workgroups on 256 CUs, each wave walking its
	// 64 rows through the whole network alone.  One block per wave spreads the same rows over four times as many waves.
	const bool small = n / 64 < 4 * 256; // fewer than one workgroup of 4-block waves per CU
	if (a.oneblob_log2) {
		if (d.width == 64 && small) return launch_fwd_oneblob<64, 1>(stream, d, a, mlp_grid(n, 1));
		if (d.width == 64) return launch_fwd_oneblob<64, nb_for_width(64)>(stream, d, a, mlp_grid(n, nb_for_width(64)));
		if (d.width == 128) return launch_fwd_oneblob<128, nb_for_width(128)>(stream, d, a, mlp_grid(n, nb_for_width(128)));
		throw std::runtime_error{"mlp_forward_io: the fused OneBlob input needs a 64- or 128-wide network"};
	}
	if (d.width == 64 && small && hidden == nullptr) return launch_fwd_act<64, 1>(stream, d, a, mlp_grid(n, 1));
	// wide inference: fragment image in LDS, 4 column blocks per wave, persistent workgroups of 8 waves (see k_mlp_fwd)
	const uint32_t image_bytes = d.n_frags_fwd * 1024;
	const char* lds_env = getenv("TCNN_AMD_MLP_FWD_LDS"); // "0": the L2-resident form (A/B runs; read per call so that tests cover both)
	const bool lds_inference = !(lds_env && lds_env[0] == '0');
	if (lds_inference && d.width == 128 && hidden == nullptr && image_bytes <= 150 * 1024 && n >= 256 * 8 * 64) {
		constexpr int NB = 4, THREADS = 512;
		const uint32_t grid = std::min<uint32_t>(256, div_round_up(n / (16 * NB), THREADS / 64));
		auto go = [&](auto kernel) {
			HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)image_bytes));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(THREADS), image_bytes, stream, d, a);
			HIP_CHECK_THROW(hipGetLastError());
		};
		if (d.activation == (uint32_t)Activation::ReLU) go(k_mlp_fwd<128, NB, (int)Activation::ReLU, true, THREADS>);
		else go(k_mlp_fwd<128, NB, -1, true, THREADS>);
		return;
	}
	switch (d.width) {
		case 16: return launch_fwd_act<16, nb_for_width(16)>(stream, d, a, mlp_grid(n, nb_for_width(16)));
		case 32: return launch_fwd_act<32, nb_for_width(32)>(stream, d, a, mlp_grid(n, nb_for_width(32)));
		case 64: return launch_fwd_act<64, nb_for_width(64)>(stream, d, a, mlp_grid(n, nb_for_width(64)));
		case 128: return launch_fwd_act<128, nb_for_width(128)>(stream, d, a, mlp_grid(n, nb_for_width(128)));
		case 256: return launch_fwd_act<256, nb_for_width(256)>(stream, d, a, mlp_grid(n, nb_for_width(256))); // CutlassMLP configs only
		default: throw std::runtime_error{"FullyFusedMLP only supports 16, 32, 64, and 128 neurons."};
	}
}

void mlp_backward(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* dL_dout, const void* out, const void* hidden, void* dhidden, void* dL_dx,
                  uint32_t dx_plane_features) {
	CHECK_THROW(n % BATCH_SIZE_GRANULARITY == 0);
	CHECK_THROW(d.n_hidden >= 1);
	if (n == 0) return;
	BwdArgs a{(const half_t*)dL_dout, (const half_t*)out, (const half_t*)hidden, (half_t*)dhidden, (half_t*)dL_dx,
	          (const h8*)((const char*)image + (size_t)d.n_frags_fwd * 1024), n, dx_plane_features};
	switch (d.width) {
		case 16: return launch_bwd_act<16, nb_for_width(16)>(stream, d, a, mlp_grid(n, nb_for_width(16)));
		case 32: return launch_bwd_act<32, nb_for_width(32)>(stream, d, a, mlp_grid(n, nb_for_width(32)));
		case 64: return launch_bwd_act<64, nb_for_width(64)>(stream, d, a, mlp_grid(n, nb_for_width(64)));
		case 128: return launch_bwd_act<128, nb_for_width(128)>(stream, d, a, mlp_grid(n, nb_for_width(128)));
		case 256: return launch_bwd_act<256, nb_for_width(256)>(stream, d, a, mlp_grid(n, nb_for_width(256)));
		default: throw std::runtime_error{"FullyFusedMLP only supports 16, 32, 64, and 128 neurons."};
	}
}

void mlp_activation_backward_output(hipStream_t stream, uint32_t n_elems, uint32_t activation, const void* dL_dout, const void* out, void* result) {
	if (n_elems == 0) return;
	hipLaunchKernelGGL(k_act_bwd_output, dim3(div_round_up(n_elems, 256)), dim3(256), 0, stream, n_elems, activation, (const half_t*)dL_dout, (const half_t*)out, (half_t*)result);
}

void mlp_reduce_slabs(hipStream_t stream, uint32_t n_params, uint32_t n_slabs, const float* slabs, void* grad_half, bool accumulate, const AdamInFlush* adam) {
	if (n_params == 0) return;
	const dim3 grid(div_round_up(n_params, (uint32_t)WR_ELEMS)), block(WR_ELEMS * WR_GROUPS);
	if (adam && !accumulate) {
		if (adam->steps16) hipLaunchKernelGGL(k_wgrad_reduce_adam<uint16_t>, grid, block, 0, stream, n_params, n_slabs, slabs, (half_t*)grad_half, *adam);
		else hipLaunchKernelGGL(k_wgrad_reduce_adam<uint32_t>, grid, block, 0, stream, n_params, n_slabs, slabs, (half_t*)grad_half, *adam);
	} else {
		hipLaunchKernelGGL(k_wgrad_reduce, grid, block, 0, stream, n_params, n_params, n_params, n_slabs, slabs, (half_t*)grad_half, accumulate ? 1 : 0);
	}
	HIP_CHECK_THROW(hipGetLastError());
}

size_t wgrad_workspace_floats(uint32_t rows, uint32_t cols, uint32_t n) { return (size_t)wgrad_grid(n) * rows * cols; }

// the shapes k_wgrad_rows / k_wgrad_cols are instantiated for; TCNN_AMD_WGRAD_ROWS=0: k_wgrad for every shape (A/B runs)
static bool wgrad_fast_shape(uint32_t rows, uint32_t cols) {
	const char* e = getenv("TCNN_AMD_WGRAD_ROWS"); // (read per call: a test compares the two settings in one process)
	const bool rows_form = !(e && e[0] == '0');
	return rows_form && ((rows == 128 || rows == 64) && (cols == 128 || cols == 64 || cols == 32) || (rows == 16 && (cols == 128 || cols == 64)));
}

static void launch_wgrad_jobs(hipStream_t stream, uint32_t n, uint32_t rows, uint32_t cols, const WgradJobs& jobs, uint32_t n_jobs, bool accumulate) {
	const uint32_t grid = wgrad_grid(n);
	const size_t shmem = (size_t)WG_CHUNK * ((rows + WG_PAD) + (cols + WG_PAD)) * sizeof(half_t);
	const dim3 g(grid, n_jobs), b(256);
#define TCNN_WGRAD_ROWS(TRW, TC) hipLaunchKernelGGL((k_wgrad_rows<TRW, TC>), g, b, shmem, stream, n, jobs)
	if (rows == 128 && cols == 128) TCNN_WGRAD_ROWS(2, 8);
	else if (rows == 128 && cols == 64) TCNN_WGRAD_ROWS(2, 4);
	else if (rows == 128 && cols == 32) TCNN_WGRAD_ROWS(2, 2);
	else if (rows == 64 && cols == 128) TCNN_WGRAD_ROWS(1, 8);
	else if (rows == 64 && cols == 64) TCNN_WGRAD_ROWS(1, 4);
	else if (rows == 64 && cols == 32) TCNN_WGRAD_ROWS(1, 2);
	else if (rows == 16 && cols == 128) hipLaunchKernelGGL((k_wgrad_cols<2>), g, b, shmem, stream, n, jobs);
	else if (rows == 16 && cols == 64) hipLaunchKernelGGL((k_wgrad_cols<1>), g, b, shmem, stream, n, jobs);
	else throw std::runtime_error{"launch_wgrad_jobs: no kernel for this shape"};
#undef TCNN_WGRAD_ROWS
	const uint32_t n_elems = rows * cols;
	hipLaunchKernelGGL(k_wgrad_reduce_jobs, dim3(div_round_up(n_elems, (uint32_t)WR_ELEMS), n_jobs), dim3(WR_ELEMS * WR_GROUPS), 0, stream, n_elems, cols, grid, jobs, accumulate ? 1 : 0);
	HIP_CHECK_THROW(hipGetLastError());
}

void mlp_wgrad(hipStream_t stream, uint32_t n, const void* dO, uint32_t ldo, uint32_t rows, const void* In, uint32_t ldi, uint32_t cols,
               void* grad_half, uint32_t ldg, bool accumulate, float* workspace) {
	const WgradPanel p{dO, ldo, rows, In, ldi, cols, grad_half, ldg, false, false};
	mlp_wgrad_panels(stream, n, &p, 1, accumulate, workspace);
}

size_t wgrad_panels_workspace_floats(const WgradPanel* panels, uint32_t count, uint32_t n) {
	size_t total = 0;
	for (uint32_t i = 0; i < count; ++i) total += wgrad_workspace_floats(panels[i].rows, panels[i].cols, n);
	return total;
}

// Several products over the same n samples (a network's layers): those of one shape the fast kernels take share a launch (and one launch of
// the slab reduction) -- they are independent, and 2 x 6 launches of ~5 - 35 us were a fifth of a 128 x 5 step at 2^16 samples.  Every
// product keeps its own slabs, its chunks and their order: the same bits as one launch per product.
void mlp_wgrad_panels(hipStream_t stream, uint32_t n, const WgradPanel* panels, uint32_t count, bool accumulate, float* workspace) {
	CHECK_THROW(n % WG_CHUNK == 0);
	std::vector<bool> done(count, false);
	std::vector<float*> slabs(count);
	size_t at = 0;
	for (uint32_t i = 0; i < count; ++i) {
		slabs[i] = workspace + at;
		at += wgrad_workspace_floats(panels[i].rows, panels[i].cols, n);
	}
	for (uint32_t i = 0; i < count; ++i) {
		if (done[i]) continue;
		const WgradPanel& p = panels[i];
		CHECK_THROW(p.rows % 16 == 0 && p.cols % 16 == 0);
		CHECK_THROW((p.rows / 16) * (p.cols / 16) <= 4 * WG_MAX_TILES);
		if (wgrad_fast_shape(p.rows, p.cols)) {
			WgradJobs jobs{};
			uint32_t n_jobs = 0;
			for (uint32_t j = i; j < count && n_jobs < WG_MAX_JOBS; ++j) {
				if (done[j] || panels[j].rows != p.rows || panels[j].cols != p.cols) continue;
				jobs.dO[n_jobs] = (const half_t*)panels[j].dO;
				jobs.In[n_jobs] = (const half_t*)panels[j].In;
				jobs.slabs[n_jobs] = slabs[j];
				jobs.grad[n_jobs] = (half_t*)panels[j].grad;
				jobs.ldo[n_jobs] = panels[j].ldo;
				jobs.ldi[n_jobs] = panels[j].ldi;
				jobs.ldg[n_jobs] = panels[j].ldg;
				jobs.tiled[n_jobs] = (panels[j].dO_tiled ? 1u : 0u) | (panels[j].In_tiled ? 2u : 0u);
				done[j] = true;
				++n_jobs;
			}
			launch_wgrad_jobs(stream, n, p.rows, p.cols, jobs, n_jobs, accumulate);
			continue;
		}
		const uint32_t grid = wgrad_grid(n);
		const size_t shmem = (size_t)WG_CHUNK * ((p.rows + WG_PAD) + (p.cols + WG_PAD)) * sizeof(half_t);
		hipLaunchKernelGGL(k_wgrad, dim3(grid), dim3(256), shmem, stream, n, (const half_t*)p.dO, p.ldo, p.rows, (const half_t*)p.In, p.ldi, p.cols, slabs[i], (p.dO_tiled ? 1u : 0u) | (p.In_tiled ? 2u : 0u));
		const uint32_t n_elems = p.rows * p.cols;
		hipLaunchKernelGGL(k_wgrad_reduce, dim3(div_round_up(n_elems, (uint32_t)WR_ELEMS)), dim3(WR_ELEMS * WR_GROUPS), 0, stream, n_elems, p.cols, p.ldg, grid, slabs[i], (half_t*)p.grad, accumulate ? 1 : 0);
		done[i] = true;
	}
}

} // namespace tcnn_amd
