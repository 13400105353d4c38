// k_train_regs.hip -- the training step's MLP part for 64-wide networks with at most 32 inputs, entirely in registers.
//
// Same job as k_train.hip (reference: src/fully_fused_mlp.cu:500-557 forward, losses/{l2,relative_l2}.h:40-75,
// fully_fused_mlp.cu:151-259 backward, :785-828 + cutlass_matmul.h:438-479 the three weight-gradient GEMMs), different shape:
//
//   * a wave owns 16 samples per trip and never synchronises with another wave: no workgroup barrier in the trip loop;
//   * forward and backward are the register chain of k_mlp.hip (the accumulator tile of one layer, transposed activations
//     [feature][sample], IS the B operand of the next layer);
//   * the weight gradients dW = dOut^T In need the SAMPLE index on the k axis, i.e. both operands transposed.  The matrix
//     cores do that themselves: the chain fragment of an activation, used as an A operand against a constant 0/1 selection
//     fragment, comes out of one MFMA as the [sample][feature] tile with samples in the accumulator registers -- exactly
//     the A / B operand layout of v_mfma_f32_16x16x16_f16 with k = sample.  The product is exact (one 1.0 x value per
//     sum), so the fp16 conversion of the result loses nothing.  19 extra MFMAs per 16 samples replace the LDS images,
//     both barriers and every transposing LDS read of k_train.hip;
//   * each wave keeps private fp32 accumulators of ALL weight-gradient tiles (28 tiles = 112 registers for 32 -> 64 -> 64
//     -> 16) for the whole kernel; the eight waves of a workgroup are summed through LDS in a fixed tree after the last
//     trip and the workgroup writes one slab, reduced across workgroups by k_wgrad_reduce (deterministic, no atomics);
//   * weight fragments (k_mlp_prep's images) are copied to LDS once per workgroup and read with one ds_read_b128 per use.
//
// MFMA work per 16 samples: 30 (chain) + 19 (transposes) + 28 (k = 16 weight-gradient products) issue slots of 16 clocks.
#include "mlp_device.h"

namespace tcnn_amd {
namespace {

struct RegsArgs {
	const half_t* x;        // encoded input: AoS [n][in_width] or level planes [in_width / F][n][F]
	const float* target;    // [n][dims] (unused with external dL/dy)
	const float* data_pdf;  // optional [n][dims]
	const half_t* ext_dy;   // external dL/doutput [n][16] (LOSS == 0)
	half_t* out;            // optional [n][16]
	half_t* dL_dout;        // optional [n][16]
	float* L;               // optional [n][16]
	float* loss_sums;       // optional [gridDim.x]: sum of the loss values of the workgroup's samples
	half_t* dL_dx;          // optional: AoS, level planes or scatter records
	float* slabs;           // optional [gridDim.x][n_params]
	const h8* image;        // k_mlp_prep's fragment images, forward then backward
	const float* rec_x;     // records: the samples' coordinates [n][rec_dims]
	uint32_t n, dims, rec_dims;
	uint32_t x_plane_f, dx_plane_f, n_params;
	float loss_scale;
};

constexpr int REGS_NW = 8; // waves per workgroup

// fragment slots inside the LDS copy of the images for a (16 IN_T) -> 64 -> ... -> 64 -> 16 network; Network's constructor
// (model.h) lays the images out the same way, checked by mlp_train_regs_supported
template <int IN_T, int NH> struct RegsLayout {
	static constexpr int T = 4, KS = 2;
	static constexpr int fwd0 = 0;                                   // T fragments (one k-step)
	static constexpr int fwd_hidden(int l) { return T + (l - 1) * T * KS; } // layer l >= 1: [t][s]
	static constexpr int fwd_out = T + (NH - 1) * T * KS;            // [s]
	static constexpr int n_fwd = fwd_out + KS;
	static constexpr int bwd0 = n_fwd;                               // W0^T: [ti][s]
	static constexpr int bwd_hidden(int l) { return bwd0 + IN_T * KS + (l - 1) * T * KS; } // W_l^T: [t][s]
	static constexpr int bwd_out = bwd0 + IN_T * KS + (NH - 1) * T * KS;                   // Wout^T: [t]
	static constexpr int n_frags = bwd_out + T;
	static constexpr int sel = n_frags;                              // 4 selection fragments: chain order tile parity 0 / 1, natural order tile 0 / 1
	static constexpr int n_tiles = T * IN_T + (NH - 1) * T * T + T;  // weight-gradient tiles
};

// f32 accumulator tile -> 4 halves (round to nearest even, like the reference's fp16 accumulators are read)
__device__ inline h4 to_h4(const f4 v) { return h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; }
__device__ inline h8 join(const h4 lo, const h4 hi) { return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

template <int IN_T, int NH, int ACT, int LOSS>
__global__ void __launch_bounds__(REGS_NW * 64, 2) k_mlp_train_regs(const MlpDesc d, const RegsArgs a) {
	using Lay = RegsLayout<IN_T, NH>;
	constexpr int T = 4, KS = 2;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* lds_frag = (h8*)smem;

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 15, q = lane >> 4;

	// ---- weight fragments and the selection fragments into LDS
	for (uint32_t i = tid; i < (uint32_t)Lay::n_frags * 64; i += REGS_NW * 64) lds_frag[i] = a.image[i];
	if (tid < 4 * 64) {
		const uint32_t which = tid >> 6;
		uint32_t j = 8; // no element
		if (which < 2) { // chain order: element j of lane (n, q) is feature 16 (j >> 2) + 4 q + (j & 3) of the k-step -> tile parity `which`, column n
			if (q == (c >> 2)) j = 4 * which + (c & 3);
		} else { // natural order: element j is feature 8 q + j -> tile which - 2, column n
			if (q == 2 * (which - 2) + (c >> 3)) j = c & 7;
		}
		h8 v;
#pragma unroll
		for (int e = 0; e < 8; ++e) v[e] = (uint32_t)e == j ? (half_t)1.0f : (half_t)0.0f;
		lds_frag[Lay::sel * 64 + tid] = v;
	}
	__syncthreads();
	// The fragment reads are loop-invariant, and left alone the compiler hoists all of them out of the trip loop (136 registers,
	// spilled to scratch).  An opaque per-trip lane offset keeps each read next to its use.
	uint32_t lane_off = lane * 16;
	auto frag = [&](const int slot) -> h8 { return *(const h8*)(smem + lane_off + slot * 1024); };

	f4 wacc[Lay::n_tiles];
#pragma unroll
	for (int i = 0; i < Lay::n_tiles; ++i) wacc[i] = f4{0, 0, 0, 0};
	float loss_sum = 0.0f;

	const uint32_t n_blocks = a.n / 16;
	const uint32_t first = blockIdx.x * REGS_NW + wave, step = gridDim.x * REGS_NW;
	const uint32_t n_total = a.n * a.dims; // loss normalisation (relative_l2.h:58)
	const uint32_t in_w = 16 * IN_T;
	// outputs live in rows 4 q + r of the output tile: r < max_r covers every row < dims
	const uint32_t max_r = a.dims >= 4 ? 4 : a.dims;

	// 8 consecutive input features 8 q .. 8 q + 7 of one sample: the B operand of layer 0 (natural k order)
	auto load_x = [&](const uint32_t sample) -> h8 {
		const uint32_t k0 = 8 * q;
		if (IN_T == 1 && q >= 2) return h8{0, 0, 0, 0, 0, 0, 0, 0};
		if (a.x_plane_f == 2) {
			uint4 v;
			v.x = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 0) * a.n + sample) * 2);
			v.y = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 1) * a.n + sample) * 2);
			v.z = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 2) * a.n + sample) * 2);
			v.w = *(const uint32_t*)(a.x + ((size_t)(k0 / 2 + 3) * a.n + sample) * 2);
			return __builtin_bit_cast(h8, v);
		} else if (a.x_plane_f == 4) {
			const uint2 lo = *(const uint2*)(a.x + ((size_t)(k0 / 4) * a.n + sample) * 4);
			const uint2 hi = *(const uint2*)(a.x + ((size_t)(k0 / 4 + 1) * a.n + sample) * 4);
			uint4 v;
			v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
			return __builtin_bit_cast(h8, v);
		} else if (a.x_plane_f == 8) {
			return *(const h8*)(a.x + ((size_t)(k0 / 8) * a.n + sample) * 8);
		}
		return *(const h8*)(a.x + (size_t)sample * in_w + k0);
	};
	// per-lane side inputs of a trip: targets (and pdf) of output rows 4 q + r, the sample's coordinates for the scatter records.
	// Loads are unconditional at clamped addresses (rows >= dims re-read row 0 and are masked where they are used): no divergent
	// branches around single loads.
	struct Aux { float t[4], pdf[4], xs[3]; h4 dy; };
	auto load_aux = [&](const uint32_t sample) -> Aux {
		Aux r;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const uint32_t j = 4 * q + i;
			const uint32_t idx = sample * a.dims + (j < a.dims ? j : 0u);
			r.t[i] = 0.0f;
			r.pdf[i] = 1.0f;
			if constexpr (LOSS != 0) {
				if (i < (int)max_r) { // wave-uniform
					r.t[i] = a.target[idx];
					if (a.data_pdf) r.pdf[i] = a.data_pdf[idx];
				}
			}
		}
		r.xs[0] = r.xs[1] = r.xs[2] = 0.0f;
		if (a.rec_x) { // wave-uniform
			r.xs[0] = a.rec_x[(size_t)sample * a.rec_dims];
			r.xs[1] = a.rec_x[(size_t)sample * a.rec_dims + 1];
			if (a.rec_dims > 2) r.xs[2] = a.rec_x[(size_t)sample * a.rec_dims + 2];
		}
		if constexpr (LOSS == 0) r.dy = *(const h4*)(a.ext_dy + (size_t)sample * 16 + 4 * q);
		else r.dy = h4{0, 0, 0, 0};
		return r;
	};

	// The next trip's inputs are requested at the START of a trip, ahead of this trip's stores: vmcnt retires in issue order, so a
	// load issued behind the stores would also wait for their write acknowledgements.
	h8 pre_x = h8{0, 0, 0, 0, 0, 0, 0, 0};
	Aux pre_aux{};
	if (first < n_blocks) {
		pre_x = load_x(first * 16 + c);
		pre_aux = load_aux(first * 16 + c);
	}

	for (uint32_t blk = first; blk < n_blocks; blk += step) {
		asm volatile("" : "+v"(lane_off));
		const uint32_t sample = blk * 16 + c;
		const h8 bx = pre_x;
		const Aux aux = pre_aux;
		if (blk + step < n_blocks) {
			pre_x = load_x((blk + step) * 16 + c);
			pre_aux = load_aux((blk + step) * 16 + c);
		}

		// =============================================================== forward chain
		f4 acc[T];
#pragma unroll
		for (int t = 0; t < T; ++t) acc[t] = mfma(frag(Lay::fwd0 + t), bx, f4{0, 0, 0, 0});
		h8 hf[NH][KS]; // post-activation hidden layers as chain fragments (kept for the backward pass and the weight gradients)
		auto finish = [&](const int l) {
			h4 v[T];
#pragma unroll
			for (int t = 0; t < T; ++t) {
				v[t] = to_h4(acc[t]);
#pragma unroll
				for (int r = 0; r < 4; ++r) v[t][r] = act_fwd_t<ACT>(d.activation, v[t][r]);
			}
#pragma unroll
			for (int s = 0; s < KS; ++s) hf[l][s] = join(v[2 * s], v[2 * s + 1]);
		};
		finish(0);
#pragma unroll
		for (int l = 1; l < NH; ++l) {
#pragma unroll
			for (int t = 0; t < T; ++t) {
				acc[t] = mfma(frag(Lay::fwd_hidden(l) + t * KS + 0), hf[l - 1][0], f4{0, 0, 0, 0});
				acc[t] = mfma(frag(Lay::fwd_hidden(l) + t * KS + 1), hf[l - 1][1], acc[t]);
			}
			finish(l);
		}

		// =============================================================== output layer + loss on the accumulator tile
		h4 gv; // dL/d(pre-activation output), rows 4 q + r
		{
			f4 o = mfma(frag(Lay::fwd_out + 0), hf[NH - 1][0], f4{0, 0, 0, 0});
			o = mfma(frag(Lay::fwd_out + 1), hf[NH - 1][1], o);
			const h4 ov = to_h4(o); // output activation None (mlp_train_regs_supported)
			if constexpr (LOSS == 0) {
				gv = aux.dy;
			} else {
				float lv[4] = {0, 0, 0, 0};
				gv = h4{0, 0, 0, 0};
#pragma unroll
				for (int r = 0; r < 4; ++r) { // l2.h:40-74 / relative_l2.h:40-75
					if (r < (int)max_r) { // wave-uniform
						const float prediction = (float)ov[r];
						const float difference = prediction - aux.t[r];
						float value, gradient;
						if constexpr (LOSS == 2) {
							const float prediction_sq_plus_epsilon = prediction * prediction + 0.01f;
							value = difference * difference / prediction_sq_plus_epsilon;
							gradient = 2 * difference / prediction_sq_plus_epsilon;
						} else {
							value = difference * difference;
							gradient = 2 * difference;
						}
						if (a.data_pdf) { // wave-uniform; a division by 1 changes nothing
							value = value / aux.pdf[r];
							gradient = gradient / aux.pdf[r];
						}
						value = value / n_total;
						const half_t grad = (half_t)(a.loss_scale * gradient / n_total);
						const bool live = 4 * q + r < a.dims;
						lv[r] = live ? value : 0.0f;
						gv[r] = live ? grad : (half_t)0.0f;
					}
				}
				loss_sum += (lv[0] + lv[1]) + (lv[2] + lv[3]);
				if (a.L) *(f4*)(a.L + (size_t)sample * 16 + 4 * q) = f4{lv[0], lv[1], lv[2], lv[3]};
				if (a.dL_dout) *(h4*)(a.dL_dout + (size_t)sample * 16 + 4 * q) = gv;
			}
			if (a.out) *(h4*)(a.out + (size_t)sample * 16 + 4 * q) = ov;
		}
		const h8 dyf = join(gv, h4{0, 0, 0, 0}); // B fragment of the first backward product (k = output index, 16 of 32 used)

		// [sample][feature] tiles by selection products.  sel 0 / 1: chain-order fragment -> tile of parity 0 / 1 of its k-step
		auto transpose_chain = [&](const h8 f, const int parity) -> h4 { return to_h4(mfma(f, frag(Lay::sel + parity), f4{0, 0, 0, 0})); };
		auto transpose_natural = [&](const h8 f, const int tile) -> h4 { return to_h4(mfma(f, frag(Lay::sel + 2 + tile), f4{0, 0, 0, 0})); };

		// =============================================================== dWout = dY^T H_last   (slots after the hidden ones)
		constexpr int W_OUT = T * IN_T + (NH - 1) * T * T;
		if (a.slabs) {
			const h4 pa = transpose_chain(dyf, 0);
#pragma unroll
			for (int tc = 0; tc < T; ++tc) wacc[W_OUT + tc] = mfma16(pa, transpose_chain(hf[NH - 1][tc / 2], tc & 1), wacc[W_OUT + tc]);
		}

		// =============================================================== backward chain
#pragma unroll
		for (int t = 0; t < T; ++t) acc[t] = mfma(frag(Lay::bwd_out + t), dyf, f4{0, 0, 0, 0});
		h8 gf[KS];
#pragma unroll
		for (int l = NH - 1; l >= 0; --l) {
			// acc = W_{l+1}^T dH_{l+1}; times act'(H_l) from the forward OUTPUT (common_device.h:241-297)
			{
				h4 g[T];
#pragma unroll
				for (int t = 0; t < T; ++t) {
					g[t] = to_h4(acc[t]);
#pragma unroll
					for (int r = 0; r < 4; ++r) g[t][r] = act_bwd_t<ACT>(d.activation, g[t][r], hf[l][t / 2][(t & 1) * 4 + r]);
				}
#pragma unroll
				for (int s = 0; s < KS; ++s) gf[s] = join(g[2 * s], g[2 * s + 1]);
			}
			if (a.slabs) { // dW_l = dH_l^T In_l
				if (l > 0) {
					h4 pb[T];
#pragma unroll
					for (int tc = 0; tc < T; ++tc) pb[tc] = transpose_chain(hf[l - 1][tc / 2], tc & 1);
#pragma unroll
					for (int tr = 0; tr < T; ++tr) {
						const h4 pa = transpose_chain(gf[tr / 2], tr & 1);
#pragma unroll
						for (int tc = 0; tc < T; ++tc) {
							const int slot = T * IN_T + (l - 1) * T * T + tr * T + tc;
							wacc[slot] = mfma16(pa, pb[tc], wacc[slot]);
						}
					}
				} else {
					h4 pb[IN_T];
#pragma unroll
					for (int tc = 0; tc < IN_T; ++tc) pb[tc] = transpose_natural(bx, tc);
#pragma unroll
					for (int tr = 0; tr < T; ++tr) {
						const h4 pa = transpose_chain(gf[tr / 2], tr & 1);
#pragma unroll
						for (int tc = 0; tc < IN_T; ++tc) wacc[tr * IN_T + tc] = mfma16(pa, pb[tc], wacc[tr * IN_T + tc]);
					}
				}
			}
			if (l > 0) {
#pragma unroll
				for (int t = 0; t < T; ++t) {
					acc[t] = mfma(frag(Lay::bwd_hidden(l) + t * KS + 0), gf[0], f4{0, 0, 0, 0});
					acc[t] = mfma(frag(Lay::bwd_hidden(l) + t * KS + 1), gf[1], acc[t]);
				}
			}
		}

		// =============================================================== dX = W0^T dH_0
		if (a.dL_dx) {
#pragma unroll
			for (int ti = 0; ti < IN_T; ++ti) {
				f4 o = mfma(frag(Lay::bwd0 + ti * KS + 0), gf[0], f4{0, 0, 0, 0});
				o = mfma(frag(Lay::bwd0 + ti * KS + 1), gf[1], o);
				const h4 v = to_h4(o);
				if (a.rec_x) store_dx_record(a.dL_dx, a.dx_plane_f, a.rec_dims, a.n, sample, 16 * ti + 4 * q, v, aux.xs);
				else store_dx(a.dL_dx, a.dx_plane_f, a.n, in_w, sample, 16 * ti + 4 * q, v);
			}
		}
	}

	// ---- loss: lanes -> wave -> workgroup in a fixed order
	if (a.loss_sums) {
		float s = loss_sum;
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
		__syncthreads(); // everyone is done with the fragments
		float* red = (float*)smem;
		if (lane == 0) red[wave] = s;
		__syncthreads();
		if (tid == 0) {
			float t = 0.0f;
			for (int w = 0; w < REGS_NW; ++w) t += red[w];
			a.loss_sums[blockIdx.x] = t;
		}
	}
	if (!a.slabs) return;

	// ---- weight gradients: fixed tree over the 8 waves through LDS (the upper half writes, the lower half adds), then the
	// sum goes to LDS in parameter order and out to the workgroup's slab with dense 16-byte stores
	f4* region = (f4*)smem; // [4][n_tiles][64]
	__syncthreads();
#pragma unroll
	for (int half = REGS_NW / 2; half >= 1; half >>= 1) {
		if (wave >= (uint32_t)half && wave < 2u * half) {
			f4* dst = region + (size_t)(wave - half) * Lay::n_tiles * 64 + lane;
#pragma unroll
			for (int i = 0; i < Lay::n_tiles; ++i) dst[i * 64] = wacc[i];
		}
		__syncthreads();
		if (wave < (uint32_t)half) {
			const f4* src = region + (size_t)wave * Lay::n_tiles * 64 + lane;
#pragma unroll
			for (int i = 0; i < Lay::n_tiles; ++i) {
				const f4 v = src[i * 64];
				wacc[i] = f4{wacc[i][0] + v[0], wacc[i][1] + v[1], wacc[i][2] + v[2], wacc[i][3] + v[3]};
			}
		}
		__syncthreads();
	}
	float* flat = (float*)smem;
	if (wave == 0) {
		// tile (row tile tr, column tile tc) of a matrix with `cols` columns: lane (c, q) holds rows 16 tr + 4 q + r of column 16 tc + c
		auto put = [&](const f4& v, const uint32_t w_off, const uint32_t cols, const int tr, const int tc) {
#pragma unroll
			for (int r = 0; r < 4; ++r) flat[w_off + (16 * tr + 4 * q + r) * cols + 16 * tc + c] = v[r];
		};
#pragma unroll
		for (int tr = 0; tr < T; ++tr)
#pragma unroll
			for (int tc = 0; tc < IN_T; ++tc) put(wacc[tr * IN_T + tc], d.layers[0].w_off, 16 * IN_T, tr, tc);
#pragma unroll
		for (int l = 1; l < NH; ++l)
#pragma unroll
			for (int tr = 0; tr < T; ++tr)
#pragma unroll
				for (int tc = 0; tc < T; ++tc) put(wacc[T * IN_T + (l - 1) * T * T + tr * T + tc], d.layers[l].w_off, 64, tr, tc);
#pragma unroll
		for (int tc = 0; tc < T; ++tc) put(wacc[T * IN_T + (NH - 1) * T * T + tc], d.layers[NH].w_off, 64, 0, tc);
	}
	__syncthreads();
	f4* slab = (f4*)(a.slabs + (size_t)blockIdx.x * a.n_params);
	for (uint32_t i = tid; i < a.n_params / 4; i += REGS_NW * 64) slab[i] = ((const f4*)flat)[i];
}

template <int IN_T, int NH> uint32_t regs_lds_bytes() {
	using Lay = RegsLayout<IN_T, NH>;
	const uint32_t frags = (Lay::n_frags + 4) * 1024;
	const uint32_t tree = (REGS_NW / 2) * Lay::n_tiles * 1024;
	return std::max(frags, tree);
}

template <int IN_T, int NH> bool regs_layout_matches(const MlpDesc& d) {
	using Lay = RegsLayout<IN_T, NH>;
	if (d.in_width != 16 * IN_T || d.width != 64 || d.out_width != 16 || d.n_hidden != NH) return false;
	// hidden activation ReLU or None, no output activation: everything else runs k_train.hip's kernels
	if ((d.activation != (uint32_t)Activation::ReLU && d.activation != (uint32_t)Activation::None) || d.output_activation != (uint32_t)Activation::None) return false;
	if ((int)d.n_frags_fwd != Lay::n_fwd || (int)(d.n_frags_fwd + d.n_frags_bwd) != Lay::n_frags) return false;
	if ((int)d.layers[0].fwd_off != Lay::fwd0 || (int)(d.n_frags_fwd + d.layers[0].bwd_off) != Lay::bwd0) return false;
	for (int l = 1; l < NH; ++l) {
		if ((int)d.layers[l].fwd_off != Lay::fwd_hidden(l) || (int)(d.n_frags_fwd + d.layers[l].bwd_off) != Lay::bwd_hidden(l)) return false;
	}
	return (int)d.layers[NH].fwd_off == Lay::fwd_out && (int)(d.n_frags_fwd + d.layers[NH].bwd_off) == Lay::bwd_out;
}

template <int IN_T, int NH> void launch_regs(hipStream_t stream, const MlpDesc& d, const RegsArgs& a, uint32_t grid, int loss) {
	const uint32_t lds = regs_lds_bytes<IN_T, NH>();
	auto go = [&](auto kernel) {
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(REGS_NW * 64), lds, stream, d, a);
		HIP_CHECK_THROW(hipGetLastError());
	};
	const bool relu = d.activation == (uint32_t)Activation::ReLU;
#define TCNN_REGS_CASE(L_) \
	if (loss == L_) { if (relu) go(k_mlp_train_regs<IN_T, NH, (int)Activation::ReLU, L_>); else go(k_mlp_train_regs<IN_T, NH, (int)Activation::None, L_>); return; }
	TCNN_REGS_CASE(0)
	TCNN_REGS_CASE(1)
	TCNN_REGS_CASE(2)
#undef TCNN_REGS_CASE
}

} // namespace

// TCNN_AMD_MLP_REGS=0 keeps k_train.hip's kernels (A/B runs; read per call so that tests can cover both forms in one process)
static bool regs_enabled() {
	const char* e = getenv("TCNN_AMD_MLP_REGS");
	return !(e && e[0] == '0');
}

bool mlp_train_regs_supported(const MlpDesc& d, uint32_t n) {
	if (!regs_enabled() || n == 0 || n % 16 != 0) return false;
	return regs_layout_matches<2, 2>(d) || regs_layout_matches<1, 2>(d) || regs_layout_matches<2, 1>(d) || regs_layout_matches<1, 1>(d);
}

uint32_t mlp_train_regs_grid(const MlpDesc& d, uint32_t n) {
	(void)d;
	return std::max(1u, std::min(256u, div_round_up(n / 16, (uint32_t)REGS_NW)));
}

void mlp_train_regs(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, uint32_t x_plane_features, const float* target, const float* data_pdf,
                    const void* external_dL_dy, uint32_t dims, LossType loss, float loss_scale, void* out, void* dL_dout, float* L, float* loss_sums, void* dL_dx,
                    uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims, float* slabs, uint32_t n_params) {
	CHECK_THROW(mlp_train_regs_supported(d, n));
	CHECK_THROW(external_dL_dy != nullptr || (target != nullptr && (loss == LossType::L2 || loss == LossType::RelativeL2)));
	CHECK_THROW(n_params % 4 == 0);
	RegsArgs a{(const half_t*)x, target, data_pdf, (const half_t*)external_dL_dy, (half_t*)out, (half_t*)dL_dout, L, loss_sums, (half_t*)dL_dx, slabs, (const h8*)image,
	           dx_record_x, n, dims, dx_record_dims, x_plane_features, dx_plane_features, n_params, loss_scale};
	const int loss_id = external_dL_dy ? 0 : (loss == LossType::L2 ? 1 : 2);
	const uint32_t grid = mlp_train_regs_grid(d, n);
	if (regs_layout_matches<2, 2>(d)) return launch_regs<2, 2>(stream, d, a, grid, loss_id);
	if (regs_layout_matches<1, 2>(d)) return launch_regs<1, 2>(stream, d, a, grid, loss_id);
	if (regs_layout_matches<2, 1>(d)) return launch_regs<2, 1>(stream, d, a, grid, loss_id);
	return launch_regs<1, 1>(stream, d, a, grid, loss_id);
}

} // namespace tcnn_amd
