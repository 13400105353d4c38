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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {

struct RegsArgs {
	const half_t* x;        // encoded input: AoS [n][in_width] or level planes [in_width / F][n][F]
	const float* target;    // [n][dims] (unused with external dL/dy)
	const float* data_pdf;  // optional [n][dims]
	const half_t* ext_dy;   // external dL/doutput [n][16] (LOSS == 0)
	half_t* out;            // optional [n][16]
	half_t* dL_dout;        // [n][dims]: the live columns of the reference's [n][16] matrix, dense ("compact"; mlp_expand_context pads them)
	float* L;               // [n][dims], likewise
	half_t* dL_dx;          // optional: AoS, level planes or scatter records
	float* slabs;           // optional [gridDim.x][n_params]
	const h8* image;        // k_mlp_prep's fragment images, forward then backward
	const float* rec_x;     // records: the samples' coordinates [n][rec_dims]
	uint32_t n, dims, rec_dims;
	uint32_t x_plane_f, dx_plane_f, n_params;
	float loss_scale;
	uint32_t prio_mode;      // 1 (default): the waves of a SIMD alternate their priority per trip; 0: no priorities; 2: the younger half at priority 1 (TCNN_AMD_MLP_PRIO, A/B runs)
	unsigned long long* dbg; // development aid (TCNN_AMD_MLP_TIMING): per workgroup, wave 0's clock at kernel start / first trip / last trip done / end
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

// f32 accumulator tile -> 4 halves (round to nearest even, like the reference's fp16 accumulators are read); as a vector
// conversion so that it becomes two v_cvt_pk_f16_f32
__device__ inline h4 to_h4(const f4 v) { return __builtin_convertvector(v, h4); }
__device__ inline h8 join(const h4 lo, const h4 hi) { return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

// hidden activation on a tile, packed: ReLU(half) = x > 0 ? x : 0 (common_device.h:92-98).  As a SIGNED INTEGER maximum of the bit
// patterns (v_pk_max_i16): negative halves, -0 included, are negative integers.  v_pk_max_f16 may return -0 for max(-0, +0), and
// the backward pass below tells "positive" from "zero" by the bits.  ACT None: identity
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int ACT> __device__ inline h4 act_fwd_tile(const h4 v) {
	if constexpr (ACT == (int)Activation::ReLU) return __builtin_bit_cast(h4, __builtin_elementwise_max(__builtin_bit_cast(s16x4, v), s16x4{0, 0, 0, 0}));
	else return v;
}
// ... and its derivative from the forward output (common_device.h:241-297): ReLU keeps the gradient where the output is
// positive.  Outputs of a ReLU are +0 or positive, so "positive" is "any bit set": min(bits, 1) = 0 / 1 -> 0 - that = all
// zeros / all ones, packed 16-bit integer ops.  (A masked gradient becomes +0 where the reference forms g * 0 = +-0.)
template <int ACT> __device__ inline h4 act_bwd_tile(const h4 g, const h4 fwd) {
	if constexpr (ACT == (int)Activation::ReLU) {
		// inline assembly: written as vector code the compiler turns min(x, 1) back into per-element compares and selects
		const uint2 f = __builtin_bit_cast(uint2, fwd), gb = __builtin_bit_cast(uint2, g);
		uint2 m;
		asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_sub_u16 %0, 0, %0 op_sel_hi:[0,1]" : "=&v"(m.x) : "v"(f.x));
		asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_sub_u16 %0, 0, %0 op_sel_hi:[0,1]" : "=&v"(m.y) : "v"(f.y));
		return __builtin_bit_cast(h4, (uint2{gb.x & m.x, gb.y & m.y}));
	} else {
		return g;
	}
}
// loads / stores at 32-bit byte offsets from wave-uniform bases (saddr + voffset addressing; mlp_train_regs_supported caps n)
template <typename V> __device__ inline V ld32(const void* base, const uint32_t byte_off) { return *(const V*)((const char*)base + byte_off); }
template <typename V> __device__ inline void st32(void* base, const uint32_t byte_off, const V v) { *(V*)((char*)base + byte_off) = v; }
// the same as a streaming store, for data nobody on the GPU reads soon (the step's outputs: out, dL_doutput, L): it leaves this
// XCD's L2 as it is written instead of waiting there, dirty, for the write-back at the end of the kernel.  NOT for the scatter
// records: measured on C3a, streamed records cost the scatter 10 us (it finds them in the caches otherwise) for 1.4 us gained here.
template <typename V> __device__ inline void st32_stream(void* base, const uint32_t byte_off, const V v) {
	typedef uint32_t nt2 __attribute__((ext_vector_type(2)));
	typedef uint32_t nt4 __attribute__((ext_vector_type(4)));
	char* p = (char*)base + byte_off;
	if constexpr (sizeof(V) == 16) __builtin_nontemporal_store(__builtin_bit_cast(nt4, v), (nt4*)p);
	else if constexpr (sizeof(V) == 8) __builtin_nontemporal_store(__builtin_bit_cast(nt2, v), (nt2*)p);
	else if constexpr (sizeof(V) == 4) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, v), (uint32_t*)p);
	else __builtin_nontemporal_store(__builtin_bit_cast(uint16_t, v), (uint16_t*)p);
}

// FAST: the common case with every format decision made at compile time -- input as level planes of 2 features, at most 4 outputs,
// no data_pdf, `out` and scatter records {x, y, two levels} written.  Not for speed of the decisions themselves: vmcnt retires in
// issue order and counts stores too, and with branches between a trip's loads and its stores the compiler's wait for the
// prefetched inputs at the top of the next trip is vmcnt(0) -- every trip then also sat out the write acknowledgements of the
// stores it had just issued (measured: a third of the trip time).  Without those branches it counts: vmcnt(8) leaves the 8
// stores of the trip in flight.
// PHASES: development build (TCNN_AMD_MLP_TIMING=2) that sums wave 0's clocks per phase of the trip into a.dbg's tail.
template <int IN_T, int NH, int ACT, int LOSS, bool FAST, bool PHASES = false>
__global__ void __launch_bounds__(REGS_NW * 64, 2) k_mlp_train_regs(const MlpDesc d, const RegsArgs a) {
	using Lay = RegsLayout<IN_T, NH>;
	constexpr int T = 4, KS = 2;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* lds_frag = (h8*)smem;

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 15, q = lane >> 4;

	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();
	// The fragment reads are loop-invariant, and left alone the compiler hoists all of them out of the trip loop (136 registers,
	// spilled to scratch).  An opaque per-trip lane offset keeps each read next to its use.
	uint32_t lane_off = lane * 16;
	auto frag = [&](const int slot) -> h8 { return *(const h8*)(smem + lane_off + slot * 1024); };

	f4 wacc[Lay::n_tiles];
#pragma unroll
	for (int i = 0; i < Lay::n_tiles; ++i) wacc[i] = f4{0, 0, 0, 0};

	const uint32_t n_blocks = a.n / 16;
	const uint32_t first = blockIdx.x * REGS_NW + wave, step = gridDim.x * REGS_NW;
	const uint32_t n_total = a.n * a.dims; // loss normalisation (relative_l2.h:58)
	const LossScales lsc = loss_scales(n_total, a.loss_scale);
	const uint32_t in_w = 16 * IN_T;
	// OUTPUT ROWS.  The output layer's fragments are permuted while they are copied to LDS (below) so that accumulator register r
	// of lane (c, q) holds output q + 4 r of sample c instead of 4 q + r: with n_output_dims <= 4 every lane quarter then has ONE
	// live output in register 0, and the loss -- a dozen dependent IEEE divisions per output -- runs once per trip instead of
	// once per output.  n_r = registers that hold live outputs.
	const uint32_t n_r = FAST ? 1u : (a.dims + 3) / 4;
	const uint32_t xpf = FAST ? 2u : a.x_plane_f;
	const bool has_pdf = FAST ? false : a.data_pdf != nullptr;
	const bool has_out = FAST ? true : a.out != nullptr;
	const bool has_dx = FAST ? true : a.dL_dx != nullptr;
	const bool dx_rec = FAST ? true : a.rec_x != nullptr;
	const uint32_t rec_dims = FAST ? 2u : a.rec_dims;

	// ---- addressing.  Every global access of a trip is "wave-uniform base of the trip's 16-sample block" + "lane offset that
	// never changes": the bases are scalar arithmetic, the lane offsets are computed once, the trip loop spends (almost) no vector
	// instruction on addresses (written naively, 30 accesses per trip cost ~250 of them).
	const uint32_t n4 = a.n * 4;
	// input: 8 consecutive features 8 q .. 8 q + 7 of sample c, the B operand of layer 0 (natural k order).  in_w = 16: the lanes
	// q >= 2 have no features; they re-read those of q - 2 and are zeroed after the load.
	const uint32_t qx = IN_T == 1 ? (q & 1u) : q;
	uint32_t x_off, x_blk; // lane offset; bytes per 16-sample block
	if (xpf == 2) { x_off = (4 * qx * a.n + c) * 4; x_blk = 64; }        // levels 4 q + i at + i n 4
	else if (xpf == 4) { x_off = (2 * qx * a.n + c) * 8; x_blk = 128; }  // levels 2 q + i at + i n 8
	else if (xpf == 8) { x_off = (qx * a.n + c) * 16; x_blk = 256; }
	else { x_off = (c * in_w + 8 * qx) * 2; x_blk = 32 * in_w; }
	auto load_x = [&](const uint32_t blk) -> h8 {
		const char* base = (const char*)a.x + (size_t)blk * x_blk;
		uint4 v;
		if (xpf == 2) { // four 4-byte loads, each a dense 64-byte run per 16 lanes
			v.x = ld32<uint32_t>(base, x_off);
			v.y = ld32<uint32_t>(base + (size_t)n4, x_off);
			v.z = ld32<uint32_t>(base + (size_t)n4 * 2, x_off);
			v.w = ld32<uint32_t>(base + (size_t)n4 * 3, x_off);
		} else if (xpf == 4) {
			const uint2 lo = ld32<uint2>(base, x_off);
			const uint2 hi = ld32<uint2>(base + (size_t)n4 * 2, x_off);
			v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
		} else {
			v = ld32<uint4>(base, x_off);
		}
		if (IN_T == 1 && q >= 2) v = uint4{0, 0, 0, 0};
		return __builtin_bit_cast(h8, v);
	};
	// dL/dinput: features k0 .. k0 + 3, k0 = 16 ti + 4 q, of sample c.  Lane offset + per-tile stride + bytes per block + the
	// distance of the second store where a tile row needs two (dx_second = 0: one store).
	uint32_t dx_off = 0, dx_tile = 0, dx_blk = 0, dx_second = 0;
	if (FAST) { // float4 rec[level pair 4 ti + q][n]
		dx_off = (q * a.n + c) * 16; dx_tile = 4 * a.n * 16; dx_blk = 256;
	} else if (dx_rec && a.dx_plane_f == 2 && rec_dims == 3) { // float4 rec[level][n], levels 8 ti + 2 q and + 1
		dx_off = (2 * q * a.n + c) * 16; dx_tile = 8 * a.n * 16; dx_blk = 256; dx_second = a.n * 16;
	} else if (dx_rec) { // float4 rec[pair or level 4 ti + q][n] (mlp_device.h store_dx_record)
		dx_off = (q * a.n + c) * 16; dx_tile = 4 * a.n * 16; dx_blk = 256;
	} else if (a.dx_plane_f == 0) { // AoS [n][in_w]
		dx_off = (c * in_w + 4 * q) * 2; dx_tile = 32; dx_blk = 32 * in_w;
	} else if (a.dx_plane_f == 2) { // half2 plane[level][n], levels 8 ti + 2 q and + 1
		dx_off = (2 * q * a.n + c) * 4; dx_tile = 8 * a.n * 4; dx_blk = 64; dx_second = a.n * 4;
	} else if (a.dx_plane_f == 4) { // half4 plane[level 4 ti + q][n]
		dx_off = (q * a.n + c) * 8; dx_tile = 4 * a.n * 8; dx_blk = 128;
	} else { // F = 8: half8 plane[level 2 ti + q / 2][n], half (q & 1) of the sample's 16 bytes
		dx_off = ((q >> 1) * a.n + c) * 16 + (q & 1u) * 8; dx_tile = 2 * a.n * 16; dx_blk = 256;
	}

	// side inputs of a trip: target (and pdf) of output rows q + 4 r, the sample's coordinates for the scatter records, or the
	// external dL/doutput.  Rows >= dims re-read the last row (masked where they are used): no divergent branches around loads.
	uint32_t t_off[4];
#pragma unroll
	for (int r = 0; r < 4; ++r) t_off[r] = (c * a.dims + min(q + 4 * r, a.dims - 1)) * 4;
	const uint32_t xs_off = c * rec_dims * 4;
	const uint32_t o_off = (c * 16 + q) * 2; // out / external dL_dout: column q (+ 4 r: + 8 r bytes) of sample c in [n][16] halves
	const uint32_t cg_off = (c * a.dims + q) * 2; // compact dL_dout [n][dims] halves (+ 4 r: + 8 r bytes); compact L [n][dims] floats: twice that
	struct Aux { float t[4], pdf[4], xs[3]; h4 dy; };
	auto load_aux = [&](const uint32_t blk) -> Aux {
		Aux r;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			r.t[i] = 0.0f;
			r.pdf[i] = 1.0f;
			r.dy[i] = (half_t)0.0f;
		}
		if constexpr (LOSS != 0) {
			const char* tb = (const char*)a.target + (size_t)blk * (64 * a.dims);
			const char* pb = (const char*)a.data_pdf + (size_t)blk * (64 * a.dims);
			r.t[0] = ld32<float>(tb, t_off[0]);
			if (has_pdf) r.pdf[0] = ld32<float>(pb, t_off[0]);
			if (n_r > 1) { // wave-uniform, rare: more than 4 outputs
#pragma unroll
				for (int i = 1; i < 4; ++i) {
					r.t[i] = ld32<float>(tb, t_off[i]);
					if (has_pdf) r.pdf[i] = ld32<float>(pb, t_off[i]);
				}
			}
		} else {
			const char* yb = (const char*)a.ext_dy + (size_t)blk * 512;
#pragma unroll
			for (int i = 0; i < 4; ++i) r.dy[i] = ld32<half_t>(yb + 8 * i, o_off);
		}
		r.xs[0] = r.xs[1] = r.xs[2] = 0.0f;
		if (dx_rec) { // wave-uniform
			const char* xb = (const char*)a.rec_x + (size_t)blk * (64 * rec_dims);
			r.xs[0] = ld32<float>(xb, xs_off);
			r.xs[1] = ld32<float>(xb + 4, xs_off);
			if (rec_dims > 2) r.xs[2] = ld32<float>(xb + 8, xs_off);
		}
		return r;
	};
	// The next trip's inputs are requested at the START of a trip, ahead of this trip's stores: vmcnt retires in issue order, so a
	// load issued behind the stores would also wait for their write acknowledgements.
	h8 pre_x = h8{0, 0, 0, 0, 0, 0, 0, 0};
	Aux pre_aux{};
	if (first < n_blocks) {
		pre_x = load_x(first);
		pre_aux = load_aux(first);
	}

	// ---- weight fragments and the selection fragments into LDS (after the first trip's loads are on their way; all of a
	// thread's fragment loads are issued before the first LDS write)
	{
		constexpr uint32_t N16 = (uint32_t)Lay::n_frags * 64;
		constexpr int FILL = (N16 + REGS_NW * 64 - 1) / (REGS_NW * 64);
		h8 tmp[FILL];
#pragma unroll
		for (int k = 0; k < FILL; ++k) tmp[k] = a.image[min(tid + k * REGS_NW * 64, N16 - 1)];
#pragma unroll
		for (int k = 0; k < FILL; ++k) {
			if (tid + k * REGS_NW * 64 < N16) lds_frag[tid + k * REGS_NW * 64] = tmp[k];
		}
	}
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
	{
		// the output-row permutation (see n_r): position rho = 4 q + r of the output tile <-> output pi(rho) = q + 4 r.
		//   forward fragments of Wout (A rows = output positions): lane (rho, qq) takes the row of lane (pi(rho), qq);
		//   fragments of Wout^T (k = output position, chain order: element j < 4 of lane quarter qq is position 4 qq + j, the
		//   elements 4..7 are the zero padding of 16 outputs to a k-step of 32): element j takes output qq + 4 j = the original
		//   element qq of lane quarter j.
		h8 v = h8{0, 0, 0, 0, 0, 0, 0, 0};
		const bool mine = tid < (KS + T) * 64;
		const uint32_t which = tid >> 6; // 0 .. KS - 1: forward k-steps; KS .. KS + T - 1: Wout^T row tiles
		if (mine) {
			if (which < (uint32_t)KS) {
				v = lds_frag[(Lay::fwd_out + which) * 64 + ((c >> 2) + 4 * (c & 3)) + 16 * q];
			} else {
				const half_t* src = (const half_t*)(lds_frag + (Lay::bwd_out + which - KS) * 64);
#pragma unroll
				for (int j = 0; j < 4; ++j) v[j] = src[(c + 16 * j) * 8 + q];
			}
		}
		__syncthreads();
		if (mine) lds_frag[(which < (uint32_t)KS ? Lay::fwd_out + which : Lay::bwd_out + which - KS) * 64 + lane] = v;
	}
	__syncthreads();
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
	unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_prev = 0;
#define TCNN_PHASE(i) do { if constexpr (PHASES) { const unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - ph_prev; ph_prev = now_; } } while (0)
	// [sample][feature] tiles by selection products.  sel 0 / 1: chain-order fragment -> tile of parity 0 / 1 of its k-step; sel 2 / 3:
	// natural-order fragment -> tile 0 / 1
	auto transpose_chain = [&](const h8 f, const int parity) -> h4 { return to_h4(mfma(f, frag(Lay::sel + parity), f4{0, 0, 0, 0})); };
	auto transpose_natural = [&](const h8 f, const int tile) -> h4 { return to_h4(mfma(f, frag(Lay::sel + 2 + tile), f4{0, 0, 0, 0})); };

	// The two waves of a SIMD (waves w and w + 4) are arbitrated oldest first: left alone the older one takes every contended issue
	// slot, finishes its trips thousands of clocks before its partner and then waits at the final reduction (measured: the loop
	// ends of a workgroup's waves spread over a third of the loop time).  Alternating the priority per trip keeps them level.
	if (a.prio_mode == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
	uint32_t prio_phase = wave >= 4 ? 1u : 0u;
	for (uint32_t blk = first; blk < n_blocks; blk += step) {
		if constexpr (PHASES) ph_prev = __builtin_readcyclecounter();
		if (a.prio_mode != 0 && a.prio_mode != 2) { // wave-uniform
			if (prio_phase & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
			prio_phase ^= 1u;
		}
		asm volatile("" : "+v"(lane_off));
		const h8 bx = pre_x;
		const Aux aux = pre_aux;
		{ // unconditionally (the last trip re-reads its own block): a branch here would cost the counted wait, see FAST
			const uint32_t next = min(blk + step, n_blocks - 1);
			pre_x = load_x(next);
			pre_aux = load_aux(next);
		}

		// =============================================================== forward chain.  The [sample][feature] copies of the inputs
		// of every layer (B operands of the weight-gradient products) are formed as soon as their source exists: they do not depend
		// on the loss, whose long scalar-style arithmetic they then overlap.
		f4 acc[T];
#pragma unroll
		for (int t = 0; t < T; ++t) acc[t] = mfma(frag(Lay::fwd0 + t), bx, f4{0, 0, 0, 0});
		h4 px[IN_T];
#pragma unroll
		for (int tc = 0; tc < IN_T; ++tc) px[tc] = transpose_natural(bx, tc);
		h8 hf[NH][KS];   // post-activation hidden layers as chain fragments (kept for the backward pass)
		h4 ph_t[NH][T];  // ... and as [sample][feature] tiles
		auto finish = [&](const int l) {
			h4 v[T];
#pragma unroll
			for (int t = 0; t < T; ++t) v[t] = act_fwd_tile<ACT>(to_h4(acc[t]));
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
#pragma unroll
			for (int tc = 0; tc < T; ++tc) ph_t[l - 1][tc] = transpose_chain(hf[l - 1][tc / 2], tc & 1);
			finish(l);
		}

		TCNN_PHASE(0);
		// =============================================================== output layer + loss on the accumulator tile
		h4 gv = h4{0, 0, 0, 0}; // dL/d(pre-activation output), outputs q + 4 r
		{
			f4 o = mfma(frag(Lay::fwd_out + 0), hf[NH - 1][0], f4{0, 0, 0, 0});
			o = mfma(frag(Lay::fwd_out + 1), hf[NH - 1][1], o);
#pragma unroll
			for (int tc = 0; tc < T; ++tc) ph_t[NH - 1][tc] = transpose_chain(hf[NH - 1][tc / 2], tc & 1);
			const h4 ov = to_h4(o); // output activation None (mlp_train_regs_supported)
			if constexpr (LOSS == 0) {
				gv = aux.dy;
			} else {
				// l2.h:40-74 / relative_l2.h:40-75 on one refined reciprocal (loss_l2_fused, mlp_device.h).  Values and gradients of the live
				// outputs go to the compact context matrices [n][dims].
				char* gb = (char*)a.dL_dout + (size_t)blk * (32 * a.dims);
				char* lb = (char*)a.L + (size_t)blk * (64 * a.dims);
				auto loss_row = [&](const int r) {
					float value;
					half_t grad;
					loss_l2_fused<LOSS == 2>((float)ov[r], aux.t[r], lsc, value, grad, has_pdf, aux.pdf[r]);
					const bool live = q + 4 * r < a.dims;
					gv[r] = live ? grad : (half_t)0.0f;
					if (live) st32_stream(gb + 8 * r, cg_off, grad);
					if (live) st32_stream(lb + 16 * r, 2 * cg_off, value);
				};
				loss_row(0);
				if (n_r > 1) { // wave-uniform, rare: more than 4 outputs
#pragma unroll
					for (int r = 1; r < 4; ++r) loss_row(r);
				}
			}
			if (has_out) { // [n][16]: this lane holds the columns q, q + 4, q + 8, q + 12 of its sample's row
				char* ob = (char*)a.out + (size_t)blk * 512;
#pragma unroll
				for (int r = 0; r < 4; ++r) st32_stream(ob + 8 * r, o_off, ov[r]);
			}
		}
		const h8 dyf = join(gv, h4{0, 0, 0, 0}); // B fragment of the first backward product (k = output position, 16 of 32 used)

		TCNN_PHASE(1);
		// =============================================================== dWout = dY^T H_last   (slots after the hidden ones)
		constexpr int W_OUT = T * IN_T + (NH - 1) * T * T;
		{
			const h4 pa = transpose_chain(dyf, 0);
#pragma unroll
			for (int tc = 0; tc < T; ++tc) wacc[W_OUT + tc] = mfma16(pa, ph_t[NH - 1][tc], wacc[W_OUT + tc]);
		}

		TCNN_PHASE(2);
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
					const h8 f = hf[l][t / 2];
					const h4 fwd = (t & 1) ? h4{f[4], f[5], f[6], f[7]} : h4{f[0], f[1], f[2], f[3]};
					g[t] = act_bwd_tile<ACT>(to_h4(acc[t]), fwd);
				}
#pragma unroll
				for (int s = 0; s < KS; ++s) gf[s] = join(g[2 * s], g[2 * s + 1]);
			}
			// dW_l = dH_l^T In_l
			if (l > 0) {
#pragma unroll
				for (int tr = 0; tr < T; ++tr) {
					const h4 pa = transpose_chain(gf[tr / 2], tr & 1);
#pragma unroll
					for (int tc = 0; tc < T; ++tc) {
						const int slot = T * IN_T + (l - 1) * T * T + tr * T + tc;
						wacc[slot] = mfma16(pa, ph_t[l - 1][tc], wacc[slot]);
					}
				}
#pragma unroll
				for (int t = 0; t < T; ++t) {
					acc[t] = mfma(frag(Lay::bwd_hidden(l) + t * KS + 0), gf[0], f4{0, 0, 0, 0});
					acc[t] = mfma(frag(Lay::bwd_hidden(l) + t * KS + 1), gf[1], acc[t]);
				}
			} else {
#pragma unroll
				for (int tr = 0; tr < T; ++tr) {
					const h4 pa = transpose_chain(gf[tr / 2], tr & 1);
#pragma unroll
					for (int tc = 0; tc < IN_T; ++tc) wacc[tr * IN_T + tc] = mfma16(pa, px[tc], wacc[tr * IN_T + tc]);
				}
			}
		}

		TCNN_PHASE(3);
		// =============================================================== dX = W0^T dH_0
		if (has_dx) {
#pragma unroll
			for (int ti = 0; ti < IN_T; ++ti) {
				f4 o = mfma(frag(Lay::bwd0 + ti * KS + 0), gf[0], f4{0, 0, 0, 0});
				o = mfma(frag(Lay::bwd0 + ti * KS + 1), gf[1], o);
				const h4 v = to_h4(o);
				char* base = (char*)a.dL_dx + (size_t)blk * dx_blk + (size_t)ti * dx_tile;
				const uint2 g = __builtin_bit_cast(uint2, v);
				if (dx_rec) { // scatter records (mlp_device.h store_dx_record)
					typedef uint32_t u4 __attribute__((ext_vector_type(4)));
					const uint32_t x0 = __builtin_bit_cast(uint32_t, aux.xs[0]), x1 = __builtin_bit_cast(uint32_t, aux.xs[1]), x2 = __builtin_bit_cast(uint32_t, aux.xs[2]);
					if (dx_second) {
						st32(base, dx_off, u4{x0, x1, x2, g.x});
						st32(base + dx_second, dx_off, u4{x0, x1, x2, g.y});
					} else {
						st32(base, dx_off, u4{x0, x1, g.x, g.y});
					}
				} else if (dx_second) {
					st32(base, dx_off, g.x);
					st32(base + dx_second, dx_off, g.y);
				} else {
					st32(base, dx_off, g);
				}
			}
		}
		TCNN_PHASE(4);
	}
#undef TCNN_PHASE
	if constexpr (PHASES) {
		if (a.dbg && tid == 0) {
			for (int i = 0; i < 5; ++i) a.dbg[(size_t)gridDim.x * (4 + REGS_NW) + blockIdx.x * 8 + i] = ph[i];
		}
	}

	if (a.dbg && lane == 0) a.dbg[gridDim.x * 4 + blockIdx.x * REGS_NW + wave] = __builtin_readcyclecounter();
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

	// ---- weight gradients: 8 waves -> 4 -> the slab, through LDS regions that lie BEHIND the fragments (a wave that is done
	// dumps its accumulators without waiting for anybody).  Fixed order of additions: bitwise reproducible.
	//   round A: waves 4..7 dump, waves 0..3 add region w;   then wave w sums the four regions for the tiles i = w (mod 8) and
	//   stores them into the workgroup's slab.
	f4* region = (f4*)(smem + (Lay::n_frags + 4) * 1024); // [4][n_tiles][64]
	if (wave >= 4) {
		f4* dst = region + (size_t)(wave - 4) * Lay::n_tiles * 64 + lane;
#pragma unroll
		for (int i = 0; i < Lay::n_tiles; ++i) dst[i * 64] = wacc[i];
	}
	__syncthreads();
	if (wave < 4) {
		f4* reg = region + (size_t)wave * Lay::n_tiles * 64 + lane;
#pragma unroll
		for (int i = 0; i < Lay::n_tiles; ++i) {
			const f4 v = reg[i * 64];
			reg[i * 64] = f4{wacc[i][0] + v[0], wacc[i][1] + v[1], wacc[i][2] + v[2], wacc[i][3] + v[3]}; // own lane's slot: no hazard
		}
	}
	__syncthreads();
	{
		float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
		// tile (row tile tr, column tile tc) of a matrix with `cols` columns: lane (c, q) holds rows 16 tr + 4 q + r of column
		// 16 tc + c -- of the output layer: output q + 4 r (the row permutation above)
		auto finish_tile = [&](const int i, const uint32_t w_off, const uint32_t cols, const int tr, const int tc, const bool out_layer) {
			if ((uint32_t)(i % REGS_NW) != wave) return; // wave-uniform
			const f4* src = region + (size_t)i * 64 + lane;
			const f4 r0 = src[0], r1 = src[(size_t)Lay::n_tiles * 64], r2 = src[(size_t)2 * Lay::n_tiles * 64], r3 = src[(size_t)3 * Lay::n_tiles * 64];
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const uint32_t row = out_layer ? q + 4 * r : 16 * tr + 4 * q + r;
				slab[w_off + row * cols + 16 * tc + c] = (r0[r] + r1[r]) + (r2[r] + r3[r]);
			}
		};
#pragma unroll
		for (int tr = 0; tr < T; ++tr)
#pragma unroll
			for (int tc = 0; tc < IN_T; ++tc) finish_tile(tr * IN_T + tc, d.layers[0].w_off, 16 * IN_T, tr, tc, false);
#pragma unroll
		for (int l = 1; l < NH; ++l)
#pragma unroll
			for (int tr = 0; tr < T; ++tr)
#pragma unroll
				for (int tc = 0; tc < T; ++tc) finish_tile(T * IN_T + (l - 1) * T * T + tr * T + tc, d.layers[l].w_off, 64, tr, tc, false);
#pragma unroll
		for (int tc = 0; tc < T; ++tc) finish_tile(T * IN_T + (NH - 1) * T * T + tc, d.layers[NH].w_off, 64, 0, tc, true);
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
}

template <int IN_T, int NH> uint32_t regs_lds_bytes() {
	using Lay = RegsLayout<IN_T, NH>;
	return (Lay::n_frags + 4) * 1024 + (REGS_NW / 2) * Lay::n_tiles * 1024; // fragments, then the regions of the final reduction
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
	constexpr int RELU = (int)Activation::ReLU, NONE = (int)Activation::None;
	// the compile-time formats of k_mlp_train_regs<FAST> (TCNN_AMD_MLP_FAST=0: the general form, for A/B runs and tests)
	const bool fast = switches().mlp_fast && loss != 0 && relu && a.x_plane_f == 2 && a.dims <= 4 && a.data_pdf == nullptr && a.out != nullptr && a.dL_dx != nullptr &&
	                  a.rec_x != nullptr && a.rec_dims == 2 && a.dx_plane_f == 2;
	if constexpr (IN_T == 2 && NH == 2) {
#ifdef TCNN_AMD_DEV
		static const bool phases = getenv("TCNN_AMD_MLP_TIMING") && getenv("TCNN_AMD_MLP_TIMING")[0] == '2';
		if (phases && a.dbg && fast && loss == 2) return go(k_mlp_train_regs<IN_T, NH, RELU, 2, true, true>);
#endif
	}
	if (fast) return loss == 1 ? go(k_mlp_train_regs<IN_T, NH, RELU, 1, true>) : go(k_mlp_train_regs<IN_T, NH, RELU, 2, true>);
#define TCNN_REGS_CASE(L_) \
	if (loss == L_) return relu ? go(k_mlp_train_regs<IN_T, NH, RELU, L_, false>) : go(k_mlp_train_regs<IN_T, NH, NONE, L_, false>);
	TCNN_REGS_CASE(0)
	TCNN_REGS_CASE(1)
	TCNN_REGS_CASE(2)
#undef TCNN_REGS_CASE
}

// compact context matrices -> the reference's padded ones: dL_dout [n][16] halves, L [n][16] floats, zero beyond `dims`
__global__ void __launch_bounds__(256) k_expand_context(const uint32_t n, const uint32_t dims, const half_t* __restrict__ cg, const float* __restrict__ cl, half_t* __restrict__ dL_dout,
                                                        float* __restrict__ L) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * 16) return;
	const uint32_t s = i >> 4, j = i & 15;
	dL_dout[i] = j < dims ? cg[s * dims + j] : (half_t)0.0f;
	L[i] = j < dims ? cl[s * dims + j] : 0.0f;
}

} // namespace

// TCNN_AMD_MLP_REGS=0 keeps k_train.hip's kernels (A/B runs; Switches, read once per model)
static bool regs_enabled() { return switches().mlp_regs; }

bool mlp_train_regs_supported(const MlpDesc& d, uint32_t n) {
	if (!regs_enabled() || n == 0 || n % 16 != 0 || n > (1u << 22)) return false; // 32-bit byte offsets into [n][...] matrices
	return regs_layout_matches<2, 2>(d) || regs_layout_matches<1, 2>(d) || regs_layout_matches<2, 1>(d) || regs_layout_matches<1, 1>(d);
}

uint32_t mlp_train_regs_grid(const MlpDesc& d, uint32_t n) {
	(void)d;
	uint32_t cap = 256;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_MLP_GRID")) cap = std::max(1, atoi(e)); // laboratory knob (how the trip time depends on the number of busy CUs: it does not)
#endif
	return std::max(1u, std::min(cap, div_round_up(n / 16, (uint32_t)REGS_NW)));
}

void mlp_train_regs(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, uint32_t x_plane_features, const float* target, const float* data_pdf,
                    const void* external_dL_dy, uint32_t dims, LossType loss, float loss_scale, void* out, void* compact_dL_dout, float* compact_L, void* dL_dx,
                    uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims, float* slabs, uint32_t n_params) {
	CHECK_THROW(mlp_train_regs_supported(d, n));
	CHECK_THROW(external_dL_dy != nullptr || (target != nullptr && (loss == LossType::L2 || loss == LossType::RelativeL2)));
	CHECK_THROW(slabs != nullptr && dims >= 1 && dims <= 16);
	CHECK_THROW(external_dL_dy || (compact_dL_dout != nullptr && compact_L != nullptr));
	// BASELINE configs 3 in the formats of the grid encoding's training step: the 32x32x16 kernel (k_train_r32.hip), same slabs
	if (mlp_train_r32_applies(d, n, x_plane_features, data_pdf, external_dL_dy, dims, loss, out, dL_dx, dx_plane_features, dx_record_x, dx_record_dims)) {
		return mlp_train_r32(stream, d, image, n, x, target, dims, loss, loss_scale, out, compact_dL_dout, compact_L, dL_dx, dx_record_x, slabs, n_params, mlp_train_r32_grid(n));
	}
	RegsArgs a{(const half_t*)x, target, data_pdf, (const half_t*)external_dL_dy, (half_t*)out, (half_t*)compact_dL_dout, compact_L, (half_t*)dL_dx, slabs, (const h8*)image,
	           dx_record_x, n, dims, dx_record_dims, x_plane_features, dx_plane_features, n_params, loss_scale, 1u, nullptr};
	a.prio_mode = switches().mlp_prio;
	const int loss_id = external_dL_dy ? 0 : (loss == LossType::L2 ? 1 : 2);
	const uint32_t grid = mlp_train_regs_grid(d, n);
#ifdef TCNN_AMD_DEV // laboratory build (build.py --dev): in-kernel clocks of the 5th launch
	static const bool timing = getenv("TCNN_AMD_MLP_TIMING") != nullptr;
	static int timing_left = 5;
	if (timing && timing_left > 0) {
		HIP_CHECK_THROW(hipMalloc(&a.dbg, (size_t)grid * (32 + 8 * REGS_NW + 64)));
		HIP_CHECK_THROW(hipMemset(a.dbg, 0, (size_t)grid * (32 + 8 * REGS_NW + 64)));
	}
#else
	int timing_left = 0; (void)timing_left;
#endif
	if (regs_layout_matches<2, 2>(d)) launch_regs<2, 2>(stream, d, a, grid, loss_id);
	else if (regs_layout_matches<1, 2>(d)) launch_regs<1, 2>(stream, d, a, grid, loss_id);
	else if (regs_layout_matches<2, 1>(d)) launch_regs<2, 1>(stream, d, a, grid, loss_id);
	else launch_regs<1, 1>(stream, d, a, grid, loss_id);
	if (a.dbg) {
		std::vector<unsigned long long> h((size_t)grid * (4 + REGS_NW + 8));
		HIP_CHECK_THROW(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
		if (--timing_left == 0) {
			double fill = 0, loop = 0, tail = 0, skew = 0;
			for (uint32_t g = 0; g < grid; ++g) { // last wave's loop end - first wave's loop end
				const unsigned long long* e = h.data() + (size_t)grid * 4 + (size_t)g * REGS_NW;
				skew += (double)(*std::max_element(e, e + REGS_NW) - *std::min_element(e, e + REGS_NW));
			}
			unsigned long long t_min = ~0ull, t_max = 0;
			for (uint32_t g = 0; g < grid; ++g) {
				fill += (double)(h[g * 4 + 1] - h[g * 4]);
				loop += (double)(h[g * 4 + 2] - h[g * 4 + 1]);
				tail += (double)(h[g * 4 + 3] - h[g * 4 + 2]);
				t_min = std::min(t_min, h[g * 4]);
				t_max = std::max(t_max, h[g * 4 + 3]);
			}
			fprintf(stderr, "k_mlp_train_regs wave 0 clocks, mean over %u workgroups: fill %.0f trips %.0f (%u blocks of 16 per wave) tail %.0f (of which the waves' loop ends are spread over %.0f)\n", grid,
			        fill / grid, loop / grid, div_round_up(n / 16, grid * REGS_NW), tail / grid, skew / grid);
			(void)t_min; (void)t_max;
			{ // when each wave of a workgroup leaves the trip loop (clocks after wave 0's kernel start, mean over the workgroups)
				double end_w[REGS_NW] = {};
				for (uint32_t g = 0; g < grid; ++g)
					for (int w = 0; w < REGS_NW; ++w) end_w[w] += (double)(h[(size_t)grid * 4 + (size_t)g * REGS_NW + w] - h[g * 4]);
				fprintf(stderr, "  loop end per wave:");
				for (int w = 0; w < REGS_NW; ++w) fprintf(stderr, " %.0f", end_w[w] / grid);
				fprintf(stderr, "\n");
			}
			double p[5] = {0, 0, 0, 0, 0};
			for (uint32_t g = 0; g < grid; ++g)
				for (int i = 0; i < 5; ++i) p[i] += (double)h[(size_t)grid * (4 + REGS_NW) + (size_t)g * 8 + i];
			if (p[0] > 0) fprintf(stderr, "  phases (wave 0, summed over its trips): forward %.0f output+loss %.0f dWout %.0f backward+dW %.0f dX+stores %.0f\n", p[0] / grid, p[1] / grid, p[2] / grid, p[3] / grid, p[4] / grid);
		}
		(void)hipFree(a.dbg);
	}
}

void mlp_expand_context(hipStream_t stream, uint32_t n, uint32_t dims, const void* compact_dL_dout, const float* compact_L, void* dL_dout, float* L) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_expand_context, dim3(div_round_up(n * 16, 256)), dim3(256), 0, stream, n, dims, (const half_t*)compact_dL_dout, compact_L, (half_t*)dL_dout, L);
	HIP_CHECK_THROW(hipGetLastError());
}

} // namespace tcnn_amd
