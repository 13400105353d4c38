// k_train_r32.hip -- the training step's MLP part for 32 -> 64 -> 64 -> 16 networks (BASELINE configs 3) on v_mfma_f32_32x32x16_f16.
//
// Same job as k_train_regs.hip (reference: src/fully_fused_mlp.cu:500-557 forward, losses/{l2,relative_l2}.h:40-75,
// fully_fused_mlp.cu:151-259 backward, :785-828 + cutlass_matmul.h:438-479 the three weight-gradient GEMMs), built around the
// larger matrix instruction: per FLOP half as many matrix instructions as the 16x16x32 form, and each of them leaves the SIMD's
// vector issue port free for 24 of its 32 clocks instead of 8 of 16 (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").
//
//   * a wave owns 32 samples per trip and never synchronises with another wave inside the trip loop;
//   * forward and backward are a register chain: a layer's result tile (features in the registers, the sample on the lane),
//     converted to halves, IS the B operand of the next layer (mlp_side_jobs.h, R32Frags: the weight fragments absorb the k order);
//   * the weight gradients dW = dOut In^T sum over the SAMPLE, which sits on the lane: both operands are needed transposed.
//     Every activation and every gradient tile is written once, as it stands in the registers (8 bytes = 4 features of one
//     sample), into a wave-private LDS image [feature group][sample] and read back with ds_read_b64_tr_b16 as an operand
//     fragment with the feature on the lane and 8 samples in the registers: no selection products, no extra conversions;
//     the images also keep the forward activations for the backward pass (ReLU mask), so they do not occupy registers meanwhile;
//   * each wave keeps private fp32 accumulators of all 8 weight-gradient tiles (128 registers) for the whole kernel; the eight
//     waves are summed through LDS in a fixed tree, one slab per workgroup (bitwise reproducible), k_wgrad_reduce sums the slabs;
//   * output "positions" (R32Frags): result register g of lane half h holds output 2 g + h, so with <= 4 outputs the loss runs on
//     two registers per lane.
//
// Matrix work per 32 samples: 30 (chain) + 16 (weight gradients) instructions of 32 clocks.
#include "r32_device.h"
#include "r32_train.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {


constexpr int R32_NW = 8;                 // waves per workgroup
constexpr int R32_NF = 30;                // weight fragments (R32Frags of 32 -> 64 -> 64 -> 16)
constexpr int R32_WAVE0 = R32_NF * 1024;
constexpr int R32_WAVE_BYTES = 15 * 1024; // X 2 K | H0 4 K | H1 4 K | dH 4 K | dY 1 K
constexpr int IMG_X = 0, IMG_H0 = 2048, IMG_H1 = 6144, IMG_DH = 10240, IMG_DY = 14336;
constexpr int R32_LDS_BYTES = R32_WAVE0 + R32_NW * R32_WAVE_BYTES; // 153 600

template <typename V> __device__ inline V ld32(const void* base, const uint32_t byte_off) { return *(const V*)((const char*)base + byte_off); }
template <typename V> __device__ inline void st32(void* base, const uint32_t byte_off, const V v) { *(V*)((char*)base + byte_off) = v; }
template <typename V> __device__ inline void st32_stream(void* base, const uint32_t byte_off, const V v) {
	typedef uint32_t nt4 __attribute__((ext_vector_type(4)));
	char* p = (char*)base + byte_off;
	if constexpr (sizeof(V) == 16) __builtin_nontemporal_store(__builtin_bit_cast(nt4, v), (nt4*)p);
	else if constexpr (sizeof(V) == 4) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, v), (uint32_t*)p);
	else __builtin_nontemporal_store(__builtin_bit_cast(uint16_t, v), (uint16_t*)p);
}

// LOSS 1: L2, 2: RelativeL2
// DIAG (timing-only builds, TCNN_AMD_MLP_DIAG; results are wrong; bit 5: no weight-gradient products): bit 0: weight fragments are not read from LDS, bit 1: the transposing
// reads are not done, bit 2: the image writes are not done, bit 3: the loss is not evaluated; bit 4: ReLU' as min / sub / and;
// bit 6: clocks per region of the trip (waves 0 and 4) into a.dbg (results are right with bits 4 and 6)
//
// The trip is written as a software pipeline of regions closed by sched_barrier(0): a region holds the matrix instructions of one
// step of the chain together with the vector / LDS work of the step before (conversion, activation, image writes of the tile the
// previous instructions finished) and the LDS reads of the operands of the region after, so that in program order vector
// instructions sit between matrix instructions (a wave issues in order: behind two back-to-back matrix instructions nothing of it
// issues until the second one has the pipe) and every LDS operand is requested one region (64 .. 256 clocks) before its use.
#define R32_SB() __builtin_amdgcn_sched_barrier(0)
#define R32_STAMP(i) do { if constexpr ((DIAG & 64) != 0) { const unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - ph_prev; ph_prev = now_; } } while (0)
constexpr int R32_REGIONS = 15;
// within a region: one matrix instruction, then V vector and D LDS instructions (sched_group_barrier masks: 0x8 MFMA, 0x2 VALU, 0x80 DS)
#define R32_MVD(V, D) do { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, V, 0); __builtin_amdgcn_sched_group_barrier(0x80, D, 0); } while (0)
// REC: dL/d(encoded input) leaves as 16-byte scatter records {x, y, gradients of a level pair} (the bit-plane scatter, k_grid_scatter.hip);
// else as plain level planes half2 [16][n] -- half the bytes -- for the list-fed scatter, whose elements know entries and weights (k_grid_scatter_lists.hip)
template <int LOSS, bool REC, int DIAG = 0>
__global__ void __launch_bounds__(R32_NW * 64, 2) k_mlp_train_r32(const R32Args a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 31, h = lane >> 5;
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();
	if constexpr ((DIAG & 64) == 0) { // the device-wide 100 MHz clock (the cycle counter is per XCD), in the slots of the region stamps
		if (a.dbg && tid == 0) a.dbg[(size_t)gridDim.x * (4 + R32_NW) + (size_t)blockIdx.x * 2 * R32_REGIONS] = __builtin_amdgcn_s_memrealtime();
	}

	const uint32_t n_blocks = a.n / 32;
	const uint32_t first = blockIdx.x * R32_NW + wave, step = gridDim.x * R32_NW;
	const uint32_t n_total = a.n * a.dims; // loss normalisation (relative_l2.h:58)
	const LossScales lsc = loss_scales(n_total, a.loss_scale);
	const uint32_t n4 = a.n * 4;

	// ---- global addressing: raw buffer accesses -- descriptor (scalar, per matrix) + a lane offset that never changes + a scalar offset of the
	// trip's 32-sample block: the trip loop spends one scalar addition per stream on addresses and no vector instruction (as plain
	// pointers the compiler carried 64-bit vector addresses: ~40 vector and ~50 scalar instructions per trip).  Offsets are 32-bit
	// (mlp_train_r32_applies caps n); an access beyond a matrix would be dropped by the range check instead of faulting.
	// STORES of 16 bytes take the block's offset in the vector offset (one addition) and no scalar offset: with a scalar-register
	// offset the compiler assumes the hardware needs no wait state between such a store and a vector instruction that overwrites
	// its data registers (GCNHazardRecognizer: "only if the instruction is not using a register in the soffset field"); on gfx950 it
	// does -- measured: the third dword of two of a trip's four records came out as whatever the next conversion wrote there.
	const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(a.n * 64), 0x00020000);
	const auto rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)a.target, 0, (int)(a.n * a.dims * 4), 0x00020000);
	const auto rs_xs = __builtin_amdgcn_make_buffer_rsrc((void*)a.rec_x, 0, (int)(a.n * 8), 0x00020000);
	const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(a.n * 32), 0x00020000);
	const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)a.dL_dout, 0, (int)(a.n * a.dims * 2), 0x00020000);
	const auto rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)a.L, 0, (int)(a.n * a.dims * 4), 0x00020000);
	const auto rs_rec = __builtin_amdgcn_make_buffer_rsrc((void*)a.rec, 0, (int)(a.n * (REC ? 128 : 64)), 0x00020000);
	const uint32_t x_off = (4 * h * a.n + c) * 4; // levels 8 s + 4 h + i at + (8 s + i) n 4
	struct In { h8 x[2]; float t[2]; float2 xs; };
	uint32_t t_off[2];
#pragma unroll
	for (int r = 0; r < 2; ++r) t_off[r] = (c * a.dims + min(2 * r + h, a.dims - 1)) * 4; // outputs >= dims re-read the last one (masked where used)
	auto load_in = [&](const uint32_t blk_) -> In {
		// scalar for the compiler too: it is the scalar offset of every access below (left to the compiler the trip's block number lived in a
		// vector register and every one of these loads and stores sat in a waterfall loop of its own -- v_readfirstlane, compare, branch: 22 per trip)
		const uint32_t blk = __builtin_amdgcn_readfirstlane(blk_);
		In r;
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			u32x4 v;
#pragma unroll
			for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, x_off, blk * 128 + n4 * (8 * s + i), 0); // features 16 s + 8 h + 2 i, + 1
			r.x[s] = __builtin_bit_cast(h8, v);
		}
		const uint32_t tb = blk * (128 * a.dims);
		r.t[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, t_off[0], tb, 0));
		r.t[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, t_off[1], tb, 0));
		if constexpr (REC) r.xs = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_xs, c * 8, blk * 256, 0));
		else r.xs = float2{0, 0};
		return r;
	};
	In pre{};
	if (first < n_blocks) pre = load_in(first);

	// ---- weight fragments into LDS
	{
		constexpr uint32_t N16 = R32_NF * 64;
		constexpr int FILL = (N16 + R32_NW * 64 - 1) / (R32_NW * 64);
		h8 tmp[FILL];
#pragma unroll
		for (int k = 0; k < FILL; ++k) tmp[k] = a.image[min(tid + k * R32_NW * 64, N16 - 1)];
#pragma unroll
		for (int k = 0; k < FILL; ++k) {
			if (tid + k * R32_NW * 64 < N16) ((h8*)smem)[tid + k * R32_NW * 64] = tmp[k];
		}
	}
	__syncthreads();
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();

	// ---- LDS addressing.  Weight fragment f: f KiB + 16 lane.  An opaque per-trip copy of the lane offset keeps the (loop-invariant)
	// fragment reads inside the trip loop (hoisted they would need 120 registers).
	uint32_t lane16 = lane * 16;
	h8 fake = __builtin_bit_cast(h8, u32x4{lane, lane16, tid, wave});
	auto frag = [&](const int f) -> h8 {
		if constexpr (DIAG & 1) { asm volatile("" : "+v"(fake)); return fake; }
		return *(const h8*)(smem + lane16 + f * 1024);
	};
	// Images, per wave and 32-feature tile (2 KiB): plane g (4 features) at 256 g, sample n inside it at 8 ((n + 4 g) & 31) -- the
	// rotation makes the transposing reads (4 samples x 8 planes per 32 lanes) and the plane-wise writes both conflict-free.
	//   writes: this lane holds, per k-step s and half e of a chain fragment, the 4 features of plane 4 s + 2 e + h of sample c
	//   (of the natural-order input fragment: plane 4 s + 2 h + e).
	const uint32_t wbase = R32_WAVE0 + wave * R32_WAVE_BYTES;
	uint32_t w_chain[4], w_nat[4]; // [2 s + e]
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t gc = 4 * (k >> 1) + 2 * (k & 1) + h, gn = 4 * (k >> 1) + 2 * h + (k & 1);
		w_chain[k] = wbase + gc * 256 + ((c + 4 * gc) & 31) * 8;
		w_nat[k] = wbase + gn * 256 + ((c + 4 * gn) & 31) * 8;
	}
	//   transposing reads (ds_read_b64_tr_b16: per 16 lanes a block of 4 rows x 16 columns; lane 4 q + p supplies the address of row q,
	//   columns 4 p .. 4 p + 3, and receives column (lane & 15) of the 4 rows)
	//   32x32x16 operand of sample k-step s': element j = sample 16 s' + 8 hh + j of feature (lane & 31), hh = lane >> 5: reads e = 0, 1 of 4 samples
	//   16x16x32 operand (all 32 samples): element j = sample 8 (lane >> 4) + j of feature 16 half + (lane & 15)
	uint32_t r_tr[4], r_16[4]; // [2 s' + e], [2 half + e]
	{
		const uint32_t grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, hh = grp >> 1;
		const uint32_t g = 4 * (grp & 1) + p;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = 16 * (k >> 1) + 8 * hh + 4 * (k & 1) + q;
			r_tr[k] = wbase + g * 256 + ((row + 4 * g) & 31) * 8;
			const uint32_t g16 = 4 * (k >> 1) + p, row16 = 8 * grp + 4 * (k & 1) + q;
			r_16[k] = wbase + g16 * 256 + ((row16 + 4 * g16) & 31) * 8;
		}
	}
	auto img_write = [&](const int img, const uint32_t (&w)[4], const int s, const h8 v) {
		if constexpr (DIAG & 4) { asm volatile("" :: "v"(v)); return; }
		*(h4*)(smem + w[2 * s + 0] + img) = h4{v[0], v[1], v[2], v[3]};
		*(h4*)(smem + w[2 * s + 1] + img) = h4{v[4], v[5], v[6], v[7]};
	};
	auto img_own = [&](const int img, const int s) -> h8 { // this lane's own chain fragment back from an image
		if constexpr (DIAG & 2) { asm volatile("" : "+v"(fake)); return fake; }
		const h4 lo = *(const h4*)(smem + w_chain[2 * s + 0] + img), hi = *(const h4*)(smem + w_chain[2 * s + 1] + img);
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr2 = [&](const uint32_t a0, const uint32_t a1) -> h8 {
		if constexpr (DIAG & 2) { asm volatile("" : "+v"(fake)); return fake; }
		const h4 lo = lds_read_tr((const half_t*)(smem + a0)), hi = lds_read_tr((const half_t*)(smem + a1));
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr_frag = [&](const int img, const int sp) -> h8 { return tr2(r_tr[2 * sp] + img, r_tr[2 * sp + 1] + img); };   // 32x32x16 operand
	auto tr_frag16 = [&](const int img, const int half) -> h8 { return tr2(r_16[2 * half] + img, r_16[2 * half + 1] + img); }; // 16x16x32 operand

	// ---- lane offsets of the stores
	const uint32_t cg_off0 = (c * a.dims + h) * 2;              // compact dL_dout: output 2 r + h (+ 4 r bytes); compact L: twice that
	const uint32_t o_off = c * 32 + h * 16;                     // out [n][16] halves: this lane stores the row's half h (16 bytes)
	const uint32_t rec_off = REC ? (h * a.n + c) * 16            // records: level pair 2 g + h at + g 2 n 16
	                             : (2 * h * a.n + c) * 4;        // planes: level 4 g + 2 h + j at + (4 g + j) n 4

	// weight-gradient accumulators: dW0 row tiles 0, 1; dW1 tiles (tr, tc); dWout as four 16 x 16 tiles (positions x 16 hidden features)
	f16v wacc[6];
	f4 wout[4];
#pragma unroll
	for (int i = 0; i < 6; ++i) wacc[i] = zero16();
#pragma unroll
	for (int i = 0; i < 4; ++i) wout[i] = f4{0, 0, 0, 0};

	// fragment slots (R32Frags of this network)
	constexpr int F0 = 0, F1 = 4, FO = 12, BO = 16, B1 = 18, B0 = 26;
	const f16v Z = zero16();

	unsigned long long ph[R32_REGIONS] = {}, ph_prev = 0;
	h8 f0[4]; // layer 0's fragments: requested at the end of the trip before
#pragma unroll
	for (int i = 0; i < 4; ++i) f0[i] = frag(F0 + i);

	// The two waves of a SIMD (waves w and w + 4) are arbitrated oldest first; left alone the older one takes every contended issue slot and
	// ends its trips long before its partner (k_train_regs.hip).  prio_mode 1 alternates their priorities per trip.
	if (a.prio_mode == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
	if (wave >= 4) { // wave-uniform
		for (uint32_t i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
	}
	uint32_t prio_phase = wave >= 4 ? 1u : 0u;
	for (uint32_t blk_v = first; blk_v < n_blocks; blk_v += step) {
		const uint32_t blk = __builtin_amdgcn_readfirstlane(blk_v); // (see load_in)
		if (a.prio_mode == 1) { // wave-uniform
			if (prio_phase & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
			prio_phase ^= 1u;
		}
		if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(1); // the regions with matrix instructions above the loss (vector only)
		asm volatile("" : "+v"(lane16));
		if constexpr ((DIAG & 64) != 0) ph_prev = __builtin_readcyclecounter();
		const In in = pre;
		// ------------------------------------------------------------------------------------------------ forward
		{ // unconditionally (the last trip re-reads its own block): no branch between a trip's loads and its stores, so that the
		  // compiler's wait for these loads at the top of the next trip is a counted one that leaves the stores in flight
			const uint32_t next = min(blk + step, n_blocks - 1);
			pre = load_in(next);
		}
		img_write(IMG_X, w_nat, 0, in.x[0]);
		img_write(IMG_X, w_nat, 1, in.x[1]);
		f16v a0 = mfma32(f0[0], in.x[0], Z);
		a0 = mfma32(f0[1], in.x[1], a0);
		h8 wa0 = frag(F1 + 0), wa1 = frag(F1 + 1);
		R32_SB(); R32_STAMP(0);
		f16v a1 = mfma32(f0[2], in.x[0], Z);
		a1 = mfma32(f0[3], in.x[1], a1);
		const h8 h00 = relu8(pack8(a0, 0)), h01 = relu8(pack8(a0, 1));
		img_write(IMG_H0, w_chain, 0, h00);
		img_write(IMG_H0, w_chain, 1, h01);
		h8 wa2 = frag(F1 + 4), wa3 = frag(F1 + 5);
		R32_MVD(8, 3); R32_MVD(8, 3);
		R32_SB(); R32_STAMP(1);
		f16v b0 = mfma32(wa0, h00, Z);
		b0 = mfma32(wa1, h01, b0);
		f16v b1 = mfma32(wa2, h00, Z);
		b1 = mfma32(wa3, h01, b1);
		const h8 h02 = relu8(pack8(a1, 0)), h03 = relu8(pack8(a1, 1));
		img_write(IMG_H0 + 2048, w_chain, 0, h02);
		img_write(IMG_H0 + 2048, w_chain, 1, h03);
		h8 wb0 = frag(F1 + 2), wb1 = frag(F1 + 3), wb2 = frag(F1 + 6), wb3 = frag(F1 + 7);
		R32_MVD(4, 2); R32_MVD(4, 2); R32_MVD(4, 2); R32_MVD(4, 2);
		R32_SB(); R32_STAMP(2);
		b0 = mfma32(wb0, h02, b0);
		b0 = mfma32(wb1, h03, b0);
		wa0 = frag(FO + 0); wa1 = frag(FO + 1);
		R32_SB(); R32_STAMP(3);
		b1 = mfma32(wb2, h02, b1);
		b1 = mfma32(wb3, h03, b1);
		const h8 h10 = relu8(pack8(b0, 0)), h11 = relu8(pack8(b0, 1));
		img_write(IMG_H1, w_chain, 0, h10);
		img_write(IMG_H1, w_chain, 1, h11);
		wa2 = frag(FO + 2); wa3 = frag(FO + 3);
		R32_MVD(8, 3); R32_MVD(8, 3);
		R32_SB(); R32_STAMP(4);
		f16v o = mfma32(wa0, h10, Z);
		o = mfma32(wa1, h11, o);
		const h8 h12 = relu8(pack8(b1, 0)), h13 = relu8(pack8(b1, 1));
		img_write(IMG_H1 + 2048, w_chain, 0, h12);
		img_write(IMG_H1 + 2048, w_chain, 1, h13);
		wb0 = frag(BO + 0); wb1 = frag(BO + 1);
		R32_MVD(8, 3); R32_MVD(8, 3);
		R32_SB(); R32_STAMP(5);
		o = mfma32(wa2, h12, o);
		o = mfma32(wa3, h13, o);
		// operands of dWout = dY H1^T: H1^T, 16 features per fragment
		h8 bH[4];
#pragma unroll
		for (int tc = 0; tc < 4; ++tc) bH[tc] = tr_frag16(IMG_H1 + 2048 * (tc >> 1), tc & 1);
		R32_SB(); R32_STAMP(6);

		// ------------------------------------------------------------------------------------------------ loss on the result tile
		if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(0);
		const h8 ov = pack8(o, 0); // element g: output 2 g + h (output activation None)
		h8 dyf = h8{0, 0, 0, 0, 0, 0, 0, 0}; // dL/doutput, the B fragment of the first backward product (k = position)
		{
			// l2.h:40-74 / relative_l2.h:40-75 on one refined reciprocal (loss_l2_fused, mlp_device.h), the two output slots of a lane side by
			// side; values and gradients of the live outputs go to the compact context matrices [n][dims]
			float value[2];
			half_t grad[2];
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const float prediction = (float)ov[r];
				if constexpr (DIAG & 8) {
					value[r] = prediction - in.t[r];
					grad[r] = (half_t)(a.loss_scale * prediction / n_total);
				} else loss_l2_fused<LOSS == 2>(prediction, in.t[r], lsc, value[r], grad[r]);
			}
			asm volatile("" : "+v"(value[0]), "+v"(value[1])); // both chains are evaluated here, in one block, not inside the masked stores below
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const bool live = 2 * r + h < a.dims;
				dyf[r] = live ? grad[r] : (half_t)0.0f;
				if (live) {
					__builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, grad[r]), rs_g, cg_off0 + 4 * r, blk * (64 * a.dims), 2);
					__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, value[r]), rs_l, 2 * cg_off0 + 8 * r, blk * (128 * a.dims), 2);
				}
			}
		}
		{ // out [n][16]: words (2 g, 2 g + 1) of this lane and of its partner lane (the other half) interleave into the row
			const u32x4 u = __builtin_bit_cast(u32x4, ov); // u[k] = outputs (4 k + h, 4 k + 2 + h)
			uint32_t w[4];
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				// lanes < 32 keep u[k] and receive the partner's u[k]; lanes >= 32 receive the partner's u[k + 2] and keep their own
				const auto sw = __builtin_amdgcn_permlane32_swap(u[k], u[k + 2], false, false);
				const uint32_t even = sw[0], odd = sw[1]; // outputs (4 k' + 0, 4 k' + 2) and (4 k' + 1, 4 k' + 3), k' = k + 2 h
				w[2 * k + 0] = __builtin_amdgcn_perm(odd, even, 0x05040100u); // (even.lo, odd.lo)
				w[2 * k + 1] = __builtin_amdgcn_perm(odd, even, 0x07060302u); // (even.hi, odd.hi)
			}
			__builtin_amdgcn_raw_buffer_store_b128(u32x4{w[0], w[1], w[2], w[3]}, rs_out, o_off + blk * 1024, 0, R32_OUT_AUX);
		}
		img_write(IMG_DY, w_chain, 0, dyf); // positions 4 h .. 4 h + 3 (plane h) and 8 + 4 h .. (plane 2 + h)
		const h8 aY = tr_frag16(IMG_DY, 0);  // dY^T: 16 positions x 32 samples
		if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(1);
		R32_SB(); R32_STAMP(7);

		// ------------------------------------------------------------------------------------------------ backward chain and weight gradients
		f16v g0 = mfma32(wb0, dyf, Z);
		f16v g1 = mfma32(wb1, dyf, Z);
		asm volatile("" : "+v"(g1)); // here, not sunk to its use two regions further
		wa0 = frag(B1 + 0); wa1 = frag(B1 + 1); wa2 = frag(B1 + 4); wa3 = frag(B1 + 5);
		R32_SB(); R32_STAMP(8);
		// dWout (rows = positions, 16 hidden features per tile) beside the first tile of dH1 = (Wout^T dY) act'(H1) (common_device.h:241-297: from the forward OUTPUT)
#pragma unroll
		for (int tc = 0; tc < 4; ++tc) if constexpr ((DIAG & 32) == 0) wout[tc] = mfma(aY, bH[tc], wout[tc]);
		const h8 d10 = relu_bwd8<(DIAG & 16) != 0>(pack8(g0, 0), h10), d11 = relu_bwd8<(DIAG & 16) != 0>(pack8(g0, 1), h11);
		img_write(IMG_DH, w_chain, 0, d10);
		img_write(IMG_DH, w_chain, 1, d11);
		R32_SB(); R32_STAMP(9);
		f16v e0 = mfma32(wa0, d10, Z);
		e0 = mfma32(wa1, d11, e0);
		f16v e1 = mfma32(wa2, d10, Z);
		e1 = mfma32(wa3, d11, e1);
		const h8 d12 = relu_bwd8<(DIAG & 16) != 0>(pack8(g1, 0), h12), d13 = relu_bwd8<(DIAG & 16) != 0>(pack8(g1, 1), h13);
		img_write(IMG_DH + 2048, w_chain, 0, d12);
		img_write(IMG_DH + 2048, w_chain, 1, d13);
		wb0 = frag(B1 + 2); wb1 = frag(B1 + 3); wb2 = frag(B1 + 6); wb3 = frag(B1 + 7);
		R32_SB(); R32_STAMP(10);
		e0 = mfma32(wb0, d12, e0);
		e0 = mfma32(wb1, d13, e0);
		e1 = mfma32(wb2, d12, e1);
		e1 = mfma32(wb3, d13, e1);
		// operands of dW1 = dH1 H0^T
		h8 tA[4], tB[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			tA[k] = tr_frag(IMG_DH + 2048 * (k >> 1), k & 1);
			tB[k] = tr_frag(IMG_H0 + 2048 * (k >> 1), k & 1);
		}
		R32_SB(); R32_STAMP(11);
#pragma unroll
		for (int tr = 0; tr < 2; ++tr)
#pragma unroll
			for (int tc = 0; tc < 2; ++tc) {
				if constexpr ((DIAG & 32) == 0) {
				wacc[2 + 2 * tr + tc] = mfma32(tA[2 * tr + 0], tB[2 * tc + 0], wacc[2 + 2 * tr + tc]);
				wacc[2 + 2 * tr + tc] = mfma32(tA[2 * tr + 1], tB[2 * tc + 1], wacc[2 + 2 * tr + tc]);
				}
			}
		const h8 d00 = relu_bwd8<(DIAG & 16) != 0>(pack8(e0, 0), img_own(IMG_H0, 0)), d01 = relu_bwd8<(DIAG & 16) != 0>(pack8(e0, 1), img_own(IMG_H0, 1));
		const h8 d02 = relu_bwd8<(DIAG & 16) != 0>(pack8(e1, 0), img_own(IMG_H0 + 2048, 0)), d03 = relu_bwd8<(DIAG & 16) != 0>(pack8(e1, 1), img_own(IMG_H0 + 2048, 1));
		img_write(IMG_DH, w_chain, 0, d00); // behind the reads of dH1 above: LDS operations of a wave execute in order
		img_write(IMG_DH, w_chain, 1, d01);
		img_write(IMG_DH + 2048, w_chain, 0, d02);
		img_write(IMG_DH + 2048, w_chain, 1, d03);
		wa0 = frag(B0 + 0); wa1 = frag(B0 + 1); wa2 = frag(B0 + 2); wa3 = frag(B0 + 3);
		R32_SB(); R32_STAMP(12);
		// dX = W0^T dH0
		f16v dx = mfma32(wa0, d00, Z);
		dx = mfma32(wa1, d01, dx);
		dx = mfma32(wa2, d02, dx);
		dx = mfma32(wa3, d03, dx);
		// operands of dW0 = dH0 X^T, and layer 0's fragments for the next trip
#pragma unroll
		for (int k = 0; k < 4; ++k) tA[k] = tr_frag(IMG_DH + 2048 * (k >> 1), k & 1);
		tB[0] = tr_frag(IMG_X, 0);
		tB[1] = tr_frag(IMG_X, 1);
#pragma unroll
		for (int i = 0; i < 4; ++i) f0[i] = frag(F0 + i);
		R32_SB(); R32_STAMP(13);
#pragma unroll
		for (int tr = 0; tr < 2; ++tr) {
			if constexpr ((DIAG & 32) == 0) {
			wacc[tr] = mfma32(tA[2 * tr + 0], tB[0], wacc[tr]);
			wacc[tr] = mfma32(tA[2 * tr + 1], tB[1], wacc[tr]);
			}
		}
		// scatter records {x, y, gradients of levels 2 p, 2 p + 1}: registers 4 g .. 4 g + 3 are features 8 g + 4 h .. + 3, i.e. level pair p = 2 g + h
		if constexpr (REC) {
			const u32x4 lo = __builtin_bit_cast(u32x4, pack8(dx, 0)), hi = __builtin_bit_cast(u32x4, pack8(dx, 1));
			const uint32_t x0 = __builtin_bit_cast(uint32_t, in.xs.x), x1 = __builtin_bit_cast(uint32_t, in.xs.y);
			const uint32_t rb = blk * 512, pair2 = a.n * 32; // pair2: two level pairs further
			__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, lo[0], lo[1]}, rs_rec, rec_off + rb, 0, R32_REC_AUX);
			__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, lo[2], lo[3]}, rs_rec, rec_off + (rb + pair2), 0, R32_REC_AUX);
			__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, hi[0], hi[1]}, rs_rec, rec_off + (rb + 2 * pair2), 0, R32_REC_AUX);
			__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, hi[2], hi[3]}, rs_rec, rec_off + (rb + 3 * pair2), 0, R32_REC_AUX);
		} else { // level planes: word 2 g + j of the tile's 8 is level 4 g + 2 h + j of sample c -- eight dense 128-byte runs per half wave
			const u32x4 lo = __builtin_bit_cast(u32x4, pack8(dx, 0)), hi = __builtin_bit_cast(u32x4, pack8(dx, 1));
			const uint32_t w[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
			for (int k = 0; k < 8; ++k) __builtin_amdgcn_raw_buffer_store_b32(w[k], rs_rec, rec_off, blk * 128 + n4 * (4 * (k >> 1) + (k & 1)), R32_REC_AUX);
		}
		R32_SB(); R32_STAMP(14);
	}
	if constexpr ((DIAG & 64) != 0) {
		if (a.dbg && lane == 0 && (wave == 0 || wave == 4)) {
			for (int i = 0; i < R32_REGIONS; ++i) a.dbg[(size_t)gridDim.x * (4 + R32_NW) + ((size_t)blockIdx.x * 2 + (wave >> 2)) * R32_REGIONS + i] = ph[i];
		}
	}
	if (a.dbg && lane == 0) a.dbg[gridDim.x * 4 + blockIdx.x * R32_NW + wave] = __builtin_readcyclecounter();
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

	// ---- weight gradients: the 8 waves' accumulators -> the workgroup's slab, through LDS (the images and fragments are dead: barrier
	// first), in two passes of 14 register quads: every wave dumps its quads, then wave w adds up the eight copies of the quads
	// q = w (mod 8) in a fixed tree (bitwise reproducible) and stores them.  All eight waves write and read in every phase.
	constexpr int NQ = 6 * 4 + 4, HALF = NQ / 2; // register quads per wave: 6 tiles of 16 registers, 4 tiles of 4
	auto quad = [&](const int qi) -> f4 {
		if (qi < 24) { const int i = qi >> 2, qd = qi & 3; return f4{wacc[i][4 * qd], wacc[i][4 * qd + 1], wacc[i][4 * qd + 2], wacc[i][4 * qd + 3]}; }
		return wout[qi - 24];
	};
	f4* region = (f4*)smem; // [8 waves][14 quads][64 lanes]
	float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
	const uint32_t grp = lane >> 4, li = lane & 15;
	__syncthreads();
#pragma unroll
	for (int pass = 0; pass < 2; ++pass) {
		f4* dst = region + (size_t)wave * (HALF * 64) + lane;
#pragma unroll
		for (int q = 0; q < HALF; ++q) dst[q * 64] = quad(pass * HALF + q);
		__syncthreads();
#pragma unroll
		for (int q = 0; q < HALF; ++q) {
			if ((uint32_t)(q & 7) != wave) continue; // wave-uniform
			const int qi = pass * HALF + q;
			const f4* src = region + (size_t)q * 64 + lane;
			f4 r[8];
#pragma unroll
			for (int w = 0; w < 8; ++w) r[w] = src[(size_t)w * (HALF * 64)];
			f4 sum;
#pragma unroll
			for (int e = 0; e < 4; ++e) sum[e] = ((r[0][e] + r[1][e]) + (r[2][e] + r[3][e])) + ((r[4][e] + r[5][e]) + (r[6][e] + r[7][e]));
			if (qi < 24) { // 32 x 32 tile i, registers 4 qd .. 4 qd + 3: rows (e) + 8 qd + 4 h of the tile, column c
				const int i = qi >> 2, qd = qi & 3;
				const uint32_t w_off = i < 2 ? a.w_off[0] : a.w_off[1], cols = i < 2 ? 32u : 64u;
				const uint32_t tr = i < 2 ? i : (i - 2) >> 1, tc = i < 2 ? 0u : (i - 2) & 1u;
#pragma unroll
				for (int e = 0; e < 4; ++e) slab[w_off + (32 * tr + e + 8 * qd + 4 * h) * cols + 32 * tc + c] = sum[e];
			} else { // dWout, 16 x 16 tile tc: register e of lane group grp is position 4 grp + e = output 2 e + 8 (grp >> 1) + (grp & 1), column 16 tc + li
				const int tc = qi - 24;
#pragma unroll
				for (int e = 0; e < 4; ++e) slab[a.w_off[2] + (2 * e + 8 * (grp >> 1) + (grp & 1)) * 64 + 16 * tc + li] = sum[e];
			}
		}
		if (pass == 0) __syncthreads(); // before the second pass overwrites the copies
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
	if constexpr ((DIAG & 64) == 0) {
		if (a.dbg && tid == 0) a.dbg[(size_t)gridDim.x * (4 + R32_NW) + (size_t)blockIdx.x * 2 * R32_REGIONS + 1] = __builtin_amdgcn_s_memrealtime();
	}
}
#undef R32_SB
#undef R32_STAMP
#undef R32_MVD

} // namespace

// TCNN_AMD_MLP_R32=0 keeps k_train_regs.hip's kernel (A/B runs; Switches, read once per model)
static bool r32_enabled() { return switches().mlp_r32; }

// the one shape this kernel is instantiated for, with the formats of the grid encoding's training step: input as level planes of 2
// features, <= 4 outputs, ReLU, L2 / RelativeL2, `out` and 2-D scatter records written, no data_pdf
bool mlp_train_r32_applies(const MlpDesc& d, uint32_t n, uint32_t x_plane_features, const float* data_pdf, const void* external_dL_dy, uint32_t dims, LossType loss, const void* out,
                           const void* dL_dx, uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims) {
	if (!r32_enabled() || !r32_shape_ok(d) || d.in_width != 32 || d.n_hidden != 2 || d.n_frags_r32 != (uint32_t)R32_NF) return false;
	if (d.activation != (uint32_t)Activation::ReLU || d.output_activation != (uint32_t)Activation::None) return false;
	if (n == 0 || n % 32 != 0 || n > (1u << 22)) return false; // 32-bit byte offsets into [n][...] matrices
	return x_plane_features == 2 && data_pdf == nullptr && external_dL_dy == nullptr && dims >= 1 && dims <= 4 && (loss == LossType::L2 || loss == LossType::RelativeL2) && out != nullptr &&
	       dL_dx != nullptr && dx_plane_features == 2 && (dx_record_x == nullptr || dx_record_dims == 2); // records {x, y, gradients} or plain level planes
}

// Which of the two kernels: k_mlp_train_r32a (k_train_r32a.hip: weights in registers, weight-gradient tiles shared out over a workgroup's
// waves, no final sum over waves) is the faster one while that final sum is a large part of the launch -- up to 2 trips per wave,
// 131 072 samples (MLP kernel at 2^14 / 2^16 / 2^17 samples: 7.9 / 12.0 / 16.0 us against 10.1 / 12.6 / 16.7); at 2^18 the two are equal
// (24.7 us) and k_mlp_train_r32 stays.  TCNN_AMD_MLP_R32A=1 / 0 forces one (A/B runs, tests).
static bool r32a_chosen(uint32_t n) {
	const int forced = switches().mlp_r32a;
	if (forced >= 0) return forced == 1;
	return n <= 131072u;
}
// workgroups = weight-gradient slabs of the kernel mlp_train_r32 launches for this batch
uint32_t mlp_train_r32_grid(uint32_t n) {
	if (r32a_chosen(n)) {
		uint32_t cap = 512u;
#ifdef TCNN_AMD_DEV
		static const uint32_t dev_cap = getenv("TCNN_AMD_MLP_GRID") ? (uint32_t)std::max(1, atoi(getenv("TCNN_AMD_MLP_GRID"))) : 512u; // laboratory knob
		cap = dev_cap;
#endif
		return std::max(1u, std::min(cap, div_round_up(n / 32, (uint32_t)R32A_NW))); // two workgroups of four waves per CU
	}
	return std::max(1u, std::min(256u, div_round_up(n / 32, (uint32_t)R32_NW)));
}

void mlp_train_r32(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, const float* target, uint32_t dims, LossType loss, float loss_scale, void* out,
                   void* compact_dL_dout, float* compact_L, void* dL_dx, const float* dx_record_x, float* slabs, uint32_t n_params, uint32_t grid) {
	CHECK_THROW(slabs != nullptr && compact_dL_dout != nullptr && compact_L != nullptr && target != nullptr);
	R32Args a{};
	a.x = (const half_t*)x;
	a.target = target;
	a.out = (half_t*)out;
	a.dL_dout = (half_t*)compact_dL_dout;
	a.L = compact_L;
	a.rec = (u32x4*)dL_dx;
	a.rec_x = dx_record_x;
	a.slabs = slabs;
	a.image = (const h8*)((const char*)image + (size_t)(d.n_frags_fwd + d.n_frags_bwd) * 1024);
	a.n = n;
	a.dims = dims;
	a.n_params = n_params;
	for (int l = 0; l < 3; ++l) a.w_off[l] = d.layers[l].w_off;
	a.loss_scale = loss_scale;
	auto go = [&](auto kernel) {
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R32_LDS_BYTES));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(R32_NW * 64), R32_LDS_BYTES, stream, a);
		HIP_CHECK_THROW(hipGetLastError());
	};
	a.prio_mode = switches().mlp_prio;
	CHECK_THROW(grid == mlp_train_r32_grid(n));
#ifndef TCNN_AMD_DEV
	if (r32a_chosen(n)) return mlp_train_r32a_launch(stream, a, grid, loss == LossType::L2 ? 1 : 2);
	if (loss == LossType::L2) { if (a.rec_x) go(k_mlp_train_r32<1, true>); else go(k_mlp_train_r32<1, false>); }
	else { if (a.rec_x) go(k_mlp_train_r32<2, true>); else go(k_mlp_train_r32<2, false>); }
#else
	// ---- laboratory build (build.py --dev): in-kernel clocks (TCNN_AMD_MLP_TIMING), timing-only kernel variants (TCNN_AMD_MLP_DIAG), start
	// delays (TCNN_AMD_MLP_STAGGER), workgroup placement (TCNN_AMD_MLP_WHERE).  None of this is in the product library.
	static const bool timing = getenv("TCNN_AMD_MLP_TIMING") != nullptr;
	static int timing_left = 5;
	if (timing && timing_left > 0) {
		HIP_CHECK_THROW(hipMalloc(&a.dbg, (size_t)grid * 1024));
		HIP_CHECK_THROW(hipMemset(a.dbg, 0, (size_t)grid * 1024));
	}
	if (const char* e = getenv("TCNN_AMD_MLP_STAGGER")) a.stagger = (uint32_t)atoi(e);
	static const int diag_env = getenv("TCNN_AMD_MLP_DIAG") ? atoi(getenv("TCNN_AMD_MLP_DIAG")) : 0;
	const int diag = a.rec_x ? diag_env : 0; // (the timing-only variants exist in the record form)
	if (r32a_chosen(n)) {
		mlp_train_r32a_launch(stream, a, grid, loss == LossType::L2 ? 1 : 2);
		if (a.dbg) {
			std::vector<unsigned long long> hst((size_t)grid * (7 + 4 * R32A_NW));
			HIP_CHECK_THROW(hipMemcpy(hst.data(), a.dbg, hst.size() * 8, hipMemcpyDeviceToHost));
			if (--timing_left == 0) {
				double fill = 0, loop = 0, tail = 0;
				unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
				for (uint32_t g = 0; g < grid; ++g) {
					const unsigned long long t_fill = hst[(size_t)grid * (6 + 4 * R32A_NW) + g];
					fill += (double)(t_fill - hst[g * 4]);
					loop += (double)(hst[g * 4 + 2] - t_fill);
					tail += (double)(hst[g * 4 + 3] - hst[g * 4 + 2]);
					const unsigned long long st = hst[(size_t)grid * 4 + g * 2], en = hst[(size_t)grid * 4 + g * 2 + 1];
					s0 = std::min(s0, st); s1 = std::max(s1, st);
					e0 = std::min(e0, en); e1 = std::max(e1, en);
				}
				fprintf(stderr, "k_mlp_train_r32a wave 0 clocks, mean over %u workgroups: fill %.0f trips %.0f (%u blocks of 32 per wave) slab stores %.0f\n", grid, fill / grid, loop / grid,
				        div_round_up(n / 32, grid * R32A_NW), tail / grid);
				fprintf(stderr, "  workgroup starts spread over %.2f us, ends from %.2f to %.2f us after the first start\n", (s1 - s0) * 0.01, (e0 - s0) * 0.01, (e1 - s0) * 0.01);
				{ // who ends when: by half of the grid (the two workgroups of a CU) and by XCD (workgroup b runs on XCD b % 8)
					double by_half[2] = {}, by_xcd[8] = {};
					for (uint32_t g = 0; g < grid; ++g) {
						const double en = (hst[(size_t)grid * 4 + g * 2 + 1] - s0) * 0.01;
						by_half[g >= grid / 2] += en / (grid / 2);
						by_xcd[g % 8] += en / (grid / 8);
					}
					if (getenv("TCNN_AMD_MLP_WHERE")) { // one line per workgroup: XCC, SE, CU, start and end in us
						for (uint32_t g = 0; g < grid; ++g) {
							const unsigned long long hw = hst[g * 4 + 1];
							fprintf(stderr, "  wg %u xcc %llu se %llu cu %llu start %.2f end %.2f\n", g, (hw >> 32) & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hst[(size_t)grid * 4 + g * 2] - s0) * 0.01,
							        (hst[(size_t)grid * 4 + g * 2 + 1] - s0) * 0.01);
						}
					}
					fprintf(stderr, "  mean end: first half of the grid %.2f us, second half %.2f us; by XCD", by_half[0], by_half[1]);
					for (int x = 0; x < 8; ++x) fprintf(stderr, " %.2f", by_xcd[x]);
					fprintf(stderr, "\n");
				}
				for (int w = 0; w < R32A_NW; ++w) {
					constexpr int NPH = 4;
					double ph[NPH] = {};
					for (uint32_t g = 0; g < grid; ++g)
						for (int i = 0; i < NPH; ++i) ph[i] += (double)hst[(size_t)grid * 6 + ((size_t)g * R32A_NW + w) * NPH + i];
					fprintf(stderr, "  wave %d clocks summed over its trips: chain %.0f, at the barrier %.0f, weight gradients %.0f, at the barrier %.0f", w, ph[0] / grid, ph[1] / grid, ph[2] / grid, ph[3] / grid);
					for (int i = 4; i < NPH; ++i) fprintf(stderr, " %.0f", ph[i] / grid);
					fprintf(stderr, "\n");
				}
			}
			(void)hipFree(a.dbg);
		}
		return;
	}
	if (diag == 1) go(k_mlp_train_r32<2, true, 1>);
	else if (diag == 2) go(k_mlp_train_r32<2, true, 2>);
	else if (diag == 3) go(k_mlp_train_r32<2, true, 3>);
	else if (diag == 7) go(k_mlp_train_r32<2, true, 7>);
	else if (diag == 8) go(k_mlp_train_r32<2, true, 8>);
	else if (diag == 15) go(k_mlp_train_r32<2, true, 15>);
	else if (diag == 16) go(k_mlp_train_r32<2, true, 16>);
	else if (diag == 64) go(k_mlp_train_r32<2, true, 64>);
	else if (diag == 34) go(k_mlp_train_r32<2, true, 34>); // the chain wave of a role split: no weight-gradient products, no transposing reads
	else if (loss == LossType::L2) { if (a.rec_x) go(k_mlp_train_r32<1, true>); else go(k_mlp_train_r32<1, false>); }
	else { if (a.rec_x) go(k_mlp_train_r32<2, true>); else go(k_mlp_train_r32<2, false>); }
	if (a.dbg) {
		std::vector<unsigned long long> hst((size_t)grid * (4 + R32_NW + 2 * R32_REGIONS));
		HIP_CHECK_THROW(hipMemcpy(hst.data(), a.dbg, hst.size() * 8, hipMemcpyDeviceToHost));
		if (--timing_left == 0) {
			double fill = 0, loop = 0, tail = 0;
			for (uint32_t g = 0; g < grid; ++g) {
				fill += (double)(hst[g * 4 + 1] - hst[g * 4]);
				loop += (double)(hst[g * 4 + 2] - hst[g * 4 + 1]);
				tail += (double)(hst[g * 4 + 3] - hst[g * 4 + 2]);
			}
			fprintf(stderr, "k_mlp_train_r32 wave 0 clocks, mean over %u workgroups: fill %.0f trips %.0f (%u blocks of 32 per wave) tail %.0f\n", grid, fill / grid, loop / grid,
			        div_round_up(n / 32, grid * R32_NW), tail / grid);
			if (diag != 64) { // the launch as a whole, on the device-wide 100 MHz clock
				unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
				for (uint32_t g = 0; g < grid; ++g) {
					const unsigned long long st = hst[(size_t)grid * (4 + R32_NW) + (size_t)g * 2 * R32_REGIONS], en = hst[(size_t)grid * (4 + R32_NW) + (size_t)g * 2 * R32_REGIONS + 1];
					s0 = std::min(s0, st); s1 = std::max(s1, st);
					e0 = std::min(e0, en); e1 = std::max(e1, en);
				}
				fprintf(stderr, "  workgroup starts spread over %.2f us, ends from %.2f to %.2f us after the first start\n", (s1 - s0) * 0.01, (e0 - s0) * 0.01, (e1 - s0) * 0.01);
			}
			double end_w[R32_NW] = {};
			for (uint32_t g = 0; g < grid; ++g)
				for (int w = 0; w < R32_NW; ++w) end_w[w] += (double)(hst[(size_t)grid * 4 + (size_t)g * R32_NW + w] - hst[g * 4]);
			fprintf(stderr, "  loop end per wave (clocks after wave 0's start):");
			for (int w = 0; w < R32_NW; ++w) fprintf(stderr, " %.0f", end_w[w] / grid);
			fprintf(stderr, "\n");
			if (diag == 64) {
				for (int half = 0; half < 2; ++half) {
					fprintf(stderr, "  wave %d, clocks per region summed over its trips:", 4 * half);
					for (int i = 0; i < R32_REGIONS; ++i) {
						double v = 0;
						for (uint32_t g = 0; g < grid; ++g) v += (double)hst[(size_t)grid * (4 + R32_NW) + ((size_t)g * 2 + half) * R32_REGIONS + i];
						fprintf(stderr, " %.0f", v / grid);
					}
					fprintf(stderr, "\n");
				}
			}
		}
		(void)hipFree(a.dbg);
	}
#endif
}

} // namespace tcnn_amd
