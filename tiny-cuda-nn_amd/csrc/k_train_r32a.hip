// k_train_r32a.hip -- k_mlp_train_r32 (k_train_r32.hip: the training step's MLP part for 32 -> 64 -> 64 -> 16, BASELINE configs 3) with the
// network's weights in REGISTERS and the weight-gradient tiles shared out over a workgroup's waves.
//
// Why: k_mlp_train_r32 keeps per wave all 8 weight-gradient tiles (112 of its 256 registers), so its 30 weight fragments live in LDS
// and every use is a 16-byte LDS read -- 30 KiB per wave and trip on a CU whose LDS pipe the 8 waves keep
// busy -- and the eight waves' tiles are summed through LDS at the end (17 % of the kernel), after a fill phase (9 %).  Here:
//   * 4 waves per workgroup, 2 workgroups per CU (still two waves per SIMD, of DIFFERENT workgroups, each in its own phase);
//   * wave w owns dW1's tile (w >> 1, w & 1); waves 0 / 1 also dW0's row tiles 0 / 1, waves 2 / 3 two of dWout's 16-column tiles
//     each (as the upper half of a 32 x 32 tile: every wave runs the same instructions): 32 accumulator registers, FINAL for the workgroup -- stored into the slab as they stand, no sum over waves;
//   * a wave sums its tiles over the samples of all four waves: two workgroup barriers per trip (images complete / consumed), and
//     dH1 / dH0 have images of their own (19 KiB per wave, 76 KiB per workgroup);
//   * the accumulators and 24 of the 30 weight fragments sit in the ACCUMULATION registers (AGPRs: with two waves per SIMD the
//     compiler splits a wave's 256 registers 128 / 128): a matrix instruction takes its A operand from there directly, the
//     accumulating ones are inline assembly with AGPR results (build.py: -amdgpu-mfma-vgpr-form keeps every other result in the
//     ordinary registers the conversions read).  The other 6 fragments (layer 0's four, Wout^T's two) are 16-byte loads from the
//     image in global memory (L1-resident), requested a phase ahead.  (All 30 from global: the CU's L1 path -- 64 bytes per
//     clock -- is the wall, 30 KiB per wave and trip; measured 9.6 k clocks per trip against k_mlp_train_r32's 6.2 k.)
// Measured ("The 32x32x16 kernels"): what the missing fill and final sum save, the two barriers per trip and the doubled
// transposing reads give back at 4 trips per wave (24.7 us both at 2^18 samples); with fewer trips this kernel is the faster one
// (2^14 / 2^16 / 2^17 samples: 7.9 / 12.0 / 16.0 us against 10.1 / 12.6 / 16.7) and mlp_train_r32 launches it up to 131 072 samples.
// The chain's instructions and their order are k_mlp_train_r32's: outputs, context matrices and scatter records are bit-identical;
// the weight gradients differ in the order of the fp32 sum (per slab 4 x trips blocks in tile order instead of 8 waves' partial sums).
#include "r32_train.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace tcnn_amd {
namespace {

constexpr int A_IMG_X = 0, A_IMG_H0 = 2048, A_IMG_H1 = 6144, A_IMG_DH1 = 10240, A_IMG_DH0 = 14336, A_IMG_DY = 18432;
constexpr int A_WAVE_BYTES = 19 * 1024;              // X 2 K | H0 4 K | H1 4 K | dH1 4 K | dH0 4 K | dY 1 K
constexpr int A_LDS_BYTES = R32A_NW * A_WAVE_BYTES + 1024; // 78 848 (the KiB: see the second tile of waves 2, 3): two workgroups per CU

__device__ inline void mfma32_acc(f16v& acc, const h8 a, const h8 b) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }

#define A_SB() __builtin_amdgcn_sched_barrier(0)
constexpr int A_NPH = 4; // stamps (each costs ~200 clocks): [0] chain, [1] barrier, [2] weight gradients, [3] barrier
#define A_STAMP(i) do { if (a.dbg) { const unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - ph_prev; ph_prev = now_; } } while (0)
// within a region: one matrix instruction, then V vector and D LDS instructions (sched_group_barrier masks: 0x8 MFMA, 0x2 VALU, 0x80 DS)
#define A_MVD(V, D) do { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, V, 0); __builtin_amdgcn_sched_group_barrier(0x80, D, 0); } while (0)

// LOSS 1: L2, 2: RelativeL2
template <int LOSS, bool REC> // REC: scatter records out, else plain level planes (k_mlp_train_r32)
__global__ void __launch_bounds__(R32A_NW * 64, 2) k_mlp_train_r32a(const R32Args a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 31, h = lane >> 5;
	if (a.dbg && tid == 0) {
		a.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();
		a.dbg[(size_t)gridDim.x * 4 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
		a.dbg[blockIdx.x * 4 + 1] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); // XCC_ID, HW_ID: where the workgroup runs
	}

	const uint32_t n_blocks = a.n / 32;
	const uint32_t per_trip = gridDim.x * R32A_NW;
	const uint32_t n_trips = (n_blocks + per_trip - 1) / per_trip; // the same for every wave of every workgroup: the barriers are workgroup-wide
	const uint32_t n_total = a.n * a.dims;                         // loss normalisation (relative_l2.h:58)
	const LossScales lsc = loss_scales(n_total, a.loss_scale);
	const uint32_t n4 = a.n * 4;

	// ---- the weights.  Fragment slots (R32Frags of this network); resident: F1, FO, B1, B0 (24 fragments)
	constexpr int F0 = 0, F1 = 4, FO = 12, BO = 16, B1 = 18, B0 = 26;
	// The workgroup's four waves bring the 30 KiB from global memory ONCE, through LDS (the image area, not in use yet): every wave
	// loading its own copy was 4 x the bytes through the L2 -> CU path, which set the length of this phase (5.4 k clocks)
	constexpr int FILL = (30 * 64 + R32A_NW * 64 - 1) / (R32A_NW * 64);
	h8 stage[FILL];
#pragma unroll
	for (int k = 0; k < FILL; ++k) stage[k] = a.image[min(tid + k * R32A_NW * 64, 30u * 64u - 1u)];
	// the other six: one 16-byte load per lane and use.  An opaque per-trip copy of the lane keeps the (loop-invariant) loads inside the trip loop
	uint32_t lane_o = lane;
	auto frag = [&](const int f) -> h8 { return a.image[f * 64 + lane_o]; };

	// ---- global addressing as in k_mlp_train_r32: raw buffer accesses, a lane offset that never changes + the scalar offset of the
	// trip's block; 16-byte stores carry the block offset in the vector offset (the hazard described there)
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
		const uint32_t blk = __builtin_amdgcn_readfirstlane(blk_); // the scalar offsets below stay scalar (behind the trip's branch the compiler had moved them to the vector unit: a waterfall loop per load)
		In r;
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			u32x4 v;
#pragma unroll
			for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, x_off, blk * 128 + n4 * (8 * s + i), 0);
			r.x[s] = __builtin_bit_cast(h8, v);
		}
		const uint32_t tb = blk * (128 * a.dims);
		r.t[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, t_off[0], tb, 0));
		r.t[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, t_off[1], tb, 0));
		if constexpr (REC) r.xs = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_xs, c * 8, blk * 256, 0));
		else r.xs = float2{0, 0};
		return r;
	};
	const uint32_t first = blockIdx.x * R32A_NW + wave;
	In pre = load_in(min(first, n_blocks - 1));
	h8 f0[4]; // layer 0's fragments: requested at the end of the trip before
#pragma unroll
	for (int i = 0; i < 4; ++i) f0[i] = frag(F0 + i);
#pragma unroll
	for (int k = 0; k < FILL; ++k) {
		if (tid + k * R32A_NW * 64 < 30u * 64u) ((h8*)smem)[tid + k * R32A_NW * 64] = stage[k];
	}
	__syncthreads();
	h8 w[30];
#pragma unroll
	for (int f = 0; f < 30; ++f) {
		if ((f >= F0 && f < F1) || (f >= BO && f < B1)) continue;
		w[f] = ((const h8*)smem)[f * 64 + lane];
		asm volatile("" : "+a"(w[f]));
	}
	__syncthreads(); // the first trip's images overwrite the staging copy
	const unsigned long long t_fill = a.dbg ? __builtin_readcyclecounter() : 0ull;

	// ---- images (k_train_r32.hip): per wave and 32-feature tile 2 KiB, plane g (4 features) at 256 g, sample n inside it at 8 ((n + 4 g) & 31)
	const uint32_t wbase = wave * A_WAVE_BYTES;
	uint32_t w_chain[4], w_nat[4]; // [2 s + e]
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t gc = 4 * (k >> 1) + 2 * (k & 1) + h, gn = 4 * (k >> 1) + 2 * h + (k & 1);
		w_chain[k] = wbase + gc * 256 + ((c + 4 * gc) & 31) * 8;
		w_nat[k] = wbase + gn * 256 + ((c + 4 * gn) & 31) * 8;
	}
	uint32_t r_tr[4], r_16[4]; // [2 s' + e], [2 half + e]: relative to a tile of ANY wave's images
	{
		const uint32_t grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, hh = grp >> 1;
		const uint32_t g = 4 * (grp & 1) + p;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = 16 * (k >> 1) + 8 * hh + 4 * (k & 1) + q;
			r_tr[k] = g * 256 + ((row + 4 * g) & 31) * 8;
			const uint32_t g16 = 4 * (k >> 1) + p, row16 = 8 * grp + 4 * (k & 1) + q;
			r_16[k] = g16 * 256 + ((row16 + 4 * g16) & 31) * 8;
		}
	}
	auto img_write = [&](const int img, const uint32_t (&wr)[4], const int s, const h8 v) {
		*(h4*)(smem + wr[2 * s + 0] + img) = h4{v[0], v[1], v[2], v[3]};
		*(h4*)(smem + wr[2 * s + 1] + img) = h4{v[4], v[5], v[6], v[7]};
	};
	auto img_own = [&](const int img, const int s) -> h8 { // this lane's own chain fragment back from an image
		const h4 lo = *(const h4*)(smem + w_chain[2 * s + 0] + img), hi = *(const h4*)(smem + w_chain[2 * s + 1] + img);
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr2 = [&](const uint32_t a0, const uint32_t a1) -> h8 {
		const h4 lo = lds_read_tr((const half_t*)(smem + a0)), hi = lds_read_tr((const half_t*)(smem + a1));
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr_frag = [&](const uint32_t img, const int sp) -> h8 { return tr2(r_tr[2 * sp] + img, r_tr[2 * sp + 1] + img); };       // 32x32x16 operand
	auto tr_frag16 = [&](const uint32_t img, const int half) -> h8 { return tr2(r_16[2 * half] + img, r_16[2 * half + 1] + img); }; // 16x16x32 operand

	const uint32_t cg_off0 = (c * a.dims + h) * 2;
	const uint32_t o_off = c * 32 + h * 16;
	const uint32_t rec_off = REC ? (h * a.n + c) * 16 : (2 * h * a.n + c) * 4; // records: level pair 2 g + h; planes: level 4 g + 2 h + j

	// this wave's tiles: dW1 (w >> 1, w & 1), and a second one through the same instructions:
	//   waves 0, 1: dW0's row tile w = dH0 (rows 32 w ..) X^T;
	//   waves 2, 3: dWout's columns 32 (w - 2) .. = dY H1^T as a 32 x 32 tile whose rows 16 .. 31 are never stored -- the A operand
	//   reads "positions" 16 .. 31 from the KiB behind the 16-position dY image, whatever is there (a row of A only reaches the same
	//   row of the result); the last wave's KiB behind is the padding at the end of the workgroup's LDS
	f16v w1acc = zero16(), w2acc = zero16();
	const uint32_t tr1 = wave >> 1, tc1 = wave & 1u;
	const uint32_t img_a2 = wave < 2 ? A_IMG_DH0 + 2048 * wave : A_IMG_DY, img_b2 = wave < 2 ? A_IMG_X : A_IMG_H1 + 2048 * (wave - 2); // wave-uniform
	const f16v Z = zero16();

	// the CU's two workgroups in different phases from the start (TCNN_AMD_MLP_STAGGER: units of 64 clocks, the second half of the grid)
	if (blockIdx.x >= gridDim.x / 2) { // workgroup-uniform
		for (uint32_t i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
	}

	unsigned long long ph[A_NPH] = {}, ph_prev = __builtin_readcyclecounter(); // TCNN_AMD_MLP_TIMING: clocks in the chain / waiting / weight gradients / waiting, summed over the trips
	if (a.prio_mode == 2 && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(1); // TCNN_AMD_MLP_PRIO=2: the younger workgroup of a CU above the older one throughout
	if (a.prio_mode == 3 && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(3);
	for (uint32_t trip = 0; trip < n_trips; ++trip) {
		const uint32_t blk = __builtin_amdgcn_readfirstlane(trip * per_trip + first); // scalar for the compiler too: it is the scalar offset of every global access of the trip
		const bool valid = blk < n_blocks; // wave-uniform
		// the two workgroups of a CU (dispatch order: b and b + gridDim.x / 2) take turns at the higher issue priority, trip by trip: left
		// alone the older one wins every contended slot and the younger one finishes the launch alone (k_train_regs.hip)
		if (a.prio_mode == 1) { // workgroup-uniform
			if ((trip + (blockIdx.x >= gridDim.x / 2 ? 1u : 0u)) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
		}
		asm volatile("" : "+v"(lane_o));
		const In in = pre;
		pre = load_in(min(blk + per_trip, n_blocks - 1)); // a whole trip ahead
		if (valid) {
			// ------------------------------------------------------------------------------------------------ forward
			img_write(A_IMG_X, w_nat, 0, in.x[0]);
			img_write(A_IMG_X, w_nat, 1, in.x[1]);
			f16v a0 = mfma32(f0[0], in.x[0], Z);
			a0 = mfma32(f0[1], in.x[1], a0);
			A_SB();
			f16v a1 = mfma32(f0[2], in.x[0], Z);
			a1 = mfma32(f0[3], in.x[1], a1);
			const h8 h00 = relu8(pack8(a0, 0)), h01 = relu8(pack8(a0, 1));
			img_write(A_IMG_H0, w_chain, 0, h00);
			img_write(A_IMG_H0, w_chain, 1, h01);
			const h8 wbo0 = frag(BO + 0), wbo1 = frag(BO + 1);
			A_MVD(8, 2); A_MVD(8, 2);
			A_SB();
			f16v b0 = mfma32(w[F1 + 0], h00, Z);
			b0 = mfma32(w[F1 + 1], h01, b0);
			f16v b1 = mfma32(w[F1 + 4], h00, Z);
			b1 = mfma32(w[F1 + 5], h01, b1);
			const h8 h02 = relu8(pack8(a1, 0)), h03 = relu8(pack8(a1, 1));
			img_write(A_IMG_H0 + 2048, w_chain, 0, h02);
			img_write(A_IMG_H0 + 2048, w_chain, 1, h03);
			A_MVD(4, 1); A_MVD(4, 1); A_MVD(4, 1); A_MVD(4, 1);
			A_SB();
			b0 = mfma32(w[F1 + 2], h02, b0);
			b0 = mfma32(w[F1 + 3], h03, b0);
			A_SB();
			b1 = mfma32(w[F1 + 6], h02, b1);
			b1 = mfma32(w[F1 + 7], h03, b1);
			const h8 h10 = relu8(pack8(b0, 0)), h11 = relu8(pack8(b0, 1));
			img_write(A_IMG_H1, w_chain, 0, h10);
			img_write(A_IMG_H1, w_chain, 1, h11);
			A_MVD(8, 2); A_MVD(8, 2);
			A_SB();
			f16v o = mfma32(w[FO + 0], h10, Z);
			o = mfma32(w[FO + 1], h11, o);
			const h8 h12 = relu8(pack8(b1, 0)), h13 = relu8(pack8(b1, 1));
			img_write(A_IMG_H1 + 2048, w_chain, 0, h12);
			img_write(A_IMG_H1 + 2048, w_chain, 1, h13);
			A_MVD(8, 2); A_MVD(8, 2);
			A_SB();
			o = mfma32(w[FO + 2], h12, o);
			o = mfma32(w[FO + 3], h13, o);
			A_SB();

			// ------------------------------------------------------------------------------------------------ loss on the result tile
			const h8 ov = pack8(o, 0); // element g: output 2 g + h (output activation None)
			h8 dyf = h8{0, 0, 0, 0, 0, 0, 0, 0}; // dL/doutput, the B fragment of the first backward product (k = position)
			{
				// l2.h:40-74 / relative_l2.h:40-75, the same operations in the same order, the two output slots of a lane side by side
				float value[2];
				half_t grad[2];
#pragma unroll
				for (int r = 0; r < 2; ++r) {
					loss_l2_fused<LOSS == 2>((float)ov[r], in.t[r], lsc, value[r], grad[r]); // l2.h:40-74 / relative_l2.h:40-75 on one refined reciprocal (mlp_device.h)
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
				uint32_t wd[4];
#pragma unroll
				for (int k = 0; k < 2; ++k) {
					const auto sw = __builtin_amdgcn_permlane32_swap(u[k], u[k + 2], false, false);
					const uint32_t even = sw[0], odd = sw[1];
					wd[2 * k + 0] = __builtin_amdgcn_perm(odd, even, 0x05040100u);
					wd[2 * k + 1] = __builtin_amdgcn_perm(odd, even, 0x07060302u);
				}
				__builtin_amdgcn_raw_buffer_store_b128(u32x4{wd[0], wd[1], wd[2], wd[3]}, rs_out, o_off + blk * 1024, 0, 2);
			}
			img_write(A_IMG_DY, w_chain, 0, dyf); // positions 4 h .. 4 h + 3 (plane h) and 8 + 4 h .. (plane 2 + h)
			A_SB();

			// ------------------------------------------------------------------------------------------------ backward chain
			f16v g0 = mfma32(wbo0, dyf, Z);
			f16v g1 = mfma32(wbo1, dyf, Z);
			asm volatile("" : "+v"(g1)); // here, not sunk to its use
			A_SB();
			// dH1 = (Wout^T dY) act'(H1) (common_device.h:241-297: from the forward OUTPUT)
			const h8 d10 = relu_bwd8(pack8(g0, 0), img_own(A_IMG_H1, 0)), d11 = relu_bwd8(pack8(g0, 1), img_own(A_IMG_H1, 1));
			img_write(A_IMG_DH1, w_chain, 0, d10);
			img_write(A_IMG_DH1, w_chain, 1, d11);
			A_SB();
			f16v e0 = mfma32(w[B1 + 0], d10, Z);
			e0 = mfma32(w[B1 + 1], d11, e0);
			f16v e1 = mfma32(w[B1 + 4], d10, Z);
			e1 = mfma32(w[B1 + 5], d11, e1);
			const h8 d12 = relu_bwd8(pack8(g1, 0), img_own(A_IMG_H1 + 2048, 0)), d13 = relu_bwd8(pack8(g1, 1), img_own(A_IMG_H1 + 2048, 1));
			img_write(A_IMG_DH1 + 2048, w_chain, 0, d12);
			img_write(A_IMG_DH1 + 2048, w_chain, 1, d13);
			A_MVD(6, 1); A_MVD(6, 1); A_MVD(6, 1); A_MVD(6, 1);
			A_SB();
			e0 = mfma32(w[B1 + 2], d12, e0);
			e0 = mfma32(w[B1 + 3], d13, e0);
			e1 = mfma32(w[B1 + 6], d12, e1);
			e1 = mfma32(w[B1 + 7], d13, e1);
			A_SB();
			const h8 d00 = relu_bwd8(pack8(e0, 0), img_own(A_IMG_H0, 0)), d01 = relu_bwd8(pack8(e0, 1), img_own(A_IMG_H0, 1));
			img_write(A_IMG_DH0, w_chain, 0, d00);
			img_write(A_IMG_DH0, w_chain, 1, d01);
			// dX = W0^T dH0
			f16v dx = mfma32(w[B0 + 0], d00, Z);
			dx = mfma32(w[B0 + 1], d01, dx);
			const h8 d02 = relu_bwd8(pack8(e1, 0), img_own(A_IMG_H0 + 2048, 0)), d03 = relu_bwd8(pack8(e1, 1), img_own(A_IMG_H0 + 2048, 1));
			img_write(A_IMG_DH0 + 2048, w_chain, 0, d02);
			img_write(A_IMG_DH0 + 2048, w_chain, 1, d03);
			dx = mfma32(w[B0 + 2], d02, dx);
			dx = mfma32(w[B0 + 3], d03, dx);
			A_SB();
			// scatter records {x, y, gradients of levels 2 p, 2 p + 1}: registers 4 g .. 4 g + 3 are features 8 g + 4 h .. + 3, i.e. level pair p = 2 g + h
			if constexpr (REC) {
				const u32x4 lo = __builtin_bit_cast(u32x4, pack8(dx, 0)), hi = __builtin_bit_cast(u32x4, pack8(dx, 1));
				const uint32_t x0 = __builtin_bit_cast(uint32_t, in.xs.x), x1 = __builtin_bit_cast(uint32_t, in.xs.y);
				const uint32_t rb = blk * 512, pair2 = a.n * 32; // pair2: two level pairs further
				__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, lo[0], lo[1]}, rs_rec, rec_off + rb, 0, R32_REC_AUX);
				__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, lo[2], lo[3]}, rs_rec, rec_off + (rb + pair2), 0, R32_REC_AUX);
				__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, hi[0], hi[1]}, rs_rec, rec_off + (rb + 2 * pair2), 0, R32_REC_AUX);
				__builtin_amdgcn_raw_buffer_store_b128(u32x4{x0, x1, hi[2], hi[3]}, rs_rec, rec_off + (rb + 3 * pair2), 0, R32_REC_AUX);
			} else { // level planes: word 2 g + j of the tile's 8 is level 4 g + 2 h + j of sample c
				const u32x4 lo = __builtin_bit_cast(u32x4, pack8(dx, 0)), hi = __builtin_bit_cast(u32x4, pack8(dx, 1));
				const uint32_t wd[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
				for (int k = 0; k < 8; ++k) __builtin_amdgcn_raw_buffer_store_b32(wd[k], rs_rec, rec_off, blk * 128 + n4 * (4 * (k >> 1) + (k & 1)), R32_REC_AUX);
			}
		} else {
			// no block for this wave in the last trip: images of zeros make its share of every product vanish
			const h8 zero = h8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
			for (int t = 0; t < 2; ++t)
#pragma unroll
				for (int s = 0; s < 2; ++s) {
					img_write(A_IMG_H0 + 2048 * t, w_chain, s, zero);
					img_write(A_IMG_H1 + 2048 * t, w_chain, s, zero);
					img_write(A_IMG_DH1 + 2048 * t, w_chain, s, zero);
					img_write(A_IMG_DH0 + 2048 * t, w_chain, s, zero);
					if (t == 0) img_write(A_IMG_X, w_chain, s, zero);
				}
			img_write(A_IMG_DY, w_chain, 0, zero);
		}
		// layer 0's fragments for the next trip: in flight over the weight-gradient phase
#pragma unroll
		for (int i = 0; i < 4; ++i) f0[i] = frag(F0 + i);
		A_STAMP(0);
		__syncthreads();
		A_STAMP(1);

		// ---------------------------------------------------------------------------------------------------- this wave's weight-gradient tiles
		// over the 4 x 32 samples of the workgroup's trip; v: the wave whose images are read
#pragma unroll
		for (int v = 0; v < R32A_NW; ++v) {
			// (requesting source v + 1's operands before source v's products needs 32 more registers: spills)
			const uint32_t vb = v * A_WAVE_BYTES;
			const h8 a10 = tr_frag(vb + A_IMG_DH1 + 2048 * tr1, 0), a11 = tr_frag(vb + A_IMG_DH1 + 2048 * tr1, 1);
			const h8 b10 = tr_frag(vb + A_IMG_H0 + 2048 * tc1, 0), b11 = tr_frag(vb + A_IMG_H0 + 2048 * tc1, 1);
			const h8 a20 = tr_frag(vb + img_a2, 0), a21 = tr_frag(vb + img_a2, 1);
			const h8 b20 = tr_frag(vb + img_b2, 0), b21 = tr_frag(vb + img_b2, 1);
			mfma32_acc(w1acc, a10, b10);
			mfma32_acc(w1acc, a11, b11);
			mfma32_acc(w2acc, a20, b20);
			mfma32_acc(w2acc, a21, b21);
			A_SB();
		}
		A_STAMP(2);
		__syncthreads(); // before the next trip overwrites the images
		A_STAMP(3);
	}
	if (a.dbg && lane == 0) {
		for (int i = 0; i < A_NPH; ++i) a.dbg[(size_t)gridDim.x * 6 + ((size_t)blockIdx.x * R32A_NW + wave) * A_NPH + i] = ph[i];
	}
	if (a.dbg && tid == 0) {
		a.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();
		a.dbg[(size_t)gridDim.x * 6 + (size_t)gridDim.x * R32A_NW * A_NPH + blockIdx.x] = t_fill;
	}

	// ---- this wave's tiles into the workgroup's slab: no other wave holds a share of them
	{
		float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
	#pragma unroll
		for (int g = 0; g < 16; ++g) slab[a.w_off[1] + (32 * tr1 + (g & 3) + 8 * (g >> 2) + 4 * h) * 64 + 32 * tc1 + c] = w1acc[g];
		if (wave < 2) {
#pragma unroll
			for (int g = 0; g < 16; ++g) slab[a.w_off[0] + (32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h) * 32 + c] = w2acc[g];
		} else {
			// dWout: register g < 8 is position (g & 3) + 8 (g >> 2) + 4 h = output 2 ((g & 3) + 4 (g >> 2)) + h (mlp_side_jobs.h, r32_prep_value), column 32 (w - 2) + c
#pragma unroll
			for (int g = 0; g < 8; ++g) slab[a.w_off[2] + (2 * ((g & 3) + 4 * (g >> 2)) + h) * 64 + 32 * (wave - 2) + c] = w2acc[g];
		}
	}
	if (a.dbg && tid == 0) {
		a.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
		a.dbg[(size_t)gridDim.x * 4 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
	}
}
#undef A_SB
#undef A_STAMP
#undef A_MVD

} // namespace

void mlp_train_r32a_launch(hipStream_t stream, const R32Args& a, uint32_t grid, int loss_id) {
	auto go = [&](auto kernel) {
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS_BYTES));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(R32A_NW * 64), A_LDS_BYTES, stream, a);
		HIP_CHECK_THROW(hipGetLastError());
	};
	if (loss_id == 1) { if (a.rec_x) go(k_mlp_train_r32a<1, true>); else go(k_mlp_train_r32a<1, false>); }
	else { if (a.rec_x) go(k_mlp_train_r32a<2, true>); else go(k_mlp_train_r32a<2, false>); }
}

} // namespace tcnn_amd
