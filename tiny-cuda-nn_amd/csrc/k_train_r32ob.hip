// k_train_r32ob.hip -- the training step's MLP part for BASELINE config 2: OneBlob(64 bins, 2 dims) -> 64 -> 64 -> 16, the encoding
// evaluated inside the kernel, on v_mfma_f32_32x32x16_f16.
//
// Same job as k_mlp_train<64, ..., OB> (k_train.hip; reference: encodings/oneblob.h:47-67, src/fully_fused_mlp.cu:500-557 forward,
// losses/{l2,relative_l2}.h:40-75, fully_fused_mlp.cu:151-259 backward, :785-828 the weight-gradient GEMMs), shaped like
// k_train_r32.hip (register chain, 32 samples per wave and trip, weight-gradient operands transposed through wave-private LDS
// images) with what 128 inputs change:
//   * dW0 alone is 8 tiles of 32 x 32: with dW1 and dWout 208 accumulator registers if every wave keeps every tile.  A wave runs alone on
//     its SIMD (4 waves per workgroup, 512 registers each), accumulators in the ACCUMULATION registers (AGPRs; the matrix instructions that
//     add into them are inline assembly, every other one is a builtin that the compiler places in ordinary registers: -mllvm
//     -amdgpu-mfma-vgpr-form, build.py).  Two forms, chosen by the launch's length (mlp_train_r32ob):
//       - k_mlp_train_r32ob (up to 4 trips per wave; BASELINE config 2 has 2): the TILES are shared out over the workgroup's waves as in
//         k_train_r32w.hip -- wave w owns 2 tiles of dW0, 1 of dW1 and half of dWout, 64 registers, summed over the samples of all four
//         waves: every wave reads every wave's images between two workgroup barriers per trip (dH1 and dH0 have images of their own),
//         every wave runs the same number of trips, and a wave stores its tiles into the slab as they stand: no sum over waves at the end;
//       - k_mlp_train_r32ob_acc (longer launches): every wave keeps all 13 tiles of its own samples, the tile products run inside the
//         backward chain, and the four waves' tiles are summed through LDS at the end (7.5 k clocks, 1.25 k fewer per trip);
//   * the input: per sample and dimension only the five bins around x differ from exactly +0 (oneblob_device.h).  Lane (sample, h)
//     evaluates dimension h's six bin edges, writes the five halves into the sample's row of the (otherwise zero) input image in
//     LDS, reads its eight k-steps of layer 0 back from that image (the same image serves dW0 transposed), and takes its five
//     halves out again at the end of the trip.  Inputs outside [0, 1] (where the window argument does not hold) take a
//     wave-uniform slow path that evaluates all 64 bins;
//   * no dL/dinput (the encoding has no parameters), context matrices in the reference's padded form [n][16].
#include "r32_device.h"
#include "mlp_side_jobs.h"
#include "oneblob_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {

struct ObArgs {
	MatView x;              // coordinates [n][2]
	const float* target;    // [n][dims]
	half_t* out;            // [n][16]
	half_t* dL_dout;        // [n][16]
	float* L;               // [n][16]
	float* slabs;           // [gridDim.x][n_params]
	const h8* image;        // R32 fragments of the network
	uint32_t n, dims, n_params;
	uint32_t w_off[3];
	float loss_scale;
	unsigned long long* dbg;
};

constexpr int OB_NW = 4;                  // waves per workgroup: one per SIMD
constexpr int OB_NF = 38;                 // fragments used: layer 0 [2][8], layer 1 [2][4], output [4], Wout^T [2], W1^T [2][4] (W0^T is not needed)
constexpr int OB_WAVE0 = OB_NF * 1024;
constexpr int OB_IMG_X = 0, OB_IMG_H0 = 8192, OB_IMG_H1 = 12288, OB_IMG_DH1 = 16384, OB_IMG_DH0 = 20480, OB_IMG_DY = 24576;
constexpr int OB_WAVE_BYTES = 25 * 1024;  // X 8 K | H0 4 K | H1 4 K | dH1 4 K | dH0 4 K | dY 1 K
constexpr int OB_LDS_BYTES = OB_WAVE0 + OB_NW * OB_WAVE_BYTES + 1024; // 142 336; the last KiB: what dWout's 32-row read of the last wave's 16-row dY image runs into
constexpr uint32_t OB_LOG2_BINS = 6;

// accumulate into AGPRs (see the header comment); no hazard to pad: the operands come from LDS reads, which the compiler waits for,
// and back-to-back accumulation into the same registers needs no wait
__device__ inline void mfma32_acc(f16v& acc, const h8 a, const h8 b) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }

#define OB_SB() __builtin_amdgcn_sched_barrier(0)

// LOSS 1: L2, 2: RelativeL2
template <int LOSS>
__global__ void __launch_bounds__(OB_NW * 64, 1) k_mlp_train_r32ob(const ObArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 31, h = lane >> 5;
	if (a.dbg && tid == 0) {
		a.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();
		a.dbg[(size_t)gridDim.x * 4 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime(); // the device-wide 100 MHz clock
	}

	const uint32_t n_blocks = a.n / 32;
	const uint32_t first = blockIdx.x * OB_NW + wave, per_trip = gridDim.x * OB_NW;
	const uint32_t n_trips = (n_blocks + per_trip - 1) / per_trip; // the same for every wave of every workgroup: the barriers are workgroup-wide
	const uint32_t n_total = a.n * a.dims;
	const LossScales lsc = loss_scales(n_total, a.loss_scale);

	// ---- inputs of a trip: this lane's coordinate (dimension h of sample c) and its targets
	uint32_t t_off[2];
#pragma unroll
	for (int r = 0; r < 2; ++r) t_off[r] = (c * a.dims + min(2 * r + h, a.dims - 1)) * 4;
	struct In { float xv; float t[2]; };
	auto load_in = [&](const uint32_t blk) -> In {
		In r;
		r.xv = a.x.data[(size_t)(blk * 32 + c) * a.x.stride_sample + (size_t)h * a.x.stride_dim];
		const char* tb = (const char*)a.target + (size_t)blk * (128 * a.dims);
		r.t[0] = *(const float*)(tb + t_off[0]);
		r.t[1] = *(const float*)(tb + t_off[1]);
		return r;
	};
	In pre = load_in(min(first, n_blocks - 1));

	// ---- weight fragments into LDS; this wave's input image zeroed (it stays zero except for the trip's window, see below)
	{
		constexpr uint32_t N16 = OB_NF * 64;
		constexpr int FILL = (N16 + OB_NW * 64 - 1) / (OB_NW * 64);
		h8 tmp[FILL];
#pragma unroll
		for (int k = 0; k < FILL; ++k) tmp[k] = a.image[min(tid + k * OB_NW * 64, N16 - 1)];
#pragma unroll
		for (int k = 0; k < FILL; ++k) {
			if (tid + k * OB_NW * 64 < N16) ((h8*)smem)[tid + k * OB_NW * 64] = tmp[k];
		}
	}
	const uint32_t wbase = OB_WAVE0 + wave * OB_WAVE_BYTES;
#pragma unroll
	for (int k = 0; k < 8; ++k) *(h8*)(smem + wbase + OB_IMG_X + (k * 64 + lane) * 16) = h8{0, 0, 0, 0, 0, 0, 0, 0};
	if (tid < 64) *(h8*)(smem + OB_WAVE0 + OB_NW * OB_WAVE_BYTES + tid * 16) = h8{0, 0, 0, 0, 0, 0, 0, 0}; // the KiB behind the last wave's images
	__syncthreads();
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();

	uint32_t lane16 = lane * 16;
	auto frag = [&](const int f) -> h8 { return *(const h8*)(smem + lane16 + f * 1024); };
	// images: see k_train_r32.hip (per 32-feature tile 2 KiB: plane g of 4 features at 256 g, sample n at 8 ((n + 4 g) & 31))
	uint32_t w_chain[4], w_nat[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t gc = 4 * (k >> 1) + 2 * (k & 1) + h, gn = 4 * (k >> 1) + 2 * h + (k & 1);
		w_chain[k] = wbase + gc * 256 + ((c + 4 * gc) & 31) * 8;
		w_nat[k] = wbase + gn * 256 + ((c + 4 * gn) & 31) * 8;
	}
	uint32_t r_tr[4]; // transposing reads, relative to a tile of ANY wave's images
	{
		const uint32_t grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, hh = grp >> 1;
		const uint32_t g = 4 * (grp & 1) + p;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = 16 * (k >> 1) + 8 * hh + 4 * (k & 1) + q;
			r_tr[k] = OB_WAVE0 + g * 256 + ((row + 4 * g) & 31) * 8;
		}
	}
	auto img_write = [&](const int img, const int s, const h8 v) {
		*(h4*)(smem + w_chain[2 * s + 0] + img) = h4{v[0], v[1], v[2], v[3]};
		*(h4*)(smem + w_chain[2 * s + 1] + img) = h4{v[4], v[5], v[6], v[7]};
	};
	auto img_own = [&](const int img, const uint32_t (&w)[4], const int s) -> h8 { // this lane's own fragment back from an image
		const h4 lo = *(const h4*)(smem + w[2 * s + 0] + img), hi = *(const h4*)(smem + w[2 * s + 1] + img);
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr_frag = [&](const uint32_t img, const int sp) -> h8 {
		const h4 lo = lds_read_tr((const half_t*)(smem + r_tr[2 * sp] + img)), hi = lds_read_tr((const half_t*)(smem + r_tr[2 * sp + 1] + img));
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	// feature f (0..127) of sample c in the input image: tile f >> 5, plane (f & 31) >> 2, element f & 3
	auto x_addr = [&](const uint32_t f) -> uint32_t {
		const uint32_t g = (f & 31u) >> 2;
		return wbase + OB_IMG_X + (f >> 5) * 2048 + g * 256 + ((c + 4 * g) & 31) * 8 + (f & 3u) * 2;
	};

	const uint32_t row_off = c * 32 + h * 16; // [n][16] halves: this lane's half of the row (16 bytes); [n][16] floats: twice that

	// This wave's weight-gradient tiles (32 x 32 each), summed over the samples of all four waves -- FINAL for the workgroup:
	//   acc[0], acc[1]: dW0 (64 x 128) row tile w >> 1, column tiles 2 (w & 1), 2 (w & 1) + 1  = dH0 X^T
	//   acc[2]:         dW1 (64 x 64) tile (w >> 1, w & 1)                                     = dH1 H0^T
	//   acc[3]:         dWout (16 x 64) columns 32 (w & 1) .. as the upper half of a tile whose A operand reads "positions" 16 .. 31 from
	//                   the KiB behind the 16-position dY image (a row of A only reaches the same row of the result); waves 0 and 1
	//                   store it, waves 2 and 3 compute the same two tiles again and drop them (the same instructions in every wave)
	f16v acc[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) acc[i] = zero16();
	const uint32_t tr_w = wave >> 1, tc_w = wave & 1u;
	const uint32_t img_a0 = OB_IMG_DH0 + 2048 * tr_w, img_b0 = OB_IMG_X + 4096 * tc_w, img_a2 = OB_IMG_DH1 + 2048 * tr_w, img_b2 = OB_IMG_H0 + 2048 * tc_w,
	               img_b3 = OB_IMG_H1 + 2048 * tc_w; // wave-uniform

	constexpr int F0 = 0, F1 = 16, FO = 24, BO = 28, B1 = 30;
	const f16v Z = zero16();

	for (uint32_t trip = 0; trip < n_trips; ++trip) {
		const uint32_t blk = __builtin_amdgcn_readfirstlane(trip * per_trip + first);
		const bool valid = blk < n_blocks; // wave-uniform
		asm volatile("" : "+v"(lane16));
		const In in = pre;
		pre = load_in(min(blk + per_trip, n_blocks - 1));
		// the encoding's window of this trip (oneblob.h:47-67): bin b of dimension h = C(edge b + 1) - C(edge b); for x in [0, 1] only the
		// bins first .. first + 4 (mod 64) differ from +0
		const bool unit = oneblob_in_unit_interval(in.xv);
		const bool fast = __all(unit) != 0; // wave-uniform
		const uint32_t first_bin = oneblob_window_first(in.xv, OB_LOG2_BINS);
		if (valid) {
			if (fast) {
				float e[6];
#pragma unroll
				for (int k = 0; k < 6; ++k) e[k] = oneblob_edge(in.xv, first_bin + k, OB_LOG2_BINS);
#pragma unroll
				for (int o = 0; o < 5; ++o) {
					const uint32_t bin = (first_bin + o) & 63u;
					float r = e[o + 1];
					if (bin == 63u) r += 1; // the last bin's right edge is bin 0's left edge + 1
					*(half_t*)(smem + x_addr(64 * h + bin)) = (half_t)(r - e[o]);
				}
			} else {
				for (uint32_t bin = 0; bin < 64; ++bin) *(half_t*)(smem + x_addr(64 * h + bin)) = (half_t)oneblob_bin(in.xv, bin, OB_LOG2_BINS);
			}
			// Software pipeline as in k_train_r32.hip, with one wave per SIMD and registers to spare: every LDS operand is requested one
			// region (sched_barrier) before the matrix instructions that use it.
			h8 wf[8], wg[8];
#pragma unroll
			for (int s = 0; s < 8; ++s) wf[s] = frag(F0 + s);
			OB_SB();

			// -------------------------------------------------------------------------------------------- forward
			h8 xs[8];
#pragma unroll
			for (int s = 0; s < 8; ++s) xs[s] = img_own(OB_IMG_X + 2048 * (s >> 1), w_nat, s & 1); // features 16 s + 8 h + j of this lane's sample
#pragma unroll
			for (int s = 0; s < 8; ++s) wg[s] = frag(F0 + 8 + s);
			f16v a0 = mfma32(wf[0], xs[0], Z);
#pragma unroll
			for (int s = 1; s < 8; ++s) a0 = mfma32(wf[s], xs[s], a0);
			OB_SB();
#pragma unroll
			for (int s = 0; s < 8; ++s) wf[s] = frag(F1 + s);
			f16v a1 = mfma32(wg[0], xs[0], Z);
#pragma unroll
			for (int s = 1; s < 8; ++s) a1 = mfma32(wg[s], xs[s], a1);
			const h8 h00 = relu8(pack8(a0, 0)), h01 = relu8(pack8(a0, 1));
			img_write(OB_IMG_H0, 0, h00);
			img_write(OB_IMG_H0, 1, h01);
			OB_SB();
			f16v b0 = mfma32(wf[0], h00, Z);
			b0 = mfma32(wf[1], h01, b0);
			f16v b1 = mfma32(wf[4], h00, Z);
			b1 = mfma32(wf[5], h01, b1);
			const h8 h02 = relu8(pack8(a1, 0)), h03 = relu8(pack8(a1, 1));
			img_write(OB_IMG_H0 + 2048, 0, h02);
			img_write(OB_IMG_H0 + 2048, 1, h03);
#pragma unroll
			for (int s = 0; s < 4; ++s) wg[s] = frag(FO + s);
			OB_SB();
			b0 = mfma32(wf[2], h02, b0);
			b0 = mfma32(wf[3], h03, b0);
			b1 = mfma32(wf[6], h02, b1);
			b1 = mfma32(wf[7], h03, b1);
			wg[4] = frag(BO + 0);
			wg[5] = frag(BO + 1);
			OB_SB();
			const h8 h10 = relu8(pack8(b0, 0)), h11 = relu8(pack8(b0, 1));
			img_write(OB_IMG_H1, 0, h10);
			img_write(OB_IMG_H1, 1, h11);
			f16v o = mfma32(wg[0], h10, Z);
			o = mfma32(wg[1], h11, o);
			const h8 h12 = relu8(pack8(b1, 0)), h13 = relu8(pack8(b1, 1));
			img_write(OB_IMG_H1 + 2048, 0, h12);
			img_write(OB_IMG_H1 + 2048, 1, h13);
			o = mfma32(wg[2], h12, o);
			o = mfma32(wg[3], h13, o);
#pragma unroll
			for (int s = 0; s < 8; ++s) wf[s] = frag(B1 + s);
			OB_SB();

			// -------------------------------------------------------------------------------------------- loss, context matrices, out
			const h8 ov = pack8(o, 0); // element g: output 2 g + h
			h8 dyf = h8{0, 0, 0, 0, 0, 0, 0, 0};
			{
				float value[2];
				half_t grad[2];
#pragma unroll
				for (int r = 0; r < 2; ++r) {
					loss_l2_fused<LOSS == 2>((float)ov[r], in.t[r], lsc, value[r], grad[r]); // l2.h:40-74 / relative_l2.h:40-75 on one refined reciprocal (mlp_device.h)
					const bool live = 2 * r + h < a.dims;
					if (!live) { value[r] = 0.0f; grad[r] = (half_t)0.0f; }
					dyf[r] = grad[r];
				}
				// the padded matrices [n][16]: outputs 0..3 come from the two lanes of the sample (this lane: 2 r + h), the rest is zero.
				// lanes h = 0 collect them; every lane stores its half of the row
				typedef _Float16 h2 __attribute__((ext_vector_type(2)));
				const uint32_t gpk = __builtin_bit_cast(uint32_t, (h2{grad[0], grad[1]})); // (out h, out 2 + h)
				const auto sg = __builtin_amdgcn_permlane32_swap(gpk, gpk, false, false);    // lanes < 32: [0] own, [1] the partner's
				const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, value[0]), __builtin_bit_cast(uint32_t, value[0]), false, false);
				const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, value[1]), __builtin_bit_cast(uint32_t, value[1]), false, false);
				u32x4 grow = u32x4{0, 0, 0, 0}, l0 = u32x4{0, 0, 0, 0};
				if (h == 0) {
					grow[0] = __builtin_amdgcn_perm(sg[1], sg[0], 0x05040100u); // (g0, g1)
					grow[1] = __builtin_amdgcn_perm(sg[1], sg[0], 0x07060302u); // (g2, g3)
					l0 = u32x4{s0[0], s0[1], s1[0], s1[1]};                     // L0, L1, L2, L3
				}
				*(u32x4*)((char*)a.dL_dout + (size_t)blk * 1024 + row_off) = grow;
				float* lrow = (float*)((char*)a.L + (size_t)blk * 2048 + 2 * row_off);
				*(u32x4*)lrow = l0;
				*(u32x4*)(lrow + 4) = u32x4{0, 0, 0, 0};
			}
			{
				const u32x4 u = __builtin_bit_cast(u32x4, ov);
				uint32_t w[4];
#pragma unroll
				for (int k = 0; k < 2; ++k) {
					const auto sw = __builtin_amdgcn_permlane32_swap(u[k], u[k + 2], false, false);
					w[2 * k + 0] = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);
					w[2 * k + 1] = __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u);
				}
				*(u32x4*)((char*)a.out + (size_t)blk * 1024 + row_off) = u32x4{w[0], w[1], w[2], w[3]};
			}
			img_write(OB_IMG_DY, 0, dyf);
			OB_SB();

			// -------------------------------------------------------------------------------------------- backward chain
			f16v g0 = mfma32(wg[4], dyf, Z);
			f16v g1 = mfma32(wg[5], dyf, Z);
			OB_SB();
			const h8 d10 = relu_bwd8(pack8(g0, 0), h10), d11 = relu_bwd8(pack8(g0, 1), h11);
			img_write(OB_IMG_DH1, 0, d10);
			img_write(OB_IMG_DH1, 1, d11);
			f16v e0 = mfma32(wf[0], d10, Z);
			e0 = mfma32(wf[1], d11, e0);
			f16v e1 = mfma32(wf[4], d10, Z);
			e1 = mfma32(wf[5], d11, e1);
			const h8 d12 = relu_bwd8(pack8(g1, 0), h12), d13 = relu_bwd8(pack8(g1, 1), h13);
			img_write(OB_IMG_DH1 + 2048, 0, d12);
			img_write(OB_IMG_DH1 + 2048, 1, d13);
			OB_SB();
			e0 = mfma32(wf[2], d12, e0);
			e0 = mfma32(wf[3], d13, e0);
			e1 = mfma32(wf[6], d12, e1);
			e1 = mfma32(wf[7], d13, e1);
			OB_SB();
			// dH0 = (W1^T dH1) act'(H0) (no dL/dinput: the encoding has no parameters)
			const h8 d00 = relu_bwd8(pack8(e0, 0), h00), d01 = relu_bwd8(pack8(e0, 1), h01), d02 = relu_bwd8(pack8(e1, 0), h02), d03 = relu_bwd8(pack8(e1, 1), h03);
			img_write(OB_IMG_DH0, 0, d00);
			img_write(OB_IMG_DH0, 1, d01);
			img_write(OB_IMG_DH0 + 2048, 0, d02);
			img_write(OB_IMG_DH0 + 2048, 1, d03);
		} else {
			// no block for this wave in the last trip: images of zeros make its share of every product vanish (its input image is zero already)
			const h8 zero = h8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
			for (int t = 0; t < 2; ++t)
#pragma unroll
				for (int s = 0; s < 2; ++s) {
					img_write(OB_IMG_H0 + 2048 * t, s, zero);
					img_write(OB_IMG_H1 + 2048 * t, s, zero);
					img_write(OB_IMG_DH1 + 2048 * t, s, zero);
					img_write(OB_IMG_DH0 + 2048 * t, s, zero);
				}
			img_write(OB_IMG_DY, 0, zero);
		}
		__syncthreads();

		// ---------------------------------------------------------------------------------------------------- this wave's weight-gradient tiles
		// over the 4 x 32 samples of the workgroup's trip; v: the wave whose images are read.  Operands of source v + 1 are requested
		// before the products of source v (512 registers: room for both sets)
		{
			h8 A0[2][2], B0[2][4], A2[2][2], B2[2][2], A3[2][2], B3[2][2]; // [buffer][k-step (x column tile for B0)]
			auto request = [&](const int v, const int buf) {
				const uint32_t vb = v * OB_WAVE_BYTES;
#pragma unroll
				for (int sp = 0; sp < 2; ++sp) {
					A0[buf][sp] = tr_frag(vb + img_a0, sp);
					B0[buf][sp] = tr_frag(vb + img_b0, sp);
					B0[buf][2 + sp] = tr_frag(vb + img_b0 + 2048, sp);
					A2[buf][sp] = tr_frag(vb + img_a2, sp);
					B2[buf][sp] = tr_frag(vb + img_b2, sp);
					A3[buf][sp] = tr_frag(vb + OB_IMG_DY, sp);
					B3[buf][sp] = tr_frag(vb + img_b3, sp);
				}
			};
			request(0, 0);
#pragma unroll
			for (int v = 0; v < OB_NW; ++v) {
				const int buf = v & 1;
				if (v + 1 < OB_NW) request(v + 1, buf ^ 1);
#pragma unroll
				for (int sp = 0; sp < 2; ++sp) {
					mfma32_acc(acc[0], A0[buf][sp], B0[buf][sp]);
					mfma32_acc(acc[1], A0[buf][sp], B0[buf][2 + sp]);
					mfma32_acc(acc[2], A2[buf][sp], B2[buf][sp]);
					mfma32_acc(acc[3], A3[buf][sp], B3[buf][sp]);
				}
				OB_SB();
			}
		}
		__syncthreads(); // before the next trip overwrites the images
		// the trip's window out of the input image again
		if (valid) {
			if (fast) {
#pragma unroll
				for (int o = 0; o < 5; ++o) *(half_t*)(smem + x_addr(64 * h + ((first_bin + o) & 63u))) = (half_t)0.0f;
			} else {
				for (uint32_t bin = 0; bin < 64; ++bin) *(half_t*)(smem + x_addr(64 * h + bin)) = (half_t)0.0f;
			}
		}
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

	// ---- this wave's tiles into the workgroup's slab: no other wave holds a share of them.  Tile register g of lane (c, h): row
	// (g & 3) + 8 (g >> 2) + 4 h, column c
	{
		float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int g = 0; g < 16; ++g) slab[a.w_off[0] + (32 * tr_w + (g & 3) + 8 * (g >> 2) + 4 * h) * 128 + 32 * (2 * tc_w + t) + c] = acc[t][g];
#pragma unroll
		for (int g = 0; g < 16; ++g) slab[a.w_off[1] + (32 * tr_w + (g & 3) + 8 * (g >> 2) + 4 * h) * 64 + 32 * tc_w + c] = acc[2][g];
		if (wave < 2) { // dWout: register g < 8 is position (g & 3) + 8 (g >> 2) + 4 h = output 2 ((g & 3) + 4 (g >> 2)) + h (mlp_side_jobs.h, r32_prep_value)
#pragma unroll
			for (int g = 0; g < 8; ++g) slab[a.w_off[2] + (2 * ((g & 3) + 4 * (g >> 2)) + h) * 64 + 32 * tc_w + c] = acc[3][g];
		}
	}
	if (a.dbg && tid == 0) {
		a.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
		a.dbg[(size_t)gridDim.x * 4 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
	}
}
// ---------------------------------------------------------------------------------------------------------------------------------------
// k_mlp_train_r32ob_acc: the form for LONG launches (5 trips per wave and more).  Every wave keeps all 13 weight-gradient tiles of its own
// samples (208 accumulator registers in AGPRs) and never meets another wave inside the trip loop -- the tile products run inside the
// backward chain's regions instead of behind a barrier: 8.4 k clocks per trip against the shared-tiles form's 9.65 k -- and the four
// waves' tiles are summed through LDS at the end (7.5 k clocks, which the shared-tiles form does not have: it wins below 5 trips).
constexpr int OBA_IMG_X = 0, OBA_IMG_H0 = 8192, OBA_IMG_H1 = 12288, OBA_IMG_DH = 16384, OBA_IMG_DY = 20480;
constexpr int OBA_WAVE_BYTES = 21 * 1024;  // X 8 K | H0 4 K | H1 4 K | dH 4 K | dY 1 K
constexpr int OBA_LDS_BYTES = OB_WAVE0 + OB_NW * OBA_WAVE_BYTES; // 124 928

// accumulate into AGPRs (see the header comment); no hazard to pad: the operands come from LDS reads, which the compiler waits for,
// and back-to-back accumulation into the same registers needs no wait
__device__ inline void mfma16_acc(f4& acc, const h8 a, const h8 b) { asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }


// LOSS 1: L2, 2: RelativeL2
template <int LOSS>
__global__ void __launch_bounds__(OB_NW * 64, 1) k_mlp_train_r32ob_acc(const ObArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 31, h = lane >> 5;
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();

	const uint32_t n_blocks = a.n / 32;
	const uint32_t first = blockIdx.x * OB_NW + wave, step = gridDim.x * OB_NW;
	const uint32_t n_total = a.n * a.dims;
	const LossScales lsc = loss_scales(n_total, a.loss_scale);

	// ---- inputs of a trip: this lane's coordinate (dimension h of sample c) and its targets
	uint32_t t_off[2];
#pragma unroll
	for (int r = 0; r < 2; ++r) t_off[r] = (c * a.dims + min(2 * r + h, a.dims - 1)) * 4;
	struct In { float xv; float t[2]; };
	auto load_in = [&](const uint32_t blk) -> In {
		In r;
		r.xv = a.x.data[(size_t)(blk * 32 + c) * a.x.stride_sample + (size_t)h * a.x.stride_dim];
		const char* tb = (const char*)a.target + (size_t)blk * (128 * a.dims);
		r.t[0] = *(const float*)(tb + t_off[0]);
		r.t[1] = *(const float*)(tb + t_off[1]);
		return r;
	};
	In pre{};
	if (first < n_blocks) pre = load_in(first);

	// ---- weight fragments into LDS; this wave's input image zeroed (it stays zero except for the trip's window, see below)
	{
		constexpr uint32_t N16 = OB_NF * 64;
		constexpr int FILL = (N16 + OB_NW * 64 - 1) / (OB_NW * 64);
		h8 tmp[FILL];
#pragma unroll
		for (int k = 0; k < FILL; ++k) tmp[k] = a.image[min(tid + k * OB_NW * 64, N16 - 1)];
#pragma unroll
		for (int k = 0; k < FILL; ++k) {
			if (tid + k * OB_NW * 64 < N16) ((h8*)smem)[tid + k * OB_NW * 64] = tmp[k];
		}
	}
	const uint32_t wbase = OB_WAVE0 + wave * OBA_WAVE_BYTES;
#pragma unroll
	for (int k = 0; k < 8; ++k) *(h8*)(smem + wbase + OBA_IMG_X + (k * 64 + lane) * 16) = h8{0, 0, 0, 0, 0, 0, 0, 0};
	__syncthreads();
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();

	uint32_t lane16 = lane * 16;
	auto frag = [&](const int f) -> h8 { return *(const h8*)(smem + lane16 + f * 1024); };
	// images: see k_train_r32.hip (per 32-feature tile 2 KiB: plane g of 4 features at 256 g, sample n at 8 ((n + 4 g) & 31))
	uint32_t w_chain[4], w_nat[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t gc = 4 * (k >> 1) + 2 * (k & 1) + h, gn = 4 * (k >> 1) + 2 * h + (k & 1);
		w_chain[k] = wbase + gc * 256 + ((c + 4 * gc) & 31) * 8;
		w_nat[k] = wbase + gn * 256 + ((c + 4 * gn) & 31) * 8;
	}
	uint32_t r_tr[4], r_16[4];
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
	auto img_write = [&](const int img, const int s, const h8 v) {
		*(h4*)(smem + w_chain[2 * s + 0] + img) = h4{v[0], v[1], v[2], v[3]};
		*(h4*)(smem + w_chain[2 * s + 1] + img) = h4{v[4], v[5], v[6], v[7]};
	};
	auto img_own = [&](const int img, const uint32_t (&w)[4], const int s) -> h8 { // this lane's own fragment back from an image
		const h4 lo = *(const h4*)(smem + w[2 * s + 0] + img), hi = *(const h4*)(smem + w[2 * s + 1] + img);
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr2 = [&](const uint32_t a0, const uint32_t a1) -> h8 {
		const h4 lo = lds_read_tr((const half_t*)(smem + a0)), hi = lds_read_tr((const half_t*)(smem + a1));
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr_frag = [&](const int img, const int sp) -> h8 { return tr2(r_tr[2 * sp] + img, r_tr[2 * sp + 1] + img); };
	auto tr_frag16 = [&](const int img, const int half) -> h8 { return tr2(r_16[2 * half] + img, r_16[2 * half + 1] + img); };
	// feature f (0..127) of sample c in the input image: tile f >> 5, plane (f & 31) >> 2, element f & 3
	auto x_addr = [&](const uint32_t f) -> uint32_t {
		const uint32_t g = (f & 31u) >> 2;
		return wbase + OBA_IMG_X + (f >> 5) * 2048 + g * 256 + ((c + 4 * g) & 31) * 8 + (f & 3u) * 2;
	};

	const uint32_t row_off = c * 32 + h * 16; // [n][16] halves: this lane's half of the row (16 bytes); [n][16] floats: twice that

	f16v w0acc[8], w1acc[4]; // dW0 tiles (tr, tc) = [4 tr + tc], dW1 tiles (tr, tc)
	f4 wout[4];
#pragma unroll
	for (int i = 0; i < 8; ++i) w0acc[i] = zero16();
#pragma unroll
	for (int i = 0; i < 4; ++i) { w1acc[i] = zero16(); wout[i] = f4{0, 0, 0, 0}; }

	constexpr int F0 = 0, F1 = 16, FO = 24, BO = 28, B1 = 30;
	const f16v Z = zero16();

	for (uint32_t blk = first; blk < n_blocks; blk += step) {
		asm volatile("" : "+v"(lane16));
		const In in = pre;
		{
			const uint32_t next = min(blk + step, n_blocks - 1);
			pre = load_in(next);
		}
		// ------------------------------------------------------------------------------------------------ the encoding (oneblob.h:47-67)
		// bin b of dimension h = C(edge b + 1) - C(edge b); for x in [0, 1] only the bins first .. first + 4 (mod 64) differ from +0
		const bool unit = oneblob_in_unit_interval(in.xv);
		const bool fast = __all(unit) != 0; // wave-uniform
		const uint32_t first_bin = oneblob_window_first(in.xv, OB_LOG2_BINS);
		if (fast) {
			float e[6];
#pragma unroll
			for (int k = 0; k < 6; ++k) e[k] = oneblob_edge(in.xv, first_bin + k, OB_LOG2_BINS);
#pragma unroll
			for (int o = 0; o < 5; ++o) {
				const uint32_t bin = (first_bin + o) & 63u;
				float r = e[o + 1];
				if (bin == 63u) r += 1; // the last bin's right edge is bin 0's left edge + 1
				*(half_t*)(smem + x_addr(64 * h + bin)) = (half_t)(r - e[o]);
			}
		} else {
			for (uint32_t bin = 0; bin < 64; ++bin) *(half_t*)(smem + x_addr(64 * h + bin)) = (half_t)oneblob_bin(in.xv, bin, OB_LOG2_BINS);
		}
		// Software pipeline as in k_train_r32.hip, with one wave per SIMD and registers to spare: every LDS operand is requested one
		// region (sched_barrier) before the matrix instructions that use it.
		h8 wf[8], wg[8];
#pragma unroll
		for (int s = 0; s < 8; ++s) wf[s] = frag(F0 + s);
		OB_SB();

		// ------------------------------------------------------------------------------------------------ forward
		h8 xs[8];
#pragma unroll
		for (int s = 0; s < 8; ++s) xs[s] = img_own(OBA_IMG_X + 2048 * (s >> 1), w_nat, s & 1); // features 16 s + 8 h + j of this lane's sample
#pragma unroll
		for (int s = 0; s < 8; ++s) wg[s] = frag(F0 + 8 + s);
		f16v a0 = mfma32(wf[0], xs[0], Z);
#pragma unroll
		for (int s = 1; s < 8; ++s) a0 = mfma32(wf[s], xs[s], a0);
		OB_SB();
#pragma unroll
		for (int s = 0; s < 8; ++s) wf[s] = frag(F1 + s);
		f16v a1 = mfma32(wg[0], xs[0], Z);
#pragma unroll
		for (int s = 1; s < 8; ++s) a1 = mfma32(wg[s], xs[s], a1);
		const h8 h00 = relu8(pack8(a0, 0)), h01 = relu8(pack8(a0, 1));
		img_write(OBA_IMG_H0, 0, h00);
		img_write(OBA_IMG_H0, 1, h01);
		OB_SB();
		f16v b0 = mfma32(wf[0], h00, Z);
		b0 = mfma32(wf[1], h01, b0);
		f16v b1 = mfma32(wf[4], h00, Z);
		b1 = mfma32(wf[5], h01, b1);
		const h8 h02 = relu8(pack8(a1, 0)), h03 = relu8(pack8(a1, 1));
		img_write(OBA_IMG_H0 + 2048, 0, h02);
		img_write(OBA_IMG_H0 + 2048, 1, h03);
#pragma unroll
		for (int s = 0; s < 4; ++s) wg[s] = frag(FO + s);
		OB_SB();
		b0 = mfma32(wf[2], h02, b0);
		b0 = mfma32(wf[3], h03, b0);
		b1 = mfma32(wf[6], h02, b1);
		b1 = mfma32(wf[7], h03, b1);
		wg[4] = frag(BO + 0);
		wg[5] = frag(BO + 1);
		OB_SB();
		const h8 h10 = relu8(pack8(b0, 0)), h11 = relu8(pack8(b0, 1));
		img_write(OBA_IMG_H1, 0, h10);
		img_write(OBA_IMG_H1, 1, h11);
		f16v o = mfma32(wg[0], h10, Z);
		o = mfma32(wg[1], h11, o);
		const h8 h12 = relu8(pack8(b1, 0)), h13 = relu8(pack8(b1, 1));
		img_write(OBA_IMG_H1 + 2048, 0, h12);
		img_write(OBA_IMG_H1 + 2048, 1, h13);
		o = mfma32(wg[2], h12, o);
		o = mfma32(wg[3], h13, o);
		h8 bH[4];
#pragma unroll
		for (int tc = 0; tc < 4; ++tc) bH[tc] = tr_frag16(OBA_IMG_H1 + 2048 * (tc >> 1), tc & 1);
#pragma unroll
		for (int s = 0; s < 8; ++s) wf[s] = frag(B1 + s);
		OB_SB();

		// ------------------------------------------------------------------------------------------------ loss, context matrices, out
		const h8 ov = pack8(o, 0); // element g: output 2 g + h
		h8 dyf = h8{0, 0, 0, 0, 0, 0, 0, 0};
		{
			float value[2];
			half_t grad[2];
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				loss_l2_fused<LOSS == 2>((float)ov[r], in.t[r], lsc, value[r], grad[r]); // l2.h:40-74 / relative_l2.h:40-75 on one refined reciprocal (mlp_device.h)
				const bool live = 2 * r + h < a.dims;
				if (!live) { value[r] = 0.0f; grad[r] = (half_t)0.0f; }
				dyf[r] = grad[r];
			}
			// the padded matrices [n][16]: outputs 0..3 come from the two lanes of the sample (this lane: 2 r + h), the rest is zero.
			// lanes h = 0 collect them; every lane stores its half of the row
			typedef _Float16 h2 __attribute__((ext_vector_type(2)));
			const uint32_t gpk = __builtin_bit_cast(uint32_t, (h2{grad[0], grad[1]})); // (out h, out 2 + h)
			const auto sg = __builtin_amdgcn_permlane32_swap(gpk, gpk, false, false);    // lanes < 32: [0] own, [1] the partner's
			const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, value[0]), __builtin_bit_cast(uint32_t, value[0]), false, false);
			const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, value[1]), __builtin_bit_cast(uint32_t, value[1]), false, false);
			u32x4 grow = u32x4{0, 0, 0, 0}, l0 = u32x4{0, 0, 0, 0};
			if (h == 0) {
				grow[0] = __builtin_amdgcn_perm(sg[1], sg[0], 0x05040100u); // (g0, g1)
				grow[1] = __builtin_amdgcn_perm(sg[1], sg[0], 0x07060302u); // (g2, g3)
				l0 = u32x4{s0[0], s0[1], s1[0], s1[1]};                     // L0, L1, L2, L3
			}
			*(u32x4*)((char*)a.dL_dout + (size_t)blk * 1024 + row_off) = grow;
			float* lrow = (float*)((char*)a.L + (size_t)blk * 2048 + 2 * row_off);
			*(u32x4*)lrow = l0;
			*(u32x4*)(lrow + 4) = u32x4{0, 0, 0, 0};
		}
		{
			const u32x4 u = __builtin_bit_cast(u32x4, ov);
			uint32_t w[4];
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const auto sw = __builtin_amdgcn_permlane32_swap(u[k], u[k + 2], false, false);
				w[2 * k + 0] = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);
				w[2 * k + 1] = __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u);
			}
			*(u32x4*)((char*)a.out + (size_t)blk * 1024 + row_off) = u32x4{w[0], w[1], w[2], w[3]};
		}
		img_write(OBA_IMG_DY, 0, dyf);
		const h8 aY = tr_frag16(OBA_IMG_DY, 0);
		OB_SB();

		// ------------------------------------------------------------------------------------------------ backward chain and weight gradients
		f16v g0 = mfma32(wg[4], dyf, Z);
		f16v g1 = mfma32(wg[5], dyf, Z);
#pragma unroll
		for (int tc = 0; tc < 4; ++tc) mfma16_acc(wout[tc], aY, bH[tc]);
		h8 tB[4]; // H0^T, for dW1
#pragma unroll
		for (int k = 0; k < 4; ++k) tB[k] = tr_frag(OBA_IMG_H0 + 2048 * (k >> 1), k & 1);
		OB_SB();
		const h8 d10 = relu_bwd8(pack8(g0, 0), h10), d11 = relu_bwd8(pack8(g0, 1), h11);
		img_write(OBA_IMG_DH, 0, d10);
		img_write(OBA_IMG_DH, 1, d11);
		f16v e0 = mfma32(wf[0], d10, Z);
		e0 = mfma32(wf[1], d11, e0);
		f16v e1 = mfma32(wf[4], d10, Z);
		e1 = mfma32(wf[5], d11, e1);
		const h8 d12 = relu_bwd8(pack8(g1, 0), h12), d13 = relu_bwd8(pack8(g1, 1), h13);
		img_write(OBA_IMG_DH + 2048, 0, d12);
		img_write(OBA_IMG_DH + 2048, 1, d13);
		OB_SB();
		e0 = mfma32(wf[2], d12, e0);
		e0 = mfma32(wf[3], d13, e0);
		e1 = mfma32(wf[6], d12, e1);
		e1 = mfma32(wf[7], d13, e1);
		h8 tA[4]; // dH1^T
#pragma unroll
		for (int k = 0; k < 4; ++k) tA[k] = tr_frag(OBA_IMG_DH + 2048 * (k >> 1), k & 1);
		OB_SB();
		// dW1 = dH1 H0^T beside dH0 = (W1^T dH1) act'(H0)
#pragma unroll
		for (int tr = 0; tr < 2; ++tr)
#pragma unroll
			for (int tc = 0; tc < 2; ++tc) {
				mfma32_acc(w1acc[2 * tr + tc], tA[2 * tr + 0], tB[2 * tc + 0]);
				mfma32_acc(w1acc[2 * tr + tc], tA[2 * tr + 1], tB[2 * tc + 1]);
			}
		const h8 d00 = relu_bwd8(pack8(e0, 0), h00), d01 = relu_bwd8(pack8(e0, 1), h01), d02 = relu_bwd8(pack8(e1, 0), h02), d03 = relu_bwd8(pack8(e1, 1), h03);
		img_write(OBA_IMG_DH, 0, d00); // behind the reads of dH1 above: LDS operations of a wave execute in order
		img_write(OBA_IMG_DH, 1, d01);
		img_write(OBA_IMG_DH + 2048, 0, d02);
		img_write(OBA_IMG_DH + 2048, 1, d03);
#pragma unroll
		for (int k = 0; k < 4; ++k) tA[k] = tr_frag(OBA_IMG_DH + 2048 * (k >> 1), k & 1); // dH0^T
#pragma unroll
		for (int tc = 0; tc < 4; ++tc) { // X^T: wf / wg are free now
			wf[2 * tc] = tr_frag(OBA_IMG_X + 2048 * tc, 0);
			wf[2 * tc + 1] = tr_frag(OBA_IMG_X + 2048 * tc, 1);
		}
		OB_SB();
		// dW0 = dH0 X^T: 2 x 4 tiles
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
#pragma unroll
			for (int tr = 0; tr < 2; ++tr) {
				mfma32_acc(w0acc[4 * tr + tc], tA[2 * tr + 0], wf[2 * tc]);
				mfma32_acc(w0acc[4 * tr + tc], tA[2 * tr + 1], wf[2 * tc + 1]);
			}
		// the trip's window out of the input image again (behind the reads above)
		if (fast) {
#pragma unroll
			for (int o = 0; o < 5; ++o) *(half_t*)(smem + x_addr(64 * h + ((first_bin + o) & 63u))) = (half_t)0.0f;
		} else {
			for (uint32_t bin = 0; bin < 64; ++bin) *(half_t*)(smem + x_addr(64 * h + bin)) = (half_t)0.0f;
		}
		OB_SB();
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

	// ---- weight gradients: the 4 waves' accumulators -> the workgroup's slab, through LDS, in 2 passes of 26 register quads (every wave
	// dumps its quads, then wave w adds up the four copies of the quads q = w (mod 4) in a fixed tree and stores them)
	constexpr int NQ = 8 * 4 + 4 * 4 + 4, PASS = 26, N_PASS = NQ / PASS; // 52 quads
	static_assert(NQ == PASS * N_PASS, "");
	auto quad = [&](const int qi) -> f4 {
		if (qi < 32) { const int i = qi >> 2, qd = qi & 3; return f4{w0acc[i][4 * qd], w0acc[i][4 * qd + 1], w0acc[i][4 * qd + 2], w0acc[i][4 * qd + 3]}; }
		if (qi < 48) { const int i = (qi - 32) >> 2, qd = qi & 3; return f4{w1acc[i][4 * qd], w1acc[i][4 * qd + 1], w1acc[i][4 * qd + 2], w1acc[i][4 * qd + 3]}; }
		return wout[qi - 48];
	};
	f4* region = (f4*)smem; // [4 waves][26 quads][64 lanes]
	float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
	const uint32_t grp = lane >> 4, li = lane & 15;
	__syncthreads();
#pragma unroll
	for (int pass = 0; pass < N_PASS; ++pass) {
		f4* dst = region + (size_t)wave * (PASS * 64) + lane;
#pragma unroll
		for (int q = 0; q < PASS; ++q) dst[q * 64] = quad(pass * PASS + q);
		__syncthreads();
#pragma unroll
		for (int q = 0; q < PASS; ++q) {
			if ((uint32_t)(q & 3) != wave) continue; // wave-uniform
			const int qi = pass * PASS + q;
			const f4* src = region + (size_t)q * 64 + lane;
			const f4 r0 = src[0], r1 = src[PASS * 64], r2 = src[2 * PASS * 64], r3 = src[3 * PASS * 64];
			f4 sum;
#pragma unroll
			for (int e = 0; e < 4; ++e) sum[e] = (r0[e] + r1[e]) + (r2[e] + r3[e]);
			if (qi < 48) { // 32 x 32 tile, registers 4 qd .. 4 qd + 3: rows e + 8 qd + 4 h of the tile, column c
				const int qd = qi & 3;
				const bool l0 = qi < 32;
				const int i = l0 ? qi >> 2 : (qi - 32) >> 2;
				const uint32_t w_off = l0 ? a.w_off[0] : a.w_off[1], cols = l0 ? 128u : 64u;
				const uint32_t tr = l0 ? i >> 2 : i >> 1, tc = l0 ? i & 3 : i & 1;
#pragma unroll
				for (int e = 0; e < 4; ++e) slab[w_off + (32 * tr + e + 8 * qd + 4 * h) * cols + 32 * tc + c] = sum[e];
			} else { // dWout, 16 x 16 tile tc: register e of lane group grp is output 2 e + 8 (grp >> 1) + (grp & 1), column 16 tc + li
				const int tc = qi - 48;
#pragma unroll
				for (int e = 0; e < 4; ++e) slab[a.w_off[2] + (2 * e + 8 * (grp >> 1) + (grp & 1)) * 64 + 16 * tc + li] = sum[e];
			}
		}
		if (pass + 1 < N_PASS) __syncthreads();
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
}

#undef OB_SB

} // namespace

// TCNN_AMD_MLP_R32=0 keeps k_train.hip's kernel
static bool r32ob_enabled() { return switches().mlp_r32; }

bool mlp_train_r32ob_shape(const MlpDesc& d, uint32_t n, uint32_t n_bins, uint32_t n_dims) {
	if (!r32ob_enabled() || d.width != 64 || d.in_width != 128 || d.out_width != 16 || d.n_hidden != 2 || d.n_frags_r32 == 0) return false;
	if (d.activation != (uint32_t)Activation::ReLU || d.output_activation != (uint32_t)Activation::None) return false;
	return n > 0 && n % 32 == 0 && n <= (1u << 22) && n_bins == 64 && n_dims == 2;
}
uint32_t mlp_train_r32ob_grid(uint32_t n) { return std::max(1u, std::min(256u, div_round_up(n / 32, (uint32_t)OB_NW))); }

bool mlp_train_r32ob_applies(const MlpDesc& d, uint32_t n, const MlpOneBlobInput* oneblob, const float* data_pdf, const void* external_dL_dy, uint32_t dims, LossType loss, const void* out,
                             const void* dL_dx, const float* slabs) {
	return oneblob && mlp_train_r32ob_shape(d, n, oneblob->n_bins, oneblob->n_dims) && data_pdf == nullptr && external_dL_dy == nullptr && dims >= 1 && dims <= 4 &&
	       (loss == LossType::L2 || loss == LossType::RelativeL2) && out != nullptr && dL_dx == nullptr && slabs != nullptr;
}

void mlp_train_r32ob(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const MlpOneBlobInput& oneblob, const float* target, uint32_t dims, LossType loss, float loss_scale,
                     void* out, void* dL_dout, float* L, float* slabs, uint32_t n_params) {
	CHECK_THROW(slabs != nullptr && dL_dout != nullptr && L != nullptr && target != nullptr && out != nullptr);
	ObArgs a{};
	a.x = oneblob.x;
	a.target = target;
	a.out = (half_t*)out;
	a.dL_dout = (half_t*)dL_dout;
	a.L = L;
	a.slabs = slabs;
	a.image = (const h8*)((const char*)image + (size_t)(d.n_frags_fwd + d.n_frags_bwd) * 1024);
	a.n = n;
	a.dims = dims;
	a.n_params = n_params;
	for (int l = 0; l < 3; ++l) a.w_off[l] = d.layers[l].w_off;
	a.loss_scale = loss_scale;
	const uint32_t grid = mlp_train_r32ob_grid(n);
#ifdef TCNN_AMD_DEV // laboratory build (build.py --dev): in-kernel clocks of the 5th launch
	static const bool timing = getenv("TCNN_AMD_MLP_TIMING") != nullptr;
	static int timing_left = 5;
	if (timing && timing_left > 0) {
		HIP_CHECK_THROW(hipMalloc(&a.dbg, (size_t)grid * 48));
		HIP_CHECK_THROW(hipMemset(a.dbg, 0, (size_t)grid * 48));
	}
#else
	int timing_left = 0; (void)timing_left;
#endif
	auto go = [&](auto kernel, int lds_bytes) {
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(OB_NW * 64), lds_bytes, stream, a);
		HIP_CHECK_THROW(hipGetLastError());
	};
	// Which form: the shared tiles spare the final sum over waves (7.5 k clocks) and cost 1.25 k clocks more per trip (two barriers, the tile
	// products behind the chain instead of inside it) -- they win up to 4 trips per wave (131 072 samples), the per-wave accumulators beyond
	// (the reference's benchmark protocol at 2^21 samples: 6.8e9 against 5.95e9 samples/s).  TCNN_AMD_MLP_R32OB_FORM=shared|acc forces one.
	const uint32_t trips = div_round_up(n / 32, grid * OB_NW);
	bool shared = trips <= 4;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_MLP_R32OB_FORM")) shared = e[0] == 's';
#endif
	if (shared) {
		if (loss == LossType::L2) go(k_mlp_train_r32ob<1>, OB_LDS_BYTES);
		else go(k_mlp_train_r32ob<2>, OB_LDS_BYTES);
	} else {
		if (loss == LossType::L2) go(k_mlp_train_r32ob_acc<1>, OBA_LDS_BYTES);
		else go(k_mlp_train_r32ob_acc<2>, OBA_LDS_BYTES);
	}
	if (a.dbg) {
		std::vector<unsigned long long> hst((size_t)grid * 6);
		HIP_CHECK_THROW(hipMemcpy(hst.data(), a.dbg, hst.size() * 8, hipMemcpyDeviceToHost));
		if (--timing_left == 0) {
			double fill = 0, loop = 0, tail = 0;
			for (uint32_t g = 0; g < grid; ++g) {
				fill += (double)(hst[g * 4 + 1] - hst[g * 4]);
				loop += (double)(hst[g * 4 + 2] - hst[g * 4 + 1]);
				tail += (double)(hst[g * 4 + 3] - hst[g * 4 + 2]);
			}
			fprintf(stderr, "k_mlp_train_r32ob%s wave 0 clocks, mean over %u workgroups: fill %.0f trips %.0f (%u blocks of 32 per wave) %s %.0f\n", shared ? "" : "_acc", grid, fill / grid,
			        loop / grid, trips, shared ? "slab stores" : "final sum", tail / grid);
			unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
			for (uint32_t g = 0; g < grid; ++g) {
				const unsigned long long st = hst[(size_t)grid * 4 + g * 2], en = hst[(size_t)grid * 4 + g * 2 + 1];
				s0 = std::min(s0, st); s1 = std::max(s1, st);
				e0 = std::min(e0, en); e1 = std::max(e1, en);
			}
			if (shared) fprintf(stderr, "  workgroup starts spread over %.2f us, ends from %.2f to %.2f us after the first start\n", (s1 - s0) * 0.01, (e0 - s0) * 0.01, (e1 - s0) * 0.01);
		}
		(void)hipFree(a.dbg);
	}
}

} // namespace tcnn_amd
