// k_train_r32w.hip -- the training step's MLP part for 128-wide networks fed by a grid encoding with 4 features per level:
// 64 -> 128 -> 128 -> 16 (BASELINE config 5), on v_mfma_f32_32x32x16_f16.
//
// Same job as k_mlp_train<128, ...> (k_train.hip; reference: src/fully_fused_mlp.cu:500-557 forward with the WIDTH = 128 instance of
// :47-129, losses/{l2,relative_l2}.h:40-75, fully_fused_mlp.cu:151-259 backward, :785-828 the weight-gradient GEMMs), shaped like
// k_train_r32.hip -- register chain, 32 samples per wave and trip, every activation / gradient tile written once into a wave's LDS image
// and read back transposed for the weight gradients -- with what width 128 changes:
//   * the weight gradients are 8 + 16 tiles of 32 x 32 and 8 of 16 x 16: 416 accumulator registers if every wave kept all of them.
//     The TILES are shared out instead: wave w of a workgroup's four owns row tile w of dW0 and dW1 and two column tiles of dWout
//     (104 registers, in AGPRs) and sums them over the samples of ALL four waves, reading every wave's images -- so the images are
//     exchanged under two workgroup barriers per trip, and no reduction across waves is left at the end: a wave stores its tiles
//     into the workgroup's slab as they stand;
//   * four waves x 37 KiB of images fill the LDS: the 108 KiB of weight fragments stay in L2 and are loaded per use, one group of eight
//     1-KiB fragments ahead of the eight matrix instructions that consume them (one wave per SIMD: 512 registers make room);
//   * input as level planes of 4 features, dL/dinput written back as level planes of 4 features (what the binned scatter of a 3-D grid
//     reads), context matrices in the reference's padded form [n][16].
#include "r32_device.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {

struct WArgs {
	const half_t* x;        // level planes half4 [16][n]
	const float* target;    // [n][dims]
	half_t* out;            // [n][16]
	half_t* dL_dout;        // [n][16]
	float* L;               // [n][16]
	half_t* dL_dx;          // level planes half4 [16][n]
	float* slabs;           // [gridDim.x][n_params]
	const h8* image;        // R32 fragments
	uint32_t n, dims, n_params;
	uint32_t w_off[3];
	float loss_scale;
	unsigned long long* dbg;
};

constexpr int W_NW = 4;
constexpr int W_IMG_X = 0, W_IMG_H0 = 4096, W_IMG_H1 = 12288, W_IMG_DH1 = 20480, W_IMG_DH0 = 28672, W_IMG_DY = 36864;
constexpr int W_WAVE_BYTES = 37 * 1024;
constexpr int W_LDS_BYTES = W_NW * W_WAVE_BYTES; // 151 552
// fragment slots (R32Frags of 64 -> 128 -> 128 -> 16)
constexpr int WF0 = 0, WF1 = 16, WFO = 48, WBO = 56, WB1 = 60, WB0 = 92, W_NFRAGS = 108;

__device__ inline void mfma32_acc(f16v& acc, const h8 a, const h8 b) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
__device__ inline void mfma16_acc(f4& acc, const h8 a, const h8 b) { asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }

#define W_SB() __builtin_amdgcn_sched_barrier(0)

// LOSS 1: L2, 2: RelativeL2
template <int LOSS>
__global__ void __launch_bounds__(W_NW * 64, 1) k_mlp_train_r32w(const WArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 31, h = lane >> 5;
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();

	const uint32_t n_blocks = a.n / 32;
	const uint32_t n_trips = (n_blocks + gridDim.x * W_NW - 1) / (gridDim.x * W_NW); // the same for every wave of every workgroup: the barriers are workgroup-wide
	const uint32_t n_total = a.n * a.dims;
	const LossScales lsc = loss_scales(n_total, a.loss_scale);
	const uint32_t n8 = a.n * 8; // bytes per level plane

	// weight fragment f: one 16-byte load per lane from the image (L2-resident: 108 KiB read by every wave of the chip)
	auto frag = [&](const int f) -> h8 { return a.image[f * 64 + lane]; };

	// images of wave v: see k_train_r32.hip (per 32-feature tile 2 KiB: plane g of 4 features at 256 g, sample n at 8 ((n + 4 g) & 31))
	const uint32_t wbase = wave * W_WAVE_BYTES;
	uint32_t w_chain[4], w_nat[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t gc = 4 * (k >> 1) + 2 * (k & 1) + h, gn = 4 * (k >> 1) + 2 * h + (k & 1);
		w_chain[k] = gc * 256 + ((c + 4 * gc) & 31) * 8;
		w_nat[k] = gn * 256 + ((c + 4 * gn) & 31) * 8;
	}
	uint32_t r_tr[4], r_16[4]; // transposing reads, relative to a tile (32x32x16 operand) / to the two halves of a tile (16x16x32 operand)
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
	auto img_write = [&](const uint32_t img, const uint32_t (&w)[4], const int s, const h8 v) { // img: byte offset of the tile in LDS
		*(h4*)(smem + img + w[2 * s + 0]) = h4{v[0], v[1], v[2], v[3]};
		*(h4*)(smem + img + w[2 * s + 1]) = h4{v[4], v[5], v[6], v[7]};
	};
	auto img_own = [&](const uint32_t img, const int s) -> h8 {
		const h4 lo = *(const h4*)(smem + img + w_chain[2 * s + 0]), hi = *(const h4*)(smem + img + w_chain[2 * s + 1]);
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr2 = [&](const uint32_t a0, const uint32_t a1) -> h8 {
		const h4 lo = lds_read_tr((const half_t*)(smem + a0)), hi = lds_read_tr((const half_t*)(smem + a1));
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};
	auto tr_frag = [&](const uint32_t img, const int sp) -> h8 { return tr2(img + r_tr[2 * sp], img + r_tr[2 * sp + 1]); };
	auto tr_frag16 = [&](const uint32_t img, const int half) -> h8 { return tr2(img + r_16[2 * half], img + r_16[2 * half + 1]); };

	const uint32_t x_off = (2 * h * a.n + c) * 8;  // input planes: levels 4 s + 2 h + i at + (4 s + i) n 8
	const uint32_t row_off = c * 32 + h * 16;      // [n][16] halves: this lane's half of the row; [n][16] floats: twice that
	const uint32_t dx_off = (h * a.n + c) * 8;     // dL/dinput planes: level 2 g + h (+ 16 ti) at + g 2 n 8

	// this wave's tiles: dW0 (row tile `wave`, column tiles 0, 1), dW1 (row tile `wave`, column tiles 0..3), dWout (16-column tiles 2 wave, 2 wave + 1)
	f16v w0acc[2], w1acc[4];
	f4 wout[2];
#pragma unroll
	for (int i = 0; i < 2; ++i) { w0acc[i] = zero16(); wout[i] = f4{0, 0, 0, 0}; }
#pragma unroll
	for (int i = 0; i < 4; ++i) w1acc[i] = zero16();
	const f16v Z = zero16();

	for (uint32_t trip = 0; trip < n_trips; ++trip) {
		const uint32_t blk = (trip * gridDim.x + blockIdx.x) * W_NW + wave;
		const bool valid = blk < n_blocks; // wave-uniform
		if (valid) {
			// ------------------------------------------------------------------------------------------------ inputs
			h8 xs[4];
			float tg[2];
			{
				const char* xb = (const char*)a.x + (size_t)blk * 256;
#pragma unroll
				for (int s = 0; s < 4; ++s) {
					typedef uint32_t u2 __attribute__((ext_vector_type(2)));
					const u2 lo = *(const u2*)(xb + (size_t)n8 * (4 * s) + x_off), hi = *(const u2*)(xb + (size_t)n8 * (4 * s + 1) + x_off);
					xs[s] = __builtin_bit_cast(h8, (u32x4{lo[0], lo[1], hi[0], hi[1]})); // features 16 s + 8 h + j
				}
				const char* tb = (const char*)a.target + (size_t)blk * (128 * a.dims);
#pragma unroll
				for (int r = 0; r < 2; ++r) tg[r] = *(const float*)(tb + (c * a.dims + min(2 * r + h, a.dims - 1)) * 4);
			}
#pragma unroll
			for (int s = 0; s < 4; ++s) img_write(wbase + W_IMG_X + 2048 * (s >> 1), w_nat, s & 1, xs[s]);

			// ------------------------------------------------------------------------------------------------ forward
			h8 wf[8], wg[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) wf[k] = frag(WF0 + k);
			h8 h0[8]; // H0 as chain fragments, k-step 2 t + s
#pragma unroll
			for (int tp = 0; tp < 2; ++tp) { // two row tiles per group of eight fragments
#pragma unroll
				for (int k = 0; k < 8; ++k) wg[k] = tp == 0 ? frag(WF0 + 8 + k) : frag(WF1 + k);
				W_SB();
#pragma unroll
				for (int tt = 0; tt < 2; ++tt) {
					const int t = 2 * tp + tt;
					f16v acc = mfma32(wf[4 * tt + 0], xs[0], Z);
#pragma unroll
					for (int s = 1; s < 4; ++s) acc = mfma32(wf[4 * tt + s], xs[s], acc);
					h0[2 * t] = relu8(pack8(acc, 0));
					h0[2 * t + 1] = relu8(pack8(acc, 1));
					img_write(wbase + W_IMG_H0 + 2048 * t, w_chain, 0, h0[2 * t]);
					img_write(wbase + W_IMG_H0 + 2048 * t, w_chain, 1, h0[2 * t + 1]);
				}
#pragma unroll
				for (int k = 0; k < 8; ++k) wf[k] = wg[k];
				W_SB();
			}
			// wf = layer 1, row tile 0
			h8 h1[8];
#pragma unroll
			for (int t = 0; t < 4; ++t) {
#pragma unroll
				for (int k = 0; k < 8; ++k) wg[k] = t < 3 ? frag(WF1 + 8 * (t + 1) + k) : frag(WFO + k);
				W_SB();
				f16v acc = mfma32(wf[0], h0[0], Z);
#pragma unroll
				for (int ks = 1; ks < 8; ++ks) acc = mfma32(wf[ks], h0[ks], acc);
				h1[2 * t] = relu8(pack8(acc, 0));
				h1[2 * t + 1] = relu8(pack8(acc, 1));
				img_write(wbase + W_IMG_H1 + 2048 * t, w_chain, 0, h1[2 * t]);
				img_write(wbase + W_IMG_H1 + 2048 * t, w_chain, 1, h1[2 * t + 1]);
#pragma unroll
				for (int k = 0; k < 8; ++k) wf[k] = wg[k];
				W_SB();
			}
			// wf = output layer
#pragma unroll
			for (int k = 0; k < 4; ++k) wg[k] = frag(WBO + k);
			W_SB();
			f16v o = mfma32(wf[0], h1[0], Z);
#pragma unroll
			for (int ks = 1; ks < 8; ++ks) o = mfma32(wf[ks], h1[ks], o);
#pragma unroll
			for (int k = 0; k < 8; ++k) wf[k] = frag(WB1 + k);
			W_SB();

			// ------------------------------------------------------------------------------------------------ loss, context matrices, out
			const h8 ov = pack8(o, 0); // element g: output 2 g + h
			h8 dyf = h8{0, 0, 0, 0, 0, 0, 0, 0};
			{
				float value[2];
				half_t grad[2];
#pragma unroll
				for (int r = 0; r < 2; ++r) {
					loss_l2_fused<LOSS == 2>((float)ov[r], tg[r], lsc, value[r], grad[r]); // l2.h:40-74 / relative_l2.h:40-75 on one refined reciprocal (mlp_device.h)
					const bool live = 2 * r + h < a.dims;
					if (!live) { value[r] = 0.0f; grad[r] = (half_t)0.0f; }
					dyf[r] = grad[r];
				}
				typedef _Float16 h2 __attribute__((ext_vector_type(2)));
				const uint32_t gpk = __builtin_bit_cast(uint32_t, (h2{grad[0], grad[1]}));
				const auto sg = __builtin_amdgcn_permlane32_swap(gpk, gpk, false, false);
				const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, value[0]), __builtin_bit_cast(uint32_t, value[0]), false, false);
				const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, value[1]), __builtin_bit_cast(uint32_t, value[1]), false, false);
				u32x4 grow = u32x4{0, 0, 0, 0}, l0 = u32x4{0, 0, 0, 0};
				if (h == 0) {
					grow[0] = __builtin_amdgcn_perm(sg[1], sg[0], 0x05040100u);
					grow[1] = __builtin_amdgcn_perm(sg[1], sg[0], 0x07060302u);
					l0 = u32x4{s0[0], s0[1], s1[0], s1[1]};
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
			img_write(wbase + W_IMG_DY, w_chain, 0, dyf);
			W_SB();

			// ------------------------------------------------------------------------------------------------ backward chain
			// dH1 = (Wout^T dY) act'(H1) (common_device.h:241-297: from the forward OUTPUT)
			h8 d1[8];
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const f16v g = mfma32(wg[t], dyf, Z);
				d1[2 * t] = relu_bwd8(pack8(g, 0), h1[2 * t]);
				d1[2 * t + 1] = relu_bwd8(pack8(g, 1), h1[2 * t + 1]);
				img_write(wbase + W_IMG_DH1 + 2048 * t, w_chain, 0, d1[2 * t]);
				img_write(wbase + W_IMG_DH1 + 2048 * t, w_chain, 1, d1[2 * t + 1]);
			}
			// dH0 = (W1^T dH1) act'(H0); wf = W1^T row tile 0
			h8 d0[8];
#pragma unroll
			for (int t = 0; t < 4; ++t) {
#pragma unroll
				for (int k = 0; k < 8; ++k) wg[k] = t < 3 ? frag(WB1 + 8 * (t + 1) + k) : frag(WB0 + k);
				W_SB();
				f16v acc = mfma32(wf[0], d1[0], Z);
#pragma unroll
				for (int ks = 1; ks < 8; ++ks) acc = mfma32(wf[ks], d1[ks], acc);
				d0[2 * t] = relu_bwd8(pack8(acc, 0), h0[2 * t]);
				d0[2 * t + 1] = relu_bwd8(pack8(acc, 1), h0[2 * t + 1]);
				img_write(wbase + W_IMG_DH0 + 2048 * t, w_chain, 0, d0[2 * t]);
				img_write(wbase + W_IMG_DH0 + 2048 * t, w_chain, 1, d0[2 * t + 1]);
#pragma unroll
				for (int k = 0; k < 8; ++k) wf[k] = wg[k];
				W_SB();
			}
			// dX = W0^T dH0 -> level planes of 4 features: registers 4 g .. 4 g + 3 of row tile ti are features 32 ti + 8 g + 4 h .. + 3 = level 8 ti + 2 g + h
#pragma unroll
			for (int ti = 0; ti < 2; ++ti) {
				if (ti == 0) {
#pragma unroll
					for (int k = 0; k < 8; ++k) wg[k] = frag(WB0 + 8 + k);
				}
				W_SB();
				f16v acc = mfma32(wf[0], d0[0], Z);
#pragma unroll
				for (int ks = 1; ks < 8; ++ks) acc = mfma32(wf[ks], d0[ks], acc);
				const u32x4 lo = __builtin_bit_cast(u32x4, pack8(acc, 0)), hi = __builtin_bit_cast(u32x4, pack8(acc, 1));
				char* db = (char*)a.dL_dx + (size_t)blk * 256 + (size_t)n8 * (8 * ti) + dx_off;
				typedef uint32_t u2 __attribute__((ext_vector_type(2)));
				*(u2*)(db) = u2{lo[0], lo[1]};
				*(u2*)(db + (size_t)n8 * 2) = u2{lo[2], lo[3]};
				*(u2*)(db + (size_t)n8 * 4) = u2{hi[0], hi[1]};
				*(u2*)(db + (size_t)n8 * 6) = u2{hi[2], hi[3]};
#pragma unroll
				for (int k = 0; k < 8; ++k) wf[k] = wg[k];
				W_SB();
			}
		} else {
			// no block for this wave in the last trip: images of zeros make its share of every product vanish
			const h8 zero = h8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
			for (int t = 0; t < 4; ++t)
#pragma unroll
				for (int s = 0; s < 2; ++s) {
					img_write(wbase + W_IMG_DH1 + 2048 * t, w_chain, s, zero);
					img_write(wbase + W_IMG_DH0 + 2048 * t, w_chain, s, zero);
					img_write(wbase + W_IMG_H0 + 2048 * t, w_chain, s, zero); // a wave that never had a block: nothing but what the LDS held
					img_write(wbase + W_IMG_H1 + 2048 * t, w_chain, s, zero);
					if (t < 2) img_write(wbase + W_IMG_X + 2048 * t, w_chain, s, zero);
				}
			img_write(wbase + W_IMG_DY, w_chain, 0, zero);
		}
		__syncthreads();

		// ---------------------------------------------------------------------------------------------------- weight gradients: this wave's tiles
		// over the 4 x 32 samples of the workgroup's trip.  v: the wave whose images are read.
#pragma unroll
		for (int v = 0; v < W_NW; ++v) {
			const uint32_t vb = v * W_WAVE_BYTES;
			// dW1 row tile `wave` = dH1 (tile wave) H0^T; dW0 row tile `wave` = dH0 (tile wave) X^T
			const h8 a10 = tr_frag(vb + W_IMG_DH1 + 2048 * wave, 0), a11 = tr_frag(vb + W_IMG_DH1 + 2048 * wave, 1);
			const h8 a00 = tr_frag(vb + W_IMG_DH0 + 2048 * wave, 0), a01 = tr_frag(vb + W_IMG_DH0 + 2048 * wave, 1);
			h8 b0[4], b1[4];
#pragma unroll
			for (int tc = 0; tc < 4; ++tc) {
				b0[tc] = tr_frag(vb + W_IMG_H0 + 2048 * tc, 0);
				b1[tc] = tr_frag(vb + W_IMG_H0 + 2048 * tc, 1);
			}
			const h8 x00 = tr_frag(vb + W_IMG_X, 0), x01 = tr_frag(vb + W_IMG_X, 1), x10 = tr_frag(vb + W_IMG_X + 2048, 0), x11 = tr_frag(vb + W_IMG_X + 2048, 1);
			// dWout (positions x 16 hidden features), column tiles 2 wave, 2 wave + 1: features 32 wave + 16 i of H1 = tile `wave`, half i
			const h8 aY = tr_frag16(vb + W_IMG_DY, 0);
			const h8 y0 = tr_frag16(vb + W_IMG_H1 + 2048 * wave, 0), y1 = tr_frag16(vb + W_IMG_H1 + 2048 * wave, 1);
#pragma unroll
			for (int tc = 0; tc < 4; ++tc) {
				mfma32_acc(w1acc[tc], a10, b0[tc]);
				mfma32_acc(w1acc[tc], a11, b1[tc]);
			}
			mfma32_acc(w0acc[0], a00, x00);
			mfma32_acc(w0acc[0], a01, x01);
			mfma32_acc(w0acc[1], a00, x10);
			mfma32_acc(w0acc[1], a01, x11);
			mfma16_acc(wout[0], aY, y0);
			mfma16_acc(wout[1], aY, y1);
		}
		__syncthreads(); // before the next trip overwrites the images
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

	// ---- this wave's tiles into the workgroup's slab: no other wave holds a share of them
	{
		float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
		const uint32_t grp = lane >> 4, li = lane & 15;
		// 32 x 32 tile, register g: row (g & 3) + 8 (g >> 2) + 4 h of the tile, column c
#pragma unroll
		for (int tc = 0; tc < 2; ++tc)
#pragma unroll
			for (int g = 0; g < 16; ++g) slab[a.w_off[0] + (32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h) * 64 + 32 * tc + c] = w0acc[tc][g];
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
#pragma unroll
			for (int g = 0; g < 16; ++g) slab[a.w_off[1] + (32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h) * 128 + 32 * tc + c] = w1acc[tc][g];
		// dWout 16 x 16 tile: register e of lane group grp is output 2 e + 8 (grp >> 1) + (grp & 1), column 16 (2 wave + i) + li
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int e = 0; e < 4; ++e) slab[a.w_off[2] + (2 * e + 8 * (grp >> 1) + (grp & 1)) * 128 + 16 * (2 * wave + i) + li] = wout[i][e];
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
}
#undef W_SB

} // namespace

static bool r32w_enabled() { return switches().mlp_r32; }

bool mlp_train_r32w_shape(const MlpDesc& d, uint32_t n) {
	if (!r32w_enabled() || d.width != 128 || d.in_width != 64 || d.out_width != 16 || d.n_hidden != 2 || d.n_frags_r32 != (uint32_t)W_NFRAGS) return false;
	if (d.activation != (uint32_t)Activation::ReLU || d.output_activation != (uint32_t)Activation::None) return false;
	return n > 0 && n % 32 == 0 && n <= (1u << 22);
}

bool mlp_train_r32w_applies(const MlpDesc& d, uint32_t n, uint32_t x_plane_features, const float* data_pdf, const void* external_dL_dy, uint32_t dims, LossType loss, const void* out,
                            const void* dL_dx, uint32_t dx_plane_features, const float* dx_record_x, const float* slabs, bool oneblob) {
	return mlp_train_r32w_shape(d, n) && !oneblob && x_plane_features == 4 && data_pdf == nullptr && external_dL_dy == nullptr && dims >= 1 && dims <= 4 &&
	       (loss == LossType::L2 || loss == LossType::RelativeL2) && out != nullptr && dL_dx != nullptr && dx_plane_features == 4 && dx_record_x == nullptr && slabs != nullptr;
}

void mlp_train_r32w(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, const float* target, uint32_t dims, LossType loss, float loss_scale, void* out,
                    void* dL_dout, float* L, void* dL_dx, float* slabs, uint32_t n_params, uint32_t grid) {
	CHECK_THROW(grid >= 1 && slabs != nullptr && dL_dout != nullptr && L != nullptr && target != nullptr && out != nullptr && dL_dx != nullptr);
	WArgs a{};
	a.x = (const half_t*)x;
	a.target = target;
	a.out = (half_t*)out;
	a.dL_dout = (half_t*)dL_dout;
	a.L = L;
	a.dL_dx = (half_t*)dL_dx;
	a.slabs = slabs;
	a.image = (const h8*)((const char*)image + (size_t)(d.n_frags_fwd + d.n_frags_bwd) * 1024);
	a.n = n;
	a.dims = dims;
	a.n_params = n_params;
	for (int l = 0; l < 3; ++l) a.w_off[l] = d.layers[l].w_off;
	a.loss_scale = loss_scale;
#ifdef TCNN_AMD_DEV // laboratory build (build.py --dev): in-kernel clocks of the 5th launch
	static const bool timing = getenv("TCNN_AMD_MLP_TIMING") != nullptr;
	static int timing_left = 5;
	if (timing && timing_left > 0) {
		HIP_CHECK_THROW(hipMalloc(&a.dbg, (size_t)grid * 32));
		HIP_CHECK_THROW(hipMemset(a.dbg, 0, (size_t)grid * 32));
	}
#else
	int timing_left = 0; (void)timing_left;
#endif
	auto go = [&](auto kernel) {
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(W_NW * 64), W_LDS_BYTES, stream, a);
		HIP_CHECK_THROW(hipGetLastError());
	};
	if (loss == LossType::L2) go(k_mlp_train_r32w<1>);
	else go(k_mlp_train_r32w<2>);
	if (a.dbg) {
		std::vector<unsigned long long> hst((size_t)grid * 4);
		HIP_CHECK_THROW(hipMemcpy(hst.data(), a.dbg, hst.size() * 8, hipMemcpyDeviceToHost));
		if (--timing_left == 0) {
			double loop = 0, tail = 0;
			for (uint32_t g = 0; g < grid; ++g) {
				loop += (double)(hst[g * 4 + 2] - hst[g * 4]);
				tail += (double)(hst[g * 4 + 3] - hst[g * 4 + 2]);
			}
			fprintf(stderr, "k_mlp_train_r32w wave 0 clocks, mean over %u workgroups: trips %.0f (%u blocks of 32 per wave) slab stores %.0f\n", grid, loop / grid,
			        div_round_up(n / 32, grid * W_NW), tail / grid);
		}
		(void)hipFree(a.dbg);
	}
}

} // namespace tcnn_amd
