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
#include "mlp_device.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct R32Args {
	const half_t* x;        // level planes half2 [16][n]
	const float* target;    // [n][dims]
	half_t* out;            // [n][16]
	half_t* dL_dout;        // compact [n][dims]
	float* L;               // compact [n][dims]
	u32x4* rec;             // scatter records [8][n]: {x, y, gradients of levels 2 p, 2 p + 1}
	const float* rec_x;     // [n][2]
	float* slabs;           // [gridDim.x][n_params]
	const h8* image;        // R32 fragments
	uint32_t n, dims, n_params;
	uint32_t w_off[3];      // element offsets of W0, W1, Wout inside a slab
	float loss_scale;
	unsigned long long* dbg; // TCNN_AMD_MLP_TIMING: per workgroup wave 0's clock at start / loop start / loop end / end
};

constexpr int R32_NW = 8;                 // waves per workgroup
constexpr int R32_NF = 30;                // weight fragments (R32Frags of 32 -> 64 -> 64 -> 16)
constexpr int R32_ZERO = R32_NF * 1024;   // 1 KiB of zeros (the upper half of the 16-position dL/doutput image)
constexpr int R32_WAVE0 = R32_ZERO + 1024;
constexpr int R32_WAVE_BYTES = 15 * 1024; // X 2 K | H0 4 K | H1 4 K | dH 4 K | dY 1 K
constexpr int IMG_X = 0, IMG_H0 = 2048, IMG_H1 = 6144, IMG_DH = 10240, IMG_DY = 14336;
constexpr int R32_LDS_BYTES = R32_WAVE0 + R32_NW * R32_WAVE_BYTES; // 154 624
constexpr int R32_NT = 8;                 // weight-gradient tiles: dW0 [2][1], dW1 [2][2], dWout [1][2]

__device__ inline f16v mfma32(const h8 a, const h8 b, const f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ inline f16v zero16() {
	f16v z;
#pragma unroll
	for (int i = 0; i < 16; ++i) z[i] = 0.0f;
	return z;
}
// result registers 8 s .. 8 s + 7 -> the B fragment of the next layer's k-step (round to nearest even, like the reference's fp16 accumulators are read)
__device__ inline h8 pack8(const f16v v, const int s) {
	const f8v t = {v[8 * s + 0], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6], v[8 * s + 7]};
	return __builtin_convertvector(t, h8);
}
// ReLU(half) = x > 0 ? x : 0 (common_device.h:92-98) as a signed integer maximum of the bit patterns (never -0: the backward pass tells
// "positive" from "zero" by the bits)
__device__ inline h8 relu8(const h8 v) { return __builtin_bit_cast(h8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), s16x8{0, 0, 0, 0, 0, 0, 0, 0})); }
// ... and its derivative from the forward output (common_device.h:241-297): the gradient where the output has any bit set, +0 elsewhere
// (min(bits, 1) = 0 / 1, times the gradient's bits as an integer product)
__device__ inline h8 relu_bwd8(const h8 g, const h8 fwd) {
	const u32x4 f = __builtin_bit_cast(u32x4, fwd), gb = __builtin_bit_cast(u32x4, g);
	u32x4 r;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		uint32_t m, o;
		asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(m) : "v"(f[i]));
		asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o) : "v"(gb[i]), "v"(m));
		r[i] = o;
	}
	return __builtin_bit_cast(h8, r);
}
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
// DIAG (timing-only builds, TCNN_AMD_MLP_DIAG; results are wrong): bit 0: weight fragments are not read from LDS, bit 1: the transposing
// reads are not done, bit 2: the image writes are not done, bit 3: the loss is not evaluated
template <int LOSS, int DIAG = 0>
__global__ void __launch_bounds__(R32_NW * 64, 2) k_mlp_train_r32(const R32Args a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t c = lane & 31, h = lane >> 5;
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();

	const uint32_t n_blocks = a.n / 32;
	const uint32_t first = blockIdx.x * R32_NW + wave, step = gridDim.x * R32_NW;
	const uint32_t n_total = a.n * a.dims; // loss normalisation (relative_l2.h:58)
	const uint32_t n4 = a.n * 4;

	// ---- global addressing: wave-uniform base of the trip's 32-sample block + a lane offset that never changes
	const uint32_t x_off = (4 * h * a.n + c) * 4; // levels 8 s + 4 h + i at + (8 s + i) n 4
	struct In { h8 x[2]; float t[2]; float2 xs; };
	uint32_t t_off[2];
#pragma unroll
	for (int r = 0; r < 2; ++r) t_off[r] = (c * a.dims + min(2 * r + h, a.dims - 1)) * 4; // outputs >= dims re-read the last one (masked where used)
	auto load_in = [&](const uint32_t blk) -> In {
		In r;
		const char* xb = (const char*)a.x + (size_t)blk * 128;
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			u32x4 v;
#pragma unroll
			for (int i = 0; i < 4; ++i) v[i] = ld32<uint32_t>(xb + (size_t)n4 * (8 * s + i), x_off); // features 16 s + 8 h + 2 i, + 1
			r.x[s] = __builtin_bit_cast(h8, v);
		}
		const char* tb = (const char*)a.target + (size_t)blk * (128 * a.dims);
		r.t[0] = ld32<float>(tb, t_off[0]);
		r.t[1] = ld32<float>(tb, t_off[1]);
		r.xs = ld32<float2>((const char*)a.rec_x + (size_t)blk * 256, c * 8);
		return r;
	};
	In pre{};
	if (first < n_blocks) pre = load_in(first);

	// ---- weight fragments into LDS, the zero block, nothing else to prepare
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
		if (tid < 64) ((h8*)(smem + R32_ZERO))[tid] = h8{0, 0, 0, 0, 0, 0, 0, 0};
	}
	__syncthreads();
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();

	// ---- LDS addressing.  Weight fragment f: f KiB + 16 lane.  An opaque per-trip copy of the lane offset keeps the (loop-invariant)
	// fragment reads inside the trip loop, next to their uses (hoisted they would need 120 registers).
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
	//   columns 4 p .. 4 p + 3, and receives column (lane & 15) of the 4 rows): operand fragment of sample k-step s', element j =
	//   sample 16 s' + 8 hh + j of feature (lane & 31), hh = lane >> 5 -- two reads (e = 0, 1) of 4 samples each.
	uint32_t r_tr[4], r_dy[4]; // [2 s' + e]; r_dy: the 16-position image, lanes of the features 16..31 read the zero block
	{
		const uint32_t grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, hh = grp >> 1;
		const uint32_t g = 4 * (grp & 1) + p;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = 16 * (k >> 1) + 8 * hh + 4 * (k & 1) + q;
			r_tr[k] = wbase + g * 256 + ((row + 4 * g) & 31) * 8;
			r_dy[k] = (grp & 1) ? (uint32_t)R32_ZERO + li * 8 : wbase + IMG_DY + g * 256 + ((row + 4 * g) & 31) * 8;
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
	auto tr_frag = [&](const int img, const uint32_t (&r)[4], const int sp) -> h8 {
		if constexpr (DIAG & 2) { asm volatile("" : "+v"(fake)); return fake; }
		const h4 lo = lds_read_tr((const half_t*)(smem + r[2 * sp + 0] + img)), hi = lds_read_tr((const half_t*)(smem + r[2 * sp + 1] + img));
		return h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	};

	// ---- lane offsets of the stores
	const uint32_t cg_off0 = (c * a.dims + h) * 2;              // compact dL_dout: output 2 r + h (+ 4 r bytes); compact L: twice that
	const uint32_t o_off = c * 32 + h * 16;                     // out [n][16] halves: this lane stores the row's half h (16 bytes)
	const uint32_t rec_off = (h * a.n + c) * 16;                // records: level pair 2 g + h at + g 2 n 16

	f16v wacc[R32_NT];
#pragma unroll
	for (int i = 0; i < R32_NT; ++i) wacc[i] = zero16();

	// fragment slots (R32Frags of this network)
	constexpr int F0 = 0, F1 = 4, FO = 12, BO = 16, B1 = 18, B0 = 26;

	for (uint32_t blk = first; blk < n_blocks; blk += step) {
		asm volatile("" : "+v"(lane16));
		const In in = pre;
		{ // unconditionally (the last trip re-reads its own block): no branch between a trip's loads and its stores, so that the
		  // compiler's wait for these loads at the top of the next trip is a counted one that leaves the stores in flight
			const uint32_t next = min(blk + step, n_blocks - 1);
			pre = load_in(next);
		}

		// =============================================================== forward
		img_write(IMG_X, w_nat, 0, in.x[0]);
		img_write(IMG_X, w_nat, 1, in.x[1]);
		f16v acc[2];
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			acc[t] = mfma32(frag(F0 + 2 * t + 0), in.x[0], zero16());
			acc[t] = mfma32(frag(F0 + 2 * t + 1), in.x[1], acc[t]);
		}
		h8 hf[4]; // chain fragments of the current layer's output, k-step 2 t + s
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				hf[2 * t + s] = relu8(pack8(acc[t], s));
				img_write(IMG_H0 + 2048 * t, w_chain, s, hf[2 * t + s]);
			}
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			acc[t] = mfma32(frag(F1 + 4 * t + 0), hf[0], zero16());
#pragma unroll
			for (int ks = 1; ks < 4; ++ks) acc[t] = mfma32(frag(F1 + 4 * t + ks), hf[ks], acc[t]);
		}
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				hf[2 * t + s] = relu8(pack8(acc[t], s));
				img_write(IMG_H1 + 2048 * t, w_chain, s, hf[2 * t + s]);
			}

		// =============================================================== output layer + loss on the result tile
		h8 ov;
		{
			f16v o = mfma32(frag(FO + 0), hf[0], zero16());
#pragma unroll
			for (int ks = 1; ks < 4; ++ks) o = mfma32(frag(FO + ks), hf[ks], o);
			ov = pack8(o, 0); // element g: output 2 g + h (output activation None)
		}
		h8 dyf = h8{0, 0, 0, 0, 0, 0, 0, 0}; // dL/doutput, the B fragment of the first backward product (k = position)
		{
			// l2.h:40-74 / relative_l2.h:40-75, the same operations in the same order; values and gradients of the live outputs go
			// to the compact context matrices [n][dims]
			char* gb = (char*)a.dL_dout + (size_t)blk * (64 * a.dims);
			char* lb = (char*)a.L + (size_t)blk * (128 * a.dims);
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const float prediction = (float)ov[r];
				const float difference = prediction - in.t[r];
				float value, gradient;
				if constexpr (DIAG & 8) {
					value = difference;
					gradient = prediction;
				} else if constexpr (LOSS == 2) {
					const float prediction_sq_plus_epsilon = prediction * prediction + 0.01f;
					value = difference * difference / prediction_sq_plus_epsilon / n_total;
					gradient = 2 * difference / prediction_sq_plus_epsilon;
				} else {
					value = difference * difference / n_total;
					gradient = 2 * difference;
				}
				const half_t grad = (half_t)(a.loss_scale * gradient / n_total);
				const bool live = 2 * r + h < a.dims;
				dyf[r] = live ? grad : (half_t)0.0f;
				if (live) st32_stream(gb + 4 * r, cg_off0, grad);
				if (live) st32_stream(lb + 8 * r, 2 * cg_off0, value);
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
			st32_stream((char*)a.out + (size_t)blk * 1024, o_off, u32x4{w[0], w[1], w[2], w[3]});
		}
		img_write(IMG_DY, w_chain, 0, dyf); // positions 4 h .. 4 h + 3 (plane h) and 8 + 4 h .. (plane 2 + h)

		// =============================================================== backward chain and weight gradients
		// dWout = dY H1^T: rows = positions, columns = the 64 hidden features
		{
			const h8 a0 = tr_frag(0, r_dy, 0), a1 = tr_frag(0, r_dy, 1);
#pragma unroll
			for (int tc = 0; tc < 2; ++tc) {
				wacc[6 + tc] = mfma32(a0, tr_frag(IMG_H1 + 2048 * tc, r_tr, 0), wacc[6 + tc]);
				wacc[6 + tc] = mfma32(a1, tr_frag(IMG_H1 + 2048 * tc, r_tr, 1), wacc[6 + tc]);
			}
		}
#pragma unroll
		for (int t = 0; t < 2; ++t) acc[t] = mfma32(frag(BO + t), dyf, zero16());
		h8 gf[4];
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				gf[2 * t + s] = relu_bwd8(pack8(acc[t], s), hf[2 * t + s]); // times act'(H1) from the forward OUTPUT (common_device.h:241-297)
				img_write(IMG_DH + 2048 * t, w_chain, s, gf[2 * t + s]);
			}
		// dW1 = dH1 H0^T
#pragma unroll
		for (int tr = 0; tr < 2; ++tr) {
			const h8 a0 = tr_frag(IMG_DH + 2048 * tr, r_tr, 0), a1 = tr_frag(IMG_DH + 2048 * tr, r_tr, 1);
#pragma unroll
			for (int tc = 0; tc < 2; ++tc) {
				wacc[2 + 2 * tr + tc] = mfma32(a0, tr_frag(IMG_H0 + 2048 * tc, r_tr, 0), wacc[2 + 2 * tr + tc]);
				wacc[2 + 2 * tr + tc] = mfma32(a1, tr_frag(IMG_H0 + 2048 * tc, r_tr, 1), wacc[2 + 2 * tr + tc]);
			}
		}
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			acc[t] = mfma32(frag(B1 + 4 * t + 0), gf[0], zero16());
#pragma unroll
			for (int ks = 1; ks < 4; ++ks) acc[t] = mfma32(frag(B1 + 4 * t + ks), gf[ks], acc[t]);
		}
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				gf[2 * t + s] = relu_bwd8(pack8(acc[t], s), img_own(IMG_H0 + 2048 * t, s));
				img_write(IMG_DH + 2048 * t, w_chain, s, gf[2 * t + s]); // behind the reads of dH1 above: LDS operations of a wave execute in order
			}
		// dW0 = dH0 X^T
		{
			const h8 b0 = tr_frag(IMG_X, r_tr, 0), b1 = tr_frag(IMG_X, r_tr, 1);
#pragma unroll
			for (int tr = 0; tr < 2; ++tr) {
				wacc[tr] = mfma32(tr_frag(IMG_DH + 2048 * tr, r_tr, 0), b0, wacc[tr]);
				wacc[tr] = mfma32(tr_frag(IMG_DH + 2048 * tr, r_tr, 1), b1, wacc[tr]);
			}
		}
		// dX = W0^T dH0 -> scatter records {x, y, gradients of levels 2 p, 2 p + 1}: registers 4 g .. 4 g + 3 are features 8 g + 4 h .. + 3,
		// i.e. level pair p = 2 g + h
		{
			f16v o = mfma32(frag(B0 + 0), gf[0], zero16());
#pragma unroll
			for (int ks = 1; ks < 4; ++ks) o = mfma32(frag(B0 + ks), gf[ks], o);
			const u32x4 lo = __builtin_bit_cast(u32x4, pack8(o, 0)), hi = __builtin_bit_cast(u32x4, pack8(o, 1));
			const uint32_t x0 = __builtin_bit_cast(uint32_t, in.xs.x), x1 = __builtin_bit_cast(uint32_t, in.xs.y);
			char* rb = (char*)a.rec + (size_t)blk * 512;
			const size_t pair2 = (size_t)a.n * 32; // two level pairs further
			st32(rb, rec_off, u32x4{x0, x1, lo[0], lo[1]});
			st32(rb + pair2, rec_off, u32x4{x0, x1, lo[2], lo[3]});
			st32(rb + 2 * pair2, rec_off, u32x4{x0, x1, hi[0], hi[1]});
			st32(rb + 3 * pair2, rec_off, u32x4{x0, x1, hi[2], hi[3]});
		}
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

	// ---- weight gradients: 8 waves -> 4 -> the slab, through LDS (the images and fragments are dead: barrier first), in a fixed order
	//   waves 4..7 dump, waves 0..3 add region w in place; then wave w sums the four regions of tile w and stores it into the slab
	__syncthreads();
	f4* region = (f4*)smem; // [4 waves][8 tiles][4 register quads][64 lanes]
	if (wave >= 4) {
		f4* dst = region + (size_t)(wave - 4) * (R32_NT * 4 * 64) + lane;
#pragma unroll
		for (int i = 0; i < R32_NT; ++i)
#pragma unroll
			for (int qd = 0; qd < 4; ++qd) dst[(i * 4 + qd) * 64] = f4{wacc[i][4 * qd], wacc[i][4 * qd + 1], wacc[i][4 * qd + 2], wacc[i][4 * qd + 3]};
	}
	__syncthreads();
	if (wave < 4) {
		f4* reg = region + (size_t)wave * (R32_NT * 4 * 64) + lane;
#pragma unroll
		for (int i = 0; i < R32_NT; ++i)
#pragma unroll
			for (int qd = 0; qd < 4; ++qd) {
				const f4 v = reg[(i * 4 + qd) * 64];
				reg[(i * 4 + qd) * 64] = f4{wacc[i][4 * qd] + v[0], wacc[i][4 * qd + 1] + v[1], wacc[i][4 * qd + 2] + v[2], wacc[i][4 * qd + 3] + v[3]}; // own slot: no hazard
			}
	}
	__syncthreads();
	{
		float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
		// tile `wave`: 0, 1: dW0 rows 32 tr (32 columns); 2..5: dW1 tile (tr, tc); 6, 7: dWout columns 32 tc, rows = positions (registers 0..7: output 2 g + h)
		const uint32_t i = wave;
		const uint32_t w_off = i < 2 ? a.w_off[0] : i < 6 ? a.w_off[1] : a.w_off[2];
		const uint32_t cols = i < 2 ? 32u : 64u;
		const uint32_t tr = i < 2 ? i : i < 6 ? (i - 2) >> 1 : 0u, tc = i < 2 ? 0u : i < 6 ? (i - 2) & 1u : i - 6;
#pragma unroll
		for (int qd = 0; qd < 4; ++qd) {
			if (i >= 6 && qd >= 2) break; // positions 16..31: no outputs
			const f4* src = region + (size_t)(i * 4 + qd) * 64 + lane;
			const f4 r0 = src[0], r1 = src[R32_NT * 4 * 64], r2 = src[2 * R32_NT * 4 * 64], r3 = src[3 * R32_NT * 4 * 64];
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const uint32_t g = 4 * qd + e;
				const uint32_t row = i >= 6 ? 2 * g + h : 32 * tr + (g & 3) + 8 * (g >> 2) + 4 * h;
				slab[w_off + row * cols + 32 * tc + c] = (r0[e] + r1[e]) + (r2[e] + r3[e]);
			}
		}
	}
	if (a.dbg && tid == 0) a.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
}

} // namespace

// TCNN_AMD_MLP_R32=0 keeps k_train_regs.hip's kernel (A/B runs; read per call so that tests can cover both forms in one process)
static bool r32_enabled() {
	const char* e = getenv("TCNN_AMD_MLP_R32");
	return !(e && e[0] == '0');
}

// the one shape this kernel is instantiated for, with the formats of the grid encoding's training step: input as level planes of 2
// features, <= 4 outputs, ReLU, L2 / RelativeL2, `out` and 2-D scatter records written, no data_pdf
bool mlp_train_r32_applies(const MlpDesc& d, uint32_t n, uint32_t x_plane_features, const float* data_pdf, const void* external_dL_dy, uint32_t dims, LossType loss, const void* out,
                           const void* dL_dx, uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims) {
	if (!r32_enabled() || !r32_shape_ok(d) || d.in_width != 32 || d.n_hidden != 2 || d.n_frags_r32 != (uint32_t)R32_NF) return false;
	if (d.activation != (uint32_t)Activation::ReLU || d.output_activation != (uint32_t)Activation::None) return false;
	if (n == 0 || n % 32 != 0 || n > (1u << 22)) return false; // 32-bit byte offsets into [n][...] matrices
	return x_plane_features == 2 && data_pdf == nullptr && external_dL_dy == nullptr && dims >= 1 && dims <= 4 && (loss == LossType::L2 || loss == LossType::RelativeL2) && out != nullptr &&
	       dL_dx != nullptr && dx_plane_features == 2 && dx_record_x != nullptr && dx_record_dims == 2;
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
	static const bool timing = getenv("TCNN_AMD_MLP_TIMING") != nullptr;
	static int timing_left = 5;
	if (timing && timing_left > 0) {
		HIP_CHECK_THROW(hipMalloc(&a.dbg, (size_t)grid * 32));
		HIP_CHECK_THROW(hipMemset(a.dbg, 0, (size_t)grid * 32));
	}
	auto go = [&](auto kernel) {
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R32_LDS_BYTES));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(R32_NW * 64), R32_LDS_BYTES, stream, a);
		HIP_CHECK_THROW(hipGetLastError());
	};
	static const int diag = getenv("TCNN_AMD_MLP_DIAG") ? atoi(getenv("TCNN_AMD_MLP_DIAG")) : 0;
	if (diag == 1) go(k_mlp_train_r32<2, 1>);
	else if (diag == 2) go(k_mlp_train_r32<2, 2>);
	else if (diag == 3) go(k_mlp_train_r32<2, 3>);
	else if (diag == 7) go(k_mlp_train_r32<2, 7>);
	else if (diag == 8) go(k_mlp_train_r32<2, 8>);
	else if (diag == 15) go(k_mlp_train_r32<2, 15>);
	else if (loss == LossType::L2) go(k_mlp_train_r32<1>);
	else go(k_mlp_train_r32<2>);
	if (a.dbg) {
		std::vector<unsigned long long> hst((size_t)grid * 4);
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
		}
		(void)hipFree(a.dbg);
	}
}

} // namespace tcnn_amd
