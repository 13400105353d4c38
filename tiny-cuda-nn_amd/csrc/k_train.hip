// k_train.hip -- the Trainer's hot path as ONE kernel: MLP forward -> loss -> backward -> weight gradients.
//
// Replaces, for trainer->training_step() (reference, /root/reference):
//   src/fully_fused_mlp.cu:500-557   kernel_mlp_fused (forward, stores every hidden activation to HBM)
//   include/tiny-cuda-nn/losses/{l2,relative_l2}.h:40-75   loss kernels (read prediction, write values + gradients)
//   src/fully_fused_mlp.cu:151-259   kernel_mlp_fused_backward (re-reads the activations, stores dL/dhidden to HBM)
//   src/fully_fused_mlp.cu:785,819,828 + cutlass_matmul.h:438-479   three split-K CUTLASS GEMMs that re-read all of it
// The reference's decomposition moves ~2 KB of HBM traffic per sample; here activations never leave the CU.
//
// Structure (one workgroup = NW waves, S = NW * NB * 16 samples per trip, persistent over the batch):
//   phase A, per wave, no synchronisation: the register-chained MFMA forward of k_mlp.hip (activations transposed, the
//     accumulator tile of one layer IS the B operand of the next), the loss on the accumulator tile of the output layer,
//     and the backward chain with the transposed weight images.  Every activation / gradient tile is also dropped into
//     LDS as [sample][feature] rows -- the only copy that ever exists.
//   barrier
//   phase B, cooperative: dW_l += dOut_l^T In_l over the S samples of the trip.  The MFMA operands need the SAMPLE index
//     on the k axis, i.e. the transposed view of those LDS rows: gfx950's ds_read_b64_tr_b16 delivers exactly that.
//     Output tiles are distributed over the waves; their accumulators stay in registers for the whole kernel.
//   barrier
//   At the end every workgroup writes one fp32 slab of partial weight gradients; k_wgrad_reduce sums the slabs in a fixed
//   order (bitwise reproducible, no atomics) and rounds to fp16.
#include "mlp_device.h"
#include "oneblob_device.h"

namespace tcnn_amd {
namespace {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL store of the wave
// (s_waitcnt vmcnt(0)); the trips' stores (outputs, loss values, scatter records) need no such wait -- nobody in the
// workgroup reads them -- and waiting for their write acknowledgements twice per trip is pure latency.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int TR_PAD = 8; // halves of row padding of the LDS images (rows stay 16-byte aligned)
// The hidden-activation / hidden-gradient images are only touched 8 bytes at a time (h4 rows in, ds_read_b64_tr_b16 out), so
// 128-wide networks pad them by 4: 8 waves x 16 samples then fit the 160 KB (2 waves per SIMD instead of 1).
constexpr int hs_pad(int width) { return width == 128 ? 4 : TR_PAD; }

struct TrainArgs {
	const half_t* x;        // [n][in_width] encoded input
	const float* target;    // [n][dims] (unused with ext_dy)
	const float* data_pdf;  // optional [n][dims]
	const half_t* ext_dy;   // optional external dL/doutput [n][out_width] (already loss-scaled)
	half_t* out;            // [n][out_width]
	half_t* dL_dout;        // [n][out_width] (written unless ext_dy)
	float* L;               // [n][out_width] (written unless ext_dy)
	half_t* dL_dx;          // optional, AoS or level planes
	float* slabs;           // [gridDim.x][n_params] partial weight gradients (nullptr: no weight gradients)
	const h8* image;        // forward fragments followed by backward fragments
	uint32_t n, dims, loss_type;
	float loss_scale;
	uint32_t dx_plane_f, n_params;
	uint32_t image_in_lds;
	const float* rec_x;     // dL_dx as 16-byte scatter records (mlp_device.h store_dx_record): the samples' coordinates, AoS [n][rec_dims]
	uint32_t rec_dims;
	uint32_t x_plane_f;     // 0: x is AoS [n][in_width]; F: x is level planes [in_width / F][n][F] (k_grid_planes.hip)
	MatView ob_x;           // OB kernels: the coordinates whose OneBlob encoding is the network's input (x is not read)
	uint32_t ob_dims, ob_log2;
	unsigned long long* dbg; // development aid (TCNN_AMD_MLP_TIMING): shader clocks per phase, summed over workgroup 0 / wave 0's trips
};

// phase timer (development aid): adds the clocks since the previous mark to slot i
#define TCNN_T(i) do { if (a.dbg) { const unsigned long long tcnn_now = __builtin_readcyclecounter(); tcnn_t[i] += tcnn_now - tcnn_t_prev; tcnn_t_prev = tcnn_now; } } while (0)

// A operand of an MFMA: register slot (REGW) or a 16-byte load from the fragment image
#define TCNN_FRAG(slot, load_expr) ([&]() -> h8 { if constexpr (REGW) return rw[(slot)]; else return (load_expr); }())

// PW ("private weight gradients"): every wave accumulates ALL weight-gradient tiles over its own 16 samples per trip
// (v_mfma_f32_16x16x16_f16, k = the wave's 16 rows of the LDS images) instead of sharing the tiles and the k axis with the
// other waves.  Nothing a wave reads was written by another wave, so the trip loop needs no workgroup barrier at all -- the
// two barriers per trip were where the shared form spent most of its time.  Costs 2x the (cheap) MFMA work and 128
// accumulator registers; the waves' partial sums are combined through LDS once, after the last trip.
// Requires NB == 1, W == 64, n_hidden <= 2, in_width <= 32, out_width <= 32 (static tile slots: 8 + 16 + 8).
// REGW ("weights in registers"): a 64-wide network with two hidden layers, <= 32 inputs and 16 outputs has 14 forward and
// 16 backward weight fragments -- 120 VGPRs per lane.  Held in registers for the whole kernel, every MFMA of a trip takes its
// A operand from a register instead of waiting on a 1 KB LDS (or L2) fetch, and the fragment images need no LDS space.
// OB: the input is the OneBlob encoding of a.ob_x, evaluated in the layer-0 loop (see k_mlp_fwd in k_mlp.hip for the scheme)
template <int W, int NB, int NW, int MAXT, int ACT, bool PW = false, bool REGW = false, bool OB = false>
__global__ void __launch_bounds__(NW * 64) k_mlp_train(const MlpDesc d, const TrainArgs a) {
	constexpr int T = W / 16;
	constexpr int KS = (T + 1) / 2;
	constexpr int S = NW * NB * 16;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	half_t* lds = (half_t*)smem;

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6); // provably wave-uniform: everything derived from it stays in SGPRs
	const uint32_t c = lane & 15, q = lane >> 4;
	const uint32_t in_w = d.in_width, out_w = d.out_width, nh = d.n_hidden;
	const uint32_t m_out = out_w / 16; // 1 or 2 output tiles (host guarantees out_width <= 32)

	// LDS images, [S rows][features + pad] halves each
	const uint32_t xs_stride = in_w + TR_PAD, hs_stride = W + hs_pad(W), dys_stride = out_w + TR_PAD;
	const uint32_t xs_off = 0;
	const uint32_t hs_off = xs_off + S * xs_stride;           // + l * S * hs_stride
	const uint32_t dhs_off = hs_off + nh * S * hs_stride;     // + l * S * hs_stride
	const uint32_t dys_off = dhs_off + nh * S * hs_stride;

	// ---- static assignment of weight-gradient tiles to waves (phase B): wave w owns tiles [first, first + my_tiles) of the
	// enumeration "layer by layer, row-major".  The tile -> (layer, row tile, col tile) arithmetic is scalar (SALU) work.
	uint32_t my_tiles = 0, first_tile = 0;
	if (a.slabs) {
		uint32_t total = 0;
		for (uint32_t l = 0; l < d.n_layers; ++l) total += (d.layers[l].rows / 16) * (d.layers[l].cols / 16);
		const uint32_t per = (total + NW - 1) / NW;
		first_tile = wave * per;
		my_tiles = first_tile < total ? min(per, total - first_tile) : 0;
	}
	struct TileDesc { uint32_t a, as, b, bs, out, ld; };
	auto tile_desc = [&](uint32_t id) {
		uint32_t l = 0;
		while (l + 1 < d.n_layers && id >= (d.layers[l].rows / 16) * (d.layers[l].cols / 16)) {
			id -= (d.layers[l].rows / 16) * (d.layers[l].cols / 16);
			++l;
		}
		const uint32_t tc_n = d.layers[l].cols / 16;
		const uint32_t tr = id / tc_n, tc = id - tr * tc_n;
		const bool last = l == d.n_layers - 1;
		TileDesc t;
		t.a = (last ? dys_off : dhs_off + l * S * hs_stride) + 16 * tr;       // dOut image
		t.as = last ? dys_stride : hs_stride;
		t.b = (l == 0 ? xs_off : hs_off + (l - 1) * S * hs_stride) + 16 * tc;  // In image
		t.bs = l == 0 ? xs_stride : hs_stride;
		t.out = d.layers[l].w_off + 16 * tr * d.layers[l].cols + 16 * tc;
		t.ld = d.layers[l].cols;
		return t;
	};
	f4 wacc[MAXT];
#pragma unroll
	for (int i = 0; i < MAXT; ++i) wacc[i] = f4{0, 0, 0, 0};

	// weight fragment images: copied into LDS once per workgroup when they fit next to the activation images (a.image_in_lds),
	// else read from global memory (L2) on every use
	const h8* image = a.image;
	if (a.image_in_lds) {
		h8* dst = (h8*)(lds + dys_off + S * dys_stride);
		const uint32_t n16 = (d.n_frags_fwd + d.n_frags_bwd) * 64;
		for (uint32_t i = tid; i < n16; i += NW * 64) dst[i] = a.image[i];
		image = dst;
		__syncthreads();
	}
	const h8* img_f = image + lane;
	const h8* img_b = image + (size_t)d.n_frags_fwd * 64 + lane;
	// REGW slots: 0..3 W0 (t) | 4..11 W1 (t * 2 + s) | 12..13 Wout (s) | 14..17 Wout^T (t) | 18..25 W1^T (t * 2 + s) | 26..29 W0^T (ti * 2 + s)
	h8 rw[REGW ? 30 : 1];
	if constexpr (REGW) {
#pragma unroll
		for (int i = 0; i < 4; ++i) rw[i] = img_f[(size_t)(d.layers[0].fwd_off + i) * 64];
#pragma unroll
		for (int i = 0; i < 8; ++i) rw[4 + i] = img_f[(size_t)(d.layers[1].fwd_off + i) * 64];
#pragma unroll
		for (int i = 0; i < 2; ++i) rw[12 + i] = img_f[(size_t)(d.layers[2].fwd_off + i) * 64];
#pragma unroll
		for (int i = 0; i < 4; ++i) rw[14 + i] = img_b[(size_t)(d.layers[2].bwd_off + i) * 64];
#pragma unroll
		for (int i = 0; i < 8; ++i) rw[18 + i] = img_b[(size_t)(d.layers[1].bwd_off + i) * 64];
#pragma unroll
		for (int i = 0; i < 4; ++i) rw[26 + i] = img_b[(size_t)(d.layers[0].bwd_off + i) * 64];
	}
	const uint32_t n_total = a.n * a.dims; // loss normalisation (relative_l2.h:58)
	const LossScales lsc = loss_scales(n_total, a.loss_scale);
	const uint32_t n_trips = a.n / S;
	const uint32_t row0 = wave * NB * 16; // this wave's first row inside the LDS images

	// 8 consecutive input features k0..k0+7 of one sample (the B operand of layer 0), from the AoS matrix or from level planes
	auto load_x = [&](const uint32_t sample, const uint32_t k0) -> h8 {
		if (a.x_plane_f == 2) { // four 4-byte loads, each a dense 64-byte run per 16 lanes
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
	// The first k-step of the NEXT trip's input is requested while this trip's weight-gradient phase runs: all waves of a
	// workgroup start a trip together, so nobody else would cover that latency.
	h8 pre[NB];
	bool have_pre = false;
	// ... and so are the targets of the first output tile and the coordinates for the scatter records: vmcnt returns in order,
	// so a load issued where its value is needed also waits for the write acknowledgement of every store issued before it
	// (outputs, loss values, records) -- measured: 28 % of the kernel sat in the loss section, waiting for 3 floats.
	float pre_t[NB][4], pre_pdf[NB][4], pre_x[NB][3];
	auto load_aux = [&](const uint32_t trip_s0) {
#pragma unroll
		for (int b = 0; b < NB; ++b) {
			const uint32_t sample = trip_s0 + 16 * b + c;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const uint32_t j = 4 * q + r;
				pre_t[b][r] = (!a.ext_dy && j < a.dims) ? a.target[sample * a.dims + j] : 0.0f;
				pre_pdf[b][r] = (!a.ext_dy && a.data_pdf && j < a.dims) ? a.data_pdf[sample * a.dims + j] : 1.0f;
			}
#pragma unroll
			for (int k = 0; k < 3; ++k) pre_x[b][k] = (a.rec_x && k < (int)a.rec_dims) ? a.rec_x[(size_t)sample * a.rec_dims + k] : 0.0f;
		}
	};
	if (blockIdx.x < n_trips) load_aux(blockIdx.x * S + row0);

	unsigned long long tcnn_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	for (uint32_t trip = blockIdx.x; trip < n_trips; trip += gridDim.x) {
		const uint32_t s0 = trip * S + row0; // first sample of this wave

		unsigned long long tcnn_t_prev = a.dbg ? __builtin_readcyclecounter() : 0;
		// =================================================================== phase A: forward
		f4 acc[T][NB];
#pragma unroll
		for (int t = 0; t < T; ++t)
#pragma unroll
			for (int b = 0; b < NB; ++b) acc[t][b] = f4{0, 0, 0, 0};
		{
			const uint32_t ks0 = d.layers[0].ks_fwd;
			const h8* img = img_f + (size_t)d.layers[0].fwd_off * 64;
			uint32_t ob_first[OB ? NB : 1];
			float ob_win[OB ? NB : 1][5], ob_xv[OB ? NB : 1];
			for (uint32_t s = 0; s < ks0; ++s) {
				h8 bf[NB];
				const uint32_t k0 = 32 * s + 8 * q;
				if constexpr (OB) { // n_bins >= 32: the four quarters of a k-step are four 8-bin chunks of one dimension's row
					const uint32_t n_bins = 1u << a.ob_log2;
					const uint32_t dim = (32 * s) >> a.ob_log2; // wave-uniform
					if (dim < a.ob_dims && ((32 * s) & (n_bins - 1)) == 0) { // first k-step of this dimension: the five bins around x, shared by the sample's four lanes
#pragma unroll
						for (int b = 0; b < NB; ++b) {
							const float xv = a.ob_x.data[(size_t)(s0 + 16 * b + c) * a.ob_x.stride_sample + (size_t)dim * a.ob_x.stride_dim];
							ob_xv[b] = xv;
							ob_first[b] = oneblob_window_first(xv, a.ob_log2);
							const float m0 = oneblob_bin(xv, (ob_first[b] + q) & (n_bins - 1), a.ob_log2);
							const float m1 = oneblob_bin(xv, (ob_first[b] + 4) & (n_bins - 1), a.ob_log2);
#pragma unroll
							for (int o = 0; o < 4; ++o) ob_win[b][o] = __shfl(m0, (int)(c + 16 * o), 64);
							ob_win[b][4] = m1;
						}
					}
					const uint32_t b0 = k0 & (n_bins - 1);
#pragma unroll
					for (int b = 0; b < NB; ++b) {
						h8 v;
						if (dim >= a.ob_dims) { // padding columns: ones
							v = h8{1, 1, 1, 1, 1, 1, 1, 1};
						} else if (oneblob_in_unit_interval(ob_xv[b])) {
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
							for (int k = 0; k < 9; ++k) e[k] = oneblob_edge(ob_xv[b], b0 + k, a.ob_log2);
							if (b0 + 8 == n_bins) e[8] += 1;
#pragma unroll
							for (int k = 0; k < 8; ++k) v[k] = (half_t)(e[k + 1] - e[k]);
						}
						if (k0 < in_w) {
							bf[b] = v;
							*(h8*)(lds + xs_off + (row0 + 16 * b + c) * xs_stride + k0) = v;
						} else {
							bf[b] = h8{0, 0, 0, 0, 0, 0, 0, 0};
						}
					}
				} else {
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					if (k0 < in_w) {
						bf[b] = (s == 0 && have_pre) ? pre[b] : load_x(s0 + 16 * b + c, k0);
						*(h8*)(lds + xs_off + (row0 + 16 * b + c) * xs_stride + k0) = bf[b];
					} else {
						bf[b] = h8{0, 0, 0, 0, 0, 0, 0, 0};
					}
				}
				}
#pragma unroll
				for (int t = 0; t < T; ++t) {
					const h8 af = TCNN_FRAG(t, img[(size_t)(t * ks0 + s) * 64]);
#pragma unroll
					for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af, bf[b], acc[t][b]);
				}
			}
		}
		h8 hf[KS][NB];
		// activation, pack into the next layer's B fragments, and drop the [sample][feature] rows into LDS
		auto finish_layer = [&](uint32_t l) {
#pragma unroll
			for (int t = 0; t < T; ++t)
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					h4 v;
#pragma unroll
					for (int r = 0; r < 4; ++r) v[r] = act_fwd_t<ACT>(d.activation, (half_t)acc[t][b][r]);
					*(h4*)(lds + hs_off + l * S * hs_stride + (row0 + 16 * b + c) * hs_stride + 16 * t + 4 * q) = v;
#pragma unroll
					for (int r = 0; r < 4; ++r) hf[t / 2][b][(t & 1) * 4 + r] = v[r];
				}
			if constexpr (T & 1) {
#pragma unroll
				for (int b = 0; b < NB; ++b)
#pragma unroll
					for (int r = 0; r < 4; ++r) hf[KS - 1][b][4 + r] = (half_t)0.0f;
			}
		};
		finish_layer(0);
		for (uint32_t l = 1; l < nh; ++l) {
			const h8* img = img_f + (size_t)d.layers[l].fwd_off * 64;
#pragma unroll
			for (int t = 0; t < T; ++t)
#pragma unroll
				for (int b = 0; b < NB; ++b) acc[t][b] = f4{0, 0, 0, 0};
#pragma unroll
			for (int s = 0; s < KS; ++s) {
#pragma unroll
				for (int t = 0; t < T; ++t) {
					const h8 af = TCNN_FRAG(4 + t * KS + s, img[(size_t)(t * KS + s) * 64]);
#pragma unroll
					for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af, hf[s][b], acc[t][b]);
				}
			}
			finish_layer(l);
		}

		TCNN_T(0);
		// =================================================================== output layer + loss (on the accumulator tile)
		h8 dyf[NB]; // dL/d(pre-activation output) as the B fragment of the first backward product (k = output index)
#pragma unroll
		for (int b = 0; b < NB; ++b) dyf[b] = h8{0, 0, 0, 0, 0, 0, 0, 0};
		{
			const MlpLayer Lo = d.layers[d.n_layers - 1];
			const h8* img = img_f + (size_t)Lo.fwd_off * 64;
#pragma unroll
			for (int to = 0; to < 2; ++to) {
				if (to < (int)m_out) {
					f4 o[NB];
#pragma unroll
					for (int b = 0; b < NB; ++b) o[b] = f4{0, 0, 0, 0};
#pragma unroll
					for (int s = 0; s < KS; ++s) {
						const h8 af = TCNN_FRAG(12 + to * KS + s, img[(size_t)(to * KS + s) * 64]);
#pragma unroll
						for (int b = 0; b < NB; ++b) o[b] = mfma(af, hf[s][b], o[b]);
					}
#pragma unroll
					for (int b = 0; b < NB; ++b) {
						const uint32_t sample = s0 + 16 * b + c;
						const uint32_t j0 = 16 * to + 4 * q; // this lane's 4 output indices
						h4 ov, gv;
						float lv[4];
#pragma unroll
						for (int r = 0; r < 4; ++r) ov[r] = activation_fwd(d.output_activation, (half_t)o[b][r]);
						if (a.ext_dy) {
							gv = *(const h4*)(a.ext_dy + (size_t)sample * out_w + j0);
						} else {
#pragma unroll
							for (int r = 0; r < 4; ++r) { // l2.h:40-74 / relative_l2.h:40-75
								const uint32_t j = j0 + r;
								float value = 0.0f;
								half_t grad = (half_t)0.0f;
								if (j < a.dims) {
									const uint32_t target_idx = sample * a.dims + j;
									const float prediction = (float)ov[r];
									const float pdf = to == 0 ? pre_pdf[b][r] : (a.data_pdf ? a.data_pdf[target_idx] : 1);
									const float tgt = to == 0 ? pre_t[b][r] : a.target[target_idx];
									if (a.loss_type == (uint32_t)LossType::RelativeL2) loss_l2_fused<true>(prediction, tgt, lsc, value, grad, a.data_pdf != nullptr, pdf);
									else loss_l2_fused<false>(prediction, tgt, lsc, value, grad, a.data_pdf != nullptr, pdf);
								}
								lv[r] = value;
								gv[r] = grad;
							}
							if (a.L) *(f4*)(a.L + (size_t)sample * out_w + j0) = f4{lv[0], lv[1], lv[2], lv[3]};
							if (a.dL_dout) *(h4*)(a.dL_dout + (size_t)sample * out_w + j0) = gv;
						}
						if (a.out) *(h4*)(a.out + (size_t)sample * out_w + j0) = ov;
						// output-activation transfer (fully_fused_mlp.cu:757-762)
						if (d.output_activation != (uint32_t)Activation::None) {
#pragma unroll
							for (int r = 0; r < 4; ++r) gv[r] = activation_bwd(d.output_activation, gv[r], ov[r]);
						}
						*(h4*)(lds + dys_off + (row0 + 16 * b + c) * dys_stride + j0) = gv;
#pragma unroll
						for (int r = 0; r < 4; ++r) dyf[b][to * 4 + r] = gv[r];
					}
				}
			}
		}

		TCNN_T(1);
		// =================================================================== backward chain
#pragma unroll
		for (int t = 0; t < T; ++t)
#pragma unroll
			for (int b = 0; b < NB; ++b) acc[t][b] = f4{0, 0, 0, 0};
		{
			const h8* img = img_b + (size_t)d.layers[d.n_layers - 1].bwd_off * 64; // Wout^T: T row tiles, one k-step (out_width <= 32)
#pragma unroll
			for (int t = 0; t < T; ++t) {
				const h8 af = TCNN_FRAG(14 + t, img[(size_t)t * 64]);
#pragma unroll
				for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af, dyf[b], acc[t][b]);
			}
		}
		for (int l = (int)nh - 1; l >= 0; --l) {
			// acc = W_{l+1}^T dH_{l+1}: multiply by act'(H_l) from the forward output (own rows of the LDS image), keep, pack
#pragma unroll
			for (int t = 0; t < T; ++t)
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					const uint32_t off = (row0 + 16 * b + c) * hs_stride + 16 * t + 4 * q;
					const h4 hv = *(const h4*)(lds + hs_off + l * S * hs_stride + off);
					h4 g;
#pragma unroll
					for (int r = 0; r < 4; ++r) g[r] = act_bwd_t<ACT>(d.activation, (half_t)acc[t][b][r], hv[r]);
					*(h4*)(lds + dhs_off + l * S * hs_stride + off) = g;
#pragma unroll
					for (int r = 0; r < 4; ++r) hf[t / 2][b][(t & 1) * 4 + r] = g[r];
				}
			if constexpr (T & 1) {
#pragma unroll
				for (int b = 0; b < NB; ++b)
#pragma unroll
					for (int r = 0; r < 4; ++r) hf[KS - 1][b][4 + r] = (half_t)0.0f;
			}
			if (l > 0) {
				const h8* img = img_b + (size_t)d.layers[l].bwd_off * 64;
#pragma unroll
				for (int t = 0; t < T; ++t)
#pragma unroll
					for (int b = 0; b < NB; ++b) acc[t][b] = f4{0, 0, 0, 0};
#pragma unroll
				for (int s = 0; s < KS; ++s) {
#pragma unroll
					for (int t = 0; t < T; ++t) {
						const h8 af = TCNN_FRAG(18 + t * KS + s, img[(size_t)(t * KS + s) * 64]);
#pragma unroll
						for (int b = 0; b < NB; ++b) acc[t][b] = mfma(af, hf[s][b], acc[t][b]);
					}
				}
			}
		}
		TCNN_T(2);
		if (a.dL_dx) { // dX = W0^T dH_0
			const h8* img = img_b + (size_t)d.layers[0].bwd_off * 64;
			auto dx_tile = [&](const uint32_t ti, const h8 (&af)[KS]) {
				f4 o[NB];
#pragma unroll
				for (int b = 0; b < NB; ++b) o[b] = f4{0, 0, 0, 0};
#pragma unroll
				for (int s = 0; s < KS; ++s) {
#pragma unroll
					for (int b = 0; b < NB; ++b) o[b] = mfma(af[s], hf[s][b], o[b]);
				}
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					const h4 v = h4{(half_t)o[b][0], (half_t)o[b][1], (half_t)o[b][2], (half_t)o[b][3]};
					if (a.rec_x) {
						const uint32_t sample = s0 + 16 * b + c;
						const float xs[3] = {pre_x[b][0], pre_x[b][1], pre_x[b][2]};
						store_dx_record(a.dL_dx, a.dx_plane_f, a.rec_dims, a.n, sample, 16 * ti + 4 * q, v, xs);
					} else {
						store_dx(a.dL_dx, a.dx_plane_f, a.n, in_w, s0 + 16 * b + c, 16 * ti + 4 * q, v);
					}
				}
			};
			if constexpr (REGW) { // at most 2 input tiles, fragment slots 26 + ti * KS + s
				static_assert(KS == 2, "REGW is laid out for 64-wide networks");
				{
					const h8 af[KS] = {rw[26], rw[27]};
					dx_tile(0, af);
				}
				if (in_w > 16) {
					const h8 af[KS] = {rw[28], rw[29]};
					dx_tile(1, af);
				}
			} else {
				for (uint32_t ti = 0; ti < in_w / 16; ++ti) {
					h8 af[KS];
#pragma unroll
					for (int s = 0; s < KS; ++s) af[s] = img[(size_t)(ti * KS + s) * 64];
					dx_tile(ti, af);
				}
			}
		}

		{ // request the next trip's input now (see `pre` above)
			const uint32_t next = trip + gridDim.x;
			have_pre = next < n_trips;
			if (!OB && have_pre && 8 * q < in_w) {
#pragma unroll
				for (int b = 0; b < NB; ++b) pre[b] = load_x(next * S + row0 + 16 * b + c, 8 * q);
			}
			if (have_pre) load_aux(next * S + row0);
		}

		// =================================================================== phase B: weight gradients from the LDS images
		if constexpr (PW) {
			if (a.slabs) {
				// transposing reads (see the shared form below), k = this wave's 16 rows: lane (li, grp) gets rows 4 grp .. 4 grp + 3 of column li
				const uint32_t grp = lane >> 4, li = lane & 15;
				const uint32_t rowp = row0 + 4 * grp + (li >> 2), colo = 4 * (li & 3);
				auto frag = [&](uint32_t image_off, uint32_t stride, uint32_t tile) { return lds_read_tr(lds + image_off + rowp * stride + 16 * tile + colo); };
				// slots 0..7: dW0 = dH_0^T X (T x in_w/16 tiles)
				{
					h4 bx[2];
#pragma unroll
					for (int tc = 0; tc < 2; ++tc) bx[tc] = 16 * tc < (int)in_w ? frag(xs_off, xs_stride, tc) : h4{0, 0, 0, 0};
#pragma unroll
					for (int tr = 0; tr < T; ++tr) {
						const h4 af = frag(dhs_off, hs_stride, tr);
#pragma unroll
						for (int tc = 0; tc < 2; ++tc) wacc[tr * 2 + tc] = mfma16(af, bx[tc], wacc[tr * 2 + tc]);
					}
				}
				// slots 8..23: dW1 = dH_1^T H_0 (T x T tiles), only with two hidden layers
				if (nh == 2) {
					h4 bh[T];
#pragma unroll
					for (int tc = 0; tc < T; ++tc) bh[tc] = frag(hs_off, hs_stride, tc);
#pragma unroll
					for (int tr = 0; tr < T; ++tr) {
						const h4 af = frag(dhs_off + S * hs_stride, hs_stride, tr);
#pragma unroll
						for (int tc = 0; tc < T; ++tc) wacc[8 + tr * T + tc] = mfma16(af, bh[tc], wacc[8 + tr * T + tc]);
					}
				}
				// slots 24..31: dWout = dY^T H_last (out_w/16 x T tiles)
				{
					h4 bh[T];
#pragma unroll
					for (int tc = 0; tc < T; ++tc) bh[tc] = frag(hs_off + (nh - 1) * S * hs_stride, hs_stride, tc);
#pragma unroll
					for (int to = 0; to < (MAXT - 24) / T; ++to) {
						if (to < (int)m_out) {
							const h4 af = frag(dys_off, dys_stride, to);
#pragma unroll
							for (int tc = 0; tc < T; ++tc) wacc[24 + to * T + tc] = mfma16(af, bh[tc], wacc[24 + to * T + tc]);
						}
					}
				}
			}
			continue; // no barrier: the next trip overwrites only this wave's own rows
		}
		TCNN_T(3);
		lds_barrier();
		TCNN_T(4);
		if (a.slabs) {
			const uint32_t grp = lane >> 4, li = lane & 15;
			const uint32_t row_in_step = 8 * grp + (li >> 2), colo = 4 * (li & 3);
#pragma unroll
			for (int i = 0; i < MAXT; ++i) {
				if (i < (int)my_tiles) { // wave-uniform
					const TileDesc td = tile_desc(first_tile + i);
					const half_t* pa = lds + td.a + colo;
					const half_t* pb = lds + td.b + colo;
#pragma unroll
					for (int ks = 0; ks < S / 32; ++ks) {
						const uint32_t row = 32 * ks + row_in_step;
						const h4 a_lo = lds_read_tr(pa + row * td.as);
						const h4 a_hi = lds_read_tr(pa + (row + 4) * td.as);
						const h4 b_lo = lds_read_tr(pb + row * td.bs);
						const h4 b_hi = lds_read_tr(pb + (row + 4) * td.bs);
						const h8 af = h8{a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
						const h8 bf = h8{b_lo[0], b_lo[1], b_lo[2], b_lo[3], b_hi[0], b_hi[1], b_hi[2], b_hi[3]};
						wacc[i] = mfma(af, bf, wacc[i]);
					}
				}
			}
		}
		TCNN_T(5);
		lds_barrier();
		TCNN_T(6);
	}

	if constexpr (PW) {
		if (a.slabs) {
			// combine the waves' partial sums in LDS (plain read-add-write, one wave at a time: LDS float atomics are slow), then
			// write the workgroup's slab with coalesced stores
			float* red = (float*)smem;
			const uint32_t grp = lane >> 4, li = lane & 15;
			__syncthreads(); // every wave is done with the images
			for (uint32_t w = 0; w < (uint32_t)NW; ++w) {
				if (wave == w) {
					auto put = [&](const f4& v, uint32_t out, uint32_t ld) {
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							float* p = red + out + (size_t)(4 * grp + r) * ld + li;
							*p = w == 0 ? v[r] : *p + v[r];
						}
					};
					const MlpLayer L0 = d.layers[0];
#pragma unroll
					for (int tr = 0; tr < T; ++tr)
#pragma unroll
						for (int tc = 0; tc < 2; ++tc)
							if (16 * tc < (int)in_w) put(wacc[tr * 2 + tc], L0.w_off + 16 * tr * L0.cols + 16 * tc, L0.cols);
					if (nh == 2) {
						const MlpLayer L1 = d.layers[1];
#pragma unroll
						for (int tr = 0; tr < T; ++tr)
#pragma unroll
							for (int tc = 0; tc < T; ++tc) put(wacc[8 + tr * T + tc], L1.w_off + 16 * tr * L1.cols + 16 * tc, L1.cols);
					}
					const MlpLayer Lo = d.layers[d.n_layers - 1];
#pragma unroll
					for (int to = 0; to < (MAXT - 24) / T; ++to)
						if (to < (int)m_out) {
#pragma unroll
							for (int tc = 0; tc < T; ++tc) put(wacc[24 + to * T + tc], Lo.w_off + 16 * to * Lo.cols + 16 * tc, Lo.cols);
						}
				}
				__syncthreads();
			}
			float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
			for (uint32_t i = tid; i < a.n_params; i += NW * 64) slab[i] = red[i];
		}
		return;
	}
	if (a.dbg && blockIdx.x == 0 && tid == 0) {
		for (int i = 0; i < 8; ++i) a.dbg[i] = tcnn_t[i];
	}
	if (a.slabs) {
		float* slab = a.slabs + (size_t)blockIdx.x * a.n_params;
		const uint32_t grp = lane >> 4, li = lane & 15;
#pragma unroll
		for (int i = 0; i < MAXT; ++i) {
			if (i < (int)my_tiles) {
				const TileDesc td = tile_desc(first_tile + i);
#pragma unroll
				for (int r = 0; r < 4; ++r) slab[td.out + (size_t)(4 * grp + r) * td.ld + li] = wacc[i][r];
			}
		}
	}
}

struct TrainConfig {
	int nb, nw, maxt;
	uint32_t lds_bytes, s;
	bool ok, image_in_lds;
	bool pw; // private weight gradients (barrier-free trips), see k_mlp_train
	bool regw; // all weight fragments in registers, see k_mlp_train
};

// (NB, NW, MAXT) triples that are instantiated; pick_config only ever returns one of them
struct TrainVariant { int width_class, nb, nw, maxt; }; // width_class: 64 or 128
constexpr TrainVariant TRAIN_VARIANTS[] = {
	{64, 1, 8, 8}, {64, 2, 4, 8}, {64, 2, 4, 16}, {64, 1, 4, 16}, {64, 1, 4, 32},
	{128, 1, 8, 16}, {128, 1, 8, 32}, {128, 1, 4, 32},
};

inline TrainConfig pick_config(const MlpDesc& d) {
	TrainConfig cfg{};
	cfg.ok = false;
	if (d.out_width > 32 || d.n_hidden < 1 || (d.width != 64 && d.width != 128)) return cfg;
	uint32_t total_tiles = 0;
	for (uint32_t l = 0; l < d.n_layers; ++l) total_tiles += (d.layers[l].rows / 16) * (d.layers[l].cols / 16);
	const uint32_t image_bytes = (d.n_frags_fwd + d.n_frags_bwd) * 1024;
	const uint32_t budget = 160 * 1024 - 512;
	// TCNN_AMD_MLP_PW=1 selects the barrier-free private weight-gradient form.  Off by default: measured on C3a it takes 92 us
	// against 78 us for the shared form -- the trips are bound by dependent LDS-read -> MFMA chains at 2 waves per SIMD, not
	// by the barriers, and the private form doubles the MFMA and accumulator work.  Read per call so that tests can cover both.
	const char* pw_env = getenv("TCNN_AMD_MLP_PW");
	const bool pw_allowed = pw_env && pw_env[0] == '1';
	if (pw_allowed && d.width == 64 && d.n_hidden <= 2 && d.in_width <= 32 && d.out_width <= 32) {
		const uint32_t s = 8 * 16;
		const uint32_t images = 2 * s * ((d.in_width + TR_PAD) + 2 * d.n_hidden * (d.width + hs_pad((int)d.width)) + (d.out_width + TR_PAD));
		size_t n_params = 0;
		for (uint32_t l = 0; l < d.n_layers; ++l) n_params += (size_t)d.layers[l].rows * d.layers[l].cols;
		if (images + image_bytes <= budget && n_params * sizeof(float) <= images) { // the final reduction reuses the image space
			cfg.nb = 1;
			cfg.nw = 8;
			cfg.maxt = d.out_width <= 16 ? 28 : 32; // 8 + 16 + 4 (or 8) tile slots
			cfg.s = s;
			cfg.lds_bytes = images + image_bytes;
			cfg.image_in_lds = true;
			cfg.pw = true;
			cfg.ok = true;
			return cfg;
		}
	}
	// 64 x 2 networks with <= 32 inputs and 16 outputs: all 30 weight fragments live in registers (TCNN_AMD_MLP_REGW=0: A/B runs)
	const char* regw_env = getenv("TCNN_AMD_MLP_REGW");
	if (!(regw_env && regw_env[0] == '0') && d.width == 64 && d.n_hidden == 2 && d.in_width <= 32 && d.out_width == 16 && d.layers[0].ks_fwd == 1) {
		const uint32_t s = 8 * 16;
		cfg.nb = 1;
		cfg.nw = 8;
		cfg.maxt = 8;
		cfg.s = s;
		cfg.lds_bytes = 2 * s * ((d.in_width + TR_PAD) + 2 * d.n_hidden * (d.width + hs_pad((int)d.width)) + (d.out_width + TR_PAD));
		cfg.image_in_lds = false;
		cfg.regw = true;
		cfg.ok = true;
		return cfg;
	}
	// Occupancy first: the variant that puts the most waves on a CU (workgroups per CU as mlp_train_fused_grid launches them,
	// at most 2) wins; among equals, the one whose weight images fit in LDS next to the activation images, then table order.
	// (Preferring "images in LDS" outright picked a 4-wave workgroup for C2 -- 128 inputs, 56 KB of fragments -- and ran at one
	// wave per SIMD: 84 us per step against 68 us with 8 waves and the images in L2.)
	const char* img_env = getenv("TCNN_AMD_MLP_IMAGE_LDS"); // development aid: "0" keeps the weight images in L2
	uint32_t best_waves = 0;
	for (int pass = (img_env && img_env[0] == '0') ? 1 : 0; pass < 2; ++pass) {
		for (const TrainVariant& v : TRAIN_VARIANTS) {
			if (v.width_class != (int)d.width) continue;
			if (const char* e = getenv("TCNN_AMD_MLP_VARIANT")) { // development aid: "nb,nw,maxt" forces one instantiated variant
				int nb = 0, nw = 0, maxt = 0;
				if (sscanf(e, "%d,%d,%d", &nb, &nw, &maxt) == 3 && (nb != v.nb || nw != v.nw || maxt != v.maxt)) continue;
			}
			const uint32_t s = v.nw * v.nb * 16;
			const uint32_t bytes = 2 * s * ((d.in_width + TR_PAD) + 2 * d.n_hidden * (d.width + hs_pad((int)d.width)) + (d.out_width + TR_PAD)) + (pass == 0 ? image_bytes : 0);
			const uint32_t per = (total_tiles + v.nw - 1) / v.nw;
			if (bytes > budget || (int)per > v.maxt) continue;
			const uint32_t waves = (uint32_t)v.nw * std::min(2u, std::max(1u, (160u * 1024u) / bytes));
			if (waves <= best_waves) continue; // earlier candidates (images in LDS, table order) keep ties
			best_waves = waves;
			cfg.nb = v.nb;
			cfg.nw = v.nw;
			cfg.maxt = v.maxt;
			cfg.s = s;
			cfg.lds_bytes = bytes;
			cfg.image_in_lds = pass == 0;
			cfg.ok = true;
		}
	}
	return cfg;
}

template <int W, int NB, int NW, int MAXT>
void launch_train(hipStream_t stream, const MlpDesc& d, const TrainArgs& a, uint32_t grid, uint32_t lds_bytes) {
	auto go = [&](auto kernel) {
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(NW * 64), lds_bytes, stream, d, a);
		HIP_CHECK_THROW(hipGetLastError());
	};
	if (d.activation == (uint32_t)Activation::ReLU) go(k_mlp_train<W, NB, NW, MAXT, (int)Activation::ReLU>);
	else go(k_mlp_train<W, NB, NW, MAXT, -1>);
}

void dispatch_train(hipStream_t stream, const MlpDesc& d, const TrainArgs& a, const TrainConfig& cfg, uint32_t grid) {
	if (cfg.pw) {
		auto go = [&](auto kernel) {
			HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(8 * 64), cfg.lds_bytes, stream, d, a);
			HIP_CHECK_THROW(hipGetLastError());
		};
		if (cfg.maxt == 28) {
			if (d.activation == (uint32_t)Activation::ReLU) go(k_mlp_train<64, 1, 8, 28, (int)Activation::ReLU, true>);
			else go(k_mlp_train<64, 1, 8, 28, -1, true>);
		} else {
			if (d.activation == (uint32_t)Activation::ReLU) go(k_mlp_train<64, 1, 8, 32, (int)Activation::ReLU, true>);
			else go(k_mlp_train<64, 1, 8, 32, -1, true>);
		}
		return;
	}
	if (cfg.regw) {
		auto go = [&](auto kernel) {
			HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(8 * 64), cfg.lds_bytes, stream, d, a);
			HIP_CHECK_THROW(hipGetLastError());
		};
		if (d.activation == (uint32_t)Activation::ReLU) go(k_mlp_train<64, 1, 8, 8, (int)Activation::ReLU, false, true>);
		else go(k_mlp_train<64, 1, 8, 8, -1, false, true>);
		return;
	}
	if (a.ob_log2) { // the OneBlob input is evaluated in the kernel: one shape (C2's)
		CHECK_THROW((int)d.width == 64 && cfg.nb == 1 && cfg.nw == 8 && cfg.maxt == 8 && !cfg.pw && !cfg.regw);
		auto go = [&](auto kernel) {
			HIP_CHECK_THROW(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes));
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(8 * 64), cfg.lds_bytes, stream, d, a);
			HIP_CHECK_THROW(hipGetLastError());
		};
		if (d.activation == (uint32_t)Activation::ReLU) go(k_mlp_train<64, 1, 8, 8, (int)Activation::ReLU, false, false, true>);
		else go(k_mlp_train<64, 1, 8, 8, -1, false, false, true>);
		return;
	}
#define TCNN_TRAIN_CASE(W_, NB_, NW_, MAXT_) \
	if ((int)d.width == W_ && cfg.nb == NB_ && cfg.nw == NW_ && cfg.maxt == MAXT_) return launch_train<W_, NB_, NW_, MAXT_>(stream, d, a, grid, cfg.lds_bytes);
	TCNN_TRAIN_CASE(64, 1, 8, 8)
	TCNN_TRAIN_CASE(64, 2, 4, 8)
	TCNN_TRAIN_CASE(64, 2, 4, 16)
	TCNN_TRAIN_CASE(64, 1, 4, 16)
	TCNN_TRAIN_CASE(64, 1, 4, 32)
	TCNN_TRAIN_CASE(128, 1, 8, 16)
	TCNN_TRAIN_CASE(128, 1, 8, 32)
	TCNN_TRAIN_CASE(128, 1, 4, 32)
#undef TCNN_TRAIN_CASE
	throw std::runtime_error{"mlp_train_fused: no kernel instance for this configuration"};
}

} // namespace

bool mlp_train_fused_oneblob_supported(const MlpDesc& d, uint32_t n, uint32_t n_bins) {
	if (n_bins < 32 || (n_bins & (n_bins - 1)) != 0 || mlp_train_regs_supported(d, n)) return false;
	const TrainConfig cfg = pick_config(d);
	return cfg.ok && n % cfg.s == 0 && n > 0 && d.width == 64 && cfg.nb == 1 && cfg.nw == 8 && cfg.maxt == 8 && !cfg.pw && !cfg.regw;
}

bool mlp_train_fused_supported(const MlpDesc& d, uint32_t n) {
	if (mlp_train_regs_supported(d, n)) return true;
	const TrainConfig cfg = pick_config(d);
	return cfg.ok && n % cfg.s == 0 && n > 0;
}

uint32_t mlp_train_fused_grid(const MlpDesc& d, uint32_t n, uint32_t oneblob_bins, uint32_t oneblob_dims) {
	if (oneblob_bins && mlp_train_r32ob_shape(d, n, oneblob_bins, oneblob_dims)) return mlp_train_r32ob_grid(n); // mlp_train_fused: the same test
	if (mlp_train_regs_supported(d, n)) return mlp_train_regs_grid(d, n);
	const TrainConfig cfg = pick_config(d);
	if (!cfg.ok) return 0;
	const uint32_t trips = n / cfg.s;
	const uint32_t per_cu = std::max(1u, (160u * 1024u) / std::max(cfg.lds_bytes, 1u));
	const uint32_t cap = 256 * std::min(per_cu, 2u);
	return std::max(1u, std::min(trips, cap));
}

void mlp_train_fused(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, uint32_t x_plane_features, const float* target, const float* data_pdf,
                     const void* external_dL_dy, uint32_t dims, LossType loss, float loss_scale, void* out, void* dL_dout, float* L, bool compact_context, void* dL_dx,
                     uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims, float* slabs, uint32_t n_params, const MlpOneBlobInput* oneblob) {
	CHECK_THROW(!oneblob || mlp_train_fused_oneblob_supported(d, n, oneblob->n_bins));
	if (!compact_context && mlp_train_r32ob_applies(d, n, oneblob, data_pdf, external_dL_dy, dims, loss, out, dL_dx, slabs)) {
		return mlp_train_r32ob(stream, d, image, n, *oneblob, target, dims, loss, loss_scale, out, dL_dout, L, slabs, n_params);
	}
	if (!compact_context && mlp_train_r32w_applies(d, n, x_plane_features, data_pdf, external_dL_dy, dims, loss, out, dL_dx, dx_plane_features, dx_record_x, slabs, oneblob != nullptr)) {
		return mlp_train_r32w(stream, d, image, n, x, target, dims, loss, loss_scale, out, dL_dout, L, dL_dx, slabs, n_params, mlp_train_fused_grid(d, n));
	}
	if (mlp_train_regs_supported(d, n) && slabs != nullptr) { // without weight gradients (GradientMode::Ignore): the kernels below
		CHECK_THROW(compact_context || external_dL_dy);
		return mlp_train_regs(stream, d, image, n, x, x_plane_features, target, data_pdf, external_dL_dy, dims, loss, loss_scale, out, dL_dout, L, dL_dx, dx_plane_features,
		                      dx_record_x, dx_record_dims, slabs, n_params);
	}
	CHECK_THROW(!compact_context); // compact context matrices are a feature of k_train_regs.hip
	const TrainConfig cfg = pick_config(d);
	CHECK_THROW(cfg.ok && n % cfg.s == 0);
	TrainArgs a{(const half_t*)x, target, data_pdf, (const half_t*)external_dL_dy, (half_t*)out, (half_t*)dL_dout, L, (half_t*)dL_dx, slabs, (const h8*)image,
	            n, dims, (uint32_t)loss, loss_scale, dx_plane_features, n_params, cfg.image_in_lds ? 1u : 0u, dx_record_x, dx_record_dims, x_plane_features, MatView{}, 0u, 0u, nullptr};
	if (oneblob) {
		a.ob_x = oneblob->x;
		a.ob_dims = oneblob->n_dims;
		while ((1u << a.ob_log2) < oneblob->n_bins) ++a.ob_log2;
	}
#ifdef TCNN_AMD_DEV // laboratory build (build.py --dev): in-kernel clocks of the 5th launch
	static const bool timing = getenv("TCNN_AMD_MLP_TIMING") != nullptr;
	static int timing_left = 5;
	if (timing && timing_left > 0) HIP_CHECK_THROW(hipMalloc(&a.dbg, 64));
#else
	int timing_left = 0; (void)timing_left;
#endif
	dispatch_train(stream, d, a, cfg, mlp_train_fused_grid(d, n));
	if (a.dbg) {
		unsigned long long h[8];
		HIP_CHECK_THROW(hipMemcpy(h, a.dbg, 64, hipMemcpyDeviceToHost));
		if (--timing_left == 0) fprintf(stderr, "k_mlp_train wave 0 clocks over its trips: fwd %llu loss %llu bwd %llu dX+stores %llu barrier1 %llu wgrad %llu barrier2 %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
		(void)hipFree(a.dbg);
	}
}

} // namespace tcnn_amd
