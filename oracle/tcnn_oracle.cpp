/*
 * tcnn_oracle.cpp -- CPU oracle: a plain C++ restatement of the reference algorithm.
 *
 * TEST INFRASTRUCTURE ONLY (see tcnn_oracle.h).  Nothing under tiny-cuda-nn_amd/ may use this file.
 *
 * Parity pins (what ties this restatement to the reference):
 *   - pcg32 / seed_seq / xavier / strided uniform fill: checked bit-for-bit against the reference's own
 *     dependencies/pcg32/pcg32.h compiled unmodified into oracle/_ref (see oracle/Makefile, tests/test_oracle_ref.py).
 *   - hashing / grid_index / pos_fract / offset tables: checked against the known answers that SURVEY.md
 *     Appendix A.2 recorded from the reference's common_device.h (tests/golden/reference_kat.json).
 *   - kernels that only exist as CUDA (grid.h, fully_fused_mlp.cu, adam.h, losses): restated line by line from the
 *     cited source; the reference ships no golden vectors for them ("parity unpinned" for those).
 *
 * Build: see oracle/Makefile (-ffp-contract=off: every fused multiply-add below is an explicit fmaf, exactly
 * where the reference writes fmaf / __hfma).
 */
#include "tcnn_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#if defined(__F16C__)
#include <immintrin.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------
// IEEE binary16 <-> binary32/64, round-to-nearest-even (what __float2half_rn / (half) casts do on device)
// ---------------------------------------------------------------------------------------------------------
inline float h2f(uint16_t h) {
#if defined(__F16C__)
	return _cvtsh_ss(h);
#else
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1f;
	uint32_t man = h & 0x3ffu;
	uint32_t bits;
	if (exp == 0) {
		if (man == 0) {
			bits = sign;
		} else {
			int e = -1;
			do { man <<= 1; ++e; } while ((man & 0x400u) == 0);
			bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
		}
	} else if (exp == 31) {
		bits = sign | 0x7f800000u | (man << 13);
	} else {
		bits = sign | ((exp + 112) << 23) | (man << 13);
	}
	float f; memcpy(&f, &bits, 4); return f;
#endif
}

inline uint16_t d2h(double d) {
	uint64_t b; memcpy(&b, &d, 8);
	const uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
	const uint32_t ebits = (uint32_t)((b >> 52) & 0x7ff);
	uint64_t m = b & 0xfffffffffffffULL;
	if (ebits == 0x7ff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
	if (ebits == 0) return sign; // +-0 and double subnormals
	const int e = (int)ebits - 1023;
	if (e > 15) return (uint16_t)(sign | 0x7c00u);
	m |= 1ULL << 52;
	int shift = e >= -14 ? 42 : 42 + (-14 - e);
	if (shift > 63) return sign;
	uint64_t q = m >> shift;
	const uint64_t rem = m & ((1ULL << shift) - 1);
	const uint64_t half = 1ULL << (shift - 1);
	if (rem > half || (rem == half && (q & 1))) ++q;
	if (e >= -14) {
		uint32_t r = ((uint32_t)(e + 14) << 10) + (uint32_t)q;
		if (r >= 0x7c00u) r = 0x7c00u;
		return (uint16_t)(sign | r);
	}
	return (uint16_t)(sign | (uint16_t)q);
}

inline uint16_t f2h(float f) {
#if defined(__F16C__)
	return (uint16_t)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
#else
	return d2h((double)f); // float -> double is exact, so this is a single rounding
#endif
}

// __hfma(a, b, c): one rounding (vec.h:210-216, 356-376)
inline uint16_t hfma(uint16_t a, uint16_t b, uint16_t c) {
	return d2h(std::fma((double)h2f(a), (double)h2f(b), (double)h2f(c)));
}
// __hmul / __hadd
inline uint16_t hmul(uint16_t a, uint16_t b) { return f2h(h2f(a) * h2f(b)); } // 11x11-bit product is exact in fp32
inline uint16_t hadd(uint16_t a, uint16_t b) { return f2h(h2f(a) + h2f(b)); } // exact in fp32 unless exponents differ > 13 -> still one rounding after exact? see note
// note on hadd: the fp32 sum of two halves can be inexact only when the smaller operand lies entirely below the
// fp32 ulp of the larger one (exponent gap > 23), in which case the half result equals the larger operand either way.

// ---------------------------------------------------------------------------------------------------------
// pcg32 (dependencies/pcg32/pcg32.h:35-166; the published PCG-XSH-RR 64/32 generator)
// ---------------------------------------------------------------------------------------------------------
constexpr uint64_t PCG_MULT = 0x5851f42d4c957f2dULL;

inline uint32_t pcg_next_uint(uint64_t* st) {
	const uint64_t old = st[0];
	st[0] = old * PCG_MULT + st[1];
	const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	const uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
inline float pcg_next_float(uint64_t* st) {
	const uint32_t u = (pcg_next_uint(st) >> 9) | 0x3f800000u;
	float f; memcpy(&f, &u, 4);
	return f - 1.0f;
}
inline void pcg_seed(uint64_t* st, uint64_t initstate, uint64_t initseq) {
	st[0] = 0;
	st[1] = (initseq << 1u) | 1u;
	pcg_next_uint(st);
	st[0] += initstate;
	pcg_next_uint(st);
}
inline void pcg_advance(uint64_t* st, int64_t delta_) {
	uint64_t cur_mult = PCG_MULT, cur_plus = st[1], acc_mult = 1u, acc_plus = 0u;
	uint64_t delta = (uint64_t)delta_;
	while (delta > 0) {
		if (delta & 1) {
			acc_mult *= cur_mult;
			acc_plus = acc_plus * cur_mult + cur_plus;
		}
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta /= 2;
	}
	st[0] = acc_mult * st[0] + acc_plus;
}

// ---------------------------------------------------------------------------------------------------------
// activations (common_device.h:87-160 forward on the fp16 accumulator, :241-297 backward from forward OUTPUTS)
// ---------------------------------------------------------------------------------------------------------
constexpr float K_ACT = 10.0f;
inline float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }

inline uint16_t activation_fwd(uint32_t act, uint16_t pre_h) {
	const float x = h2f(pre_h);
	switch (act) {
		case ORC_ACT_RELU: return x > 0.0f ? pre_h : (uint16_t)0; // __hmax(val, 0): -0 and NaN details irrelevant here
		case ORC_ACT_LEAKY_RELU: return hmul(pre_h, f2h(x > 0.0f ? 1.0f : 0.01f));
		case ORC_ACT_EXPONENTIAL: return f2h(expf(x));
		case ORC_ACT_SINE: return f2h(sinf(x));
		case ORC_ACT_SIGMOID: return f2h(logistic(x));
		case ORC_ACT_SQUAREPLUS: { const float y = x * K_ACT; return f2h(0.5f * (y + sqrtf(y * y + 4)) / K_ACT); }
		case ORC_ACT_SOFTPLUS: return f2h(logf(expf(x * K_ACT) + 1.0f) / K_ACT);
		case ORC_ACT_TANH: return f2h(tanhf(x));
		default: return pre_h;
	}
}

inline uint16_t activation_bwd(uint32_t act, uint16_t grad_h, uint16_t fwd_h) {
	const float y = h2f(fwd_h);
	switch (act) {
		case ORC_ACT_RELU: return y > 0.0f ? grad_h : hmul(grad_h, 0);
		case ORC_ACT_LEAKY_RELU: return hmul(grad_h, f2h(y > 0.0f ? 1.0f : 0.01f));
		case ORC_ACT_EXPONENTIAL: return hmul(grad_h, fwd_h);
		case ORC_ACT_SINE: return grad_h; // unsupported from outputs (common_device.h:261-265)
		case ORC_ACT_SIGMOID: return hmul(grad_h, hmul(fwd_h, f2h(1.0f - y)));
		case ORC_ACT_SQUAREPLUS: { const float t = y * K_ACT; return hmul(grad_h, f2h(t * t / (t * t + 1))); }
		case ORC_ACT_SOFTPLUS: return hmul(grad_h, f2h(1.0f - expf(-y * K_ACT)));
		case ORC_ACT_TANH: return hmul(grad_h, f2h(1.0f - (y * y)));
		default: return grad_h;
	}
}

// ---------------------------------------------------------------------------------------------------------
// grid helpers (common_device.h:631-718, 801-868)
// ---------------------------------------------------------------------------------------------------------
inline uint32_t lcg_hash(uint32_t n_dims, const uint32_t* pos, const uint32_t* primes) {
	uint32_t r = 0;
	for (uint32_t i = 0; i < n_dims; ++i) r ^= pos[i] * primes[i];
	return r;
}

inline uint32_t grid_hash(uint32_t n_dims, uint32_t hash_type, const uint32_t* pos) {
	static const uint32_t prime[7] = {1958374283u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
	static const uint32_t coherent[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
	static const uint32_t reversed[7] = {2165219737u, 1434869437u, 2097192037u, 3674653429u, 805459861u, 2654435761u, 1958374283u};
	switch (hash_type) {
		case ORC_HASH_PRIME: return lcg_hash(n_dims, pos, prime);
		case ORC_HASH_COHERENT_PRIME: return lcg_hash(n_dims, pos, coherent);
		case ORC_HASH_REVERSED_PRIME: return lcg_hash(n_dims, pos, reversed);
		case ORC_HASH_RNG: { // common_device.h:663-676
			const uint32_t bits_per_dim = 64 / n_dims;
			uint64_t step = 0;
			for (uint32_t i = 0; i < n_dims; ++i) step ^= (uint64_t)pos[i] << (i * bits_per_dim);
			uint64_t st[2];
			pcg_seed(st, 1337, 1);
			pcg_advance(st, (int64_t)step);
			return pcg_next_uint(st);
		}
	}
	return 0;
}

inline uint32_t grid_index(uint32_t n_dims, uint32_t hash_type, uint32_t grid_type, uint32_t hashmap_size, uint32_t resolution, const uint32_t* pos) {
	uint32_t stride = 1;
	uint32_t index = 0;
	// uint32 arithmetic wraps; the overflow quirk of SURVEY 8a-G3 falls out of this loop unchanged
	for (uint32_t dim = 0; dim < n_dims && stride <= hashmap_size; ++dim) {
		index += pos[dim] * stride;
		stride *= resolution;
	}
	if (grid_type == ORC_GRID_HASH && hashmap_size < stride) {
		index = grid_hash(n_dims, hash_type, pos);
	}
	return index % hashmap_size;
}

inline float grid_scale(uint32_t level, float log2_per_level_scale, uint32_t base_resolution) {
	return exp2f(level * log2_per_level_scale) * base_resolution - 1.0f;
}
inline uint32_t grid_resolution(float scale) { return (uint32_t)ceilf(scale) + 1; }

inline uint32_t powi(uint32_t base, uint32_t exponent) { // common.h powi
	uint32_t result = 1;
	for (uint32_t i = 0; i < exponent; ++i) result *= base;
	return result;
}

inline float smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
inline float smoothstep_derivative(float v) { return 6 * v * (1.0f - v); }

inline uint32_t pos_fract(float input, float scale, uint32_t interpolation, float* pos, float* pos_derivative) {
	float p = fmaf(scale, input, 0.5f);
	const float tmp = floorf(p);
	const uint32_t cell = (uint32_t)(int)tmp;
	p -= tmp;
	if (interpolation == ORC_INTERP_SMOOTHSTEP) {
		if (pos_derivative) *pos_derivative = smoothstep_derivative(p);
		*pos = smoothstep(p);
	} else {
		if (pos_derivative) *pos_derivative = 1.0f;
		*pos = p;
	}
	return cell;
}

// quartic kernel (common_device.h:905-920)
inline float quartic(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float tmp = fmaxf(1 - u * u, 0.0f);
	return ((float)15 / 16) * tmp * tmp;
}
inline float quartic_cdf_deriv(float x, float inv_radius) { return quartic(x, inv_radius) * inv_radius; }
inline float quartic_cdf(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float u2 = u * u;
	const float u4 = u2 * u2;
	return fmaxf(0.0f, fminf(1.0f, ((float)15 / 16) * u * (1 - ((float)2 / 3) * u2 + ((float)1 / 5) * u4) + 0.5f));
}

struct MlpLayout {
	std::vector<uint32_t> rows, cols;
	std::vector<size_t> offset;
	size_t total = 0;
};

MlpLayout mlp_layout(const orc_mlp_t* m) {
	MlpLayout L;
	auto push = [&](uint32_t r, uint32_t c) {
		L.rows.push_back(r); L.cols.push_back(c); L.offset.push_back(L.total); L.total += (size_t)r * c;
	};
	if (m->n_hidden_layers == 0) { // cutlass_mlp.cu:64-67
		push(m->out_width, m->in_width);
	} else {
		push(m->width, m->in_width);
		for (uint32_t i = 0; i + 1 < m->n_hidden_layers; ++i) push(m->width, m->width);
		push(m->out_width, m->width);
	}
	return L;
}

// y[rows] = W[rows x cols] * x[cols]; fp16 in, fp16 out, selectable accumulation
inline void matvec(uint32_t acc_mode, const uint16_t* W, uint32_t rows, uint32_t cols, const float* xf, uint16_t* y) {
	for (uint32_t r = 0; r < rows; ++r) {
		const uint16_t* w = W + (size_t)r * cols;
		if (acc_mode == ORC_ACC_FP32) {
			float acc = 0.0f;
			for (uint32_t k = 0; k < cols; ++k) acc = fmaf(h2f(w[k]), xf[k], acc);
			y[r] = f2h(acc);
		} else {
			// wmma-like: fp16 accumulator, one rounding per 16-deep k step (fully_fused_mlp.cu:68, cutlass_matmul.h:67)
			uint16_t acc = 0;
			for (uint32_t k0 = 0; k0 < cols; k0 += 16) {
				float part = h2f(acc);
				for (uint32_t k = k0; k < std::min(cols, k0 + 16); ++k) part = fmaf(h2f(w[k]), xf[k], part);
				acc = f2h(part);
			}
			y[r] = acc;
		}
	}
}

// y[cols] = W^T * g[rows]
inline void matvec_t(uint32_t acc_mode, const uint16_t* W, uint32_t rows, uint32_t cols, const float* gf, uint16_t* y) {
	for (uint32_t c = 0; c < cols; ++c) {
		if (acc_mode == ORC_ACC_FP32) {
			float acc = 0.0f;
			for (uint32_t r = 0; r < rows; ++r) acc = fmaf(h2f(W[(size_t)r * cols + c]), gf[r], acc);
			y[c] = f2h(acc);
		} else {
			uint16_t acc = 0;
			for (uint32_t r0 = 0; r0 < rows; r0 += 16) {
				float part = h2f(acc);
				for (uint32_t r = r0; r < std::min(rows, r0 + 16); ++r) part = fmaf(h2f(W[(size_t)r * cols + c]), gf[r], part);
				acc = f2h(part);
			}
			y[c] = acc;
		}
	}
}

} // namespace

extern "C" {

uint16_t orc_float_to_half(float f) { return f2h(f); }
float orc_half_to_float(uint16_t h) { return h2f(h); }
uint16_t orc_double_to_half(double d) { return d2h(d); }

void orc_cast_float_to_half(size_t n, const float* in, uint16_t* out) {
#pragma omp parallel for schedule(static)
	for (size_t i = 0; i < n; ++i) out[i] = f2h(in[i]);
}
void orc_cast_half_to_float(size_t n, const uint16_t* in, float* out) {
#pragma omp parallel for schedule(static)
	for (size_t i = 0; i < n; ++i) out[i] = h2f(in[i]);
}

void orc_pcg32_seed(uint64_t* st, uint64_t initstate, uint64_t initseq) { pcg_seed(st, initstate, initseq); }
uint32_t orc_pcg32_next_uint(uint64_t* st) { return pcg_next_uint(st); }
float orc_pcg32_next_float(uint64_t* st) { return pcg_next_float(st); }
void orc_pcg32_advance(uint64_t* st, int64_t delta) { pcg_advance(st, delta); }

void orc_seed_seq2(uint32_t seed, uint32_t* out2) {
	std::seed_seq seq{seed};
	std::vector<uint32_t> seeds(2);
	seq.generate(seeds.begin(), seeds.end());
	out2[0] = seeds[0];
	out2[1] = seeds[1];
}

void orc_xavier_uniform(uint64_t* st, float* out, uint32_t rows, uint32_t cols, float scale) {
	// gpu_matrix.h:284-299: fan_in + fan_out = cols + rows
	scale *= std::sqrt(6.0f / (float)(rows + cols));
	const size_t n = (size_t)rows * cols;
	for (size_t i = 0; i < n; ++i) out[i] = pcg_next_float(st) * 2.0f * scale - scale;
}

void orc_generate_random_uniform(uint64_t* st, size_t n, float* out, float lower, float upper) {
	// random.h:40-65: N_TO_GENERATE = 4, 128 threads per block, thread i advances a COPY of the rng by 4*i and writes
	// elements i + n_threads_total * j.
	const size_t n_threads = (n + 3) / 4;
	const size_t n_blocks = (n_threads + 127) / 128;
	const size_t total_threads = n_blocks * 128;
#pragma omp parallel for schedule(static)
	for (size_t i = 0; i < total_threads; ++i) {
		uint64_t local[2] = {st[0], st[1]};
		pcg_advance(local, (int64_t)(i * 4));
		for (size_t j = 0; j < 4; ++j) {
			const size_t idx = i + total_threads * j;
			if (idx >= n) break;
			const float val = pcg_next_float(local);
			out[idx] = val * (upper - lower) + lower;
		}
	}
	pcg_advance(st, (int64_t)n);
}

// ---------------------------------------------------------------------------------------------------------
int orc_grid_setup(orc_grid_t* g) {
	if (g->n_levels > ORC_MAX_LEVELS || g->n_pos_dims < 1 || g->n_pos_dims > 7) return -1;
	const float log2_scale = std::log2(g->per_level_scale); // float overload, as grid.h:694 / :784
	uint32_t offset = 0;
	for (uint32_t i = 0; i < g->n_levels; ++i) {
		const float scale = grid_scale(i, log2_scale, g->base_resolution);
		const uint32_t resolution = grid_resolution(scale);
		g->scales[i] = scale;
		g->resolutions[i] = resolution;

		const uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
		uint32_t params_in_level = std::pow((float)resolution, g->n_pos_dims) > (float)max_params ? max_params : powi(resolution, g->n_pos_dims);
		params_in_level = (params_in_level + 7u) / 8u * 8u; // next_multiple(.., 8u)
		if (g->grid_type == ORC_GRID_DENSE) {
		} else if (g->grid_type == ORC_GRID_TILED) {
			params_in_level = std::min(params_in_level, powi(g->base_resolution, g->n_pos_dims));
		} else if (g->grid_type == ORC_GRID_HASH) {
			params_in_level = std::min(params_in_level, 1u << g->log2_hashmap_size);
		} else {
			return -2;
		}
		g->offsets[i] = offset;
		offset += params_in_level;
	}
	g->offsets[g->n_levels] = offset;
	g->n_params = offset * g->n_features_per_level;
	return 0;
}

uint32_t orc_grid_hash(uint32_t n_dims, uint32_t hash_type, const uint32_t* pos_grid) { return grid_hash(n_dims, hash_type, pos_grid); }
uint32_t orc_grid_index(uint32_t n_dims, uint32_t hash_type, uint32_t grid_type, uint32_t hashmap_size, uint32_t resolution, const uint32_t* pos_grid) {
	return grid_index(n_dims, hash_type, grid_type, hashmap_size, resolution, pos_grid);
}
uint32_t orc_pos_fract(float input, float scale, uint32_t interpolation, float* frac, float* frac_derivative) {
	return pos_fract(input, scale, interpolation, frac, frac_derivative);
}

void orc_grid_forward(const orc_grid_t* g, uint32_t n, const float* x, const uint16_t* grid, uint16_t* out, uint32_t out_stride,
                      uint32_t* indices, float* dy_dx) {
	const uint32_t D = g->n_pos_dims, F = g->n_features_per_level, L = g->n_levels;
	const uint32_t n_corners = 1u << D;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t level = 0; level < L; ++level) {
			const uint16_t* lgrid = grid + (size_t)g->offsets[level] * F;
			const uint32_t hashmap_size = g->offsets[level + 1] - g->offsets[level];
			const float scale = g->scales[level];
			const uint32_t resolution = g->resolutions[level];

			float pos[8], pos_derivative[8];
			uint32_t pos_grid[8];
			for (uint32_t dim = 0; dim < D; ++dim) {
				pos_grid[dim] = pos_fract(x[(size_t)i * D + dim], scale, g->interpolation, &pos[dim], &pos_derivative[dim]);
			}

			uint16_t* o = out ? out + (size_t)i * out_stride + level * F : nullptr;

			if (g->interpolation == ORC_INTERP_NEAREST) { // grid.h:121-140
				const uint32_t index = grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, pos_grid);
				if (indices) indices[((size_t)i * L + level) * n_corners] = index;
				if (o) for (uint32_t f = 0; f < F; ++f) o[f] = lgrid[(size_t)index * F + f];
				if (dy_dx) for (uint32_t f = 0; f < F; ++f) for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)i * L * F + level * F + f) * D + d] = 0.0f;
				continue;
			}

			// N-linear interpolation, grid.h:142-169: acc = hfma((half)weight, value, acc), corner order idx = 0..2^D-1
			uint16_t result[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (uint32_t idx = 0; idx < n_corners; ++idx) {
				float weight = 1;
				uint32_t local[8];
				for (uint32_t dim = 0; dim < D; ++dim) {
					if ((idx & (1u << dim)) == 0) {
						weight *= 1 - pos[dim];
						local[dim] = pos_grid[dim];
					} else {
						weight *= pos[dim];
						local[dim] = pos_grid[dim] + 1;
					}
				}
				const uint32_t index = grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, local);
				if (indices) indices[((size_t)i * L + level) * n_corners + idx] = index;
				const uint16_t wh = f2h(weight);
				for (uint32_t f = 0; f < F; ++f) result[f] = hfma(wh, lgrid[(size_t)index * F + f], result[f]);
			}
			if (o) for (uint32_t f = 0; f < F; ++f) o[f] = result[f];

			if (dy_dx) { // grid.h:172-211
				for (uint32_t f = 0; f < F; ++f) for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)i * L * F + level * F + f) * D + d] = 0.0f;
				for (uint32_t grad_dim = 0; grad_dim < D; ++grad_dim) {
					for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
						float weight = scale;
						uint32_t local[8];
						for (uint32_t ngd = 0; ngd < D - 1; ++ngd) {
							const uint32_t dim = ngd >= grad_dim ? (ngd + 1) : ngd;
							if ((idx & (1u << ngd)) == 0) {
								weight *= 1 - pos[dim];
								local[dim] = pos_grid[dim];
							} else {
								weight *= pos[dim];
								local[dim] = pos_grid[dim] + 1;
							}
						}
						local[grad_dim] = pos_grid[grad_dim];
						const uint32_t il = grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, local);
						local[grad_dim] = pos_grid[grad_dim] + 1;
						const uint32_t ir = grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, local);
						for (uint32_t f = 0; f < F; ++f) {
							float& gr = dy_dx[((size_t)i * L * F + level * F + f) * D + grad_dim];
							gr += weight * (h2f(lgrid[(size_t)ir * F + f]) - h2f(lgrid[(size_t)il * F + f])) * pos_derivative[grad_dim];
						}
					}
				}
			}
		}
		if (out) for (uint32_t j = L * F; j < out_stride; ++j) out[(size_t)i * out_stride + j] = 0; // grid.h:749-758
	}
}

void orc_grid_backward(const orc_grid_t* g, uint32_t n, const float* x, const uint16_t* dL_dy, uint32_t dy_stride,
                       uint16_t* grad, float* grad_f32) {
	const uint32_t D = g->n_pos_dims, F = g->n_features_per_level, L = g->n_levels;
	const uint32_t n_corners = 1u << D;
	// F == 1: the reference accumulates in an fp32 scratch and casts once at the end (grid.h:660, 850-886)
	std::vector<float> scratch32;
	if (F == 1 && !grad_f32) {
		scratch32.assign(g->n_params, 0.0f);
		grad_f32 = scratch32.data();
	}
	uint16_t* grad_half_atomic = F == 1 ? nullptr : grad;
	// levels are independent: parallelise over levels, keep sample order inside a level deterministic
#pragma omp parallel for schedule(dynamic, 1)
	for (uint32_t level = 0; level < L; ++level) {
		uint16_t* lgrad = grad_half_atomic ? grad_half_atomic + (size_t)g->offsets[level] * F : nullptr;
		float* lgrad32 = grad_f32 ? grad_f32 + (size_t)g->offsets[level] * F : nullptr;
		const uint32_t hashmap_size = g->offsets[level + 1] - g->offsets[level];
		const float scale = g->scales[level];
		const uint32_t resolution = g->resolutions[level];
		for (uint32_t i = 0; i < n; ++i) {
			float pos[8];
			uint32_t pos_grid[8];
			for (uint32_t dim = 0; dim < D; ++dim) pos_grid[dim] = pos_fract(x[(size_t)i * D + dim], scale, g->interpolation, &pos[dim], nullptr);
			const uint16_t* gy = dL_dy + (size_t)i * dy_stride + level * F;

			auto add = [&](const uint32_t* local, float weight) { // grid.h:252-255: (GRAD_T)weight * grad, then atomic add
				const uint32_t index = grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, local);
				const uint16_t wh = f2h(weight);
				for (uint32_t f = 0; f < F; ++f) {
					if (lgrad) lgrad[(size_t)index * F + f] = hadd(lgrad[(size_t)index * F + f], hmul(wh, gy[f]));
					if (lgrad32) lgrad32[(size_t)index * F + f] += F == 1 ? weight * h2f(gy[f]) : h2f(hmul(wh, gy[f]));
				}
			};

			if (g->interpolation == ORC_INTERP_NEAREST) {
				add(pos_grid, 1.0f);
				continue;
			}
			for (uint32_t idx = 0; idx < n_corners; ++idx) {
				float weight = 1;
				uint32_t local[8];
				for (uint32_t dim = 0; dim < D; ++dim) {
					if ((idx & (1u << dim)) == 0) {
						weight *= 1 - pos[dim];
						local[dim] = pos_grid[dim];
					} else {
						weight *= pos[dim];
						local[dim] = pos_grid[dim] + 1;
					}
				}
				add(local, weight);
			}
		}
	}
	if (F == 1 && grad) { // grid.h:882-886
		for (size_t i = 0; i < g->n_params; ++i) grad[i] = f2h(grad_f32[i]);
	}
}

// Exact mode: every contribution is the reference's fp16 product (half)weight * dL_dy (grid.h:254); the contributions of an
// entry are summed EXACTLY (integers in units of 2^-24, the fp16 subnormal LSB) and rounded to fp16 once.  This is the
// order-independent limit of the reference's fp16 atomic accumulation and what the MI355X scatter kernel computes.
void orc_grid_backward_exact(const orc_grid_t* g, uint32_t n, const float* x, const uint16_t* dL_dy, uint32_t dy_stride, uint16_t* grad, int accumulate) {
	const uint32_t D = g->n_pos_dims, F = g->n_features_per_level, L = g->n_levels;
	const uint32_t n_corners = 1u << D;
	auto to_fixed = [](uint16_t h) -> int64_t { return (int64_t)std::llround(std::ldexp((double)h2f(h), 24)); }; // exact: fp16 * 2^24 is an integer
#pragma omp parallel for schedule(dynamic, 1)
	for (uint32_t level = 0; level < L; ++level) {
		const uint32_t hashmap_size = g->offsets[level + 1] - g->offsets[level];
		std::vector<int64_t> sum((size_t)hashmap_size * F, 0);
		uint16_t* lgrad = grad + (size_t)g->offsets[level] * F;
		if (accumulate) for (size_t k = 0; k < sum.size(); ++k) sum[k] = to_fixed(lgrad[k]);
		const float scale = g->scales[level];
		const uint32_t resolution = g->resolutions[level];
		for (uint32_t i = 0; i < n; ++i) {
			float pos[8];
			uint32_t pos_grid[8];
			for (uint32_t dim = 0; dim < D; ++dim) pos_grid[dim] = pos_fract(x[(size_t)i * D + dim], scale, g->interpolation, &pos[dim], nullptr);
			const uint16_t* gy = dL_dy + (size_t)i * dy_stride + level * F;
			auto add = [&](const uint32_t* local, float weight) {
				const uint32_t index = grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, local);
				const uint16_t wh = f2h(weight);
				for (uint32_t f = 0; f < F; ++f) sum[(size_t)index * F + f] += to_fixed(hmul(wh, gy[f]));
			};
			if (g->interpolation == ORC_INTERP_NEAREST) {
				add(pos_grid, 1.0f);
				continue;
			}
			for (uint32_t idx = 0; idx < n_corners; ++idx) {
				float weight = 1;
				uint32_t local[8];
				for (uint32_t dim = 0; dim < D; ++dim) {
					if ((idx & (1u << dim)) == 0) {
						weight *= 1 - pos[dim];
						local[dim] = pos_grid[dim];
					} else {
						weight *= pos[dim];
						local[dim] = pos_grid[dim] + 1;
					}
				}
				add(local, weight);
			}
		}
		// |sum| < 2^53 for any realistic batch: the conversion to double is exact, d2h rounds once
		for (size_t k = 0; k < sum.size(); ++k) lgrad[k] = d2h(std::ldexp((double)sum[k], -24));
	}
}

void orc_grid_backward_input(const orc_grid_t* g, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride, const float* dy_dx, float* dL_dx) {
	const uint32_t D = g->n_pos_dims, NF = g->n_levels * g->n_features_per_level;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		float result[8] = {0};
		for (uint32_t k = 0; k < NF; ++k) {
			const float dl = h2f(dL_dy[(size_t)i * dy_stride + k]);
			for (uint32_t d = 0; d < D; ++d) result[d] += dl * dy_dx[((size_t)i * NF + k) * D + d];
		}
		for (uint32_t d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = result[d];
	}
}

// Second-order terms of dL/dx = sum_k dL_dy_k * dy_k/dx (GridEncodingTemplated::backward_backward_input_impl, grid.h:902-1026):
// given dL/d(dL_dx), the gradients w.r.t. the grid (kernel_grid_backward_input_backward_grid, grid.h:352-454), w.r.t. dL_dy
// (kernel_grid_backward_input_backward_dLdoutput, grid.h:626-650) and w.r.t. x (kernel_grid_backward_input_backward_input,
// grid.h:456-624).  Operation order inside a (sample, level, feature pair) follows the kernels; across them the grid
// gradient is accumulated in sample order (fp16 like the reference's atomics, and optionally fp32), dL_dx in
// (level, feature pair) order per sample (the reference adds the per-thread partial sums with fp32 atomics in any order).
void orc_grid_backward_backward_input(const orc_grid_t* g, uint32_t n, const float* x, const float* dL_ddLdx, const uint16_t* dL_dy, uint32_t dy_stride,
                                      const uint16_t* grid, const float* dy_dx, uint16_t* grad, float* grad_f32, uint16_t* dL_ddLdy, float* dL_dx) {
	const uint32_t D = g->n_pos_dims, F = g->n_features_per_level, L = g->n_levels;
	const uint32_t FT = F < 2 ? F : 2; // N_FEATURES_PER_THREAD (grid.h:949)
	const bool nearest = g->interpolation == ORC_INTERP_NEAREST, smooth = g->interpolation == ORC_INTERP_SMOOTHSTEP;

	auto fractions = [&](uint32_t i, float scale, float* pos, float* d1, float* d2, uint32_t* cell) { // common_device.h:825-838
		for (uint32_t dim = 0; dim < D; ++dim) {
			float p = fmaf(scale, x[(size_t)i * D + dim], 0.5f);
			const float tmp = floorf(p);
			cell[dim] = (uint32_t)(int)tmp;
			p -= tmp;
			d2[dim] = smooth ? 6.0f - 12.0f * p : 0.0f;
			d1[dim] = smooth ? smoothstep_derivative(p) : 1.0f;
			pos[dim] = smooth ? smoothstep(p) : p;
		}
	};

	if ((grad || grad_f32) && !nearest) { // grid.h:352-454; d(dy_dx)/dgrid is zero without interpolation (:420-423)
		std::vector<float> scratch32;
		if (F == 1 && !grad_f32) {
			scratch32.assign(g->n_params, 0.0f);
			grad_f32 = scratch32.data();
		}
		uint16_t* grad_half_atomic = F == 1 ? nullptr : grad;
#pragma omp parallel for schedule(dynamic, 1)
		for (uint32_t level = 0; level < L; ++level) {
			uint16_t* lgrad = grad_half_atomic ? grad_half_atomic + (size_t)g->offsets[level] * F : nullptr;
			float* lgrad32 = grad_f32 ? grad_f32 + (size_t)g->offsets[level] * F : nullptr;
			const uint32_t hashmap_size = g->offsets[level + 1] - g->offsets[level];
			const float scale = g->scales[level];
			const uint32_t resolution = g->resolutions[level];
			for (uint32_t i = 0; i < n; ++i) {
				float pos[8], d1[8], d2[8];
				uint32_t cell[8];
				fractions(i, scale, pos, d1, d2, cell);
				const uint16_t* gy = dL_dy + (size_t)i * dy_stride + level * F;
				auto add = [&](const uint32_t* local, float weight) { // :393-396: (GRAD_T)weight * grad, atomic add
					const uint32_t index = grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, local);
					const uint16_t wh = f2h(weight);
					for (uint32_t f = 0; f < F; ++f) {
						if (lgrad) lgrad[(size_t)index * F + f] = hadd(lgrad[(size_t)index * F + f], hmul(wh, gy[f]));
						if (lgrad32) lgrad32[(size_t)index * F + f] += F == 1 ? weight * h2f(gy[f]) : h2f(hmul(wh, gy[f]));
					}
				};
				for (uint32_t grad_dim = 0; grad_dim < D; ++grad_dim) {
					const float grad_in = scale * dL_ddLdx[(size_t)i * D + grad_dim] * d1[grad_dim];
					for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
						float weight = grad_in;
						uint32_t local[8];
						for (uint32_t ngd = 0; ngd < D - 1; ++ngd) {
							const uint32_t dim = ngd >= grad_dim ? (ngd + 1) : ngd;
							if ((idx & (1u << ngd)) == 0) {
								weight *= 1 - pos[dim];
								local[dim] = cell[dim];
							} else {
								weight *= pos[dim];
								local[dim] = cell[dim] + 1;
							}
						}
						local[grad_dim] = cell[grad_dim];
						add(local, -weight);
						local[grad_dim] = cell[grad_dim] + 1;
						add(local, weight);
					}
				}
			}
		}
		if (F == 1 && grad) for (size_t k = 0; k < g->n_params; ++k) grad[k] = f2h(grad_f32[k]);
	}

	if (dL_ddLdy) { // grid.h:626-650
		const uint32_t NF = L * F;
#pragma omp parallel for schedule(static)
		for (uint32_t i = 0; i < n; ++i) {
			for (uint32_t k = 0; k < dy_stride; ++k) {
				float result = 0;
				if (k < NF) for (uint32_t dim = 0; dim < D; ++dim) result += dy_dx[((size_t)i * NF + k) * D + dim] * dL_ddLdx[(size_t)i * D + dim];
				dL_ddLdy[(size_t)i * dy_stride + k] = f2h(result);
			}
		}
	}

	if (dL_dx) { // grid.h:456-624
#pragma omp parallel for schedule(static)
		for (uint32_t i = 0; i < n; ++i) {
			float out[8] = {0};
			for (uint32_t level = 0; level < L && !nearest; ++level) {
				const uint16_t* lgrid = grid + (size_t)g->offsets[level] * F;
				const uint32_t hashmap_size = g->offsets[level + 1] - g->offsets[level];
				const float scale = g->scales[level];
				const uint32_t resolution = g->resolutions[level];
				float pos[8], d1[8], d2[8];
				uint32_t cell[8];
				fractions(i, scale, pos, d1, d2, cell);
				float diag[8], other[8];
				for (uint32_t gd = 0; gd < D; ++gd) {
					diag[gd] = scale * scale * dL_ddLdx[(size_t)i * D + gd] * d2[gd];
					other[gd] = scale * scale * dL_ddLdx[(size_t)i * D + gd] * d1[gd];
				}
				for (uint32_t feature = 0; feature < F; feature += FT) {
					const uint16_t* gy = dL_dy + (size_t)i * dy_stride + level * F + feature;
					auto calc = [&](const uint32_t* local, float weight) { // :531-541
						const size_t index = (size_t)grid_index(D, g->hash_type, g->grid_type, hashmap_size, resolution, local) * F + feature;
						float v = 0;
						for (uint32_t f = 0; f < FT; ++f) v += h2f(lgrid[index + f]) * h2f(gy[f]) * weight;
						return v;
					};
					for (uint32_t grad_dim = 0; grad_dim < D; ++grad_dim) {
						float grad_out = 0;
						for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
							if (smooth) { // diagonal part of the Hessian (zero for linear interpolation)
								float w = diag[grad_dim];
								uint32_t local[8];
								for (uint32_t ngd = 0; ngd < D - 1; ++ngd) {
									const uint32_t dim = ngd >= grad_dim ? (ngd + 1) : ngd;
									if ((idx & (1u << ngd)) == 0) {
										w *= 1 - pos[dim];
										local[dim] = cell[dim];
									} else {
										w *= pos[dim];
										local[dim] = cell[dim] + 1;
									}
								}
								local[grad_dim] = cell[grad_dim];
								grad_out += calc(local, -w);
								local[grad_dim] = cell[grad_dim] + 1;
								grad_out += calc(local, w);
							}
							if (D > 1) { // mixed part: d(dy/d[other])/d[grad_dim]
								for (uint32_t og = 0; og < D - 1; ++og) {
									const uint32_t rog = og >= grad_dim ? (og + 1) : og;
									float w = other[rog] * d1[grad_dim];
									uint32_t local[8];
									for (uint32_t ngd = 0; ngd < D - 1; ++ngd) {
										const uint32_t dim = ngd >= rog ? (ngd + 1) : ngd;
										if ((idx & (1u << ngd)) == 0) {
											if (dim != grad_dim) w *= 1 - pos[dim];
											else w *= -1;
											local[dim] = cell[dim];
										} else {
											if (dim != grad_dim) w *= pos[dim];
											local[dim] = cell[dim] + 1;
										}
									}
									local[rog] = cell[rog];
									grad_out += calc(local, -w);
									local[rog] = cell[rog] + 1;
									grad_out += calc(local, w);
								}
							}
						}
						out[grad_dim] += grad_out;
					}
				}
			}
			for (uint32_t d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = out[d];
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
void orc_oneblob_forward(uint32_t n, uint32_t n_dims, uint32_t n_bins, const float* x, uint16_t* out, uint32_t out_stride) {
	uint32_t log2_bins = 0;
	while ((1u << log2_bins) < n_bins) ++log2_bins;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < n_dims; ++j) {
			const float xv = x[(size_t)i * n_dims + j];
			// oneblob.h:47-67: left CDF of every bin = three wrapped kernels; right CDF = left CDF of the next bin,
			// the last bin's right CDF is bin 0's left CDF + 1.
			auto left_cdf = [&](uint32_t bin) {
				const float lb = scalbnf((float)bin, -(int)log2_bins);
				return quartic_cdf(lb - xv, (float)n_bins) + quartic_cdf(lb - xv - 1.0f, (float)n_bins) + quartic_cdf(lb - xv + 1.0f, (float)n_bins);
			};
			for (uint32_t k = 0; k < n_bins; ++k) {
				const float l = left_cdf(k);
				float r = left_cdf((k + 1) & (n_bins - 1));
				if (k == n_bins - 1) r += 1;
				out[(size_t)i * out_stride + j * n_bins + k] = f2h(r - l);
			}
		}
		for (uint32_t j = n_dims * n_bins; j < out_stride; ++j) out[(size_t)i * out_stride + j] = f2h(1.0f); // oneblob.h:207-209
	}
}

void orc_oneblob_backward_input(uint32_t n, uint32_t n_dims, uint32_t n_bins, const float* x, const uint16_t* dL_dy, uint32_t dy_stride, float* dL_dx) {
	uint32_t log2_bins = 0;
	while ((1u << log2_bins) < n_bins) ++log2_bins;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < n_dims; ++j) { // oneblob.h:130-164
			const float xv = x[(size_t)i * n_dims + j];
			float result = 0;
			float left = quartic_cdf_deriv(-xv, (float)n_bins) + quartic_cdf_deriv(-xv - 1.0f, (float)n_bins) + quartic_cdf_deriv(-xv + 1.0f, (float)n_bins);
			for (uint32_t k = 0; k < n_bins; ++k) {
				const float rb = scalbnf((float)(k + 1), -(int)log2_bins);
				const float right = quartic_cdf_deriv(rb - xv, (float)n_bins) + quartic_cdf_deriv(rb - xv - 1.0f, (float)n_bins) + quartic_cdf_deriv(rb - xv + 1.0f, (float)n_bins);
				const float deriv = left - right;
				left = right;
				result += h2f(dL_dy[(size_t)i * dy_stride + j * n_bins + k]) * deriv;
			}
			dL_dx[(size_t)i * n_dims + j] = result;
		}
	}
}

void orc_identity_forward(uint32_t n, uint32_t n_dims, float scale, float offset, const float* x, uint16_t* out, uint32_t out_stride) {
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < out_stride; ++j) { // identity.h:46-67
			out[(size_t)i * out_stride + j] = j >= n_dims ? f2h(1.0f) : f2h(x[(size_t)i * n_dims + j] * scale + offset);
		}
	}
}

void orc_identity_backward_input(uint32_t n, uint32_t n_dims, float scale, const uint16_t* dL_dy, uint32_t dy_stride, float* dL_dx) {
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < n_dims; ++j) { // identity.h:70-85: rounds through T before widening
			dL_dx[(size_t)i * n_dims + j] = h2f(f2h(h2f(dL_dy[(size_t)i * dy_stride + j]) * scale));
		}
	}
}

// ---- Frequency (encodings/frequency.h:44-101) and TriangleWave (encodings/triangle_wave.h:44-107): AoS output, pad dims = 1.
// The reference evaluates __sinf / __cosf (hardware approximations); here sinf / cosf -- parity for Frequency is by tolerance.
void orc_frequency_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* x, uint16_t* out, uint32_t out_stride, float* dy_dx) {
	const uint32_t fan_out_encoded = n_dims * n_frequencies * 2;
	const float PI = 3.14159265358979323846f;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < out_stride; ++j) {
			if (j >= fan_out_encoded) { out[(size_t)i * out_stride + j] = f2h(1.0f); continue; }
			const uint32_t feature = j / (n_frequencies * 2);
			const uint32_t log2_frequency = (j / 2) % n_frequencies;
			const float phase_shift = (j % 2) * (PI / 2);
			const float v = scalbnf(x[(size_t)i * n_dims + feature], (int)log2_frequency);
			const float input = v * PI + phase_shift;
			out[(size_t)i * out_stride + j] = f2h(sinf(input));
			if (dy_dx) dy_dx[(size_t)i * fan_out_encoded + j] = scalbnf(1.0f, (int)log2_frequency) * PI * cosf(input);
		}
	}
}

void orc_trianglewave_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* x, uint16_t* out, uint32_t out_stride, float* dy_dx) {
	const uint32_t fan_out_encoded = n_dims * n_frequencies;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < out_stride; ++j) {
			if (j >= fan_out_encoded) { out[(size_t)i * out_stride + j] = f2h(1.0f); continue; }
			const uint32_t feature = j / n_frequencies;
			const int log2_frequency = (int)(j - feature * n_frequencies);
			const float v = scalbnf(x[(size_t)i * n_dims + feature], log2_frequency - 1);
			const float val = v + log2_frequency * 0.25f; // small frequency-dependent phase shift (triangle_wave.h:71-72)
			out[(size_t)i * out_stride + j] = f2h(fabsf(val - floorf(val) - 0.5f) * 4 - 1);
			if (dy_dx) dy_dx[(size_t)i * fan_out_encoded + j] = scalbnf((int)floorf(val * 2.0f) % 2 == 0 ? -1.0f : 1.0f, log2_frequency + 1);
		}
	}
}

// frequency.h:82-101 / triangle_wave.h:83-107: dL_dx(j) = sum_k dL_dy(j * outputs_per_input + k) * dy_dx
void orc_periodic_backward_input(uint32_t n, uint32_t n_dims, uint32_t outputs_per_input, const uint16_t* dL_dy, uint32_t dy_stride, const float* dy_dx, float* dL_dx) {
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < n_dims; ++j) {
			float result = 0;
			for (uint32_t k = 0; k < outputs_per_input; ++k)
				result += h2f(dL_dy[(size_t)i * dy_stride + j * outputs_per_input + k]) * dy_dx[((size_t)i * n_dims + j) * outputs_per_input + k];
			dL_dx[(size_t)i * n_dims + j] = result;
		}
	}
}

// ---- SphericalHarmonics (encodings/spherical_harmonics.h:44-108; common_device.h:339-420 sh_enc, :421-700 sh_enc_grad).
// The reference hard-codes the 64 polynomials of degree <= 8 (generated from the recurrences of P.-P. Sloan, "Stupid Spherical
// Harmonics Tricks", appendix A1).  This restatement evaluates the SAME polynomials through those recurrences:
//   Y_l^m(x, y, z) = N_l^|m| * Q_l^|m|(z) * { Im (x + iy)^|m|  (m < 0),  1  (m = 0),  Re (x + iy)^m  (m > 0) }
//   Q_m^m = (-1)^m (2m - 1)!!,  Q_{m+1}^m = (2m + 1) z Q_m^m,  Q_l^m = ((2l - 1) z Q_{l-1}^m - (l + m - 1) Q_{l-2}^m) / (l - m)
//   N_l^m = sqrt((2l + 1) / (4 pi) * (l - m)! / (l + m)!) * (m ? sqrt 2 : 1)
// -- polynomials in z times harmonic polynomials in (x, y), identical as functions (also off the unit sphere), summed in a
// different floating-point order (differences ~1e-7 relative, below fp16 resolution).  Output index l^2 + l + m.
static void sh_norms(uint32_t degree, float* norms /* [degree][degree] indexed [l][m] */) {
	for (uint32_t l = 0; l < degree; ++l) {
		for (uint32_t m = 0; m <= l; ++m) {
			double ratio = 1.0; // (l - m)! / (l + m)!
			for (uint32_t k = l - m + 1; k <= l + m; ++k) ratio /= (double)k;
			norms[l * degree + m] = (float)(std::sqrt((2.0 * l + 1.0) / (4.0 * 3.14159265358979323846) * ratio) * (m ? std::sqrt(2.0) : 1.0));
		}
	}
}

// values (and optionally the gradient against dL_dy) of all degree^2 functions at one point
static void sh_eval(uint32_t degree, const float* norms, float x, float y, float z, float* values, const uint16_t* dL_dy, float* grad3) {
	float gx = 0, gy = 0, gz = 0;
	float c = 1, s = 0, cp = 0, sp = 0; // Re / Im (x + iy)^m and ^(m-1)
	float qmm = 1;                      // Q_m^m
	for (uint32_t m = 0; m < degree; ++m) {
		if (m > 0) {
			cp = c;
			sp = s;
			c = x * cp - y * sp;
			s = x * sp + y * cp;
			qmm = -qmm * (float)(2 * m - 1);
		}
		float q2 = 0, q1 = 0, d2 = 0, d1 = 0; // Q_{l-2}^m, Q_{l-1}^m and their z derivatives
		for (uint32_t l = m; l < degree; ++l) {
			float q, dq;
			if (l == m) { q = qmm; dq = 0; }
			else if (l == m + 1) { q = (float)(2 * m + 1) * z * q1; dq = (float)(2 * m + 1) * q1; }
			else {
				q = ((float)(2 * l - 1) * z * q1 - (float)(l + m - 1) * q2) / (float)(l - m);
				dq = ((float)(2 * l - 1) * (q1 + z * d1) - (float)(l + m - 1) * d2) / (float)(l - m);
			}
			q2 = q1; q1 = q; d2 = d1; d1 = dq;
			const float nq = norms[l * degree + m] * q, ndq = norms[l * degree + m] * dq;
			const uint32_t base = l * l + l;
			if (m == 0) {
				if (values) values[base] = nq;
				if (dL_dy) gz += h2f(dL_dy[base]) * ndq;
			} else {
				if (values) { values[base + m] = nq * c; values[base - m] = nq * s; }
				if (dL_dy) {
					const float gp = h2f(dL_dy[base + m]), gm = h2f(dL_dy[base - m]);
					// d/dx (x + iy)^m = m (x + iy)^(m-1), d/dy = i m (x + iy)^(m-1)
					gx += gp * (nq * (float)m * cp) + gm * (nq * (float)m * sp);
					gy += gp * (nq * -(float)m * sp) + gm * (nq * (float)m * cp);
					gz += gp * (ndq * c) + gm * (ndq * s);
				}
			}
		}
	}
	if (grad3) { grad3[0] = gx; grad3[1] = gy; grad3[2] = gz; }
}

// out: [n][out_stride] half; the padding columns come FIRST (spherical_harmonics.h:58-64), then the degree^2 values
void orc_sh_forward(uint32_t n, uint32_t degree, const float* x, uint16_t* out, uint32_t out_stride) {
	std::vector<float> norms(degree * degree);
	sh_norms(degree, norms.data());
	const uint32_t n_to_pad = out_stride - degree * degree;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		float values[64];
		sh_eval(degree, norms.data(), x[(size_t)i * 3] * 2.f - 1.f, x[(size_t)i * 3 + 1] * 2.f - 1.f, x[(size_t)i * 3 + 2] * 2.f - 1.f, values, nullptr, nullptr);
		for (uint32_t j = 0; j < n_to_pad; ++j) out[(size_t)i * out_stride + j] = f2h(1.0f);
		for (uint32_t j = 0; j < degree * degree; ++j) out[(size_t)i * out_stride + n_to_pad + j] = f2h(values[j]);
	}
}

// spherical_harmonics.h:78-105: dL_dx = 2 * sum_k dL_dy_k dY_k/d(direction)
void orc_sh_backward_input(uint32_t n, uint32_t degree, const float* x, const uint16_t* dL_dy, uint32_t dy_stride, float* dL_dx) {
	std::vector<float> norms(degree * degree);
	sh_norms(degree, norms.data());
	const uint32_t n_to_pad = dy_stride - degree * degree;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		float g[3];
		sh_eval(degree, norms.data(), x[(size_t)i * 3] * 2.f - 1.f, x[(size_t)i * 3 + 1] * 2.f - 1.f, x[(size_t)i * 3 + 2] * 2.f - 1.f, nullptr, dL_dy + (size_t)i * dy_stride + n_to_pad, g);
		for (uint32_t d = 0; d < 3; ++d) dL_dx[(size_t)i * 3 + d] = 2.0f * g[d];
	}
}

// ---------------------------------------------------------------------------------------------------------
size_t orc_mlp_n_params(const orc_mlp_t* m) { return mlp_layout(m).total; }

void orc_mlp_init_params(const orc_mlp_t* m, uint64_t* st, float* params, float scale) {
	const MlpLayout L = mlp_layout(m);
	for (size_t i = 0; i < L.rows.size(); ++i) orc_xavier_uniform(st, params + L.offset[i], L.rows[i], L.cols[i], scale);
}

void orc_mlp_forward(const orc_mlp_t* m, uint32_t n, const uint16_t* x, const uint16_t* params, uint16_t* hidden, uint16_t* out) {
	const MlpLayout L = mlp_layout(m);
	const uint32_t nh = m->n_hidden_layers;
	const uint32_t maxw = std::max(std::max(m->in_width, m->width), m->out_width);
#pragma omp parallel
	{
		std::vector<float> cur(maxw);
		std::vector<uint16_t> nxt(maxw);
#pragma omp for schedule(static)
		for (uint32_t i = 0; i < n; ++i) {
			for (uint32_t k = 0; k < m->in_width; ++k) cur[k] = h2f(x[(size_t)i * m->in_width + k]);
			for (uint32_t l = 0; l < nh; ++l) {
				matvec(m->acc_mode, params + L.offset[l], L.rows[l], L.cols[l], cur.data(), nxt.data());
				for (uint32_t r = 0; r < m->width; ++r) {
					const uint16_t a = activation_fwd(m->activation, nxt[r]);
					if (hidden) hidden[((size_t)l * n + i) * m->width + r] = a;
					cur[r] = h2f(a);
				}
			}
			const size_t last = L.rows.size() - 1;
			matvec(m->acc_mode, params + L.offset[last], L.rows[last], L.cols[last], cur.data(), nxt.data());
			for (uint32_t r = 0; r < m->out_width; ++r) out[(size_t)i * m->out_width + r] = activation_fwd(m->output_activation, nxt[r]);
		}
	}
}

void orc_mlp_backward(const orc_mlp_t* m, uint32_t n, const uint16_t* x, const uint16_t* params, const uint16_t* hidden, const uint16_t* out,
                      const uint16_t* dL_dout, uint16_t* dL_dx, uint16_t* grads, float* grads_f32, int accumulate) {
	const MlpLayout L = mlp_layout(m);
	const uint32_t nh = m->n_hidden_layers;
	const uint32_t W = m->width;
	// dH for every hidden layer, [nh][n][W] half -- the reference's backward_tmp (fully_fused_mlp.cu:751-755)
	std::vector<uint16_t> dH((size_t)nh * n * W);
	std::vector<uint16_t> dY((size_t)n * m->out_width);

#pragma omp parallel
	{
		std::vector<float> gf(std::max(W, m->out_width));
		std::vector<uint16_t> tmp(std::max(W, m->in_width));
#pragma omp for schedule(static)
		for (uint32_t i = 0; i < n; ++i) {
			// output activation transfer first (fully_fused_mlp.cu:757-762)
			for (uint32_t r = 0; r < m->out_width; ++r) {
				uint16_t g = dL_dout[(size_t)i * m->out_width + r];
				if (m->output_activation != ORC_ACT_NONE) g = activation_bwd(m->output_activation, g, out[(size_t)i * m->out_width + r]);
				dY[(size_t)i * m->out_width + r] = g;
				gf[r] = h2f(g);
			}
			const size_t last = L.rows.size() - 1;
			if (nh == 0) {
				if (dL_dx) matvec_t(m->acc_mode, params + L.offset[0], L.rows[0], L.cols[0], gf.data(), dL_dx + (size_t)i * m->in_width);
				continue;
			}
			// dH_{h-1} = (Wout^T dY) . act'(H_{h-1})
			matvec_t(m->acc_mode, params + L.offset[last], L.rows[last], L.cols[last], gf.data(), tmp.data());
			for (int l = (int)nh - 1; l >= 0; --l) {
				for (uint32_t r = 0; r < W; ++r) {
					const uint16_t g = activation_bwd(m->activation, tmp[r], hidden[((size_t)l * n + i) * W + r]);
					dH[((size_t)l * n + i) * W + r] = g;
					gf[r] = h2f(g);
				}
				if (l > 0) {
					matvec_t(m->acc_mode, params + L.offset[l], L.rows[l], L.cols[l], gf.data(), tmp.data());
				} else if (dL_dx) {
					matvec_t(m->acc_mode, params + L.offset[0], L.rows[0], L.cols[0], gf.data(), dL_dx + (size_t)i * m->in_width);
				}
			}
		}
	}

	if (!grads && !grads_f32) return;

	// weight gradients: dW_l = dOut_l * In_l^T summed over the batch (fully_fused_mlp.cu:785,819,828).
	// Accumulated in double here ("true" value); the reference rounds per split-K slice in fp16.
	for (size_t l = 0; l < L.rows.size(); ++l) {
		const uint32_t rows = L.rows[l], cols = L.cols[l];
		const uint16_t* dO; uint32_t dO_stride;
		const uint16_t* In; uint32_t In_stride;
		if (l == L.rows.size() - 1) { dO = dY.data(); dO_stride = m->out_width; } else { dO = dH.data() + (size_t)l * n * W; dO_stride = W; }
		if (l == 0) { In = x; In_stride = m->in_width; } else { In = hidden + (size_t)(l - 1) * n * W; In_stride = W; }
#pragma omp parallel for schedule(static)
		for (uint32_t r = 0; r < rows; ++r) {
			std::vector<double> acc(cols, 0.0);
			for (uint32_t i = 0; i < n; ++i) {
				const float g = h2f(dO[(size_t)i * dO_stride + r]);
				if (g == 0.0f) continue;
				const uint16_t* in = In + (size_t)i * In_stride;
				for (uint32_t c = 0; c < cols; ++c) acc[c] += (double)g * (double)h2f(in[c]);
			}
			for (uint32_t c = 0; c < cols; ++c) {
				const size_t idx = L.offset[l] + (size_t)r * cols + c;
				if (grads_f32) grads_f32[idx] = (float)acc[c];
				if (grads) grads[idx] = accumulate ? f2h(h2f(grads[idx]) + (float)acc[c]) : f2h((float)acc[c]);
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
void orc_loss(uint32_t type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale, const uint16_t* pred, const float* target,
              float* values, uint16_t* grads, const float* data_pdf) {
	const uint32_t n_elements = n * stride;
	const uint32_t n_total = n_elements / stride * dims;
#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n_elements; ++i) {
		const uint32_t intra = i % stride;
		const uint32_t inter = i / stride;
		if (intra >= dims) {
			if (values) values[i] = 0;
			if (grads) grads[i] = 0;
			continue;
		}
		const uint32_t target_idx = inter * dims + intra;
		const float prediction = h2f(pred[i]);
		const float pdf = data_pdf ? data_pdf[target_idx] : 1;
		const float tgt = target[target_idx];
		const float difference = prediction - tgt;
		float value, gradient;
		bool gradient_has_n_total = false;
		if (type == ORC_LOSS_RELATIVE_L2) { // relative_l2.h:60-73
			const float prediction_sq_plus_epsilon = prediction * prediction + 0.01f;
			value = difference * difference / prediction_sq_plus_epsilon / pdf / n_total;
			gradient = 2 * difference / prediction_sq_plus_epsilon / pdf;
		} else if (type == ORC_LOSS_RELATIVE_L2_LUMINANCE) { // relative_l2_luminance.h:60-84
			const uint16_t* px = pred + (i - intra);
			float r = h2f(px[0]), g = h2f(px[1]), b = h2f(px[2]);
			if (dims >= 6) {
				r += h2f(px[3]);
				g += h2f(px[4]);
				b += h2f(px[5]);
			}
			const float luminance = (0.299f * r + 0.587f * g + 0.114f * b);
			const float prediction_sq_plus_epsilon = luminance * luminance + 0.01f;
			value = difference * difference / prediction_sq_plus_epsilon / pdf / n_total;
			gradient = 2 * difference / prediction_sq_plus_epsilon / pdf;
		} else if (type == ORC_LOSS_L1) { // l1.h:60-69
			value = fabsf(difference) / pdf / n_total;
			gradient = copysignf(1.0f / pdf, difference);
		} else if (type == ORC_LOSS_RELATIVE_L1) { // relative_l1.h:60-71
			const float scale = 1.0f / (fabsf(prediction) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
		} else if (type == ORC_LOSS_MAPE) { // mape.h:60-72
			const float scale = 1.0f / (fabsf(tgt) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
		} else if (type == ORC_LOSS_SMAPE) { // smape.h:60-72
			const float scale = 1.0f / (0.5f * (fabsf(tgt) + fabsf(prediction)) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
		} else if (type == ORC_LOSS_CROSS_ENTROPY) { // cross_entropy.h:62-71
			const float factor = -tgt / pdf / n_total;
			value = factor * logf(prediction);
			gradient = factor / prediction;
			gradient_has_n_total = true;
		} else if (type == ORC_LOSS_VARIANCE) { // variance_is.h:62-71
			const float factor = tgt * tgt / pdf / n_total;
			value = factor / prediction - factor / pdf;
			gradient = -factor / (prediction * prediction);
			gradient_has_n_total = true;
		} else { // l2.h:60-72
			value = difference * difference / pdf / n_total;
			gradient = 2 * difference / pdf;
		}
		if (values) values[i] = value;
		if (grads) grads[i] = gradient_has_n_total ? f2h(loss_scale * gradient) : f2h(loss_scale * gradient / n_total);
	}
}

double orc_reduce_sum(size_t n, const float* values) {
	double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
	for (size_t i = 0; i < n; ++i) s += values[i];
	return s;
}

// ---------------------------------------------------------------------------------------------------------
void orc_adam_defaults(orc_adam_t* a) { // adam.h:312-326
	a->learning_rate = 1e-3f; a->beta1 = 0.9f; a->beta2 = 0.999f; a->epsilon = 1e-8f; a->l2_reg = 1e-8f;
	a->relative_decay = 0.0f; a->absolute_decay = 0.0f; a->clipping_magnitude = 0.0f; a->non_matrix_learning_rate_factor = 1.0f;
	a->adabound = 0; a->optimize_matrix_params = 1; a->optimize_non_matrix_params = 1;
}

void orc_adam_step(const orc_adam_t* a, size_t n, size_t n_matrix, float loss_scale, uint32_t current_step,
                   float* w_fp, uint16_t* w, const uint16_t* g, float* m1, float* m2, uint32_t* steps) {
	float lower_lr_bound = 0;
	float upper_lr_bound = std::numeric_limits<float>::max();
	if (a->adabound) { // adam.h:157-160
		lower_lr_bound = 0.1f - 0.1f / ((1 - a->beta2) * (float)current_step + 1);
		upper_lr_bound = 0.1f + 0.1f / ((1 - a->beta2) * (float)current_step);
	}
#pragma omp parallel for schedule(static)
	for (size_t i = 0; i < n; ++i) {
		float gradient = h2f(g[i]) / loss_scale;
		if (i >= n_matrix) {
			if (!a->optimize_non_matrix_params || gradient == 0) continue;
		} else {
			if (!a->optimize_matrix_params) continue;
		}
		const float weight_fp = w_fp[i];
		if (i < n_matrix) gradient += a->l2_reg * weight_fp;
		const float gradient_sq = gradient * gradient;
		const float first_moment = m1[i] = a->beta1 * m1[i] + (1 - a->beta1) * gradient;
		const float second_moment = m2[i] = a->beta2 * m2[i] + (1 - a->beta2) * gradient_sq;
		float learning_rate = a->learning_rate;
		if (i >= n_matrix) learning_rate *= a->non_matrix_learning_rate_factor;
		const uint32_t step = ++steps[i];
		learning_rate *= sqrtf(1 - powf(a->beta2, (float)step)) / (1 - powf(a->beta1, (float)step));
		const float effective_learning_rate = fminf(fmaxf(learning_rate / (sqrtf(second_moment) + a->epsilon), lower_lr_bound), upper_lr_bound);
		// weight_decay(rel * lr, abs * lr, w): common_device.h:870-873
		const float decayed_weight = (1 - a->relative_decay * learning_rate) * weight_fp - copysignf(a->absolute_decay * learning_rate, weight_fp);
		float new_weight = decayed_weight - effective_learning_rate * first_moment;
		if (a->clipping_magnitude != 0.0f) new_weight = fminf(fmaxf(new_weight, -a->clipping_magnitude), a->clipping_magnitude);
		w_fp[i] = new_weight;
		w[i] = f2h(new_weight);
	}
}

uint16_t orc_activation(uint32_t act, uint16_t pre) { return activation_fwd(act, pre); }
uint16_t orc_activation_backward(uint32_t act, uint16_t grad, uint16_t forward_out) { return activation_bwd(act, grad, forward_out); }

} // extern "C"
