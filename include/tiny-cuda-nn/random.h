// random.h -- the caller-side random number surface of the reference (include/tiny-cuda-nn/random.h:36-70 and its
// dependencies/pcg32 generator), over libtcnn_amd.so's C ABI.  Header-only, any C++14 compiler.
//
//   tcnn::pcg32 / tcnn::default_rng_t     PCG32 (O'Neill's pcg32_random_r, 64-bit LCG state + xorshift-rotate output): seed,
//                                         next_uint, next_float, advance -- the stream the reference's `default_rng_t{1337}`
//                                         produces (pinned in tests/golden/reference_kat.json: 0.147699356, 0.471029401, ...)
//   tcnn::generate_random_uniform<float>  random.h:58-70: fills a device array in the reference's order (thread i advances a copy
//                                         of the generator by 4 i and writes elements i + n_threads j) and advances `rng` by n
#pragma once

#include "../tcnn_amd.h"

#include <cstdint>
#include <cstring>
#include <stdexcept>

namespace tcnn {

struct pcg32 {
	uint64_t state_inc[2] = {0x853c49e6748fea9bULL, 0xda3e39cb94b95bdbULL}; // {state, inc}: the default-constructed generator
	static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;

	pcg32() = default;
	explicit pcg32(uint64_t initstate, uint64_t initseq = 1u) { seed(initstate, initseq); }

	void seed(uint64_t initstate, uint64_t initseq = 1u) {
		state_inc[0] = 0u;
		state_inc[1] = (initseq << 1u) | 1u;
		next_uint();
		state_inc[0] += initstate;
		next_uint();
	}
	uint32_t next_uint() {
		const uint64_t old = state_inc[0];
		state_inc[0] = old * MULT + state_inc[1];
		const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		const uint32_t rot = (uint32_t)(old >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	uint32_t next_uint(uint32_t bound) { // unbiased: rejects the lowest 2^32 mod bound values
		const uint32_t threshold = (~bound + 1u) % bound;
		for (;;) {
			const uint32_t r = next_uint();
			if (r >= threshold) return r % bound;
		}
	}
	float next_float() { // [0, 1): 23 random mantissa bits under exponent 0
		const uint32_t u = (next_uint() >> 9) | 0x3f800000u;
		float f;
		std::memcpy(&f, &u, 4);
		return f - 1.0f;
	}
	void advance(int64_t delta_) { // O(log delta) jump (Brown, "Random number generation with arbitrary strides")
		uint64_t cur_mult = MULT, cur_plus = state_inc[1], acc_mult = 1u, acc_plus = 0u;
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
		state_inc[0] = acc_mult * state_inc[0] + acc_plus;
	}
};

using default_rng_t = pcg32; // random.h:36

// random.h:58-70; T = float (the samples' use).  `stream` is a hipStream_t.
template <typename T, typename RNG>
void generate_random_uniform(void* stream, RNG& rng, size_t n_elements, T* out, const T lower = (T)0.0, const T upper = (T)1.0) {
	static_assert(sizeof(T) == 4 && (T)0.5 == 0.5f, "generate_random_uniform: float only");
	if (tcnn_generate_random_uniform(stream, rng.state_inc, n_elements, (float*)out, (float)lower, (float)upper) != TCNN_OK) throw std::runtime_error{tcnn_last_error()};
}
template <typename T, typename RNG>
void generate_random_uniform(RNG& rng, size_t n_elements, T* out, const T lower = (T)0.0, const T upper = (T)1.0) {
	generate_random_uniform<T, RNG>(nullptr, rng, n_elements, out, lower, upper);
}

} // namespace tcnn
