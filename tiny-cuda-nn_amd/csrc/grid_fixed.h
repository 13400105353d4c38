// grid_fixed.h -- exact 64-bit fixed-point form of fp16 values, shared by the gradient kernels of the grid encoding
// (k_grid_scatter.hip, k_grid_bin.hip).  LSB = 2^-24, the finest fp16 subnormal: every finite fp16 value converts
// exactly, 2^39 of them can be summed without overflow, and the sum is rounded to fp16 ONCE (round-to-nearest-even).
#pragma once

#include "grid_device.h"

namespace tcnn_amd {
namespace {

// h * 2^24 as an integer: exact for every finite fp16 value
__device__ inline long long half_to_fixed(half_t h) {
	const uint16_t b = __builtin_bit_cast(uint16_t, h);
	const uint32_t e = (b >> 10) & 31u, f = b & 1023u;
	const unsigned long long m = e ? ((unsigned long long)(1024u | f) << (e - 1)) : (unsigned long long)f;
	return (b & 0x8000u) ? -(long long)m : (long long)m;
}

// the same for the common case |h| < 64: h * 2^24 fits an int32 and the float detour is exact (3 instructions instead of ~12)
__device__ inline long long half_to_fixed_fast(half_t h) {
	const float f = (float)h;
	long long v = (long long)(int)(f * 16777216.0f);
	if (__builtin_expect(!(__builtin_fabsf(f) < 64.0f), 0)) v = half_to_fixed(h);
	return v;
}

// s * 2^-24 rounded to fp16, round-to-nearest-even, one rounding
__device__ inline half_t fixed_to_half(long long s) {
	const bool neg = s < 0;
	unsigned long long m = neg ? (unsigned long long)(-s) : (unsigned long long)s;
	if (m == 0) return (half_t)0.0f;
	const int p = 63 - __builtin_clzll(m);
	float v;
	if (p <= 10) {
		v = (float)(uint32_t)m * 5.9604644775390625e-08f; // 2^-24, exact
	} else {
		const int shift = p - 10;
		unsigned long long q = m >> shift;
		const unsigned long long rem = m & ((1ull << shift) - 1), half = 1ull << (shift - 1);
		if (rem > half || (rem == half && (q & 1ull))) ++q;
		v = ldexpf((float)(uint32_t)q, shift - 24); // <= 12 significant bits: exact; >= 65520 becomes inf in the cast below
	}
	const half_t r = (half_t)v;
	return neg ? -r : r;
}

// |s| < 2^24 (|value| < 1): s is exact as a float, the scaling is exact, and the hardware float -> half conversion is the one RNE rounding
__device__ inline half_t fixed_to_half_fast(long long s) {
	if (__builtin_expect((unsigned long long)(s + (1ll << 24)) < (1ull << 25), 1)) {
		int lo = (int)s;
		asm volatile("" : "+v"(lo)); // hidden from the optimiser, which otherwise widens (float)(int)s back into a 13-instruction 64-bit conversion
		return (half_t)((float)lo * 5.9604644775390625e-08f);
	}
	return fixed_to_half(s);
}

} // namespace
} // namespace tcnn_amd
