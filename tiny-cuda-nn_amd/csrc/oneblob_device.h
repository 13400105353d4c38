// oneblob_device.h -- the OneBlob encoding's arithmetic (oneblob.h:47-67, common_device.h:905-920), shared by the encoding's own
// kernels (k_misc.hip) and by the MLP kernels that evaluate the encoding inside their input load (k_mlp.hip).
#pragma once
#include "tcnn_common.h"

namespace tcnn_amd {

__device__ inline float quartic(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float tmp = fmaxf(1 - u * u, 0.0f);
	return ((float)15 / 16) * tmp * tmp;
}
__device__ inline float quartic_cdf_deriv(float x, float inv_radius) { return quartic(x, inv_radius) * inv_radius; }
__device__ inline float quartic_cdf(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float u2 = u * u;
	const float u4 = u2 * u2;
	return fmaxf(0.0f, fminf(1.0f, ((float)15 / 16) * u * (1 - ((float)2 / 3) * u2 + ((float)1 / 5) * u4) + 0.5f));
}

// C(l) of bin edge `bin` (modulo n_bins) for the input xv: cdf(l - x) + cdf(l - x - 1) + cdf(l - x + 1)
__device__ inline float oneblob_edge(const float xv, const uint32_t bin, const uint32_t log2_bins) {
	const uint32_t n_bins = 1u << log2_bins;
	const float nb = (float)n_bins;
	const float lb = scalbnf((float)(bin & (n_bins - 1)), -(int)log2_bins);
	return quartic_cdf(lb - xv, nb) + quartic_cdf(lb - xv - 1.0f, nb) + quartic_cdf(lb - xv + 1.0f, nb);
}
// bin `bin` (< n_bins) in the definition form: C(right edge) - C(left edge), the last bin's right edge being bin 0's left edge + 1
__device__ inline float oneblob_bin(const float xv, const uint32_t bin, const uint32_t log2_bins) {
	const float l = oneblob_edge(xv, bin, log2_bins);
	float r = oneblob_edge(xv, bin + 1, log2_bins);
	if (bin == (1u << log2_bins) - 1) r += 1;
	return r - l;
}
// For xv in [0, 1] only the five bins (oneblob_window_first(xv) + o) mod n_bins, o = 0..4, can differ from +0 (n_bins >= 8): the
// kernel's radius is one bin, and the wrap-around images x +- 1 fall on the same bins modulo n_bins.
__device__ inline bool oneblob_in_unit_interval(const float xv) { return xv >= 0.0f && xv <= 1.0f; }
__device__ inline uint32_t oneblob_window_first(const float xv, const uint32_t log2_bins) {
	const uint32_t n_bins = 1u << log2_bins;
	return (uint32_t)(int)floorf(xv * (float)n_bins) + n_bins - 2u;
}

} // namespace tcnn_amd
