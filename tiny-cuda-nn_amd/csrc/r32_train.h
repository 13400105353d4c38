// r32_train.h -- what k_train_r32.hip (host side, and the kernel with two waves per SIMD) and k_train_r32a.hip (the kernel with one wave per
// SIMD and the weights in registers) share: the kernels' argument block and the second file's launch.
#pragma once

#include "r32_device.h"

namespace tcnn_amd {

// Cache policy of the scatter-record stores (the aux operand of the raw buffer stores: 0 default, 2 nontemporal).  Round 2 measured that
// streamed records cost the scatter 10 us: its tasks then ran on whatever XCD and found the records in the writer's L2.  Since round 4 a
// record plane is gathered by ONE XCD, which pulls it into its own L2 whatever the writers' L2s hold -- and every dirty line left in
// those at the kernel's end is written back on the kernel's time.
#ifndef TCNN_R32_REC_AUX
#define TCNN_R32_REC_AUX 0
#endif
constexpr int R32_REC_AUX = TCNN_R32_REC_AUX;
#ifndef TCNN_R32_OUT_AUX
#define TCNN_R32_OUT_AUX 2
#endif
constexpr int R32_OUT_AUX = TCNN_R32_OUT_AUX; // the [n][16] output rows (2: nontemporal)

struct R32Args {
	const half_t* x;        // level planes half2 [16][n]
	const float* target;    // [n][dims]
	half_t* out;            // [n][16]
	half_t* dL_dout;        // compact [n][dims]
	float* L;               // compact [n][dims]
	u32x4* rec;             // scatter records [8][n]: {x, y, gradients of levels 2 p, 2 p + 1}; or (rec_x == nullptr) level planes half2 [16][n]
	const float* rec_x;     // [n][2]; nullptr: dL/d(encoded input) leaves as plain level planes
	float* slabs;           // [gridDim.x][n_params]
	const h8* image;        // R32 fragments
	uint32_t n, dims, n_params;
	uint32_t w_off[3];      // element offsets of W0, W1, Wout inside a slab
	float loss_scale;
	uint32_t stagger;        // TCNN_AMD_MLP_STAGGER: waves 4..7 start their first trip this many times 64 clocks late
	uint32_t prio_mode;      // TCNN_AMD_MLP_PRIO: 0 no priorities, 1 the two waves of a SIMD alternate their priority per trip, 2 the younger half at priority 1, 3 a trip's matrix regions above its loss
	unsigned long long* dbg; // TCNN_AMD_MLP_TIMING: per workgroup wave 0's clock at start / loop start / loop end / end, then every wave's loop end
};

constexpr int R32A_NW = 4; // waves per workgroup of k_mlp_train_r32a: one per SIMD

// loss_id 1: L2, 2: RelativeL2; grid = workgroups = slabs
void mlp_train_r32a_launch(hipStream_t stream, const R32Args& a, uint32_t grid, int loss_id);

} // namespace tcnn_amd
