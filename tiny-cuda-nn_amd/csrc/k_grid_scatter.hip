// k_grid_scatter.hip -- dL/dgrid for the hash / dense / tiled grid, MI355X form: "owner computes", exact, deterministic.
//
// Replaces (reference, /root/reference): include/tiny-cuda-nn/encodings/grid.h:215-320 (kernel_grid_backward) together
// with the memset of the gradient table (grid.h:858) for half-precision grids with F >= 2.
//
// Why not the reference's shape.  The reference scatters 2^D * L packed-fp16 atomics per sample into global memory
// (grid.h:252-255).  Measured on MI355X (profiles/round1_*): random float atomics execute at the memory side at ~21 G/s
// chip-wide whatever their scope or footprint, so the 16.8 M atomics of one 256k-sample step cost 0.8-2.2 ms -- up to
// 78 % of the training step.  LDS *float* atomics are no way out either: ds_pk_add_f16 / ds_add_f32 retire ~0.32 lanes
// per clock per CU (measured), while LDS *integer* atomics retire 3.7-4.6.
//
// What this file does instead.
//   * Every workgroup OWNS a contiguous chunk of one level's table, held in LDS as 64-bit fixed-point accumulators
//     (LSB = 2^-24, the finest fp16 subnormal: every fp16 product converts EXACTLY, and 2^39 such products fit).
//   * The forward pass recorded, per (level, chunk), one bit per sample: "this sample touches the chunk" (k_grid_fwd ->
//     grid_mask_to_bits).  A workgroup reads only its own bit plane, compacts the hits and recomputes indices / weights
//     for those samples alone (cheap integer + fp32 math, bit-identical to the forward pass).
//   * Each contribution is formed exactly as the reference forms it -- (half)weight * dL_dy in fp16 (grid.h:254) -- then
//     added with ds_add_u64.  Integer addition is associative: the result is the EXACT sum of the fp16 products, rounded
//     to fp16 once (round-to-nearest-even) -- deterministic, independent of scheduling, and at least as accurate as any
//     order of the reference's fp16 atomics.  The oracle reproduces it bit for bit (orc_grid_backward, exact mode).
//   * The owner writes its chunk with plain coalesced stores, which also replaces the memset of the whole table.
//     Coarse levels, where every sample hits the same few entries, are split over several workgroups by sample range;
//     their exact partial sums are merged with 64-bit integer atomics in a small scratch table and rounded by
//     k_grid_scatter_finalize.
#include "grid_fixed.h"
#include "adam_device.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace tcnn_amd {
namespace {

// One workgroup of 16 waves per CU owning 128 KB of accumulators.  The other shape the LDS allows -- two workgroups of 8 waves
// with 64 KB each, i.e. tables cut into 128 chunks per level (GRID_FILTER_MAX_CHUNKS = 128) -- was built and measured on C3a:
// k_grid_scatter 69 -> 89 us (twice the tasks, each still scanning a whole bit plane of n / 64 words and paying its own
// zeroing and flush for half the hits) and k_grid_fwd_planes 58 -> 71 us (twice the filter words to build and write).
// TCNN_SCATTER_ACC_KB / TCNN_SCATTER_THREADS / GRID_FILTER_MAX_CHUNKS (tcnn_common.h) are the compile-time knobs of that build.
#ifndef TCNN_SCATTER_ACC_KB
#define TCNN_SCATTER_ACC_KB 128
#endif
#ifndef TCNN_SCATTER_THREADS
#define TCNN_SCATTER_THREADS 1024
#endif
constexpr uint32_t SCATTER_ACC_BYTES = TCNN_SCATTER_ACC_KB * 1024; // accumulator chunk per workgroup
constexpr uint32_t SCATTER_THREADS = TCNN_SCATTER_THREADS;
constexpr uint32_t SCATTER_WAVES = SCATTER_THREADS / 64;
constexpr int SCATTER_SUB_BATCHES = 2;             // 64-sample batches whose gathers are in flight together per wave
constexpr uint32_t SCATTER_QUEUE_IDS = 64 * SCATTER_SUB_BATCHES + 64; // wave-private compaction queue (sample ids)
constexpr uint32_t SCATTER_LDS_BYTES = SCATTER_ACC_BYTES + SCATTER_WAVES * SCATTER_QUEUE_IDS * 4;
constexpr uint32_t SCATTER_MAX_CHUNKS = GRID_FILTER_MAX_CHUNKS; // bit planes per level; levels cut finer are binned (k_grid_bin.hip)

// REC: dL_dy holds 16-byte records {D coordinates, gradient halves} written by the fused MLP kernel
// (mlp_device.h store_dx_record; float4 [level][n], or [level / 2][n] where two levels fit one record): one gather per hit instead of two -- gathers cost ~2 clk per lane per CU whatever their width.
//
// ADAM: a sole owner also applies the optimizer's update to its chunk as it flushes (AdamInFlush, tcnn_common.h): the gradient is
// final there, so adam.h:48-119 runs on it at once -- same function, same inputs, same bits as k_adam run afterwards.  What this
// buys is one launch and the 2 B/param re-read of the gradient, not the overlap one might hope for; measured on C3a (DESIGN.md
// "Adam in the scatter"): a CU streams a chunk's 590 KB of optimizer state in 18 us whatever the other CUs do -- ~32 GB/s, its
// share of the HBM -- which is what the same bytes cost in k_adam, and while it does that its accumulators sit idle.
template <int D, int F, bool REC, bool ADAM>
__global__ void __launch_bounds__(SCATTER_THREADS) k_grid_scatter(
	const GridMeta* __restrict__ meta, const GridScatterTask* __restrict__ tasks, const uint32_t n, const MatView x,
	const half_t* __restrict__ dL_dy, const uint32_t dy_stride_sample, const uint32_t dy_stride_level, half_t* __restrict__ grad,
	const unsigned long long* __restrict__ chunk_bits, unsigned long long* __restrict__ scratch, const int accumulate_mode,
	unsigned long long* __restrict__ dbg_times, const AdamInFlush adam
) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	long long* acc = (long long*)smem; // [n_entries][F]
	typedef __attribute__((address_space(3))) unsigned long long lds_u64;
	lds_u64* acc_lds = (lds_u64*)smem;
	const GridScatterTask task = tasks[blockIdx.x];
	if (task.n_entries == 0) return;
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63, wave = tid >> 6;
	uint32_t* queue = (uint32_t*)(smem + SCATTER_ACC_BYTES) + wave * SCATTER_QUEUE_IDS; // wave-private
	if (dbg_times && tid == 0) dbg_times[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();

	const GridLevel lv = meta->levels[task.level];
	half_t* __restrict__ g = grad + ((size_t)lv.offset + task.entry_begin) * F;
	const uint32_t n_vals = task.n_entries * F;

	// initialise the owned chunk: zero, or the existing gradient for GradientMode::Accumulate when we are the only owner
	if (accumulate_mode && !task.flush_atomic) {
		for (uint32_t i = tid; i < n_vals; i += SCATTER_THREADS) acc[i] = half_to_fixed(g[i]);
	} else {
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		u4* a4 = (u4*)smem;
		const uint32_t n16 = n_vals / 2;
		for (uint32_t i = tid; i < n16; i += SCATTER_THREADS) a4[i] = u4{0, 0, 0, 0};
		if ((n_vals & 1u) && tid == 0) acc[n_vals - 1] = 0;
	}
	__syncthreads();
	if (dbg_times && tid == 0) dbg_times[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();

	const uint32_t interpolation = meta->interpolation;
	const uint32_t hash_type = meta->hash_type;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = meta->primes[d];
	const half_t* __restrict__ dy = dL_dy + (size_t)task.level * dy_stride_level;
	typedef typename VecOf<half_t, F>::type vecF;

	// issue the loads of one sample (coordinates + its dL/dy of this level)
	// records (mlp_device.h store_dx_record): one per level, except D = 2 with F = 2 where two levels share one
	constexpr bool PAIRED = REC && D == 2 && F == 2;
	const uint4* recs = (const uint4*)dL_dy + (size_t)(PAIRED ? task.level / 2 : task.level) * n;
	auto fetch = [&](const uint32_t i, float (&xin)[D], vecF& gv) {
		if constexpr (REC) {
			const uint4 r = recs[i];
			const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
			for (int d = 0; d < D; ++d) xin[d] = __builtin_bit_cast(float, w[d]);
			if constexpr (PAIRED) {
				gv = __builtin_bit_cast(vecF, (task.level & 1u) ? w[3] : w[2]);
			} else if constexpr (F == 2) {
				gv = __builtin_bit_cast(vecF, w[D]);
			} else {
				typedef uint32_t u2 __attribute__((ext_vector_type(2)));
				gv = __builtin_bit_cast(vecF, (u2{w[2], w[3]}));
			}
		} else {
			load_coords<D>(x, i, xin);
			gv = *(const vecF*)&dy[(size_t)i * dy_stride_sample];
		}
	};
	// full treatment of one sample: recompute its corners, add those that fall into the owned chunk (grid.h:215-320)
	auto accumulate = [&](const float (&xin)[D], const vecF& gv) {
		float pos[D], unused;
		uint32_t cell[D];
#pragma unroll
		for (int d = 0; d < D; ++d) cell[d] = pos_fract(xin[d], lv.scale, interpolation, &pos[d], &unused);

		auto add = [&](const uint32_t* local, float weight) {
			const uint32_t index = level_index<D>(lv, primes, hash_type, local) - task.entry_begin;
			if (index < task.n_entries) {
				asm volatile("" : "+v"(weight)); // keep the fp32 rounding of the weight product (see k_grid_fwd)
				const half_t w = (half_t)weight;
#pragma unroll
				for (int f = 0; f < F; ++f) {
					const half_t c = w * gv[f]; // (GRAD_T)weight * grad in fp16, grid.h:254
					// explicit LDS address space: guarantees ds_add_u64 (a generic pointer may degrade to flat_atomic_add_x2)
					__hip_atomic_fetch_add(acc_lds + index * F + f, (unsigned long long)half_to_fixed_fast(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
		};

		if (interpolation == (uint32_t)InterpolationType::Nearest) {
			add(cell, 1.0f);
			return;
		}
		// All 2^D corners are located first; then the wave adds "the next corner of mine that falls into the chunk" until no
		// lane has one left.  A sample that reaches this point typically owns 2^(D-1) such corners (one row of the cell), so
		// this takes half as many -- fully populated -- LDS atomic instructions as walking the corners in lockstep.
		constexpr int C = 1 << D;
		uint32_t cidx[C];
		float cw[C];
		uint32_t mine = 0;
#pragma unroll
		for (int idx = 0; idx < C; ++idx) {
			float weight = 1;
			uint32_t local[D];
#pragma unroll
			for (int d = 0; d < D; ++d) {
				if ((idx & (1 << d)) == 0) {
					weight *= 1 - pos[d];
					local[d] = cell[d];
				} else {
					weight *= pos[d];
					local[d] = cell[d] + 1;
				}
			}
			asm volatile("" : "+v"(weight)); // keep the fp32 rounding of the weight product (see k_grid_fwd)
			cw[idx] = weight;
			cidx[idx] = level_index<D>(lv, primes, hash_type, local) - task.entry_begin;
			if (cidx[idx] < task.n_entries) mine |= 1u << idx;
		}
		while (__builtin_amdgcn_ballot_w64(mine != 0) != 0) {
			if (mine != 0) {
				const uint32_t k = (uint32_t)__builtin_ctz(mine);
				mine &= mine - 1;
				uint32_t index = cidx[0];
				float weight = cw[0];
#pragma unroll
				for (int idx = 1; idx < C; ++idx) {
					index = k == (uint32_t)idx ? cidx[idx] : index;
					weight = k == (uint32_t)idx ? cw[idx] : weight;
				}
				const half_t w = (half_t)weight;
#pragma unroll
				for (int f = 0; f < F; ++f) {
					const half_t c = w * gv[f]; // (GRAD_T)weight * grad in fp16, grid.h:254
					__hip_atomic_fetch_add(acc_lds + index * F + f, (unsigned long long)half_to_fixed_fast(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
		}
	};

	// the same for a task whose chunk is the whole level (coarse levels): every corner is ours, no test, no compaction
	auto accumulate_whole = [&](const float (&xin)[D], const vecF& gv) {
		float pos[D], unused;
		uint32_t cell[D];
#pragma unroll
		for (int d = 0; d < D; ++d) cell[d] = pos_fract(xin[d], lv.scale, interpolation, &pos[d], &unused);
		constexpr int C = 1 << D;
		const bool nearest = interpolation == (uint32_t)InterpolationType::Nearest; // grid.h:232-246: the cell's own entry, weight 1
#pragma unroll
		for (int idx = 0; idx < C; ++idx) {
			if (idx > 0 && nearest) break;
			float weight = 1;
			uint32_t local[D];
#pragma unroll
			for (int d = 0; d < D; ++d) {
				if ((idx & (1 << d)) == 0) {
					weight *= nearest ? 1.0f : 1 - pos[d];
					local[d] = cell[d];
				} else {
					weight *= pos[d];
					local[d] = cell[d] + 1;
				}
			}
			asm volatile("" : "+v"(weight));
			const uint32_t index = level_index<D>(lv, primes, hash_type, local);
			const half_t w = (half_t)weight;
#pragma unroll
			for (int f = 0; f < F; ++f) {
				const half_t c = w * gv[f];
				__hip_atomic_fetch_add(acc_lds + index * F + f, (unsigned long long)half_to_fixed_fast(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
	};

	// Every wave owns a contiguous block of the task's samples and walks it 64 samples at a time.  Loads of batch k+1 are
	// issued before batch k is accumulated (two register sets, A and B), so a wave always has a batch in flight.
	const uint32_t n_task = task.sample_end - task.sample_begin;
	const uint32_t per_wave = ((n_task + SCATTER_WAVES - 1) / SCATTER_WAVES + 63) / 64 * 64;
	const uint32_t w_begin = min(task.sample_begin + wave * per_wave, task.sample_end);
	const uint32_t w_end = min(task.sample_end, w_begin + per_wave);
	const uint32_t n_blocks = (w_end - w_begin + 63) / 64; // wave-uniform

	// Software pipeline, two batches in flight: the loads of batch k are issued, then batch k-2 is accumulated.  The register
	// sets rotate statically (plain moves) so that the compiler can count the younger loads and wait with vmcnt(2), not vmcnt(0).
	constexpr int SB = SCATTER_SUB_BATCHES;
	float ax[SB][D], nx[SB][D];
	vecF ag[SB], ng[SB];
	bool a_valid[SB]; // per lane
#pragma unroll
	for (int s = 0; s < SB; ++s) a_valid[s] = false;
	unsigned long long t_fetch = 0, t_acc = 0, n_sub = 0; // development aid (TCNN_AMD_SCATTER_TIMING): shader clocks of wave 0
	auto submit = [&](const uint32_t (&ids)[SB], const bool (&valid)[SB]) { // ids are in range for every lane (clamped by the caller)
		const unsigned long long c0 = dbg_times ? __builtin_readcyclecounter() : 0;
#pragma unroll
		for (int s = 0; s < SB; ++s) fetch(ids[s], nx[s], ng[s]);
		const unsigned long long c1 = dbg_times ? __builtin_readcyclecounter() : 0;
#pragma unroll
		for (int s = 0; s < SB; ++s) if (a_valid[s]) accumulate(ax[s], ag[s]);
		if (dbg_times) {
			const unsigned long long c2 = __builtin_readcyclecounter();
			t_fetch += c1 - c0;
			t_acc += c2 - c1;
			++n_sub;
		}
#pragma unroll
		for (int s = 0; s < SB; ++s) {
#pragma unroll
			for (int d = 0; d < D; ++d) ax[s][d] = nx[s][d];
			ag[s] = ng[s];
			a_valid[s] = valid[s];
		}
	};

	const unsigned long long t_loop0 = dbg_times ? __builtin_readcyclecounter() : 0;
	if (n_blocks > 0) {
		if (lv.scatter_n_chunks == 1) {
			// the chunk is the whole level: every sample, every corner, in lockstep; one batch in flight while the previous one is added
			float px[D], qx[D];
			vecF pg, qg;
			bool p_valid = false;
			for (uint32_t blk = 0; blk < n_blocks; ++blk) {
				const uint32_t i = w_begin + blk * 64 + lane;
				fetch(min(i, w_end - 1), qx, qg);
				if (p_valid) accumulate_whole(px, pg);
#pragma unroll
				for (int d = 0; d < D; ++d) px[d] = qx[d];
				pg = qg;
				p_valid = i < w_end;
			}
			if (p_valid) accumulate_whole(px, pg);
		} else if (chunk_bits == nullptr) {
			// no filter available: every sample gets the full treatment
			for (uint32_t blk = 0; blk < n_blocks; blk += SB) {
				uint32_t ids[SB];
				bool valid[SB];
#pragma unroll
				for (int s = 0; s < SB; ++s) {
					const uint32_t i = w_begin + (blk + s) * 64 + lane;
					ids[s] = min(i, w_end - 1);
					valid[s] = i < w_end;
				}
				submit(ids, valid);
			}
		} else {
			// One bit per sample says whether it touches this chunk.  A wave reads 64 words at once (one word = the ballot of
			// one 64-sample block), compacts the hits into its LDS queue and runs the expensive part on full batches of 64.
			const unsigned long long* __restrict__ words =
				chunk_bits + ((size_t)task.level * SCATTER_MAX_CHUNKS + scatter_chunk(lv, task.entry_begin)) * (n / 64) + w_begin / 64;
			uint32_t queued = 0; // wave-uniform
			unsigned long long w_next = lane < n_blocks ? words[lane] : 0ull;
			for (uint32_t blk0 = 0; blk0 < n_blocks; blk0 += 64) {
				// one word (= 64 samples) per lane; every round each lane with bits left peels off its lowest one
				unsigned long long w = w_next;
				w_next = blk0 + 64 + lane < n_blocks ? words[blk0 + 64 + lane] : 0ull; // in flight while this block is processed
				const uint32_t base_id = w_begin + (blk0 + lane) * 64;
				while (true) {
					const bool has = w != 0;
					const unsigned long long ballot = __ballot(has);
					if (ballot == 0) break;
					const uint32_t blo = (uint32_t)ballot, bhi = (uint32_t)(ballot >> 32);
					if (has) {
						queue[queued + __builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0))] = base_id + (uint32_t)__builtin_ctzll(w);
						w &= w - 1;
					}
					queued += __builtin_popcount(blo) + __builtin_popcount(bhi);
					if (queued >= 64 * SB) {
						// the queue is wave-private and LDS serves one wave's instructions in order: compiler barriers suffice
						// (a memory fence here would also drain the gathers in flight)
						__atomic_signal_fence(__ATOMIC_SEQ_CST);
						__builtin_amdgcn_wave_barrier();
						uint32_t ids[SB];
						bool valid[SB];
#pragma unroll
						for (int s = 0; s < SB; ++s) {
							ids[s] = queue[s * 64 + lane];
							valid[s] = true;
						}
						const uint32_t rest = queue[64 * SB + lane];
						__builtin_amdgcn_wave_barrier();
						queued -= 64 * SB;
						if (lane < queued) queue[lane] = rest;
						__atomic_signal_fence(__ATOMIC_SEQ_CST);
						submit(ids, valid);
					}
				}
			}
			__atomic_signal_fence(__ATOMIC_SEQ_CST);
			__builtin_amdgcn_wave_barrier();
			if (queued > 0) {
				uint32_t ids[SB];
				bool valid[SB];
#pragma unroll
				for (int s = 0; s < SB; ++s) {
					valid[s] = s * 64 + lane < queued;
					ids[s] = valid[s] ? queue[s * 64 + lane] : w_begin;
				}
				submit(ids, valid);
			}
		}
#pragma unroll
		for (int s = 0; s < SB; ++s) if (a_valid[s]) accumulate(ax[s], ag[s]);
	}
	if (dbg_times && tid == 0) {
		dbg_times[blockIdx.x * 8 + 4] = t_fetch;
		dbg_times[blockIdx.x * 8 + 5] = t_acc;
		dbg_times[blockIdx.x * 8 + 6] = __builtin_readcyclecounter() - t_loop0;
		dbg_times[blockIdx.x * 8 + 7] = n_sub;
	}
	__syncthreads();
	if (dbg_times && tid == 0) dbg_times[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();

	if (task.flush_atomic) {
		// several workgroups share this chunk: merge the exact partial sums with 64-bit integer atomics; rounded later
		unsigned long long* sc = scratch + (size_t)task.scratch_begin;
		for (uint32_t i = tid; i < n_vals; i += SCATTER_THREADS) {
			const long long v = acc[i];
			if (v != 0) atomicAdd(sc + i, (unsigned long long)v);
		}
	} else if (ADAM && adam.w_fp) {
		// sole owner, optimizer step included (the host checked that the chunk is a whole number of aligned quads of parameters)
		typedef _Float16 h4 __attribute__((ext_vector_type(4)));
		const size_t p0 = ((size_t)lv.offset + task.entry_begin) * F;
		float* __restrict__ wf_p = adam.w_fp + p0;
		float* __restrict__ m1_p = adam.m1 + p0;
		float* __restrict__ m2_p = adam.m2 + p0;
		uint32_t* __restrict__ st_p = (uint32_t*)adam.steps + p0; // uint16 counts (adam.steps16): addressed through st16_p
		uint16_t* __restrict__ st16_p = (uint16_t*)adam.steps + p0;
		half_t* __restrict__ wh_p = (half_t*)adam.w_half + p0;
		// Parameters that missed an update carry their own step count and look their debiasing factor up (adam_device.h): one
		// gather per parameter, and gathers are what a CU has least of (~1.7 clocks per lane).  The most recent steps of the table
		// -- all that parameters touched in the last few thousand steps ask for -- are kept in LDS, in the space of the compaction
		// queues, which are dead by now.
		constexpr uint32_t WINDOW = SCATTER_WAVES * SCATTER_QUEUE_IDS;
		float* window = (float*)(smem + SCATTER_ACC_BYTES);
		const uint32_t common = adam.args.common_step;
		const uint32_t window_base = common + 1 > WINDOW ? common + 1 - WINDOW : 0; // window[i] = table[window_base + i], up to table[common]
		for (uint32_t i = tid; i < WINDOW; i += SCATTER_THREADS) if (window_base + i <= common) window[i] = adam.debias_table[window_base + i];
		__syncthreads();
		const float debias = window[common - window_base];
		const auto debias_of = [&](const uint32_t t) { return t >= window_base ? window[t - window_base] : adam.debias_table[t]; };
		constexpr int Q = 4; // quads per thread in flight
		const uint32_t n_quads = n_vals / 4;
		for (uint32_t q0 = tid; q0 < n_quads; q0 += Q * SCATTER_THREADS) {
			h4 gq[Q], old[Q];
			bool live[Q], has_old[Q];
			float4 wf[Q], a1[Q], a2[Q];
			uint4 st[Q];
#pragma unroll
			for (int k = 0; k < Q; ++k) {
				const uint32_t q = q0 + k * SCATTER_THREADS;
				live[k] = q < n_quads;
				has_old[k] = false;
				if (live[k]) {
					gq[k] = h4{fixed_to_half_fast(acc[4 * q]), fixed_to_half_fast(acc[4 * q + 1]), fixed_to_half_fast(acc[4 * q + 2]), fixed_to_half_fast(acc[4 * q + 3])};
					*(h4*)(g + 4 * (size_t)q) = gq[k];
					const bool z0 = gq[k][0] == (half_t)0.0f, z1 = gq[k][1] == (half_t)0.0f, z2 = gq[k][2] == (half_t)0.0f, z3 = gq[k][3] == (half_t)0.0f;
					live[k] = !(z0 && z1 && z2 && z3); // adam.h:76-79: a grid parameter with a zero gradient is left alone -- nothing else of it is read
					has_old[k] = live[k] && (z0 || z1 || z2 || z3);
				}
				if (live[k]) {
					wf[k] = *(const float4*)(wf_p + 4 * (size_t)q);
					a1[k] = *(const float4*)(m1_p + 4 * (size_t)q);
					a2[k] = *(const float4*)(m2_p + 4 * (size_t)q);
					if (adam.steps16) {
						typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
						const u16x4 sv = *(const u16x4*)(st16_p + 4 * (size_t)q);
						st[k] = uint4{sv[0], sv[1], sv[2], sv[3]};
					} else {
						st[k] = *(const uint4*)(st_p + 4 * (size_t)q);
					}
					if (has_old[k]) old[k] = *(const h4*)(wh_p + 4 * (size_t)q);
				}
			}
#pragma unroll
			for (int k = 0; k < Q; ++k) {
				if (!live[k]) continue;
				const size_t i4 = 4 * (size_t)(q0 + k * SCATTER_THREADS);
				half_t wh[4];
				bool up[4];
				adam_one(adam.args, debias_of, debias, false, gq[k][0], wf[k].x, wh[0], a1[k].x, a2[k].x, st[k].x, up[0]);
				adam_one(adam.args, debias_of, debias, false, gq[k][1], wf[k].y, wh[1], a1[k].y, a2[k].y, st[k].y, up[1]);
				adam_one(adam.args, debias_of, debias, false, gq[k][2], wf[k].z, wh[2], a1[k].z, a2[k].z, st[k].z, up[2]);
				adam_one(adam.args, debias_of, debias, false, gq[k][3], wf[k].w, wh[3], a1[k].w, a2[k].w, st[k].w, up[3]);
				*(float4*)(wf_p + i4) = wf[k];
				*(float4*)(m1_p + i4) = a1[k];
				*(float4*)(m2_p + i4) = a2[k];
				if (adam.steps16) {
					typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
					*(u16x4*)(st16_p + i4) = u16x4{(uint16_t)st[k].x, (uint16_t)st[k].y, (uint16_t)st[k].z, (uint16_t)st[k].w};
				} else {
					*(uint4*)(st_p + i4) = st[k];
				}
				// parameters that were not updated keep their half value, whatever it is: quads with a zero gradient somewhere
				// brought their old halves along, so that the quad is stored whole
				if (up[0] && up[1] && up[2] && up[3]) {
					*(h4*)(wh_p + i4) = h4{wh[0], wh[1], wh[2], wh[3]};
				} else if (has_old[k]) {
					*(h4*)(wh_p + i4) = h4{up[0] ? wh[0] : old[k][0], up[1] ? wh[1] : old[k][1], up[2] ? wh[2] : old[k][2], up[3] ? wh[3] : old[k][3]};
				} else { // a non-zero half gradient that became zero when the loss scale was divided out
#pragma unroll
					for (int e = 0; e < 4; ++e) if (up[e]) wh_p[i4 + e] = wh[e];
				}
			}
		}
	} else {
		// sole owner: round once and store (two values per 4-byte store; n_vals is even because F >= 2)
		typedef _Float16 h2 __attribute__((ext_vector_type(2)));
		for (uint32_t i = tid; i < n_vals / 2; i += SCATTER_THREADS) {
			((h2*)g)[i] = h2{fixed_to_half_fast(acc[2 * i]), fixed_to_half_fast(acc[2 * i + 1])};
		}
	}
	if (dbg_times) {
		__syncthreads();
		if (tid == 0) dbg_times[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
	}
}

// shared (split) chunks: scratch holds the exact integer sums; round, store, and leave the scratch zeroed for the next step.
// Blocks [0, n_reduce_blocks) carry the MLP's slab reduction (mlp_side_jobs.h) instead -- the step's two small reductions in one launch.
constexpr uint32_t FINALIZE_THREADS = SLAB_REDUCE_ELEMS * SLAB_REDUCE_GROUPS; // 1024: what mlp_reduce_block wants
constexpr uint32_t FINALIZE_BLOCKS_PER_RANGE = 64;
__global__ void __launch_bounds__(FINALIZE_THREADS) k_grid_scatter_finalize(const GridScatterRange* __restrict__ ranges, unsigned long long* __restrict__ scratch, half_t* __restrict__ grad, const int accumulate_mode,
                                                                           const uint32_t n_reduce_blocks, const MlpReduceJob job) {
	__shared__ float part[SLAB_REDUCE_GROUPS * SLAB_REDUCE_ELEMS];
	if (blockIdx.x < n_reduce_blocks) {
		mlp_reduce_block(part, blockIdx.x, threadIdx.x, job.n_elems, job.n_elems, job.n_elems, job.n_slabs, job.slabs, (_Float16*)job.grad, job.accumulate);
		return;
	}
	const uint32_t b = blockIdx.x - n_reduce_blocks;
	const GridScatterRange r = ranges[b / FINALIZE_BLOCKS_PER_RANGE];
	for (uint32_t i = (b % FINALIZE_BLOCKS_PER_RANGE) * FINALIZE_THREADS + threadIdx.x; i < r.n_elems; i += FINALIZE_BLOCKS_PER_RANGE * FINALIZE_THREADS) {
		long long s = (long long)scratch[r.scratch_begin + i];
		scratch[r.scratch_begin + i] = 0;
		if (accumulate_mode) s += half_to_fixed(grad[r.grad_begin + i]);
		grad[r.grad_begin + i] = fixed_to_half(s);
	}
}

// [n_levels][n][2] uint64 masks (written by k_grid_fwd: bit c of the 128-bit mask = the sample touches chunk c) -> bit planes
// [n_levels][128][n / 64] uint64: word w of plane (l, c) is the ballot "sample 64 w + lane touches chunk c of level l".
// One wave per 64 samples of one level.
__global__ void __launch_bounds__(256) k_grid_mask_to_bits(const GridMeta* __restrict__ meta, const uint32_t n, const unsigned long long* __restrict__ mask, unsigned long long* __restrict__ bits) {
	const uint32_t level = blockIdx.y;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return; // n is a multiple of 64: whole waves leave together
	const uint32_t n_chunks = meta->levels[level].scatter_n_chunks;
	if (n_chunks <= 1 || n_chunks > SCATTER_MAX_CHUNKS) return; // one chunk: no filter; more than the filter describes: binned level, no filter either
	const uint32_t lane = threadIdx.x & 63;
#pragma unroll
	for (uint32_t half = 0; half < SCATTER_MAX_CHUNKS / 64; ++half) {
		if (64 * half >= n_chunks) break;
		const unsigned long long m = mask[((size_t)level * n + i) * (SCATTER_MAX_CHUNKS / 64) + half];
		unsigned long long mine = 0;
		const uint32_t here = min(n_chunks - 64 * half, 64u);
		for (uint32_t c = 0; c < here; ++c) {
			const unsigned long long b = __ballot((m >> c) & 1ull);
			if (lane == c) mine = b;
		}
		if (lane < here) bits[((size_t)level * SCATTER_MAX_CHUNKS + 64 * half + lane) * (n / 64) + i / 64] = mine;
	}
}

// set by grid_backward_lds for the duration of one launch: device buffer uint64[n_tasks][8] that receives the per-task
// timestamps (slots 0..3: start / after zeroing / after accumulation / end, 100 MHz clock) -- input of the plan tuner
thread_local unsigned long long* g_task_times = nullptr;

template <int D, int F, bool REC = false, bool ADAM = false>
void launch_scatter(hipStream_t s, const GridMeta* dm, const GridScatterTask* tasks, uint32_t n_tasks, uint32_t n, MatView x, const void* dy, uint32_t dss, uint32_t dsl,
                    void* grad, const unsigned long long* chunk_bits, unsigned long long* scratch, bool accumulate, const AdamInFlush* adam = nullptr) {
	if constexpr (REC && !ADAM) {
		if (adam) return launch_scatter<D, F, REC, true>(s, dm, tasks, n_tasks, n, x, dy, dss, dsl, grad, chunk_bits, scratch, accumulate, adam);
	}
	CHECK_THROW(ADAM || !adam);
	static bool configured = false;
	if (!configured) { // more than 64 KiB of dynamic LDS has to be opted into once per kernel
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)k_grid_scatter<D, F, REC, ADAM>, hipFuncAttributeMaxDynamicSharedMemorySize, SCATTER_LDS_BYTES));
		configured = true;
	}
	// development aid: TCNN_AMD_SCATTER_TIMING=1 prints per-task phase times (100 MHz constant clock) for the 3rd launch
	static const bool timing = getenv("TCNN_AMD_SCATTER_TIMING") != nullptr;
	static int timing_left = 3;
	unsigned long long* dbg = nullptr;
	if (timing && timing_left > 0 && !g_task_times) HIP_CHECK_THROW(hipMalloc(&dbg, (size_t)n_tasks * 8 * 8));
	hipLaunchKernelGGL((k_grid_scatter<D, F, REC, ADAM>), dim3(n_tasks), dim3(SCATTER_THREADS), SCATTER_LDS_BYTES, s, dm, tasks, n, x, (const half_t*)dy, dss, dsl, (half_t*)grad, chunk_bits,
	                   scratch, accumulate ? 1 : 0, g_task_times ? g_task_times : dbg, adam ? *adam : AdamInFlush{});
	HIP_CHECK_THROW(hipGetLastError());
	if (dbg) {
		std::vector<unsigned long long> h((size_t)n_tasks * 8);
		std::vector<GridScatterTask> ht(n_tasks);
		HIP_CHECK_THROW(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
		HIP_CHECK_THROW(hipMemcpy(ht.data(), tasks, n_tasks * sizeof(GridScatterTask), hipMemcpyDeviceToHost));
		unsigned long long t0 = ~0ull;
		for (uint32_t i = 0; i < n_tasks; ++i) if (ht[i].n_entries) t0 = std::min(t0, h[i * 8]);
		if (--timing_left == 0) {
			for (uint32_t i = 0; i < n_tasks; ++i) {
				if (!ht[i].n_entries) continue;
				fprintf(stderr, "task %3u level %2u entries %6u samples %6u atomic %u: start %7.1f zero %6.1f accumulate %6.1f flush %6.1f us | wave0 clocks: fetch %llu acc %llu loop %llu submits %llu\n", i, ht[i].level, ht[i].n_entries,
				        ht[i].sample_end - ht[i].sample_begin, ht[i].flush_atomic, (h[i * 8] - t0) * 0.01, (h[i * 8 + 1] - h[i * 8]) * 0.01, (h[i * 8 + 2] - h[i * 8 + 1]) * 0.01,
				        (h[i * 8 + 3] - h[i * 8 + 2]) * 0.01, h[i * 8 + 4], h[i * 8 + 5], h[i * 8 + 6], h[i * 8 + 7]);
			}
		}
		(void)hipFree(dbg);
	}
}

template <int D>
void dispatch_scatter(hipStream_t s, uint32_t F, const GridMeta* dm, const GridScatterTask* tasks, uint32_t n_tasks, uint32_t n, MatView x,
                      const void* dy, uint32_t dss, uint32_t dsl, void* grad, const unsigned long long* chunk_bits, unsigned long long* scratch, bool accumulate, bool records,
                      const AdamInFlush* adam) {
	if (records) {
		if constexpr (D == 2) {
			if (F == 2) return launch_scatter<D, 2, true>(s, dm, tasks, n_tasks, n, x, dy, dss, dsl, grad, chunk_bits, scratch, accumulate, adam);
			if (F == 4) return launch_scatter<D, 4, true>(s, dm, tasks, n_tasks, n, x, dy, dss, dsl, grad, chunk_bits, scratch, accumulate, adam);
		} else if constexpr (D == 3) {
			if (F == 2) return launch_scatter<D, 2, true>(s, dm, tasks, n_tasks, n, x, dy, dss, dsl, grad, chunk_bits, scratch, accumulate, adam);
		}
		throw std::runtime_error{"grid_backward_lds: scatter records need 4 D + 2 F <= 16"};
	}
	CHECK_THROW(!adam); // only the record forms carry the optimizer step (grid_scatter_adam_ranges)
	switch (F) {
		case 2: return launch_scatter<D, 2>(s, dm, tasks, n_tasks, n, x, dy, dss, dsl, grad, chunk_bits, scratch, accumulate);
		case 4: return launch_scatter<D, 4>(s, dm, tasks, n_tasks, n, x, dy, dss, dsl, grad, chunk_bits, scratch, accumulate);
		case 8: return launch_scatter<D, 8>(s, dm, tasks, n_tasks, n, x, dy, dss, dsl, grad, chunk_bits, scratch, accumulate);
		default: throw std::runtime_error{"grid_backward_lds: needs n_features_per_level in {2, 4, 8}"};
	}
}

} // namespace

uint32_t grid_scatter_max_chunks() { return SCATTER_MAX_CHUNKS; }

uint32_t grid_scatter_record_planes(const GridMeta& meta) {
	return (meta.n_pos_dims == 2 && meta.n_features_per_level == 2) ? (meta.n_levels + 1) / 2 : meta.n_levels;
}

bool grid_scatter_records_supported(const GridMeta& meta) {
	const uint32_t D = meta.n_pos_dims, F = meta.n_features_per_level;
	return (D == 2 && (F == 2 || F == 4)) || (D == 3 && F == 2);
}

// Hit lists and the finer cut that goes with them pay where the grid has levels of MANY chunks: there the bit-plane form scans 32 KB per
// task to find the 3 % of samples that hit, and the lists' price in the forward kernel (~0.5 us per level) is small against its gathers.
// A grid whose levels all fit a few chunks (config_hash.json: T = 2^15, at most 4 chunks per level) is served better by the bit planes'
// ballot form -- its tasks stream the samples in order and test them, no gathers at all (measured on it: lists 0.149 ms per step, bit planes 0.135).
bool grid_scatter_prefers_lists(const GridMeta& meta) {
	if (!grid_scatter_records_supported(meta) || meta.hash_type == (uint32_t)HashType::Rng || meta.n_pos_dims > 3) return false;
	const uint32_t capacity = SCATTER_ACC_BYTES / (meta.n_features_per_level * 8);
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		const uint32_t k = div_round_up(meta.levels[l].size, capacity);
		if (k > 8 && k <= SCATTER_MAX_CHUNKS) return true;
	}
	return false;
}

void grid_scatter_setup_levels(GridMeta& meta) {
	const uint32_t capacity = SCATTER_ACC_BYTES / (meta.n_features_per_level * 8); // entries one workgroup can own (64-bit accumulators)
	const bool prefers_lists = grid_scatter_prefers_lists(meta);
	auto cut = [](GridLevel& lv, uint32_t entries_per_chunk) {
		lv.scatter_n_chunks = div_round_up(lv.size, entries_per_chunk);
		lv.scatter_per_chunk = next_multiple(div_round_up(lv.size, lv.scatter_n_chunks), 8u);
		lv.scatter_shift = 0xffffffffu;
		if ((lv.scatter_per_chunk & (lv.scatter_per_chunk - 1)) == 0) {
			lv.scatter_shift = 0;
			while ((1u << lv.scatter_shift) < lv.scatter_per_chunk) ++lv.scatter_shift;
		}
	};
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		GridLevel& lv = meta.levels[l];
		cut(lv, capacity);
		// More chunks than the sample filter describes: the level's gradients are binned instead (k_grid_bin.hip).  Where a visit
		// of the filtered form costs several gathers (no 16-byte records: 3-D with F = 4) binning already wins from 9 chunks on
		// (measured on C5: the 64-chunk level 2.4x faster; on C3a, with records, the filtered form is 1.5x faster at 64 chunks).
		// TCNN_AMD_BIN_MIN_CHUNKS=k moves the threshold (levels with more than k chunks are binned), for A/B runs.
		uint32_t bin_above = grid_scatter_records_supported(meta) ? SCATTER_MAX_CHUNKS : SCATTER_MAX_CHUNKS / 8u;
		if (const char* e = getenv("TCNN_AMD_BIN_MIN_CHUNKS")) bin_above = std::min<uint32_t>((uint32_t)std::max(atoi(e), 1), SCATTER_MAX_CHUNKS);
		const uint32_t bin_capacity = grid_bin_acc_bytes() / (meta.n_features_per_level * 8); // the binned kernels have their own chunk size
		lv.scatter_binned = (lv.scatter_n_chunks > bin_above && div_round_up(lv.size, bin_capacity) <= grid_bin_max_chunks() && grid_bin_supported(meta)) ? 1u : 0u;
		if (lv.scatter_binned) cut(lv, bin_capacity);
	}
	bool any_binned = false;
	for (uint32_t l = 0; l < meta.n_levels; ++l) any_binned |= meta.levels[l].scatter_binned != 0;
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		GridLevel& lv = meta.levels[l];
		// Round 4: a level that fits fewer chunks than the filter can describe is cut finer anyway -- into up to 64 chunks of at least 256
		// entries -- where the scatter gathers one record per hit (the forms with records).  Every chunk then has ONE owner whatever the
		// level's size: a level of 16 384 entries used to be 2 chunks shared by ~35 workgroups each (split over the samples), every one
		// of them flushing 16 384 64-bit global atomics into the scratch table (~10 us per task, measured) for the finalize pass to round;
		// as 64 chunks of 256 entries it is 64 tasks like those of the hashed levels, no scratch, no atomics.  Levels that fit ONE chunk
		// (8192 entries at two features per entry) stay one chunk: their tasks stream the samples in order (no gathers) and their flush is small.
		// (Round 5: where the hit lists feed the scatter the limit is the chunk's capacity, not 8192 entries -- with four features a level of
		// 4097 .. 8191 entries was TWO chunks with one owner each, 2^18 elements per task where the others have 8192: 0.79 ms for the launch.
		// A grid with binned levels never takes the lists (model.h hit_lists_usable), and its bit-plane tasks, split over the samples, serve
		// such a level better as two chunks than as sixteen: 87 against 110 us.  profiles/r05_shape_sweep.txt.)
		// TCNN_AMD_SCATTER_FINE_CUT=0: as before.
		static const bool fine_cut = [] { const char* e = getenv("TCNN_AMD_SCATTER_FINE_CUT"); return !(e && e[0] == '0'); }();
		const uint32_t cut_from = any_binned ? std::max(capacity, 8192u) : capacity;
		if (fine_cut && prefers_lists && !lv.scatter_binned && lv.size >= cut_from && lv.scatter_n_chunks < SCATTER_MAX_CHUNKS) {
			const uint32_t per_chunk = std::min(capacity, std::max(256u, next_multiple(div_round_up(lv.size, SCATTER_MAX_CHUNKS), 8u)));
			cut(lv, per_chunk);
		}
	}
}

void grid_mask_to_bits(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, uint32_t n, const uint64_t* mask, uint64_t* bits) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_grid_mask_to_bits, dim3(div_round_up(n, 256), meta.n_levels), dim3(256), 0, stream, dev_meta, n, (const unsigned long long*)mask, (unsigned long long*)bits);
}

// Splits per level from MEASURED per-level work (microseconds of workgroup time, summed over the level's tasks of a first
// launch).  Levels cut into many chunks ("fine") cannot be split further cheaply -- every extra split flushes a whole chunk
// through global atomics -- so their N_f tasks of t_f microseconds each set the rhythm: they need R rounds on the CUs that
// are left.  Levels with few chunks ("coarse": few entries, cheap shared flush) are cut into just enough LONG tasks to keep
// x = 256 - ceil(N_f / R) CUs busy for those R rounds; they are launched first.  Without this the coarse tasks were sized
// like the fine ones and the ~850 equal tasks needed a fourth, nearly empty round (20 of 88 us on C3a).
static void tuned_splits(const GridMeta& meta, uint32_t n, const std::vector<float>& level_us, std::vector<uint32_t>& splits) {
	const uint32_t n_cus = 256 * std::max(1u, (160u * 1024u) / SCATTER_LDS_BYTES); // workgroups that run at once
	double w_coarse = 0, w_fine = 0;
	uint32_t n_fine = 0;
	auto is_fine = [&](uint32_t l) { return meta.levels[l].scatter_n_chunks > 8; };
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		if (meta.levels[l].scatter_binned) continue;
		if (is_fine(l)) { w_fine += level_us[l]; n_fine += meta.levels[l].scatter_n_chunks; }
		else w_coarse += level_us[l];
	}
	double task_us; // duration the coarse tasks should have
	uint32_t coarse_cus = n_cus;
	if (n_fine > 0) {
		const double t_f = w_fine / n_fine;
		uint32_t x = n_cus / 4;
		double best = 1e30;
		for (uint32_t rounds = 1; rounds <= 64; ++rounds) { // makespan(rounds) = max(fine rounds, coarse work on the CUs that are left)
			const uint32_t fine_cus = div_round_up(n_fine, rounds);
			if (fine_cus >= n_cus) continue;
			const double makespan = std::max(rounds * t_f, w_coarse / (n_cus - fine_cus));
			if (makespan < best) {
				best = makespan;
				x = n_cus - fine_cus;
			}
		}
		coarse_cus = x;
		task_us = std::max(w_coarse / std::max(x - 1, 1u), 0.25 * t_f);
	} else {
		// only coarse levels: one task per CU.  (A model that also charges the shared chunks' flush atomics -- x = sqrt(W G / V)
		// tasks instead of 256 -- was measured and lost: C3b 0.188 -> 0.208 ms, C5's level-0 scatter 151 -> 172 us.)
		task_us = std::max(w_coarse / n_cus, 4.0);
	}
	const uint32_t max_splits = std::max(n / 1024u, 1u);
	if (getenv("TCNN_AMD_SCATTER_TIMING")) {
		fprintf(stderr, "scatter tuner: n_fine %u w_fine %.1f w_coarse %.1f task_us %.1f; level_us:", n_fine, w_fine, w_coarse, task_us);
		for (uint32_t l = 0; l < meta.n_levels; ++l) fprintf(stderr, " %.0f", level_us[l]);
		fprintf(stderr, "\n");
	}
	if (n_fine > 0) {
		// With fine levels in the plan the long tasks have a budget: the CUs set aside for them minus one spare (one long task too
		// many costs the fine tasks a whole round).  Inside it the splits minimise the LONGEST task: start from one task per chunk
		// and keep splitting the level whose tasks are longest while the budget allows.  (Rounding level time / target time per
		// level, as before, sat on a knife-edge: 1.5 % of measurement noise turned 299 / 65.7 -> 5 splits into 302 / 67.3 -> 4,
		// i.e. 75 us tasks in a 70 us kernel, in every third run.)
		const uint32_t budget = coarse_cus > 1 ? coarse_cus - 1 : 1;
		uint32_t n_coarse_tasks = 0;
		for (uint32_t l = 0; l < meta.n_levels; ++l) {
			splits[l] = 1;
			if (!is_fine(l) && !meta.levels[l].scatter_binned) n_coarse_tasks += meta.levels[l].scatter_n_chunks;
		}
		while (true) {
			uint32_t longest = meta.n_levels;
			double t_longest = 0;
			for (uint32_t l = 0; l < meta.n_levels; ++l) {
				if (is_fine(l) || meta.levels[l].scatter_binned) continue;
				const double t = level_us[l] / (meta.levels[l].scatter_n_chunks * splits[l]);
				if (t > t_longest) { t_longest = t; longest = l; }
			}
			// stop when the longest tasks cannot be cut further (budget, sample granularity) or are short enough not to matter
			if (longest == meta.n_levels || splits[longest] >= max_splits || n_coarse_tasks + meta.levels[longest].scatter_n_chunks > budget || t_longest <= 0.25 * task_us) break;
			++splits[longest];
			n_coarse_tasks += meta.levels[longest].scatter_n_chunks;
		}
		return;
	}
	// Only coarse levels: ONE round of tasks, as equal as the levels allow -- start from one task per chunk and keep splitting the level
	// whose tasks are longest while all tasks still fit the CUs at once.  (Rounding level time / target time per level, as before,
	// sat on a knife-edge here too: the sum came out at 240 or at 280 tasks depending on a per cent of measurement noise, and 280 tasks
	// on 256 CUs are two rounds -- C3b's scatter took 60 us in one process and 86 us in the next.)
	uint32_t n_tasks = 0;
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		splits[l] = 1;
		if (!is_fine(l) && !meta.levels[l].scatter_binned) n_tasks += meta.levels[l].scatter_n_chunks;
	}
	while (true) {
		uint32_t longest = meta.n_levels;
		double t_longest = 0;
		for (uint32_t l = 0; l < meta.n_levels; ++l) {
			if (is_fine(l) || meta.levels[l].scatter_binned) continue;
			const double t = level_us[l] / (meta.levels[l].scatter_n_chunks * splits[l]);
			if (t > t_longest) { t_longest = t; longest = l; }
		}
		if (longest == meta.n_levels || splits[longest] >= max_splits || n_tasks + meta.levels[longest].scatter_n_chunks > n_cus || t_longest <= std::max(0.25 * task_us, 4.0)) break;
		++splits[longest];
		n_tasks += meta.levels[longest].scatter_n_chunks;
	}
}

void grid_scatter_plan(const GridMeta& meta, uint32_t n, std::vector<GridScatterTask>& tasks, std::vector<GridScatterRange>& shared_ranges, size_t& scratch_elems,
                       const std::vector<float>* measured_level_us) {
	const uint32_t F = meta.n_features_per_level;
	const uint32_t corners = meta.interpolation == (uint32_t)InterpolationType::Nearest ? 1u : (1u << meta.n_pos_dims);
	tasks.clear();
	shared_ranges.clear();
	scratch_elems = 0;
	std::vector<uint32_t> tuned(meta.n_levels, 0);
	if (measured_level_us) tuned_splits(meta, n, *measured_level_us, tuned);

	// profiling aid: TCNN_AMD_SCATTER_LEVELS="lo,hi" restricts the plan to levels lo..hi (results are then incomplete!)
	uint32_t dbg_lo = 0, dbg_hi = meta.n_levels;
	if (const char* e = getenv("TCNN_AMD_SCATTER_LEVELS")) sscanf(e, "%u,%u", &dbg_lo, &dbg_hi);

	std::vector<std::vector<GridScatterTask>> per_level(meta.n_levels);
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		if (l < dbg_lo || l > dbg_hi) continue;
		const GridLevel& lv = meta.levels[l];
		if (lv.scatter_binned) continue; // grid_backward_binned serves this level
		const uint32_t n_chunks = lv.scatter_n_chunks;
		const uint32_t per_chunk = lv.scatter_per_chunk;
		// corner events landing in one chunk; beyond ~64k the adds dominate: split the samples over several workgroups
		const uint64_t events = (uint64_t)n * corners / n_chunks;
		uint32_t splits = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(events / 65536, 1), 32);
		splits = std::min(splits, std::max(n / 1024u, 1u));
		if (measured_level_us) splits = tuned[l];
		const uint32_t samples_per_split = next_multiple(div_round_up(n, splits), 64u);
		for (uint32_t c = 0; c < n_chunks; ++c) {
			const uint32_t begin = c * per_chunk;
			if (begin >= lv.size) break;
			const uint32_t cnt = std::min(per_chunk, lv.size - begin);
			const bool shared = splits > 1;
			uint32_t scratch_begin = 0;
			if (shared) {
				scratch_begin = (uint32_t)scratch_elems;
				const size_t grad_begin = ((size_t)lv.offset + begin) * F;
				if (!shared_ranges.empty() && shared_ranges.back().grad_begin + shared_ranges.back().n_elems == grad_begin &&
				    shared_ranges.back().scratch_begin + shared_ranges.back().n_elems == scratch_begin) {
					shared_ranges.back().n_elems += cnt * F;
				} else {
					shared_ranges.push_back(GridScatterRange{grad_begin, cnt * F, scratch_begin, 0});
				}
				scratch_elems += (size_t)cnt * F;
			}
			for (uint32_t s = 0; s < splits; ++s) {
				const uint32_t sb = s * samples_per_split;
				if (sb >= n) break;
				per_level[l].push_back(GridScatterTask{l, begin, cnt, sb, std::min(n, sb + samples_per_split), shared ? 1u : 0u, scratch_begin, 0});
			}
		}
	}

	// Speed-only placement: blocks b and b + 8 usually share an XCD (MI355X_MICROARCH.md "Workgroup dispatch"), so give all
	// tasks of a level the same residue mod 8 -- its coordinates, bit planes and gradient plane then stay in that XCD's L2.
	if (measured_level_us) {
		// tuned plan: the long coarse tasks go first (longest first), one after the other -- they run for the whole kernel
		std::vector<uint32_t> coarse;
		for (uint32_t l = 0; l < meta.n_levels; ++l) if (meta.levels[l].scatter_n_chunks <= 8 && !per_level[l].empty()) coarse.push_back(l);
		std::sort(coarse.begin(), coarse.end(), [&](uint32_t a, uint32_t b) { return (*measured_level_us)[a] / per_level[a].size() > (*measured_level_us)[b] / per_level[b].size(); });
		for (uint32_t l : coarse) {
			tasks.insert(tasks.end(), per_level[l].begin(), per_level[l].end());
			per_level[l].clear();
		}
		while (tasks.size() % 8) tasks.push_back(GridScatterTask{0, 0, 0, 0, 0, 0, 0, 0});
	}
	std::vector<std::vector<GridScatterTask>> bins(8);
	if (measured_level_us) {
		// the remaining tasks, level after level, cut into 8 equal runs: every XCD gets the same number of tasks and sees
		// at most a few levels (whole levels per XCD, as below, leave XCDs idle when levels / 8 is not an integer)
		std::vector<GridScatterTask> all;
		for (uint32_t l = 0; l < meta.n_levels; ++l) all.insert(all.end(), per_level[l].begin(), per_level[l].end());
		const size_t per_bin = (all.size() + 7) / 8;
		for (size_t i = 0; i < all.size(); ++i) bins[std::min<size_t>(i / std::max<size_t>(per_bin, 1), 7)].push_back(all[i]);
	} else {
		std::vector<uint32_t> order(meta.n_levels);
		for (uint32_t l = 0; l < meta.n_levels; ++l) order[l] = l;
		std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return per_level[a].size() > per_level[b].size(); });
		for (uint32_t l : order) {
			if (per_level[l].empty()) continue;
			size_t best = 0;
			for (size_t b = 1; b < 8; ++b) if (bins[b].size() < bins[best].size()) best = b;
			bins[best].insert(bins[best].end(), per_level[l].begin(), per_level[l].end());
		}
	}
	size_t longest = 0;
	for (const auto& b : bins) longest = std::max(longest, b.size());
	for (size_t j = 0; j < longest; ++j) {
		for (size_t b = 0; b < 8; ++b) {
			if (j < bins[b].size()) tasks.push_back(bins[b][j]);
			else tasks.push_back(GridScatterTask{0, 0, 0, 0, 0, 0, 0, 0});
		}
	}
	while (!tasks.empty() && tasks.back().n_entries == 0) tasks.pop_back();
}

std::vector<float> grid_scatter_level_costs(const GridMeta& meta, const std::vector<GridScatterTask>& tasks, const std::vector<uint64_t>& times) {
	std::vector<float> level_us(meta.n_levels, 0.0f);
	for (size_t i = 0; i < tasks.size(); ++i) {
		if (!tasks[i].n_entries) continue;
		const uint64_t t0 = times[i * 8], t3 = times[i * 8 + 3];
		if (t3 > t0) level_us[tasks[i].level] += (float)(t3 - t0) * 0.01f; // s_memrealtime: 100 MHz
	}
	return level_us;
}

// The parameter ranges (relative to the encoding's first parameter) whose optimizer step a launch with an AdamInFlush performs:
// the chunks with a single owner.  Empty when this plan cannot carry the step (a chunk that is not a whole number of aligned quads).
ParamRanges grid_scatter_adam_ranges(const GridMeta& meta, const std::vector<GridScatterTask>& tasks, bool dy_records) {
	ParamRanges r;
	if (!dy_records) return r;
	const size_t F = meta.n_features_per_level;
	for (const GridScatterTask& t : tasks) {
		if (t.n_entries == 0 || t.flush_atomic) continue;
		const size_t begin = ((size_t)meta.levels[t.level].offset + t.entry_begin) * F, end = begin + (size_t)t.n_entries * F;
		if (begin % 4 || end % 4) return ParamRanges{};
		r.emplace_back(begin, end);
	}
	std::sort(r.begin(), r.end());
	ParamRanges merged;
	for (const auto& x : r) {
		if (!merged.empty() && merged.back().second == x.first) merged.back().second = x.second;
		else merged.push_back(x);
	}
	return merged;
}

void grid_backward_lds(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, const GridScatterTask* dev_tasks, uint32_t n_tasks,
                       const GridScatterRange* dev_ranges, uint32_t n_ranges, uint64_t* scratch, uint32_t n, MatView x,
                       const void* dL_dy, uint32_t dy_stride_sample, uint32_t dy_stride_level, void* grad, const uint64_t* chunk_bits, bool accumulate, bool dy_records,
                       uint64_t* task_times, const AdamInFlush* adam, const MlpReduceJob* reduce_job) {
	if (n_tasks == 0) return;
	const unsigned long long* bits = (const unsigned long long*)chunk_bits;
	unsigned long long* sc = (unsigned long long*)scratch;
	CHECK_THROW(!dy_records || grid_scatter_records_supported(meta));
	struct TimesGuard {
		explicit TimesGuard(uint64_t* p) { g_task_times = (unsigned long long*)p; }
		~TimesGuard() { g_task_times = nullptr; }
	} guard{task_times};
	switch (meta.n_pos_dims) {
		case 2: dispatch_scatter<2>(stream, meta.n_features_per_level, dev_meta, dev_tasks, n_tasks, n, x, dL_dy, dy_stride_sample, dy_stride_level, grad, bits, sc, accumulate, dy_records, adam); break;
		case 3: dispatch_scatter<3>(stream, meta.n_features_per_level, dev_meta, dev_tasks, n_tasks, n, x, dL_dy, dy_stride_sample, dy_stride_level, grad, bits, sc, accumulate, dy_records, adam); break;
		case 4: dispatch_scatter<4>(stream, meta.n_features_per_level, dev_meta, dev_tasks, n_tasks, n, x, dL_dy, dy_stride_sample, dy_stride_level, grad, bits, sc, accumulate, dy_records, adam); break;
		default: throw std::runtime_error{"GridEncoding: number of input dims must be 2 or 3."};
	}
	grid_scatter_finalize(stream, dev_ranges, n_ranges, scratch, grad, accumulate, reduce_job);
}

void grid_scatter_finalize(hipStream_t stream, const GridScatterRange* dev_ranges, uint32_t n_ranges, uint64_t* scratch, void* grad, bool accumulate, const MlpReduceJob* reduce_job) {
	if (n_ranges == 0) return;
	const uint32_t n_reduce_blocks = reduce_job ? div_round_up(reduce_job->n_elems, (uint32_t)SLAB_REDUCE_ELEMS) : 0;
	hipLaunchKernelGGL(k_grid_scatter_finalize, dim3(n_reduce_blocks + n_ranges * FINALIZE_BLOCKS_PER_RANGE), dim3(FINALIZE_THREADS), 0, stream, dev_ranges, (unsigned long long*)scratch, (half_t*)grad,
	                   accumulate ? 1 : 0, n_reduce_blocks, reduce_job ? *reduce_job : MlpReduceJob{});
	if (reduce_job) reduce_job->taken = true;
}

} // namespace tcnn_amd
