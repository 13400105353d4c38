// k_grid_bin.hip -- dL/dgrid for levels whose table is cut into MANY LDS chunks (2^22 entries x F = 4: 1024 chunks): binned form.
//
// Replaces (reference, /root/reference): include/tiny-cuda-nn/encodings/grid.h:215-320 (kernel_grid_backward) and the memset
// of the gradient table (grid.h:858) for those levels; same arithmetic and the same exact result as k_grid_scatter.hip.
//
// Why a second shape.  The owner-computes scatter (k_grid_scatter.hip) lets the workgroup that owns a chunk walk the samples
// that touch it.  A sample touches 2^(D-1) chunks of a hashed level (its corner pairs along x are neighbours in memory, the
// rest are anywhere), so with n samples every level costs 2^(D-1) n visits, each a handful of RANDOM gathers (coordinates,
// dL/dy) plus 2^D hashes of which 2 are used -- and on MI355X a gather costs ~2 clocks per LANE per CU whatever its width
// (tools/ubench/gather.hip).  Measured on config C5 (3-D, F = 4, T = 2^22, 512k samples): 3.9 ms per step, 50 us per chunk
// for 2048 visits.  Here no kernel gathers from global memory at all:
//   k_bin_count   one workgroup per (level, tile of samples): corner indices -> histogram over chunks in LDS
//   k_bin_scan*   offsets: contributions are stored chunk-major, inside a chunk tile after tile
//   k_bin_fill    same tiles: recompute corners, form every contribution (half)weight * dL_dy in fp16 (grid.h:254), counting-sort
//                 the tile's contributions by chunk in LDS and copy each run to its place {u16 entry inside chunk, F halves}
//   k_bin_accum   one workgroup per (level, chunk): stream the chunk's contiguous list into 64-bit fixed-point LDS
//                 accumulators (integer adds: exact, order-free, deterministic), round once, store the chunk.
// Every global access is a dense stream; the only random traffic is inside LDS.  Traffic: 2 (2 + 2F) bytes per contribution.
#include "grid_fixed.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {

constexpr uint32_t BIN_TILE_CONTRIBS = 8192; // contributions sorted per workgroup (LDS: 4 + 2F bytes each)
constexpr uint32_t BIN_MAX_CHUNKS = 4096;    // chunks per level (LDS: three uint32 tables)
constexpr uint32_t BIN_COUNT_THREADS = 256;
constexpr uint32_t BIN_FILL_THREADS = 1024;
constexpr uint32_t BIN_ACC_THREADS = 1024;
constexpr uint32_t BIN_ACC_BYTES_MAX = 128 * 1024;

struct BinArgs {
	const GridMeta* meta;
	uint32_t n;
	uint32_t tile_samples;   // samples per tile
	uint32_t tile_contribs;  // LDS slots per tile (tile_samples << D)
	uint32_t n_tiles;
	uint32_t stride;         // row length of the per-tile tables (>= chunks of any binned level)
	uint32_t per_sample;     // contributions per sample (2^D, 1 for nearest-neighbour interpolation)
	uint8_t level_of_slot[MAX_N_LEVELS]; // binned levels, in order
};

// barrier for LDS traffic only: does not wait for global loads / stores in flight
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// corners of one sample in the reference's order (grid.h:232-260), weights formed exactly as k_grid_fwd / k_grid_scatter form them
template <int D, typename Fn>
__device__ inline void for_each_corner(const GridLevel& lv, const uint32_t (&primes)[D], const uint32_t hash_type, const uint32_t interpolation, const float (&xin)[D], Fn&& fn) {
	float pos[D], unused;
	uint32_t cell[D];
#pragma unroll
	for (int d = 0; d < D; ++d) cell[d] = pos_fract(xin[d], lv.scale, interpolation, &pos[d], &unused);
	if (interpolation == (uint32_t)InterpolationType::Nearest) {
		fn(level_index<D>(lv, primes, hash_type, cell), 1.0f);
		return;
	}
#pragma unroll
	for (int idx = 0; idx < (1 << D); ++idx) {
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
		fn(level_index<D>(lv, primes, hash_type, local), weight);
	}
}

// Tiles of one level are handed to the XCDs in runs: workgroup b runs on XCD b % 8, and consecutive tiles write neighbouring
// pieces of every chunk's list, so a run of tiles in one L2 assembles whole lines before they leave for HBM.
__device__ inline uint32_t tile_of_block(const uint32_t b, const uint32_t n_tiles) {
	if (n_tiles % 8 != 0) return b;
	return (b & 7u) * (n_tiles / 8) + (b >> 3);
}

template <int D>
__global__ void __launch_bounds__(BIN_COUNT_THREADS) k_bin_count(const BinArgs a, const MatView x, uint32_t* __restrict__ counts) {
	__shared__ uint32_t hist[BIN_MAX_CHUNKS];
	const uint32_t tid = threadIdx.x, slot = blockIdx.y, tile = tile_of_block(blockIdx.x, a.n_tiles);
	const GridLevel lv = a.meta->levels[a.level_of_slot[slot]];
	const uint32_t n_chunks = lv.scatter_n_chunks;
	for (uint32_t c = tid; c < n_chunks; c += BIN_COUNT_THREADS) hist[c] = 0;
	__syncthreads();
	const uint32_t interpolation = a.meta->interpolation, hash_type = a.meta->hash_type;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = a.meta->primes[d];
	const uint32_t s0 = tile * a.tile_samples, s1 = min(a.n, s0 + a.tile_samples);
	for (uint32_t i = s0 + tid; i < s1; i += BIN_COUNT_THREADS) {
		float xin[D];
		load_coords<D>(x, i, xin);
		for_each_corner<D>(lv, primes, hash_type, interpolation, xin, [&](const uint32_t index, float) {
			__hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)&hist[scatter_chunk(lv, index)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		});
	}
	__syncthreads();
	uint32_t* __restrict__ row = counts + ((size_t)slot * a.n_tiles + tile) * a.stride;
	for (uint32_t c = tid; c < n_chunks; c += BIN_COUNT_THREADS) row[c] = hist[c];
}

// counts [slot][tile][chunk] -> rel [slot][tile][chunk] = contributions of earlier tiles to the same chunk; totals [slot][chunk].
// One workgroup per 64 chunks: lane = chunk, wave = range of tiles.
__global__ void __launch_bounds__(1024) k_bin_scan_tiles(const BinArgs a, const uint32_t* __restrict__ counts, uint32_t* __restrict__ rel, uint32_t* __restrict__ totals) {
	__shared__ uint32_t wave_sum[16][64];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = blockIdx.y;
	const uint32_t chunk = blockIdx.x * 64 + lane;
	const uint32_t per_wave = (a.n_tiles + 15) / 16;
	const uint32_t t0 = min(wave * per_wave, a.n_tiles), t1 = min(t0 + per_wave, a.n_tiles);
	const uint32_t* __restrict__ src = counts + (size_t)slot * a.n_tiles * a.stride + chunk;
	uint32_t* __restrict__ dst = rel + (size_t)slot * a.n_tiles * a.stride + chunk;
	uint32_t sum = 0;
#pragma unroll 8
	for (uint32_t t = t0; t < t1; ++t) sum += src[(size_t)t * a.stride];
	wave_sum[wave][lane] = sum;
	__syncthreads();
	uint32_t run = 0, total = 0;
	for (uint32_t w = 0; w < 16; ++w) {
		const uint32_t v = wave_sum[w][lane];
		if (w < wave) run += v;
		total += v;
	}
#pragma unroll 8
	for (uint32_t t = t0; t < t1; ++t) {
		const uint32_t v = src[(size_t)t * a.stride]; // second read: L2
		dst[(size_t)t * a.stride] = run;
		run += v;
	}
	if (wave == 0) totals[(size_t)slot * a.stride + chunk] = total;
}

// totals [slot][chunk] -> base [slot][chunk] = first contribution of the chunk inside the level's list (exclusive scan over chunks)
__global__ void __launch_bounds__(1024) k_bin_scan_chunks(const BinArgs a, const uint32_t* __restrict__ totals, uint32_t* __restrict__ base) {
	__shared__ uint32_t part[1024];
	const uint32_t tid = threadIdx.x, slot = blockIdx.x;
	const uint32_t n_chunks = a.meta->levels[a.level_of_slot[slot]].scatter_n_chunks;
	constexpr uint32_t K = BIN_MAX_CHUNKS / 1024;
	uint32_t v[K], sum = 0;
#pragma unroll
	for (uint32_t k = 0; k < K; ++k) {
		const uint32_t c = tid * K + k;
		v[k] = c < n_chunks ? totals[(size_t)slot * a.stride + c] : 0u;
		sum += v[k];
	}
	part[tid] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
		const uint32_t add = tid >= d ? part[tid - d] : 0u;
		__syncthreads();
		part[tid] += add;
		__syncthreads();
	}
	uint32_t run = part[tid] - sum;
#pragma unroll
	for (uint32_t k = 0; k < K; ++k) {
		const uint32_t c = tid * K + k;
		if (c < n_chunks) base[(size_t)slot * a.stride + c] = run;
		run += v[k];
	}
}

template <int D, int F>
__global__ void __launch_bounds__(BIN_FILL_THREADS) k_bin_fill(
	const BinArgs a, const MatView x, const half_t* __restrict__ dL_dy, const uint32_t dy_stride_sample, const uint32_t dy_stride_level,
	const uint32_t* __restrict__ counts, const uint32_t* __restrict__ rel, const uint32_t* __restrict__ base, uint16_t* __restrict__ out_idx, half_t* __restrict__ out_val
) {
	typedef typename VecOf<half_t, F>::type vecF;
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	vecF* s_val = (vecF*)smem;                                            // [tile_contribs]
	uint32_t* s_idx = (uint32_t*)(smem + a.tile_contribs * sizeof(vecF)); // [tile_contribs] chunk << 16 | entry inside chunk
	uint32_t* s_loff = s_idx + a.tile_contribs;                           // [stride] first slot of the chunk inside the tile
	uint32_t* s_cursor = s_loff + a.stride;                                 // [stride] next free slot
	uint32_t* s_goff = s_cursor + a.stride;                                 // [stride] where the chunk's run goes in the level's list
	__shared__ uint32_t part[BIN_FILL_THREADS];

	const uint32_t tid = threadIdx.x, slot = blockIdx.y, tile = tile_of_block(blockIdx.x, a.n_tiles);
	const uint32_t level = a.level_of_slot[slot];
	const GridLevel lv = a.meta->levels[level];
	const uint32_t n_chunks = lv.scatter_n_chunks;
	const size_t row = ((size_t)slot * a.n_tiles + tile) * a.stride;

	// ---- local offsets: exclusive scan of this tile's histogram
	for (uint32_t c = tid; c < n_chunks; c += BIN_FILL_THREADS) {
		s_cursor[c] = counts[row + c];
		s_goff[c] = base[(size_t)slot * a.stride + c] + rel[row + c];
	}
	__syncthreads();
	const uint32_t K = (n_chunks + BIN_FILL_THREADS - 1) / BIN_FILL_THREADS; // consecutive chunks per thread
	uint32_t sum = 0;
	for (uint32_t k = 0; k < K; ++k) {
		const uint32_t c = tid * K + k;
		if (c < n_chunks) sum += s_cursor[c];
	}
	part[tid] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < BIN_FILL_THREADS; d <<= 1) {
		const uint32_t add = tid >= d ? part[tid - d] : 0u;
		__syncthreads();
		part[tid] += add;
		__syncthreads();
	}
	uint32_t run = part[tid] - sum;
	for (uint32_t k = 0; k < K; ++k) {
		const uint32_t c = tid * K + k;
		if (c < n_chunks) {
			const uint32_t cnt = s_cursor[c];
			s_loff[c] = run;
			s_cursor[c] = run;
			run += cnt;
		}
	}
	__syncthreads();

	// ---- contributions of the tile's samples, placed by chunk
	const uint32_t interpolation = a.meta->interpolation, hash_type = a.meta->hash_type;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = a.meta->primes[d];
	const half_t* __restrict__ dy = dL_dy + (size_t)level * dy_stride_level;
	const uint32_t s0 = tile * a.tile_samples, s1 = min(a.n, s0 + a.tile_samples);
	for (uint32_t i = s0 + tid; i < s1; i += BIN_FILL_THREADS) {
		float xin[D];
		load_coords<D>(x, i, xin);
		const vecF gv = *(const vecF*)&dy[(size_t)i * dy_stride_sample];
		for_each_corner<D>(lv, primes, hash_type, interpolation, xin, [&](const uint32_t index, const float weight) {
			const uint32_t chunk = scatter_chunk(lv, index);
			const uint32_t entry = index - chunk * lv.scatter_per_chunk;
			const uint32_t at = __hip_atomic_fetch_add((lds_u32*)&s_cursor[chunk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const half_t w = (half_t)weight;
			vecF c;
#pragma unroll
			for (int f = 0; f < F; ++f) c[f] = w * gv[f]; // (GRAD_T)weight * grad in fp16, grid.h:254
			s_idx[at] = chunk << 16 | entry;
			s_val[at] = c;
		});
	}
	__syncthreads();

	// ---- copy the sorted tile out: run of chunk c -> list of chunk c, behind the runs of earlier tiles
	const size_t level_base = (size_t)slot * a.n * a.per_sample;
	uint16_t* __restrict__ o_idx = out_idx + level_base;
	vecF* __restrict__ o_val = (vecF*)out_val + level_base;
	const uint32_t total = (s1 - s0) * a.per_sample;
	for (uint32_t j = tid; j < total; j += BIN_FILL_THREADS) {
		const uint32_t e = s_idx[j];
		const uint32_t chunk = e >> 16;
		const uint32_t at = s_goff[chunk] + (j - s_loff[chunk]);
		o_idx[at] = (uint16_t)e;
		o_val[at] = s_val[j];
	}
}

// One workgroup per CU walks chunks b, b + gridDim.x, ...  Per chunk: adds (LDS integer atomics), then flush: round, store,
// and re-zero the accumulators in the same pass.  The list of the NEXT chunk is loaded while the current one is flushed, and
// the (first, count) pair of the one after that while the adds run, so no phase waits for HBM.
template <int F>
__global__ void __launch_bounds__(BIN_ACC_THREADS) k_bin_accum(
	const BinArgs a, const uint32_t n_slots, const uint32_t acc_bytes, const uint32_t* __restrict__ totals, const uint32_t* __restrict__ base, const uint16_t* __restrict__ in_idx,
	const half_t* __restrict__ in_val, half_t* __restrict__ grad, const int accumulate_mode, unsigned long long* __restrict__ dbg, uint32_t* __restrict__ fallback_count, const int force_wide
) {
	typedef typename VecOf<half_t, F>::type vecF;
	typedef __attribute__((address_space(3))) unsigned long long lds_u64;
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	// Round 5: two features of an entry in ONE 64-bit LDS add as 2 x int32 (low half sign-extended into the high one), as in
	// k_grid_scatter_lists.hip -- the adds are what this kernel's time is (LDS integer atomics: 3.9 us per chunk of 4096 entries x 4 features).
	// The same proof: while sum |contribution| over everything the chunk receives (+ the largest existing gradient in Accumulate mode) stays
	// below 120 of the 128 that 2^31 units of 2^-24 hold, no half can have overflowed; a chunk whose bound fails is zeroed and added again
	// in 64 bits -- slower, never different (counted in *fallback_count).
	__shared__ uint32_t s_bound[2][2]; // per chunk parity: {sum of the waves' |contribution| sums in units of 2^-20, largest |initial value| as float bits}
	extern __shared__ __attribute__((aligned(16))) char smem[];
	long long* acc = (long long*)smem;
	lds_u64* acc_lds = (lds_u64*)smem;
	unsigned long long t_add = 0, t_flush = 0, t_mark = dbg ? __builtin_amdgcn_s_memrealtime() : 0; // development aid (TCNN_AMD_BIN_TIMING)
	const unsigned long long t_start = t_mark;
	auto lap = [&](unsigned long long& into) {
		if (dbg) {
			const unsigned long long now = __builtin_amdgcn_s_memrealtime();
			into += now - t_mark;
			t_mark = now;
		}
	};
	struct SlotInfo { uint32_t task_begin, n_chunks, per_chunk, size, offset; };
	__shared__ SlotInfo slots[MAX_N_LEVELS + 1];
	const uint32_t tid = threadIdx.x;

	if (tid < 4) ((uint32_t*)s_bound)[tid] = 0;
	if (tid == 0) { // chunks of all binned levels, numbered consecutively
		uint32_t t = 0;
		for (uint32_t s = 0; s < n_slots; ++s) {
			const GridLevel& lv = a.meta->levels[a.level_of_slot[s]];
			slots[s] = SlotInfo{t, lv.scatter_n_chunks, lv.scatter_per_chunk, lv.size, lv.offset};
			t += lv.scatter_n_chunks;
		}
		slots[n_slots].task_begin = t;
	}
	{
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		u4* a4 = (u4*)smem;
		for (uint32_t i = tid; i < acc_bytes / 16; i += BIN_ACC_THREADS) a4[i] = u4{0, 0, 0, 0};
	}
	__syncthreads();
	const uint32_t n_tasks = slots[n_slots].task_begin;

	struct Task { uint32_t slot, chunk, first, count; };
	auto locate = [&](const uint32_t t, Task& k) { // issues the loads of (first, count)
		uint32_t s = 0;
		while (s + 1 < n_slots && slots[s + 1].task_begin <= t) ++s;
		k.slot = s;
		k.chunk = t - slots[s].task_begin;
		k.first = base[(size_t)s * a.stride + k.chunk];
		k.count = totals[(size_t)s * a.stride + k.chunk];
	};
	constexpr int PRE = 4; // list items per thread held in registers (4096 items per chunk on average)
	uint32_t e[PRE];
	vecF v[PRE];
	auto prefetch = [&](const Task& k) {
		// unconditional loads from clamped addresses: a load under `if (j < count)` into a register that is still live makes the
		// compiler wait for each load before issuing the next (measured: the 8 loads cost 4 HBM latencies in a row)
		const size_t at = (size_t)k.slot * a.n * a.per_sample + k.first;
#pragma unroll
		for (int p = 0; p < PRE; ++p) {
			const uint32_t j = tid + p * BIN_ACC_THREADS;
			const size_t src = k.count ? at + min(j, k.count - 1) : 0;
			e[p] = in_idx[src];
			v[p] = ((const vecF*)in_val)[src];
		}
	};
	auto add = [&](const uint32_t entry, const vecF& c) {
#pragma unroll
		for (int f = 0; f < F; ++f) __hip_atomic_fetch_add(acc_lds + entry * F + f, (unsigned long long)half_to_fixed_fast(c[f]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	float bsum = 0, bmax = 0; // this thread's share of the chunk's bound
	auto pack2 = [](const float p0, const float p1) -> unsigned long long {
		const int f0 = (int)(p0 * 16777216.0f), f1 = (int)(p1 * 16777216.0f); // exact below 128; beyond it the bound has failed anyway
		return (unsigned long long)(uint32_t)f0 | ((unsigned long long)(uint32_t)(f1 + (f0 >> 31)) << 32); // f0 + f1 2^32 as one 64-bit integer
	};
	auto add_packed = [&](const uint32_t entry, const vecF& c) {
#pragma unroll
		for (int j = 0; j < F / 2; ++j) {
			const float p0 = (float)c[2 * j], p1 = (float)c[2 * j + 1];
			bsum += __builtin_fabsf(p0);
			bsum += __builtin_fabsf(p1);
			__hip_atomic_fetch_add(acc_lds + entry * (F / 2) + j, pack2(p0, p1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	};

	uint32_t t = blockIdx.x;
	if (t >= n_tasks) return;
	uint32_t chunk_parity = 0;
	Task cur, next;
	locate(t, cur);
	prefetch(cur);
	if (t + gridDim.x < n_tasks) locate(t + gridDim.x, next);
	for (; t < n_tasks; t += gridDim.x) {
		const SlotInfo si = slots[cur.slot];
		const uint32_t entry_begin = cur.chunk * si.per_chunk;
		const uint32_t n_vals = entry_begin < si.size ? min(si.per_chunk, si.size - entry_begin) * F : 0u;
		half_t* __restrict__ g = grad + ((size_t)si.offset + entry_begin) * F;

		// ---- adds: packed (two 32-bit sums per LDS add) while the chunk's bound holds
		const uint32_t par = chunk_parity;
		chunk_parity ^= 1u;
		const size_t list_at = (size_t)cur.slot * a.n * a.per_sample + cur.first;
		bool packed_ok = !force_wide;
		if (packed_ok) {
			bsum = 0;
			bmax = 0;
			if (accumulate_mode) { // GradientMode::Accumulate: the existing gradient joins the sum
				for (uint32_t i = tid; i < n_vals / 2; i += BIN_ACC_THREADS) {
					const float g0 = (float)g[2 * i], g1 = (float)g[2 * i + 1];
					bmax = fmaxf(bmax, fmaxf(__builtin_fabsf(g0), __builtin_fabsf(g1)));
					if (!(__builtin_fabsf(g0) < 128.0f) || !(__builtin_fabsf(g1) < 128.0f)) bmax = 1e30f; // also NaN / inf
					__hip_atomic_fetch_add(acc_lds + i, pack2(g0, g1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
#pragma unroll
			for (int p = 0; p < PRE; ++p) if (tid + p * BIN_ACC_THREADS < cur.count) add_packed(e[p], v[p]);
			for (uint32_t j = tid + PRE * BIN_ACC_THREADS; j < cur.count; j += BIN_ACC_THREADS) add_packed(in_idx[list_at + j], ((const vecF*)in_val)[list_at + j]);
			// the chunk's bound: the waves' sums meet in one LDS word (integers: the order does not matter), read behind the barrier the adds need anyway
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				bsum += __shfl_xor(bsum, o);
				bmax = fmaxf(bmax, __shfl_xor(bmax, o));
			}
			if ((tid & 63u) == 0) {
				const float capped = bsum < 127.0f ? bsum : 127.0f; // (NaN compares false: 127)
				__hip_atomic_fetch_add((lds_u32*)&s_bound[par][0], (uint32_t)(capped * 1048576.0f) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_max((lds_u32*)&s_bound[par][1], __builtin_bit_cast(uint32_t, bmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			lds_barrier(); // LDS only: a full __syncthreads() would also wait for the loads and stores in flight
			packed_ok = (float)s_bound[par][0] * (1.0f / 1048576.0f) + __builtin_bit_cast(float, s_bound[par][1]) < 120.0f;
			if (tid == 0) { s_bound[par ^ 1u][0] = 0; s_bound[par ^ 1u][1] = 0; } // the next chunk's words (nobody touches them before the flush's barrier)
		}
		if (!packed_ok) { // 64-bit sums: nothing to prove
			if (!force_wide) {
				if (fallback_count && tid == 0) atomicAdd(fallback_count, 1u);
				typedef uint32_t u4 __attribute__((ext_vector_type(4)));
				for (uint32_t i = tid; i < (n_vals * 4 + 15) / 16; i += BIN_ACC_THREADS) ((u4*)smem)[i] = u4{0, 0, 0, 0}; // what the packed sums used
				lds_barrier();
			}
			if (accumulate_mode) {
				for (uint32_t i = tid; i < n_vals; i += BIN_ACC_THREADS)
					__hip_atomic_fetch_add(acc_lds + i, (unsigned long long)half_to_fixed(g[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
#pragma unroll
			for (int p = 0; p < PRE; ++p) if (tid + p * BIN_ACC_THREADS < cur.count) add(e[p], v[p]);
			for (uint32_t j = tid + PRE * BIN_ACC_THREADS; j < cur.count; j += BIN_ACC_THREADS) add(in_idx[list_at + j], ((const vecF*)in_val)[list_at + j]);
			lds_barrier();
		}
		lap(t_add);

		// ---- next chunk's list on its way; the one after that located
		const bool has_next = t + gridDim.x < n_tasks;
		Task after{};
		if (has_next) {
			prefetch(next);
			if (t + 2 * gridDim.x < n_tasks) locate(t + 2 * gridDim.x, after);
		}

		// ---- flush: round once (two values per 4-byte store; n_vals is even because F >= 2), leave the accumulators zero
		typedef _Float16 h2 __attribute__((ext_vector_type(2)));
		typedef long long i64x2 __attribute__((ext_vector_type(2)));
		if (packed_ok) { // one 64-bit word = the two sums of a pair; both fit 32 bits (that is what the bound proved)
			auto round32 = [](const int v) -> half_t {
				if (__builtin_expect((uint32_t)(v + (1 << 24)) < (1u << 25), 1)) return (half_t)((float)v * 5.9604644775390625e-08f); // |value| < 1: exact as a float, one rounding
				return fixed_to_half((long long)v);
			};
			for (uint32_t i = tid; i < n_vals / 2; i += BIN_ACC_THREADS) {
				const unsigned long long sum = ((unsigned long long*)acc)[i];
				((unsigned long long*)acc)[i] = 0;
				const int lo = (int)(uint32_t)sum, hi = (int)(uint32_t)((sum - (unsigned long long)(long long)lo) >> 32);
				((h2*)g)[i] = h2{round32(lo), round32(hi)};
			}
		} else
		for (uint32_t i = tid; i < n_vals / 2; i += BIN_ACC_THREADS) {
			const i64x2 s = ((i64x2*)acc)[i];
			((i64x2*)acc)[i] = i64x2{0, 0};
			// one branch for the pair: |s| < 2^24 (|value| < 1) for both -> the low words are the values, exact as floats, and the
			// float -> half conversion is the one rounding.  (The low words are hidden from the optimiser, which otherwise turns
			// (float)(int)s back into a 13-instruction 64-bit conversion.)
			const bool small = (unsigned long long)(s[0] + (1ll << 24)) < (1ull << 25) && (unsigned long long)(s[1] + (1ll << 24)) < (1ull << 25);
			h2 out;
			if (__builtin_expect(small, 1)) {
				int lo0 = (int)s[0], lo1 = (int)s[1];
				asm volatile("" : "+v"(lo0), "+v"(lo1));
				out = h2{(half_t)((float)lo0 * 5.9604644775390625e-08f), (half_t)((float)lo1 * 5.9604644775390625e-08f)};
			} else {
				out = h2{fixed_to_half(s[0]), fixed_to_half(s[1])};
			}
			((h2*)g)[i] = out;
		}
		lds_barrier();
		lap(t_flush);
		cur = next;
		next = after;
	}
	if (dbg && tid == 0) {
		dbg[blockIdx.x * 4 + 0] = t_add;
		dbg[blockIdx.x * 4 + 1] = t_flush;
		dbg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime() - t_start;
	}
}

struct BinLayout {
	BinArgs args;
	uint32_t n_slots = 0, n_chunks_total = 0;
	size_t counts_off = 0, rel_off = 0, totals_off = 0, base_off = 0, idx_off = 0, val_off = 0, bytes = 0;
};

BinLayout bin_layout(const GridMeta& meta, const GridMeta* dev_meta, uint32_t n) {
	BinLayout l{};
	BinArgs& a = l.args;
	a.meta = dev_meta;
	a.n = n;
	const bool nearest = meta.interpolation == (uint32_t)InterpolationType::Nearest;
	a.per_sample = nearest ? 1u : (1u << meta.n_pos_dims);
	a.tile_contribs = BIN_TILE_CONTRIBS;
	if (meta.n_features_per_level == 8) a.tile_contribs = BIN_TILE_CONTRIBS / 2; // (20 bytes per contribution in k_bin_fill's LDS instead of 12: half the tile)
	if (const char* e = getenv("TCNN_AMD_BIN_TILE")) a.tile_contribs = std::min<uint32_t>(std::max(atoi(e), 512), a.tile_contribs); // A/B runs
	a.tile_samples = a.tile_contribs >> meta.n_pos_dims; // nearest: the LDS arrays are simply not filled
	a.n_tiles = div_round_up(n, a.tile_samples);
	uint32_t max_chunks = 0;
	for (uint32_t i = 0; i < meta.n_levels; ++i) {
		if (!meta.levels[i].scatter_binned) continue;
		a.level_of_slot[l.n_slots++] = (uint8_t)i;
		max_chunks = std::max(max_chunks, meta.levels[i].scatter_n_chunks);
		l.n_chunks_total += meta.levels[i].scatter_n_chunks;
	}
	a.stride = next_multiple(std::max(max_chunks, 1u), 64u);
	auto take = [&](size_t bytes) {
		const size_t at = l.bytes;
		l.bytes += (bytes + 255) / 256 * 256;
		return at;
	};
	const size_t rows = (size_t)l.n_slots * a.n_tiles;
	const size_t contribs = (size_t)l.n_slots * n * a.per_sample;
	l.counts_off = take(rows * a.stride * 4);
	l.rel_off = take(rows * a.stride * 4);
	l.totals_off = take((size_t)l.n_slots * a.stride * 4);
	l.base_off = take((size_t)l.n_slots * a.stride * 4);
	l.idx_off = take(contribs * 2);
	l.val_off = take(contribs * meta.n_features_per_level * 2);
	return l;
}

template <int D, int F>
void launch_binned(hipStream_t s, const BinLayout& l, MatView x, const void* dy, uint32_t dss, uint32_t dsl, void* grad, bool accumulate, char* ws, uint32_t* fallback_count) {
	const BinArgs& a = l.args;
	uint32_t* counts = (uint32_t*)(ws + l.counts_off);
	uint32_t* rel = (uint32_t*)(ws + l.rel_off);
	uint32_t* totals = (uint32_t*)(ws + l.totals_off);
	uint32_t* base = (uint32_t*)(ws + l.base_off);
	uint16_t* idx = (uint16_t*)(ws + l.idx_off);
	half_t* val = (half_t*)(ws + l.val_off);
	const uint32_t fill_lds = a.tile_contribs * (F * 2 + 4) + 3 * a.stride * 4;
	static bool configured = false;
	if (!configured) { // more than 64 KiB of dynamic LDS has to be opted into once per kernel
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)k_bin_fill<D, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (F == 8 ? BIN_TILE_CONTRIBS / 2 : BIN_TILE_CONTRIBS) * (F * 2 + 4) + 3 * BIN_MAX_CHUNKS * 4));
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)k_bin_accum<F>, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_ACC_BYTES_MAX));
		configured = true;
	}
	hipLaunchKernelGGL((k_bin_count<D>), dim3(a.n_tiles, l.n_slots), dim3(BIN_COUNT_THREADS), 0, s, a, x, counts);
	hipLaunchKernelGGL(k_bin_scan_tiles, dim3(a.stride / 64, l.n_slots), dim3(1024), 0, s, a, counts, rel, totals);
	hipLaunchKernelGGL(k_bin_scan_chunks, dim3(l.n_slots), dim3(1024), 0, s, a, totals, base);
	hipLaunchKernelGGL((k_bin_fill<D, F>), dim3(a.n_tiles, l.n_slots), dim3(BIN_FILL_THREADS), fill_lds, s, a, x, (const half_t*)dy, dss, dsl, counts, rel, base, idx, val);
	const uint32_t acc_bytes = grid_bin_acc_bytes();
	const uint32_t acc_blocks = std::min(l.n_chunks_total, 256u * std::max(1u, (160u * 1024u) / (acc_bytes + 4096u)));
	static const bool timing = getenv("TCNN_AMD_BIN_TIMING") != nullptr;
	static int timing_left = 3;
	unsigned long long* dbg = nullptr;
	if (timing && timing_left > 0) HIP_CHECK_THROW(hipMalloc(&dbg, acc_blocks * 4 * 8));
	hipLaunchKernelGGL((k_bin_accum<F>), dim3(acc_blocks), dim3(BIN_ACC_THREADS), acc_bytes, s, a, l.n_slots, acc_bytes, totals, base, idx, val, (half_t*)grad, accumulate ? 1 : 0, dbg, fallback_count, switches().scatter_wide ? 1 : 0);
	if (dbg) {
		std::vector<unsigned long long> h(acc_blocks * 4);
		HIP_CHECK_THROW(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
		if (--timing_left == 0) {
			double add = 0, flush = 0, total = 0;
			for (uint32_t b = 0; b < acc_blocks; ++b) { add += h[b * 4] * 0.01; flush += h[b * 4 + 1] * 0.01; total += h[b * 4 + 2] * 0.01; }
			fprintf(stderr, "k_bin_accum: %u workgroups, %u chunks; per workgroup: adds %.1f us, flush %.1f us, total %.1f us\n", acc_blocks, l.n_chunks_total, add / acc_blocks, flush / acc_blocks, total / acc_blocks);
		}
		(void)hipFree(dbg);
	}
	HIP_CHECK_THROW(hipGetLastError());
}

} // namespace

uint32_t grid_bin_max_chunks() { return BIN_MAX_CHUNKS; }

// LDS accumulators per workgroup of k_bin_accum = chunk size of the binned levels.  TCNN_AMD_BIN_ACC_KB: A/B runs.
uint32_t grid_bin_acc_bytes() {
	static const uint32_t v = [] {
		uint32_t kb = 128;
		if (const char* e = getenv("TCNN_AMD_BIN_ACC_KB")) kb = (uint32_t)std::min(std::max(atoi(e), 16), 128);
		return kb * 1024u;
	}();
	return v;
}

bool grid_bin_supported(const GridMeta& meta) {
	const uint32_t D = meta.n_pos_dims, F = meta.n_features_per_level;
	return (F == 2 || F == 4 || F == 8) && D >= 2 && D <= 4;
}

size_t grid_bin_workspace_bytes(const GridMeta& meta, uint32_t n) { return bin_layout(meta, nullptr, n).bytes; }

void grid_backward_binned(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, uint32_t n, MatView x, const void* dL_dy, uint32_t dy_stride_sample,
                          uint32_t dy_stride_level, void* grad, bool accumulate, void* workspace, uint32_t* fallback_count) {
	const BinLayout l = bin_layout(meta, dev_meta, n);
	if (l.n_slots == 0 || n == 0) return;
	CHECK_THROW(grid_bin_supported(meta) && workspace != nullptr);
	CHECK_THROW(l.args.stride <= BIN_MAX_CHUNKS);
	for (uint32_t s = 0; s < l.n_slots; ++s) {
		const GridLevel& lv = meta.levels[l.args.level_of_slot[s]];
		CHECK_THROW(lv.scatter_per_chunk * meta.n_features_per_level * 8 <= grid_bin_acc_bytes() && lv.scatter_per_chunk <= 65536);
	}
	const uint32_t F = meta.n_features_per_level;
	char* ws = (char*)workspace;
#define TCNN_BIN(D) \
	if (F == 2) launch_binned<D, 2>(stream, l, x, dL_dy, dy_stride_sample, dy_stride_level, grad, accumulate, ws, fallback_count); \
	else if (F == 4) launch_binned<D, 4>(stream, l, x, dL_dy, dy_stride_sample, dy_stride_level, grad, accumulate, ws, fallback_count); \
	else launch_binned<D, 8>(stream, l, x, dL_dy, dy_stride_sample, dy_stride_level, grad, accumulate, ws, fallback_count);
	switch (meta.n_pos_dims) {
		case 2: TCNN_BIN(2) break;
		case 3: TCNN_BIN(3) break;
		default: TCNN_BIN(4) break;
	}
#undef TCNN_BIN
}

} // namespace tcnn_amd
