// k_grid_planes.hip -- the training-step form of the grid encoding forward pass (kernel_grid, grid.h:49-212) for gfx950.
//
// Same arithmetic as k_grid_fwd (fp32 corner weights, fp16 fma chain in corner order -- bit-identical results), different
// SHAPE, chosen for where the time goes on MI355X: the gathers of one level hit a 2^log2_hashmap_size * F * 2 byte table
// (2 MB for T = 2^19, F = 2) that fits in ONE XCD's 4 MB L2 but not next to the other 15 levels.
//   * level-major work list, cut into per-XCD runs on the host (grid_planes_plan): workgroup b runs on XCD b % 8 and walks
//     that XCD's run in order, so an XCD gathers from one or two tables at a time and they stay in its L2;
//   * a thread owns ONE level of SPT samples (8 x 4 gathers in flight), and the output is written as level planes
//     half [L][n][F] (dense 256-B stores per wave) which the fused MLP kernel reads directly;
//   * the sample filter of the owner-computes scatter is produced here.  Round 4 (LISTS): as HIT LISTS (GridHitLists, tcnn_common.h) --
//     a workgroup's item (FP_THREADS x FP_SPT samples of one level) is counting-sorted by chunk in LDS -- a rank per element from one
//     returning LDS add on the chunk's counter, a 64-lane scan for the chunks' offsets -- and leaves as ONE contiguous, chunk-sorted run
//     in the item's own region of the level's pool (coalesced stores) with its 65 offsets beside it.  No global atomics.  The scatter
//     then walks the items' runs of its chunk instead of scanning a bit plane.  Round 5: an element is a cell ROW with its two entries and
//     its two half weights (8 bytes + a 16-bit sample number) -- this kernel has them in registers anyway, and the owner then needs neither coordinates nor hashes.
//     Before: bit planes over samples per (level, chunk), built in LDS with integer ORs (or ballots for levels with few chunks); kept
//     for callers without lists.
#include "grid_device.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {

constexpr int FP_THREADS = 128; // 2 waves per workgroup
// ... and with hit lists: a work item is FP_LIST_THREADS x FP_SPT samples, and the item's run of a chunk (samples x rows / chunks elements
// of 8 bytes) is what the chunk's owner reads in one piece -- 16 elements = 128 bytes from a 512-sample item, and every piece drags the
// rest of its first and last 128-byte line along (measured in the owners: 19 us per task from such runs, 10 from one contiguous list,
// profiles/r05_scatter_dev.txt)
#ifndef TCNN_FP_LIST_THREADS
#define TCNN_FP_LIST_THREADS 256
#endif
constexpr int FP_LIST_THREADS = TCNN_FP_LIST_THREADS;

constexpr int FP_MAX_CHUNKS = GRID_FILTER_MAX_CHUNKS; // bit planes per level: levels cut into more chunks get no filter (they are binned, k_grid_bin.hip)
constexpr int FP_PER_LANE = FP_MAX_CHUNKS / 64;       // chunks whose words one lane carries out (chunk = lane + 64 j)

// FP_SPT = samples per thread = consecutive 64-sample groups per wave; a work item is FP_THREADS * FP_SPT samples of one level.
// Shapes: 4 (16 gathers in flight per thread; 8 is kept for A/B runs), and 2 for the big 3-D grids whose 8 corners x 4 features would
// not fit the registers.
template <int D, int F, int FP_SPT, bool LISTS>
__global__ void __launch_bounds__(LISTS ? FP_LIST_THREADS : FP_THREADS) k_grid_fwd_planes(
	const GridMeta* __restrict__ meta, const uint32_t* __restrict__ work, const uint32_t max_items, const uint32_t blocks_per_xcd, const uint32_t n, const MatView x,
	const half_t* __restrict__ grid, half_t* __restrict__ out, unsigned long long* __restrict__ bits, const MlpPrepJob prep, const GridHitLists lists
) {
	typedef typename VecOf<half_t, F>::type vecF;
	constexpr int THREADS = LISTS ? FP_LIST_THREADS : FP_THREADS;
	if (prep.image) { // side job (mlp_side_jobs.h): the fragment images of the network this batch is encoded for; independent of everything below
		const uint32_t total = (prep.desc.n_frags_fwd + prep.desc.n_frags_bwd + prep.desc.n_frags_r32) * 512;
		for (uint32_t e = blockIdx.x * THREADS + threadIdx.x; e < total; e += gridDim.x * THREADS) mlp_prep_element(prep.desc, (const half_t*)prep.params, (half_t*)prep.image, e);
	}
	constexpr int FP_WAVE_SAMPLES = 64 * FP_SPT;
	constexpr int FP_ITEM_SAMPLES = THREADS * FP_SPT;
	constexpr int STAGE_ELEMS = FP_ITEM_SAMPLES << (D - 1); // one element per (sample, row) of an item
	// bit planes (!LISTS) or the item's counting sort (LISTS): the staged elements, counters, offsets
	constexpr int STAGE_U64 = STAGE_ELEMS * (int)GRID_HIT_WORDS / 2 + STAGE_ELEMS / 4; // the item's elements: 8 bytes each + their 16-bit sample numbers
	static_assert(!LISTS || FP_ITEM_SAMPLES <= 65536, "16-bit sample numbers inside an item");
	__shared__ __attribute__((aligned(16))) unsigned long long lds_raw[LISTS ? STAGE_U64 + 72 : (FP_THREADS / 64) * FP_SPT * FP_MAX_CHUNKS];
	typedef unsigned long long plane_row[FP_SPT][FP_MAX_CHUNKS];
	plane_row* planes = (plane_row*)lds_raw; // [FP_THREADS / 64][FP_SPT][FP_MAX_CHUNKS]
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	uint32_t* l_stage = (uint32_t*)lds_raw;                 // [STAGE_ELEMS][GRID_HIT_WORDS] the item's elements in chunk order
	uint16_t* l_sidx = (uint16_t*)(lds_raw + STAGE_ELEMS * (int)GRID_HIT_WORDS / 2); // [STAGE_ELEMS] their samples, relative to the item's first
	uint32_t* l_cnt = (uint32_t*)(lds_raw + STAGE_U64);     // [64] elements of this item per chunk (zero between items)
	uint32_t* l_off = l_cnt + 64;                           // [64] first staged position of the chunk
	uint32_t* l_total = l_cnt + 128;
	static_assert(FP_MAX_CHUNKS == 64, "one lane per chunk");
	if constexpr (LISTS) {
		if (threadIdx.x < 64) l_cnt[threadIdx.x] = 0;
		// the counter set of the next forward launch on this stream (nobody reads it any more: the scatter that did has finished)
		if (blockIdx.x == gridDim.x - 1 && lists.zero_counts) {
			for (uint32_t e = threadIdx.x; e < meta->n_levels * GRID_HIT_COUNT_STRIDE; e += THREADS) lists.zero_counts[e] = 0;
		}
		__syncthreads();
	}

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
	const uint32_t n_items = work[xcd];
	const uint32_t* __restrict__ items = work + 8 + (size_t)xcd * max_items;

	const uint32_t interpolation = meta->interpolation;
	const uint32_t hash_type = meta->hash_type;
	uint32_t primes[D];
#pragma unroll
	for (int d = 0; d < D; ++d) primes[d] = meta->primes[d];
	const uint32_t n_words = n / 64;
	constexpr int C = 1 << D;
	constexpr int R_ROWS = C / 2;
	constexpr uint32_t HIT_SHIFT = grid_hit_mask_shift(D);

	for (uint32_t it = slot; it < n_items; it += blocks_per_xcd) {
		const uint32_t w = items[it];
		const uint32_t level = w >> 24;
		const uint32_t base = (w & 0xffffffu) * FP_ITEM_SAMPLES + wave * FP_WAVE_SAMPLES; // first sample of this wave
		if constexpr (!LISTS) { if (base >= n) continue; } // (LISTS: the waves of a workgroup meet at barriers; a wave past the end recomputes the last sample and emits nothing)
		const GridLevel lv = meta->levels[level];
		const half_t* __restrict__ lgrid = grid + (size_t)lv.offset * F;
		half_t* __restrict__ lout = out + (size_t)level * n * F;
		const uint32_t n_chunks = lv.scatter_n_chunks;
		const bool want_bits = !LISTS && bits != nullptr && n_chunks > 1 && n_chunks <= FP_MAX_CHUNKS;
		const bool lds_or = want_bits && n_chunks > 8;
		const bool want_lists = LISTS && n_chunks > 1 && n_chunks <= FP_MAX_CHUNKS; // workgroup-uniform: depends on the level only
		uint32_t cr[FP_SPT][R_ROWS]; // per (sample, row): chunk << 16 | rank inside (item, chunk); ~0: no element
		uint32_t ipair[FP_SPT][R_ROWS]; // ... and its word 1: entry of corner A | entry of corner B << 16, relative to the chunk (0xffff: none)
		const bool nearest = interpolation == (uint32_t)InterpolationType::Nearest;

		if (lds_or) {
#pragma unroll
			for (int k = 0; k < FP_SPT; ++k)
#pragma unroll
				for (int j = 0; j < FP_PER_LANE; ++j) planes[wave][k][lane + 64 * j] = 0ull;
		}

		// ---- phase 1: positions, indices, all gathers in flight.
		// A gather costs ~2 clocks per lane per CU whatever its width (tools/ubench/gather.hip), so the two corners of a cell
		// that differ only in x are fetched as ONE 2-entry load wherever they are neighbours in memory: always for dense-style
		// levels (index(x + 1) = index(x) + 1), and for hashed levels with prime[0] == 1 when x is even (index(x + 1) =
		// index(x) ^ 1).  A lane whose second corner is not in the pair issues an extra load; masked-off lanes cost nothing.
		constexpr int R = C / 2; // rows = corner pairs (x, x + 1)
		typedef typename VecOf<half_t, 2 * F>::type vecP __attribute__((aligned(2 * F)));
		vecP P[FP_SPT][R];
		vecF E[FP_SPT][R];
		uint32_t sel[FP_SPT]; // per row 3 bits: corner 0 is the pair's high entry | corner 1 is the pair's high entry | corner 1 came by its own load
		float pos[FP_SPT][D];
		unsigned long long touched[FP_SPT];
		auto note_chunk = [&](const int k, const uint32_t index) {
			const uint32_t ch = scatter_chunk(lv, index);
			if (lds_or) {
				// bit `lane` of word [k][ch]; integer LDS atomics run at ~4 lane-ops/clk/CU
				__hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)&planes[wave][k][ch] + (lane >> 5), 1u << (lane & 31), __ATOMIC_RELAXED,
				                      __HIP_MEMORY_SCOPE_WAVEFRONT);
			} else {
				touched[k] |= 1ull << ch;
			}
		};
#pragma unroll
		for (int k = 0; k < FP_SPT; ++k) {
			const uint32_t i = min(base + k * 64 + lane, n - 1); // groups past the end recompute the last sample and store nothing
			float xin[D];
			load_coords<D>(x, i, xin);
			uint32_t cell[D];
			float pd;
#pragma unroll
			for (int d = 0; d < D; ++d) cell[d] = pos_fract(xin[d], lv.scale, interpolation, &pos[k][d], &pd);
			touched[k] = 0;
			sel[k] = 0;
			if constexpr (LISTS) {
#pragma unroll
				for (int r = 0; r < R_ROWS; ++r) { cr[k][r] = 0xffffffffu; ipair[k][r] = 0; }
			}
			const bool lists_here = LISTS && want_lists && base + k * 64 + lane < n;
#pragma unroll
			for (int r = 0; r < R; ++r) {
				if (r > 0 && nearest) { P[k][r] = P[k][0]; E[k][r] = E[k][0]; continue; }
				uint32_t local[D];
				local[0] = cell[0];
#pragma unroll
				for (int d = 1; d < D; ++d) local[d] = cell[d] + ((r >> (d - 1)) & 1);
				const uint32_t idx0 = level_index<D>(lv, primes, hash_type, local);
				local[0] = cell[0] + 1;
				const uint32_t idx1 = nearest ? idx0 : level_index<D>(lv, primes, hash_type, local);
				const uint32_t a0 = lv.hashed ? (idx0 & ~1u) : min(idx0, lv.size - 2u);
				P[k][r] = *(const vecP*)&lgrid[(size_t)a0 * F];
				const uint32_t d1 = idx1 - a0;
				const bool own = d1 > 1u;
				vecF e;
#pragma unroll
				for (int f = 0; f < F; ++f) { if constexpr (F == 1) e = (half_t)0.0f; else e[f] = (half_t)0.0f; }
				if (own) e = *(const vecF*)&lgrid[(size_t)idx1 * F];
				E[k][r] = e;
				sel[k] |= ((idx0 - a0) | ((d1 & 1u) << 1) | (own ? 4u : 0u)) << (3 * r);
				if (want_bits) {
					note_chunk(k, idx0);
					if (!nearest) note_chunk(k, idx1);
				}
				if constexpr (LISTS) {
					// One element per (sample, row), in the list of corner A's chunk; corner B rides along when it lies in the same chunk -- else (a
					// row that straddles two chunks, one in ~8000 on hashed levels) it goes to the level's straggler list.
					if (lists_here) {
						const uint32_t ch_a = scatter_chunk(lv, idx0), first = ch_a * lv.scatter_per_chunk;
						uint32_t rel_b = idx0 - first; // no corner B in this chunk: the element names A's entry twice and gives the second a zero weight
						if (!nearest) {
							const uint32_t ch_b = scatter_chunk(lv, idx1);
							if (ch_b == ch_a) rel_b = idx1 - first;
							else {
								const uint32_t at = atomicAdd(&lists.counts[level * GRID_HIT_COUNT_STRIDE], 1u);
								if (at < lists.straggler_capacity) {
									typedef uint32_t u2 __attribute__((ext_vector_type(2)));
									*(u2*)&lists.stragglers[((size_t)level * lists.straggler_capacity + at) * 2] = u2{i | (2u << (2 * r)) << HIT_SHIFT, ch_b};
								}
							}
						}
						ipair[k][r] = (idx0 - first) | rel_b << 16;
						const uint32_t rank = __hip_atomic_fetch_add((lds_u32*)&l_cnt[ch_a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						cr[k][r] = ch_a << 16 | rank;
					}
				}
			}
		}
		if constexpr (LISTS) {
			if (want_lists) {
				__syncthreads(); // every element of the item has its rank
				if (wave == 0) {
					const uint32_t cnt = l_cnt[lane];
					l_cnt[lane] = 0;
					uint32_t incl = cnt;
#pragma unroll
					for (int o = 1; o < 64; o <<= 1) {
						const uint32_t up = __shfl_up(incl, o);
						if (lane >= (uint32_t)o) incl += up;
					}
					l_off[lane] = incl - cnt;
					uint32_t* heads = lists.heads + (size_t)level * GRID_HIT_HEADS * lists.n_items + (w & 0xffffffu); // [chunk][item]
#ifdef TCNN_AMD_DEV
					if (!(lists.dev_flags & 4u))
#endif
					heads[(size_t)lane * lists.n_items] = incl - cnt;
					if (lane == 63) { heads[(size_t)64 * lists.n_items] = incl; *l_total = incl; }
				}
				__syncthreads(); // the chunks' offsets are there (the gathers of phase 1 are in flight meanwhile)
			}
		}

		// ---- phase 2: interpolation (grid.h:142-169) and the plane stores
#pragma unroll
		for (int k = 0; k < FP_SPT; ++k) {
			const uint32_t i = base + k * 64 + lane;
			vecF v[C];
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const uint32_t s = sel[k] >> (3 * r);
				vecF lo, hi;
#pragma unroll
				for (int f = 0; f < F; ++f) { lo[f] = P[k][r][f]; hi[f] = P[k][r][F + f]; }
				v[2 * r] = (s & 1u) ? hi : lo;
				v[2 * r + 1] = (s & 4u) ? E[k][r] : ((s & 2u) ? hi : lo);
			}
			half_t acc[F];
			half_t whs[C]; // the corners' weights as halves: what the scatter multiplies dL/dy with (grid.h:254)
#pragma unroll
			for (int idx = 0; idx < C; ++idx) whs[idx] = (half_t)1.0f;
			if (nearest) { // grid.h:121-140
#pragma unroll
				for (int f = 0; f < F; ++f) acc[f] = v[0][f];
			} else {
#pragma unroll
				for (int f = 0; f < F; ++f) acc[f] = (half_t)0.0f;
#pragma unroll
				for (int idx = 0; idx < C; ++idx) {
					float weight = 1;
#pragma unroll
					for (int d = 0; d < D; ++d) weight *= (idx & (1 << d)) == 0 ? 1 - pos[k][d] : pos[k][d];
					asm volatile("" : "+v"(weight)); // round to fp32 first, then to fp16 (see k_grid.hip)
					const half_t wh = (half_t)weight;
					whs[idx] = wh;
#pragma unroll
					for (int f = 0; f < F; ++f) {
						half_t val;
						val = v[idx][f];
						acc[f] = __builtin_fmaf16(wh, val, acc[f]);
					}
				}
			}
			if (base + k * 64 < n) {
				vecF o;
#pragma unroll
				for (int f = 0; f < F; ++f) { if constexpr (F == 1) o = acc[f]; else o[f] = acc[f]; }
				*(vecF*)&lout[(size_t)i * F] = o;
			}
			if constexpr (LISTS) { // this sample's elements, staged at their place in chunk order
				if (want_lists) {
#pragma unroll
					for (int r = 0; r < R_ROWS; ++r) {
						const uint32_t v = cr[k][r];
						if (v != 0xffffffffu) {
							const uint32_t at = l_off[v >> 16] + (v & 0xffffu);
							typedef uint32_t u2 __attribute__((ext_vector_type(2)));
							const uint32_t pr = ipair[k][r];
							const bool no_b = (pr >> 16) == (pr & 0xffffu); // (two corners of a row are never the same entry)
							*(u2*)(l_stage + at * GRID_HIT_WORDS) = u2{pr, (uint32_t)__builtin_bit_cast(uint16_t, whs[2 * r]) | (no_b ? 0u : (uint32_t)__builtin_bit_cast(uint16_t, whs[2 * r + 1]) << 16)};
							l_sidx[at] = (uint16_t)(wave * FP_WAVE_SAMPLES + k * 64 + lane);
						}
					}
				}
			}
		}

		// ---- phase 3 (LISTS): the item's elements, staged in chunk order, leave as one run into the item's region of the level's pool
		if constexpr (LISTS) {
			if (want_lists) {
				__syncthreads();
				typedef uint32_t u4 __attribute__((ext_vector_type(4)));
				// (a region is a whole number of 16-byte quads: the last one of a run may carry a few words of the item before)
				const uint32_t total = *l_total, quads = (total * GRID_HIT_WORDS + 3) / 4, squads = (total + 7) / 8;
				const size_t region = ((size_t)level * lists.n_items + (w & 0xffffffu)) * lists.item_capacity;
				u4* dst = (u4*)(lists.elems + region * GRID_HIT_WORDS);
				u4* sdst = (u4*)(lists.sidx + region);
#ifdef TCNN_AMD_DEV
				if (!(lists.dev_flags & 1u))
#endif
				{ // (plain stores: streamed ones cost the scatter 13 us -- it reads the lists out of the Infinity Cache, which streamed stores pass by)
#ifdef TCNN_AMD_DEV
					if (lists.dev_flags & 8u) { // sc1: written through, not kept in this XCD's L2
						const auto rs_e = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)(quads * 16u), 0x00020000);
						const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)sdst, 0, (int)(squads * 16u), 0x00020000);
						for (uint32_t p = tid; p < quads; p += THREADS) __builtin_amdgcn_raw_buffer_store_b128(((const u4*)l_stage)[p], rs_e, p * 16u, 0, 16);
						for (uint32_t p = tid; p < squads; p += THREADS) __builtin_amdgcn_raw_buffer_store_b128(((const u4*)l_sidx)[p], rs_s, p * 16u, 0, 16);
					} else
#endif
					{
					for (uint32_t p = tid; p < quads; p += THREADS) dst[p] = ((const u4*)l_stage)[p];
					for (uint32_t p = tid; p < squads; p += THREADS) sdst[p] = ((const u4*)l_sidx)[p];
					}
				}
			}
		}
		// ---- phase 3: bit planes.  Lane c ends up with the FP_SPT consecutive words of chunk c (and of chunk c + 64): dense runs per lane.
		if (want_bits) {
#pragma unroll
			for (int j = 0; j < FP_PER_LANE; ++j) {
				if (j > 0 && !lds_or) break; // the ballot form serves levels with at most 8 chunks
				unsigned long long mine[FP_SPT];
				const uint32_t ch = lane + 64 * j;
				if (lds_or) {
					__builtin_amdgcn_wave_barrier(); // LDS serves one wave's instructions in order; keep the compiler from moving the reads up
#pragma unroll
					for (int k = 0; k < FP_SPT; ++k) mine[k] = planes[wave][k][ch];
				} else {
#pragma unroll
					for (int k = 0; k < FP_SPT; ++k) {
						mine[k] = 0;
						for (uint32_t c = 0; c < n_chunks; ++c) {
							const unsigned long long b = __ballot((touched[k] >> c) & 1ull);
							if (lane == c) mine[k] = b;
						}
					}
				}
				if (ch < n_chunks) {
					unsigned long long* dst = bits + ((size_t)level * FP_MAX_CHUNKS + ch) * n_words + base / 64;
					if (base + FP_WAVE_SAMPLES <= n) {
						typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
						for (int k = 0; k < FP_SPT; k += 2) *(u64x2*)&dst[k] = u64x2{mine[k], mine[k + 1]};
					} else {
#pragma unroll
						for (int k = 0; k < FP_SPT; ++k) if (base + k * 64 < n) dst[k] = mine[k];
					}
				}
			}
		}
	}
}

template <int D, int F, int SPT>
void launch_planes(hipStream_t s, const GridMeta* dm, const uint32_t* work, uint32_t max_items, uint32_t blocks_per_xcd, uint32_t n, MatView x, const void* grid, void* out, uint64_t* bits,
                   const MlpPrepJob* prep_job, const GridHitLists* hit_lists) {
	MlpPrepJob prep{};
	if (prep_job) prep = *prep_job;
	if (hit_lists) {
		if constexpr (SPT == 8) throw std::runtime_error{"grid_forward_planes: hit lists come from the 2- and 4-samples-per-thread shapes"};
		else
		hipLaunchKernelGGL((k_grid_fwd_planes<D, F, SPT, true>), dim3(8 * blocks_per_xcd), dim3(FP_LIST_THREADS), 0, s, dm, work, max_items, blocks_per_xcd, n, x, (const half_t*)grid, (half_t*)out,
		                   nullptr, prep, *hit_lists);
	} else {
		hipLaunchKernelGGL((k_grid_fwd_planes<D, F, SPT, false>), dim3(8 * blocks_per_xcd), dim3(FP_THREADS), 0, s, dm, work, max_items, blocks_per_xcd, n, x, (const half_t*)grid, (half_t*)out,
		                   (unsigned long long*)bits, prep, GridHitLists{});
	}
	HIP_CHECK_THROW(hipGetLastError());
}

uint32_t max_scatter_chunks(const GridMeta& meta) {
	uint32_t m = 1;
	for (uint32_t l = 0; l < meta.n_levels; ++l) m = std::max(m, meta.levels[l].scatter_n_chunks);
	return m;
}

} // namespace

uint32_t grid_hit_item_samples(const GridMeta& meta) { return FP_LIST_THREADS * grid_planes_spt(meta); }

// samples per thread of the kernel shape used for this grid (see k_grid_fwd_planes)
uint32_t grid_planes_spt(const GridMeta& meta) {
	// 4 samples per thread (16 gathers in flight, half the registers, twice the waves per CU) measured against 8 on C3a, four runs
	// each: 56.1-56.9 us against 58.7-59.2.  TCNN_AMD_FWD_SPT=8 keeps the old shape (A/B runs).
	uint32_t forced = 0;
#ifdef TCNN_AMD_DEV
	static const uint32_t dev_forced = getenv("TCNN_AMD_FWD_SPT") ? (uint32_t)atoi(getenv("TCNN_AMD_FWD_SPT")) : 0u; // laboratory knob
	forced = dev_forced;
#endif
	if (max_scatter_chunks(meta) > (uint32_t)FP_MAX_CHUNKS) return 2u;
	if (forced == 8 || forced == 4 || forced == 2) return forced;
	// 3-D grids that write hit lists: four cell rows per sample make an item of 1024 samples 4096 list elements -- 40 KB of LDS staging, three
	// workgroups per CU.  Items of 512 samples (round 5, L = 16, F = 2, T = 2^19 at 2^18 samples): forward 151-161 -> 117-123 us, the gradient
	// kernels 86-89 -> 92-94 (twice the items per chunk), step 0.340 -> 0.307 ms.  In 2-D the same halving loses (C3a: gradient 55 -> 63 us).
	if (meta.n_pos_dims == 3 && grid_scatter_prefers_lists(meta)) return 2u;
	return 4u;
}

bool grid_planes_supported(const GridMeta& meta, uint32_t n) {
	const uint32_t F = meta.n_features_per_level;
	return (F == 2 || F == 4 || F == 8) && (meta.n_pos_dims == 2 || meta.n_pos_dims == 3) && meta.n_levels < 256 && n % 64 == 0 && n > 0 &&
	       div_round_up(n, FP_THREADS * grid_planes_spt(meta)) < (1u << 24);
}

// Work list: [0..7] = number of items of XCD x, then 8 runs of max_items entries (level << 24 | item index inside the level).
// Levels are laid end to end in order and the sequence is cut into 8 runs of equal estimated cost, so every XCD gathers
// from at most a few consecutive levels (usually two) and each table is pulled into at most two L2s.
void grid_planes_plan(const GridMeta& meta, uint32_t n, bool hit_lists, std::vector<uint32_t>& work, uint32_t& max_items, uint32_t& blocks_per_xcd) {
	const uint32_t threads = hit_lists ? FP_LIST_THREADS : FP_THREADS;
	const uint32_t items_per_level = div_round_up(n, threads * grid_planes_spt(meta));
	// relative cost of one item: tables beyond a few hundred KB miss the per-CU cache on nearly every corner pair
	float coarse_cost = 0.6f;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_FWD_COARSE_COST")) coarse_cost = (float)atof(e); // laboratory knob
#endif
	std::vector<float> cost(meta.n_levels);
	double total = 0;
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		const size_t bytes = (size_t)meta.levels[l].size * meta.n_features_per_level * 2;
		cost[l] = bytes > (256u << 10) ? 1.0f : coarse_cost;
		total += (double)cost[l] * items_per_level;
	}
	// profiling aid: TCNN_AMD_FWD_LEVELS="lo,hi" restricts the work list to levels lo..hi (results are then incomplete!)
	uint32_t dbg_lo = 0, dbg_hi = meta.n_levels;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_FWD_LEVELS")) sscanf(e, "%u,%u", &dbg_lo, &dbg_hi);
#endif
	(void)total;
	// A level whose table does not fit an XCD's L2 anyway (> 4 MB: config 5's 32 MB tables) is NOT given to one XCD: with per-XCD level runs
	// eight such tables are in flight at once -- 256 MB, the whole Infinity Cache, beside the plane stores -- nothing stays resident and
	// every 16-byte pair costs a DRAM burst (3.07 GB read for 537 MB of gathers, measured).  Its items are dealt out to all XCDs in turn
	// instead, behind the levels that do fit: all XCDs then work on the same one or two big tables at a time, which stay in the Infinity Cache.
	size_t wide_bytes = (size_t)4 << 20;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_FWD_WIDE_MB")) wide_bytes = (size_t)std::max(atoi(e), 0) << 20; // 0: never (A/B runs)
#endif
	auto is_wide = [&](uint32_t l) { return wide_bytes > 0 && (size_t)meta.levels[l].size * meta.n_features_per_level * 2 > wide_bytes; };
	double narrow_total = 0;
	for (uint32_t l = 0; l < meta.n_levels; ++l) if (l >= dbg_lo && l <= dbg_hi && !is_wide(l)) narrow_total += (double)cost[l] * items_per_level;
	std::vector<std::vector<uint32_t>> runs(8);
	double acc = 0;
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		if (l < dbg_lo || l > dbg_hi || is_wide(l)) continue;
		for (uint32_t i = 0; i < items_per_level; ++i) {
			const uint32_t bin = std::min<uint32_t>((uint32_t)((acc + 0.5 * cost[l]) * 8.0 / narrow_total), 7u);
			runs[bin].push_back(l << 24 | i);
			acc += cost[l];
		}
	}
	{ // even the runs out before the wide levels follow (they start together)
		size_t longest = 0;
		for (auto& r : runs) longest = std::max(longest, r.size());
		uint32_t next = 0;
		for (uint32_t l = 0; l < meta.n_levels; ++l) {
			if (l < dbg_lo || l > dbg_hi || !is_wide(l)) continue;
			for (uint32_t i = 0; i < items_per_level; ++i) {
				// fill the shortest run first until all are level, then round robin
				size_t best = next % 8;
				for (size_t b = 0; b < 8; ++b) if (runs[b].size() < runs[best].size()) best = b;
				runs[best].push_back(l << 24 | i);
				++next;
			}
		}
		(void)longest;
	}
	max_items = 0;
	for (auto& r : runs) max_items = std::max<uint32_t>(max_items, (uint32_t)r.size());
	work.assign(8 + (size_t)8 * max_items, 0xffffffffu);
	for (uint32_t b = 0; b < 8; ++b) {
		work[b] = (uint32_t)runs[b].size();
		std::copy(runs[b].begin(), runs[b].end(), work.begin() + 8 + (size_t)b * max_items);
	}
	// 16 waves per CU on each XCD's 32 CUs (8 workgroups of 128 threads), fewer when there is less work
	blocks_per_xcd = std::max(1u, std::min(256u * FP_THREADS / threads, max_items));
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_FWD_BLOCKS_PER_XCD")) blocks_per_xcd = std::max(1, atoi(e)); // laboratory knob
#endif
}

void grid_forward_planes(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, const uint32_t* dev_work, uint32_t max_items, uint32_t blocks_per_xcd, uint32_t n,
                         MatView x, const void* grid, void* out_planes, uint64_t* chunk_bits, const MlpPrepJob* prep_job, const GridHitLists* hit_lists) {
	CHECK_THROW(grid_planes_supported(meta, n));
	CHECK_THROW(!hit_lists || (hit_lists->elems && hit_lists->heads && hit_lists->stragglers && hit_lists->counts && n <= grid_hit_max_samples(meta) &&
	                           hit_lists->item_samples == grid_hit_item_samples(meta) && hit_lists->n_items == div_round_up(n, hit_lists->item_samples) &&
	                           hit_lists->sidx && hit_lists->item_capacity >= (hit_lists->item_samples << (meta.n_pos_dims - 1)) && hit_lists->item_capacity % 8 == 0));
	const uint32_t F = meta.n_features_per_level;
#define TCNN_PLANES_F(D, SPT) \
	switch (F) { \
		case 2: return launch_planes<D, 2, SPT>(stream, dev_meta, dev_work, max_items, blocks_per_xcd, n, x, grid, out_planes, chunk_bits, prep_job, hit_lists); \
		case 4: return launch_planes<D, 4, SPT>(stream, dev_meta, dev_work, max_items, blocks_per_xcd, n, x, grid, out_planes, chunk_bits, prep_job, hit_lists); \
		default: return launch_planes<D, 8, SPT>(stream, dev_meta, dev_work, max_items, blocks_per_xcd, n, x, grid, out_planes, chunk_bits, prep_job, hit_lists); \
	}
#define TCNN_PLANES(D) \
	if (grid_planes_spt(meta) == 8) { TCNN_PLANES_F(D, 8) } else if (grid_planes_spt(meta) == 4) { TCNN_PLANES_F(D, 4) } else { TCNN_PLANES_F(D, 2) }
	if (meta.n_pos_dims == 2) { TCNN_PLANES(2) } else { TCNN_PLANES(3) }
#undef TCNN_PLANES
#undef TCNN_PLANES_F
}

// Level planes [level][n][F] (what k_grid_fwd_planes writes) -> rows [n][row_words] of the same halves, as 32-bit words: 64 samples per
// workgroup through LDS, both sides coalesced (a level's 64 x F / 2 words are contiguous in the planes, a tile's 64 rows in the output).
namespace {
constexpr uint32_t PR_THREADS = 256, PR_SAMPLES = 64, PR_MAX_WORDS = 128;
__global__ void __launch_bounds__(PR_THREADS) k_planes_to_rows(const uint32_t* __restrict__ planes, uint32_t* __restrict__ rows, const uint32_t n, const uint32_t words_per_level,
                                                                const uint32_t words, const uint32_t row_words) {
	__shared__ uint32_t tile[PR_SAMPLES * (PR_MAX_WORDS + 1)];
	const uint32_t i0 = blockIdx.x * PR_SAMPLES, pitch = words + 1;
	const uint32_t per_level = PR_SAMPLES * words_per_level;
	for (uint32_t idx = threadIdx.x; idx < PR_SAMPLES * words; idx += PR_THREADS) {
		const uint32_t l = idx / per_level, r = idx - l * per_level;
		const uint32_t i = r / words_per_level, k = r - i * words_per_level;
		tile[i * pitch + l * words_per_level + k] = planes[((size_t)l * n + i0) * words_per_level + r];
	}
	__syncthreads();
	for (uint32_t idx = threadIdx.x; idx < PR_SAMPLES * words; idx += PR_THREADS) {
		const uint32_t i = idx / words, w = idx - i * words;
		rows[(size_t)(i0 + i) * row_words + w] = tile[i * pitch + w];
	}
}
// the other way round: rows [n][row_words] -> level planes (dL/dy of a caller's own network for the list-fed gradient kernel, whose streamed
// tasks read a level's gradients sample after sample)
__global__ void __launch_bounds__(PR_THREADS) k_rows_to_planes(const uint32_t* __restrict__ rows, uint32_t* __restrict__ planes, const uint32_t n, const uint32_t words_per_level,
                                                                const uint32_t words, const uint32_t row_words) {
	__shared__ uint32_t tile[PR_SAMPLES * (PR_MAX_WORDS + 1)];
	const uint32_t i0 = blockIdx.x * PR_SAMPLES, pitch = words + 1;
	const uint32_t per_level = PR_SAMPLES * words_per_level;
	for (uint32_t idx = threadIdx.x; idx < PR_SAMPLES * words; idx += PR_THREADS) {
		const uint32_t i = idx / words, w = idx - i * words;
		tile[i * pitch + w] = rows[(size_t)(i0 + i) * row_words + w];
	}
	__syncthreads();
	for (uint32_t idx = threadIdx.x; idx < PR_SAMPLES * words; idx += PR_THREADS) {
		const uint32_t l = idx / per_level, r = idx - l * per_level;
		const uint32_t i = r / words_per_level, k = r - i * words_per_level;
		planes[((size_t)l * n + i0) * words_per_level + r] = tile[i * pitch + l * words_per_level + k];
	}
}
} // namespace

bool grid_planes_to_rows_supported(const GridMeta& meta, uint32_t n, uint32_t width) {
	const uint32_t F = meta.n_features_per_level;
	return n > 0 && n % PR_SAMPLES == 0 && F % 2 == 0 && width > 0 && width % F == 0 && width / 2 <= PR_MAX_WORDS;
}

void grid_planes_to_rows(hipStream_t stream, const GridMeta& meta, uint32_t n, uint32_t width, const void* planes, void* rows, uint32_t row_stride) {
	CHECK_THROW(grid_planes_to_rows_supported(meta, n, width) && row_stride % 2 == 0 && row_stride >= width);
	hipLaunchKernelGGL(k_planes_to_rows, dim3(n / PR_SAMPLES), dim3(PR_THREADS), 0, stream, (const uint32_t*)planes, (uint32_t*)rows, n, meta.n_features_per_level / 2, width / 2, row_stride / 2);
}

void grid_rows_to_planes(hipStream_t stream, const GridMeta& meta, uint32_t n, uint32_t width, const void* rows, uint32_t row_stride, void* planes) {
	CHECK_THROW(grid_planes_to_rows_supported(meta, n, width) && row_stride % 2 == 0 && row_stride >= width);
	hipLaunchKernelGGL(k_rows_to_planes, dim3(n / PR_SAMPLES), dim3(PR_THREADS), 0, stream, (const uint32_t*)rows, (uint32_t*)planes, n, meta.n_features_per_level / 2, width / 2, row_stride / 2);
}

} // namespace tcnn_amd
