// k_grid_scatter_lists.hip -- dL/dgrid, owner computes (k_grid_scatter.hip), fed by HIT LISTS instead of bit planes.
//
// Replaces (reference, /root/reference): include/tiny-cuda-nn/encodings/grid.h:215-320 (kernel_grid_backward) + the memset of the
// gradient table (grid.h:858), for half-precision grids with F >= 2 inside the fused training step.
//
// What round 3's counters said about k_grid_scatter (profiles/r03c_sq_counters.json): per task of 8192 hits ~9 k clocks went into scanning a
// 32 KB bit plane for the 3 % of samples that hit, the waves waited 57 % of their lives, and 63 % of the LDS cycles were bank conflicts of the
// random ds_add_u64.  This kernel changes the three things those numbers name:
//   * hits arrive as a stream: the forward kernel (k_grid_planes.hip) appends one element per (sample, cell row) to the list of the chunk
//     the row's corners fall into (GridHitLists, tcnn_common.h).  The owner reads its list front to back with coalesced loads -- no scan,
//     no compaction queue -- and an element says which corners of the sample's cell are meant, so the owner hashes the rows named and
//     tests nothing (the filter form located all 2^D corners of a sample and tested each against the chunk);
//   * both features of an entry travel in ONE ds_add_u64 as 2 x int32 (low half sign-extended into the high one; decoded as
//     lo = (int32) s, hi = (s - lo) >> 32): half the LDS atomics and half the LDS footprint.  Exactness is not given up: each task sums
//     |product| over everything it adds; while that sum stays below 2^31 fixed-point units (|value| < 128) no half of any entry can have
//     overflowed and the packed sums ARE the exact sums.  A task whose bound fails throws its accumulators away and runs again with
//     64-bit accumulators, half of its entries at a time -- slower, never different;
//   * 64 KiB of accumulators per workgroup (the same 8192 entries per chunk as before, F = 2): TWO workgroups of 8 waves per CU, so one
//     zeroes or flushes while the other accumulates and two independent instruction streams share the address path.
// Arithmetic per contribution is the reference's: (half) weight * dL_dy in fp16 (grid.h:254), weights as fp32 products in dimension
// order; the sum is exact (integers), rounded to fp16 once -- bit-identical to k_grid_scatter and to the oracle's orc_grid_backward_exact.
#include "grid_fixed.h"
#include "adam_device.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tcnn_amd {
namespace {

constexpr uint32_t SL_ACC_BYTES = 64 * 1024;
constexpr uint32_t SL_THREADS = 512;
constexpr uint32_t SL_WAVES = SL_THREADS / 64;
constexpr uint32_t SL_WINDOW = 1536;                  // elements of a wave's 64 runs looked up through one fill of its owner table
constexpr uint32_t SL_WAVE_LDS = SL_WINDOW + 256;     // per wave: owner table (one byte per element: the lane whose run holds it), the runs' positions
constexpr uint32_t SL_LDS_BYTES = SL_ACC_BYTES + 256 + SL_WAVES * SL_WAVE_LDS; // accumulators, the waves' bound sums and the verdict, the waves' tables
static_assert(2 * SL_LDS_BYTES <= 160 * 1024, "two workgroups per CU");
#ifndef TCNN_SL_SB
#define TCNN_SL_SB 2
#endif
constexpr int SL_SB = TCNN_SL_SB;                     // 64-element batches whose gathers are issued together
#ifndef TCNN_SL_LEAD
#define TCNN_SL_LEAD 2
#endif
constexpr int SL_LEAD = TCNN_SL_LEAD;                 // batches gathered ahead of the one being added (register sets: LEAD + 2)
// the packed sums are trusted while sum |product| (+ the largest initial value in Accumulate mode) stays below this; the exact limit is 128
// (2^31 units of 2^-24); the margin covers the rounding of the fp32 running sums (<= 2^17 additions per thread at 6e-8 each)
constexpr float SL_PACKED_BOUND = 120.0f;

typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) float lds_f32; // (explicit: a generic pointer to LDS selected against a global one does not compile on gfx950)

template <int D, int F, bool REC>
struct ScatterCtx {
	typedef typename VecOf<half_t, F>::type vecF;
	GridLevel lv;
	uint32_t primes[D];
	uint32_t hash_type, interpolation;
	uint32_t level, n;
	const uint4* recs;          // REC: record plane of this level (pair)
	MatView x;                  // !REC
	const half_t* dy;           // !REC: dL_dy + level * dy_stride_level
	uint32_t dy_stride_sample;
};

template <int D, int F, bool REC>
__device__ inline void sl_fetch(const ScatterCtx<D, F, REC>& c, const uint32_t i, float (&xin)[D], typename ScatterCtx<D, F, REC>::vecF& gv) {
	typedef typename ScatterCtx<D, F, REC>::vecF vecF;
	constexpr bool PAIRED = REC && D == 2 && F == 2;
	if constexpr (REC) {
		const uint4 r = c.recs[i];
		const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
		for (int d = 0; d < D; ++d) xin[d] = __builtin_bit_cast(float, w[d]);
		if constexpr (PAIRED) {
			gv = __builtin_bit_cast(vecF, (c.level & 1u) ? w[3] : w[2]);
		} else if constexpr (F == 2) {
			gv = __builtin_bit_cast(vecF, w[D]);
		} else {
			typedef uint32_t u2 __attribute__((ext_vector_type(2)));
			gv = __builtin_bit_cast(vecF, (u2{w[2], w[3]}));
		}
	} else {
		load_coords<D>(c.x, i, xin);
		gv = *(const vecF*)&c.dy[(size_t)i * c.dy_stride_sample];
	}
}

// The cell of one sample: positions and weights' factors, once per element (grid.h:147-160, common_device.h:856-868)
template <int D>
struct CellPos {
	uint32_t cell[D];
	float pos[D];
};
template <int D, int F, bool REC>
__device__ inline CellPos<D> sl_cell(const ScatterCtx<D, F, REC>& c, const float (&xin)[D]) {
	CellPos<D> p;
	float unused;
#pragma unroll
	for (int d = 0; d < D; ++d) p.cell[d] = pos_fract(xin[d], c.lv.scale, c.interpolation, &p.pos[d], &unused);
	return p;
}

// One cell row of one sample: corners A (cell_0) and B (cell_0 + 1) of row `row` (bit d - 1: cell_d + 1; per lane), each added if its flag
// is set and its entry lies in [e0, e0 + n_sub).  PACKED: accumulators are uint64 [entry][F / 2] holding two int32 sums; else int64 [entry][F].
template <int D, int F, bool REC, bool PACKED>
__device__ inline void sl_add_row(const ScatterCtx<D, F, REC>& c, lds_u64* acc, const uint32_t e0, const uint32_t n_sub, const CellPos<D>& p, const typename ScatterCtx<D, F, REC>::vecF& gv,
                                  const uint32_t row, const bool want_a, const bool want_b, float& bound) {
	const bool nearest = c.interpolation == (uint32_t)InterpolationType::Nearest;
	uint32_t local[D];
	local[0] = p.cell[0];
	// weights: the reference multiplies in dimension order starting from 1 (grid.h:147-160); 1 * w is exact, so start from dimension 0's factor
	float wa = nearest ? 1.0f : 1 - p.pos[0], wb = p.pos[0];
#pragma unroll
	for (int d = 1; d < D; ++d) {
		const bool up = (row >> (d - 1)) & 1u;
		local[d] = p.cell[d] + (up ? 1u : 0u);
		const float wd = up ? p.pos[d] : 1 - p.pos[d];
		if (!nearest) { wa *= wd; wb *= wd; }
	}
	const uint32_t ia = level_index<D, false>(c.lv, c.primes, c.hash_type, local) - e0;
	local[0] += 1;
	const uint32_t ib = level_index<D, false>(c.lv, c.primes, c.hash_type, local) - e0;
	asm volatile("" : "+v"(wa), "+v"(wb)); // keep the fp32 rounding of the weight products (see k_grid_fwd)
	auto add = [&](const uint32_t index, const float weight) {
		const half_t w = (half_t)weight;
		if constexpr (PACKED) {
#pragma unroll
			for (int j = 0; j < F / 2; ++j) {
				const float p0 = (float)(half_t)(w * gv[2 * j]), p1 = (float)(half_t)(w * gv[2 * j + 1]); // (GRAD_T)weight * grad in fp16, grid.h:254
				bound += __builtin_fabsf(p0);
				bound += __builtin_fabsf(p1);
				const int f0 = (int)(p0 * 16777216.0f), f1 = (int)(p1 * 16777216.0f); // exact below 128; beyond it the bound has failed anyway
				const uint32_t lo = (uint32_t)f0, hi = (uint32_t)(f1 + (f0 >> 31));     // f0 + f1 2^32 as one 64-bit integer
				__hip_atomic_fetch_add(acc + index * (F / 2) + j, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else {
#pragma unroll
			for (int f = 0; f < F; ++f) {
				const half_t pr = w * gv[f];
				__hip_atomic_fetch_add(acc + index * F + f, (unsigned long long)half_to_fixed_fast(pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
	};
	if (want_a && ia < n_sub) add(ia, wa);
	if (want_b && ib < n_sub) add(ib, wb);
}

// All corners of an element's mask: the rows named in it, the lowest first, each lane its own.  Two rounds without a loop (around a loop the
// register allocator splits the live ranges of the records still in flight, and every copy of such a register is a wait for its load): a
// hashed level's element holds one row (the rows of a cell fall into different chunks), so the second round -- both rows of a cell in
// this chunk: dense levels -- is skipped by the whole wave; 3-D cells have four rows and take two more.
template <int D, int F, bool REC, bool PACKED>
__device__ inline void sl_add_masked(const ScatterCtx<D, F, REC>& c, lds_u64* acc, const uint32_t e0, const uint32_t n_sub, const float (&xin)[D], const typename ScatterCtx<D, F, REC>::vecF& gv,
                                     uint32_t m, float& bound) {
	constexpr int R = 1 << (D - 1);
	const CellPos<D> p = sl_cell<D, F, REC>(c, xin);
#pragma unroll
	for (int round = 0; round < R; ++round) {
		if (round > 0 && __builtin_amdgcn_ballot_w64(m != 0) == 0) break; // (unrolled: a forward branch)
		if (m != 0) {
			const uint32_t row = (uint32_t)__builtin_ctz(m) >> 1;
			const uint32_t two = (m >> (2 * row)) & 3u;
			m &= ~(3u << (2 * row));
			sl_add_row<D, F, REC, PACKED>(c, acc, e0, n_sub, p, gv, row, (two & 1u) != 0, (two & 2u) != 0, bound);
		}
	}
}

// Where a listed task's elements are: the items' runs of its chunk (GridHitLists, tcnn_common.h)
struct ScatterRuns {
	const uint32_t* elems;      // this level's pool: [n_items][item_capacity]
	const uint32_t* heads;      // this level's offsets: [n_items][GRID_HIT_HEADS], already advanced to the task's chunk
	const uint32_t* stragglers; // this level's {element, chunk} pairs
	uint32_t n_stragglers, n_items, item_capacity, chunk;
};

// The accumulation pass of one task: over the runs of its chunk in the items [begin, end) of a level cut into chunks, or over the samples
// [begin, end) of a level that is one chunk.
template <int D, int F, bool REC, bool PACKED>
__device__ inline float sl_pass(const ScatterCtx<D, F, REC>& c, lds_u64* acc, const uint32_t e0, const uint32_t n_sub, const bool listed, const ScatterRuns& runs, const bool with_stragglers,
                                const uint32_t begin, const uint32_t end, const uint32_t wave, const uint32_t lane, char* wave_lds, const uint32_t first_h0, const uint32_t first_h1) {
	typedef typename ScatterCtx<D, F, REC>::vecF vecF;
	float bound = 0;
	const bool nearest = c.interpolation == (uint32_t)InterpolationType::Nearest;
	constexpr uint32_t R = 1u << (D - 1);
	constexpr uint32_t HIT_SHIFT = grid_hit_mask_shift(D), HIT_ID_MASK = (1u << HIT_SHIFT) - 1u;
	if (!listed) {
		const uint32_t n_el = end - begin;
		const uint32_t per_wave = ((n_el + SL_WAVES - 1) / SL_WAVES + 63) / 64 * 64;
		const uint32_t w_begin = min(begin + wave * per_wave, end), w_end = min(end, w_begin + per_wave);
		if (w_begin >= w_end) return bound; // wave-uniform
		// the chunk is the whole level: every sample, every row, both corners; one batch in flight while the previous one is added
		float px[D], qx[D];
		vecF pg, qg;
		bool p_valid = false;
		auto whole = [&](const float (&xin)[D], const vecF& gv) {
			const CellPos<D> p = sl_cell<D, F, REC>(c, xin);
#pragma unroll
			for (uint32_t row = 0; row < R; ++row) {
				if (row > 0 && nearest) break; // grid.h:232-246: the cell's own entry, weight 1
				sl_add_row<D, F, REC, PACKED>(c, acc, e0, n_sub, p, gv, row, true, !nearest, bound);
			}
		};
		for (uint32_t i0 = w_begin; i0 < w_end; i0 += 64) {
			const uint32_t i = i0 + lane;
			sl_fetch<D, F, REC>(c, min(i, w_end - 1), qx, qg);
			if (p_valid) whole(px, pg);
#pragma unroll
			for (int d = 0; d < D; ++d) px[d] = qx[d];
			pg = qg;
			p_valid = i < w_end;
		}
		if (p_valid) whole(px, pg);
		return bound;
	}
	// Listed.  A wave takes 64 items at a time, one per lane: the lane reads where its item's run of this chunk starts and ends, a scan
	// over the lanes numbers the runs' elements 0 .. T - 1, and every lane writes its own number into a wave-private byte table at its
	// elements' positions (~16 each) -- element e is then TWO LDS reads away: table[e] names the lane, positions[lane] where its run
	// lies.  (First form: a 6-step search over the lanes' running totals with ds_bpermute per element -- 7 cross-lane reads and ~20 vector
	// instructions per 64 elements, and the kernel is bound by its vector instructions: 28 us per task against 21 with plain lists.)
	//
	// Then a software pipeline over batches of 64 SL_SB elements, NS = SL_LEAD + 2 register sets that rotate STATICALLY (the loop is
	// unrolled NS times; a set is never copied -- a move of a register whose load is still in flight is a wait for it): per step the
	// elements of batch j + LEAD + 1 are requested, the records of batch j + LEAD gathered by the elements that have arrived, and batch j
	// is added.  Loads return in order, so the compiler's counted waits leave everything younger in flight.  All loads are raw buffer
	// loads without a branch around them: a lane past the last element reads from beyond the descriptor's range (-> zero: no corner
	// wanted), and an element without corners gathers from beyond the record plane (-> zeros, nothing requested).  (With `if (element)
	// load` the compiler waited with vmcnt(0) right behind every gather -- the loaded value has to be merged with the default at the end
	// of the branch -- and nothing was ever in flight.)
	constexpr int NS = SL_LEAD + 2;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	const auto rs_pool = __builtin_amdgcn_make_buffer_rsrc((void*)runs.elems, 0, (int)(runs.n_items * runs.item_capacity * 4u), 0x00020000);
	const auto rs_rec = __builtin_amdgcn_make_buffer_rsrc((void*)c.recs, 0, (int)(c.n * 16u), 0x00020000);
	constexpr uint32_t BATCH = 64 * SL_SB;
	const uint32_t n_groups = (end - begin + 63) / 64;
	for (uint32_t g = wave; g < n_groups; g += SL_WAVES) { // wave-uniform
		const uint32_t item0 = begin + g * 64, item = item0 + lane;
		uint32_t h0 = first_h0, h1 = first_h1; // the wave's first group: requested when the task began, under the zeroing of the accumulators
		if (g != wave) {
			const uint32_t* hd = runs.heads + (size_t)min(item, end - 1) * GRID_HIT_HEADS;
			h0 = hd[0];
			h1 = hd[1];
		}
		const uint32_t cnt = item < end ? h1 - h0 : 0u;
		uint32_t incl = cnt;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t up = __shfl_up(incl, o);
			if (lane >= (uint32_t)o) incl += up;
		}
		const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
		uint8_t* owner = (uint8_t*)wave_lds;                 // [SL_WINDOW]
		uint32_t* run_pos = (uint32_t*)(wave_lds + SL_WINDOW); // [64]: element e of lane l's run sits at run_pos[l] + e in the level's pool
		run_pos[lane] = (item * runs.item_capacity + h0) - (incl - cnt);
		for (uint32_t w0 = 0; w0 < total; w0 += SL_WINDOW) { // (one window unless the runs are far longer than the ~16 elements of a uniform batch)
		const uint32_t w1 = min(total, w0 + SL_WINDOW);
		for (uint32_t j = max(incl - cnt, w0); j < min(incl, w1); ++j) owner[j - w0] = (uint8_t)lane;
		uint32_t el[NS][SL_SB];
		u32x4 rec[NS][SL_SB];    // REC: the record
		float xs[NS][SL_SB][D];  // else: coordinates and gradient by loads of their own (an element without corners reads sample 0's)
		vecF gs[NS][SL_SB];
		auto load_elems = [&](const int set, const uint32_t b0) {
#pragma unroll
			for (int s = 0; s < SL_SB; ++s) {
				const uint32_t e = b0 + s * 64 + lane;
				const bool have = e < w1;
				const uint32_t at = run_pos[owner[have ? e - w0 : 0u]] + e;
				el[set][s] = __builtin_amdgcn_raw_buffer_load_b32(rs_pool, have ? at * 4u : 0xfffffffcu, 0, 0);
			}
		};
		auto gather = [&](const int set) {
#pragma unroll
			for (int s = 0; s < SL_SB; ++s) {
				const uint32_t e = el[set][s];
				if constexpr (REC) rec[set][s] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, (e >> HIT_SHIFT) ? (e & HIT_ID_MASK) * 16u : 0xfffffff0u, 0, 0);
				else sl_fetch<D, F, REC>(c, (e >> HIT_SHIFT) ? (e & HIT_ID_MASK) : 0u, xs[set][s], gs[set][s]);
			}
		};
		auto accumulate = [&](const int set) {
#pragma unroll
			for (int s = 0; s < SL_SB; ++s) {
				float xin[D];
				vecF gv;
				if constexpr (REC) {
					const u32x4 rv = rec[set][s];
					const uint32_t r[4] = {rv.x, rv.y, rv.z, rv.w}; // (never r[d] on the vector itself with a loop variable: the compiler read r[0] for every d)
#pragma unroll
					for (int d = 0; d < D; ++d) xin[d] = __builtin_bit_cast(float, r[d]);
					constexpr bool PAIRED = D == 2 && F == 2;
					if constexpr (PAIRED) gv = __builtin_bit_cast(vecF, (c.level & 1u) ? r[3] : r[2]);
					else if constexpr (F == 2) gv = __builtin_bit_cast(vecF, r[D]);
					else gv = __builtin_bit_cast(vecF, (u32x2{r[2], r[3]}));
				} else {
#pragma unroll
					for (int d = 0; d < D; ++d) xin[d] = xs[set][s][d];
					gv = gs[set][s];
				}
				sl_add_masked<D, F, REC, PACKED>(c, acc, e0, n_sub, xin, gv, el[set][s] >> HIT_SHIFT, bound);
			}
		};
#pragma unroll
		for (int b = 0; b <= SL_LEAD; ++b) load_elems(b, w0 + b * BATCH);
#pragma unroll
		for (int b = 0; b < SL_LEAD; ++b) gather(b);
		const uint32_t quarter = max((w1 - w0) / 4u, 1u);
		for (uint32_t b0 = w0; b0 < w1; b0 += NS * BATCH) {
			// The two workgroups of a CU share its address path, and the arbiter serves the older wave first: left alone the older workgroup
			// runs as if it had the CU to itself (a task in 14 us) and the younger one takes the rest (30 us).  Priority by progress
			// instead: whoever is further from the end of its run goes first.
			{
				const uint32_t done = (b0 - w0) / quarter; // wave-uniform
				if (done == 0) __builtin_amdgcn_s_setprio(3);
				else if (done == 1) __builtin_amdgcn_s_setprio(2);
				else if (done == 2) __builtin_amdgcn_s_setprio(1);
				else __builtin_amdgcn_s_setprio(0);
			}
#pragma unroll
			for (int u = 0; u < NS; ++u) { // batch j = (b0 - w0) / BATCH + u lives in set u (j is a multiple of NS at u = 0)
				load_elems((u + SL_LEAD + 1) % NS, b0 + (u + SL_LEAD + 1) * BATCH);
				gather((u + SL_LEAD) % NS);
				accumulate(u);
			}
		}
		__builtin_amdgcn_s_setprio(0);
		} // window
	}
	// the level's stragglers (second corners of rows that straddle two chunks: one row in ~8000): every task of the level looks through all of them
	if (with_stragglers) {
		for (uint32_t p = wave * 64 + lane; p < runs.n_stragglers; p += SL_THREADS) {
			const uint32_t e = runs.stragglers[2 * (size_t)p], ch = runs.stragglers[2 * (size_t)p + 1];
			if (ch == runs.chunk) {
				float xin[D];
				vecF gv;
				sl_fetch<D, F, REC>(c, e & HIT_ID_MASK, xin, gv);
				const CellPos<D> cp = sl_cell<D, F, REC>(c, xin);
				uint32_t m = e >> HIT_SHIFT;
				while (m != 0) {
					const uint32_t row = (uint32_t)__builtin_ctz(m) >> 1;
					const uint32_t two = (m >> (2 * row)) & 3u;
					m &= ~(3u << (2 * row));
					sl_add_row<D, F, REC, PACKED>(c, acc, e0, n_sub, cp, gv, row, (two & 1u) != 0, (two & 2u) != 0, bound);
				}
			}
		}
	}
	return bound;
}

struct ScatterListsArgs {
	const GridMeta* meta;
	const GridScatterTask* tasks;
	uint32_t n;
	MatView x;
	const half_t* dL_dy;
	uint32_t dy_stride_sample, dy_stride_level;
	half_t* grad;
	GridHitLists lists;
	unsigned long long* scratch;
	int accumulate_mode, force_wide;
	unsigned long long* dbg_times;
	uint32_t* fallback_count;
	AdamInFlush adam; // w_fp != nullptr: the single owner of a chunk applies the optimizer's update to it as it flushes (sl_flush_adam)
};

// The single owner of a chunk has the chunk's FINAL gradient in LDS when it flushes: adam.h:48-119 runs on it at once -- the same adam_one
// on the same half gradient as k_adam afterwards, so the same bits (tests: test_adam_in_the_scatter_flush_is_bit_identical) -- and the 28 bytes
// of optimizer state per parameter stream from and to HBM while the CU's other workgroup gathers from its L2: the two kernels' bounds are
// different resources.  (Round 2 built this into k_grid_scatter, one 16-wave workgroup per CU, and measured it equal to the two launches: a
// lone workgroup's accumulators sit idle while it streams.  Two workgroups per CU are what makes it pay.)
// grad_quad(q) -> the four half gradients of parameters 4 q .. 4 q + 3 of the chunk; p0: the chunk's first parameter; window: LDS floats.
template <typename GradQuad>
__device__ inline void sl_flush_adam(const AdamInFlush& adam, const size_t p0, const uint32_t n_quads, half_t* __restrict__ g, lds_f32* window, const uint32_t window_floats, const uint32_t tid, GradQuad&& grad_quad) {
	typedef _Float16 h4 __attribute__((ext_vector_type(4)));
	float* __restrict__ wf_p = adam.w_fp + p0;
	float* __restrict__ m1_p = adam.m1 + p0;
	float* __restrict__ m2_p = adam.m2 + p0;
	uint32_t* __restrict__ st_p = (uint32_t*)adam.steps + p0; // uint16 counts (adam.steps16): addressed through st16_p
	uint16_t* __restrict__ st16_p = (uint16_t*)adam.steps + p0;
	half_t* __restrict__ wh_p = (half_t*)adam.w_half + p0;
	// the most recent steps of the debiasing table -- all that parameters touched in the last few thousand steps ask for -- in LDS
	const uint32_t common = adam.args.common_step;
	const uint32_t window_base = common + 1 > window_floats ? common + 1 - window_floats : 0; // window[i] = table[window_base + i], up to table[common]
	for (uint32_t i = tid; i < window_floats; i += SL_THREADS) if (window_base + i <= common) window[i] = adam.debias_table[window_base + i];
	__syncthreads();
	const float debias = window[common - window_base];
	const auto debias_of = [&](const uint32_t t) { return t >= window_base ? window[t - window_base] : adam.debias_table[t]; };
	constexpr int Q = 4; // quads per thread in flight
	for (uint32_t q0 = tid; q0 < n_quads; q0 += Q * SL_THREADS) {
		h4 gq[Q], old[Q];
		bool live[Q], has_old[Q];
		float4 wf[Q], a1[Q], a2[Q];
		uint4 st[Q];
#pragma unroll
		for (int k = 0; k < Q; ++k) {
			const uint32_t q = q0 + k * SL_THREADS;
			live[k] = q < n_quads;
			has_old[k] = false;
			if (live[k]) {
				gq[k] = grad_quad(q);
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
			const size_t i4 = 4 * (size_t)(q0 + k * SL_THREADS);
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
			// parameters that were not updated keep their half value, whatever it is: quads with a zero gradient somewhere brought their old
			// halves along, so that the quad is stored whole
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
}

template <int D, int F, bool REC>
__device__ inline void sl_run_task(const ScatterListsArgs& a, const uint32_t task_index, char* smem) {
	const ScatterListsArgs& a_ = a; // (a local `a` below shadows the argument block)
	const GridMeta* __restrict__ meta = a.meta;
	const uint32_t n = a.n;
	const MatView x = a.x;
	const half_t* __restrict__ dL_dy = a.dL_dy;
	const uint32_t dy_stride_sample = a.dy_stride_sample, dy_stride_level = a.dy_stride_level;
	half_t* __restrict__ grad = a.grad;
	unsigned long long* __restrict__ scratch = a.scratch;
	const int accumulate_mode = a.accumulate_mode, force_wide = a.force_wide;
	unsigned long long* __restrict__ dbg_times = a.dbg_times ? a.dbg_times + (size_t)task_index * 8 : nullptr; // this task's 8 time stamps
	uint32_t* __restrict__ fallback_count = a.fallback_count;
	lds_u64* acc = (lds_u64*)smem;
	float* wave_bound = (float*)(smem + SL_ACC_BYTES); // [2][SL_WAVES]: sums, initial maxima; then the verdict
	uint32_t* verdict = (uint32_t*)(smem + SL_ACC_BYTES + 128);
	const GridScatterTask task = a.tasks[task_index];
	if (task.n_entries == 0) return;
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	if (dbg_times && tid == 0) {
		dbg_times[0] = __builtin_amdgcn_s_memrealtime();
		dbg_times[4] = __builtin_readcyclecounter(); // shader clocks beside the 100 MHz clock: the task's clock rate (laboratory builds print it)
	}

	ScatterCtx<D, F, REC> c;
	c.lv = meta->levels[task.level];
#pragma unroll
	for (int d = 0; d < D; ++d) c.primes[d] = meta->primes[d];
	c.hash_type = meta->hash_type;
	c.interpolation = meta->interpolation;
	c.level = task.level;
	c.n = n;
	constexpr bool PAIRED = REC && D == 2 && F == 2;
	c.recs = (const uint4*)dL_dy + (size_t)(PAIRED ? task.level / 2 : task.level) * n;
	c.x = x;
	c.dy = dL_dy + (size_t)task.level * dy_stride_level;
	c.dy_stride_sample = dy_stride_sample;

	half_t* __restrict__ g = grad + ((size_t)c.lv.offset + task.entry_begin) * F;
	const uint32_t n_vals = task.n_entries * F;
	const bool sole = !task.flush_atomic;
	const bool init_from_grad = accumulate_mode && sole;

	// this task's share of the work: a range of items whose runs of its chunk it walks, or a range of samples where the level is one chunk
	const bool listed = c.lv.scatter_n_chunks > 1;
	uint32_t begin = task.sample_begin, end = task.sample_end;
	ScatterRuns runs{};
	bool with_stragglers = false;
	if (listed) {
		const GridHitLists& hl = a.lists;
		const uint32_t s = task.pad & 0xffffu, n_splits = max(task.pad >> 16, 1u);
		const uint32_t per = (hl.n_items + n_splits - 1) / n_splits;
		begin = min(s * per, hl.n_items);
		end = min(begin + per, hl.n_items);
		runs.chunk = scatter_chunk(c.lv, task.entry_begin);
		runs.elems = hl.elems + (size_t)task.level * hl.n_items * hl.item_capacity;
		runs.heads = hl.heads + (size_t)task.level * hl.n_items * GRID_HIT_HEADS + runs.chunk;
		runs.stragglers = hl.stragglers + (size_t)task.level * hl.straggler_capacity * 2;
		runs.n_stragglers = min(hl.counts[task.level * GRID_HIT_COUNT_STRIDE], hl.straggler_capacity);
		runs.n_items = hl.n_items;
		runs.item_capacity = hl.item_capacity;
		with_stragglers = s == 0;
	}

	uint32_t first_h0 = 0, first_h1 = 0; // where this lane's item (of the wave's first 64) keeps its run of the chunk
	if (listed && begin < end) {
		const uint32_t* hd = runs.heads + (size_t)min(begin + wave * 64 + lane, end - 1) * GRID_HIT_HEADS;
		first_h0 = hd[0];
		first_h1 = hd[1];
	}
	typedef uint32_t u4 __attribute__((ext_vector_type(4)));
	auto zero_acc = [&](const uint32_t bytes) {
		u4* a4 = (u4*)smem;
		for (uint32_t i = tid; i < (bytes + 15) / 16; i += SL_THREADS) a4[i] = u4{0, 0, 0, 0};
	};

	// A chunk small enough for 64-bit accumulators in one pass takes them at once: nothing to verify, nothing to repeat.  These are the
	// coarse levels, where thousands of samples add into each entry and the task-wide bound below -- sum |product| over EVERYTHING the
	// task adds -- says little about a single entry's sum.
	const uint32_t wide_parts = (n_vals * 8 + SL_ACC_BYTES - 1) / SL_ACC_BYTES;
	// ---------------------------------------------------------------------------------------------------- packed pass
	bool packed_ok = !force_wide && wide_parts > 1;
	if (packed_ok) {
		float init_max = 0;
		if (init_from_grad) { // GradientMode::Accumulate, single owner: start from the existing gradient
			for (uint32_t i = tid; i < n_vals / 2; i += SL_THREADS) {
				const float g0 = (float)g[2 * i], g1 = (float)g[2 * i + 1];
				init_max = fmaxf(init_max, fmaxf(__builtin_fabsf(g0), __builtin_fabsf(g1)));
				const int f0 = (int)(g0 * 16777216.0f), f1 = (int)(g1 * 16777216.0f);
				((unsigned long long*)smem)[i] = (unsigned long long)(uint32_t)f0 | ((unsigned long long)(uint32_t)(f1 + (f0 >> 31)) << 32);
			}
			if (!(init_max < 128.0f)) init_max = 1e30f; // also catches NaN / inf
		} else {
			zero_acc(n_vals * 4);
		}
		__syncthreads();
		if (dbg_times && tid == 0) dbg_times[1] = __builtin_amdgcn_s_memrealtime();
		float bound = sl_pass<D, F, REC, true>(c, acc, task.entry_begin, task.n_entries, listed, runs, with_stragglers, begin, end, wave, lane, smem + SL_ACC_BYTES + 256 + wave * SL_WAVE_LDS, first_h0, first_h1);
		// workgroup verdict: (sum over everything added) + (largest initial value) bounds every entry's sum
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			bound += __shfl_xor(bound, o);
			init_max = fmaxf(init_max, __shfl_xor(init_max, o));
		}
		if (lane == 0) { wave_bound[wave] = bound; wave_bound[SL_WAVES + wave] = init_max; }
		__syncthreads();
		if (tid == 0) {
			float t = 0, m = 0;
			for (uint32_t w = 0; w < SL_WAVES; ++w) { t += wave_bound[w]; m = fmaxf(m, wave_bound[SL_WAVES + w]); }
			*verdict = t + m < SL_PACKED_BOUND ? 1u : 0u; // NaN compares false: wide passes
		}
		__syncthreads();
		packed_ok = *verdict != 0;
		if (dbg_times && tid == 0) dbg_times[2] = __builtin_amdgcn_s_memrealtime();
	}

	if (packed_ok) {
		const unsigned long long* a = (const unsigned long long*)smem;
		auto lo_of = [](const unsigned long long s) { return (long long)(int)(uint32_t)s; };
		auto hi_of = [](const unsigned long long s) { return (long long)((long long)(s - (unsigned long long)(long long)(int)(uint32_t)s) >> 32); };
		if (task.flush_atomic) {
			// several workgroups share this chunk: merge the exact partial sums with 64-bit integer atomics; rounded by the finalize pass
			unsigned long long* sc = scratch + (size_t)task.scratch_begin;
			for (uint32_t i = tid; i < n_vals / 2; i += SL_THREADS) {
				const unsigned long long s = a[i];
				const long long v0 = lo_of(s), v1 = hi_of(s);
				if (v0 != 0) atomicAdd(sc + 2 * i, (unsigned long long)v0);
				if (v1 != 0) atomicAdd(sc + 2 * i + 1, (unsigned long long)v1);
			}
		} else {
			// both sums fit 32 bits (that is what the bound proved): below 2^24 in magnitude (|value| < 1, nearly always) the integer is
			// exact as a float, the scaling is exact, and the hardware's float -> half conversion is the ONE rounding (RNE); else the general path
			typedef _Float16 h2 __attribute__((ext_vector_type(2)));
			auto round32 = [](const int v) -> half_t {
				if (__builtin_expect((uint32_t)(v + (1 << 24)) < (1u << 25), 1)) return (half_t)((float)v * 5.9604644775390625e-08f);
				return fixed_to_half((long long)v);
			};
			auto pair_of = [&](const unsigned long long s) -> uint32_t {
				const int lo = (int)(uint32_t)s, hi = (int)(uint32_t)((s - (unsigned long long)(long long)lo) >> 32);
				return __builtin_bit_cast(uint32_t, (h2{round32(lo), round32(hi)}));
			};
			if (a_.adam.w_fp) { // sole owner, optimizer step included (the host checked that the chunk is a whole number of aligned quads of parameters)
				typedef _Float16 h4 __attribute__((ext_vector_type(4)));
				sl_flush_adam(a_.adam, ((size_t)c.lv.offset + task.entry_begin) * F, n_vals / 4, g, (lds_f32*)(smem + SL_ACC_BYTES + 256), SL_WAVES * SL_WAVE_LDS / 4, tid, [&](const uint32_t q) {
					const unsigned long long s0 = a[2 * q], s1 = a[2 * q + 1];
					const int l0 = (int)(uint32_t)s0, h0 = (int)(uint32_t)((s0 - (unsigned long long)(long long)l0) >> 32);
					const int l1 = (int)(uint32_t)s1, h1 = (int)(uint32_t)((s1 - (unsigned long long)(long long)l1) >> 32);
					return h4{round32(l0), round32(h0), round32(l1), round32(h1)};
				});
			} else {
			// four accumulators = eight halves = one 16-byte store per lane where the chunk allows (its first value 16-byte aligned in the table: a
			// quarter of the store instructions, which queue behind the other workgroup's gathers in the CU's address path)
			const uint32_t n_pairs = n_vals / 2;
			const bool quads = ((uintptr_t)g & 15u) == 0;
			const uint32_t n_quads = quads ? n_pairs / 4 : 0;
			for (uint32_t q = tid; q < n_quads; q += SL_THREADS) {
				typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
				const u64x2 s01 = *(const u64x2*)(a + 4 * q), s23 = *(const u64x2*)(a + 4 * q + 2);
				((u4*)g)[q] = u4{pair_of(s01.x), pair_of(s01.y), pair_of(s23.x), pair_of(s23.y)};
			}
			for (uint32_t i = 4 * n_quads + tid; i < n_pairs; i += SL_THREADS) ((uint32_t*)g)[i] = pair_of(a[i]);
			}
		}
	} else {
		// ------------------------------------------------------------------------------------------------ wide passes: 64-bit accumulators, half of the entries at a time
		if (fallback_count && tid == 0 && !force_wide && wide_parts > 1) atomicAdd(fallback_count, 1u);
		long long* acc64 = (long long*)smem;
		const uint32_t half_entries = (task.n_entries + wide_parts - 1) / wide_parts;
		for (uint32_t part = 0; part < wide_parts; ++part) {
			const uint32_t sub_lo = part * half_entries;
			if (sub_lo >= task.n_entries) break;
			const uint32_t n_sub = min(half_entries, task.n_entries - sub_lo);
			const uint32_t sub_vals = n_sub * F;
			half_t* __restrict__ gs = g + (size_t)sub_lo * F;
			__syncthreads(); // the previous part's flush has read the accumulators
			if (init_from_grad) {
				for (uint32_t i = tid; i < sub_vals; i += SL_THREADS) acc64[i] = half_to_fixed(gs[i]);
			} else {
				zero_acc(sub_vals * 8);
			}
			__syncthreads();
			(void)sl_pass<D, F, REC, false>(c, acc, task.entry_begin + sub_lo, n_sub, listed, runs, with_stragglers, begin, end, wave, lane, smem + SL_ACC_BYTES + 256 + wave * SL_WAVE_LDS, first_h0, first_h1);
			__syncthreads();
			if (task.flush_atomic) {
				unsigned long long* sc = scratch + (size_t)task.scratch_begin + (size_t)sub_lo * F;
				for (uint32_t i = tid; i < sub_vals; i += SL_THREADS) {
					const long long v = acc64[i];
					if (v != 0) atomicAdd(sc + i, (unsigned long long)v);
				}
			} else if (a_.adam.w_fp) {
				typedef _Float16 h4 __attribute__((ext_vector_type(4)));
				// (the tables behind the accumulators are dead between the pass and the next part's: the barrier inside comes after every wave's pass)
				__syncthreads();
				sl_flush_adam(a_.adam, ((size_t)c.lv.offset + task.entry_begin + sub_lo) * F, sub_vals / 4, gs, (lds_f32*)(smem + SL_ACC_BYTES + 256), SL_WAVES * SL_WAVE_LDS / 4, tid, [&](const uint32_t q) {
					return h4{fixed_to_half_fast(acc64[4 * q]), fixed_to_half_fast(acc64[4 * q + 1]), fixed_to_half_fast(acc64[4 * q + 2]), fixed_to_half_fast(acc64[4 * q + 3])};
				});
			} else {
				typedef _Float16 h2 __attribute__((ext_vector_type(2)));
				for (uint32_t i = tid; i < sub_vals / 2; i += SL_THREADS) ((h2*)gs)[i] = h2{fixed_to_half_fast(acc64[2 * i]), fixed_to_half_fast(acc64[2 * i + 1])};
			}
		}
		if (dbg_times && tid == 0 && (force_wide || wide_parts == 1)) dbg_times[1] = dbg_times[2] = __builtin_amdgcn_s_memrealtime();
	}
	if (dbg_times) {
		__syncthreads();
		if (tid == 0) {
			dbg_times[3] = __builtin_amdgcn_s_memrealtime();
			dbg_times[5] = __builtin_readcyclecounter();
		}
	}
}

// One workgroup per task, in the order of the plan (grid_scatter_lists_plan): blocks are handed to the XCDs round robin, block b to XCD
// b % 8, so the plan's eight per-XCD task lists are interleaved.  (Persistent workgroups pulling from per-XCD queues were built and
// measured: 71 - 78 us against 60 -- a workgroup that stays resident keeps its place in the oldest-first arbitration of its CU for the
// whole launch, the pops cost a global round trip per task, and stealing across XCDs runs a task away from its record plane at half speed.)
template <int D, int F, bool REC>
__global__ void __launch_bounds__(SL_THREADS, 4) k_grid_scatter_lists(const ScatterListsArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	sl_run_task<D, F, REC>(a, blockIdx.x, smem);
}

template <int D, int F, bool REC>
void launch_lists(hipStream_t s, ScatterListsArgs a, uint32_t n_tasks) {
	static bool configured = false;
	if (!configured) { // more than 64 KiB of dynamic LDS has to be opted into once per kernel
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)k_grid_scatter_lists<D, F, REC>, hipFuncAttributeMaxDynamicSharedMemorySize, SL_LDS_BYTES));
		configured = true;
	}
#ifdef TCNN_AMD_DEV
	// laboratory build: TCNN_AMD_SCATTER_TIMING=1 prints per-task phase times (100 MHz constant clock) of the 3rd launch (tools/scatter_timing.py)
	static const bool timing = getenv("TCNN_AMD_SCATTER_TIMING") != nullptr;
	static int timing_left = 3;
	unsigned long long* dbg = nullptr;
	if (timing && timing_left > 0 && !a.dbg_times) {
		HIP_CHECK_THROW(hipMalloc(&dbg, (size_t)n_tasks * 8 * 8));
		HIP_CHECK_THROW(hipMemset(dbg, 0, (size_t)n_tasks * 8 * 8));
		a.dbg_times = dbg;
	}
#endif
	hipLaunchKernelGGL((k_grid_scatter_lists<D, F, REC>), dim3(n_tasks), dim3(SL_THREADS), SL_LDS_BYTES, s, a);
	HIP_CHECK_THROW(hipGetLastError());
#ifdef TCNN_AMD_DEV
	if (dbg) {
		std::vector<unsigned long long> h((size_t)n_tasks * 8);
		std::vector<GridScatterTask> ht(n_tasks);
		HIP_CHECK_THROW(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
		HIP_CHECK_THROW(hipMemcpy(ht.data(), a.tasks, n_tasks * sizeof(GridScatterTask), hipMemcpyDeviceToHost));
		unsigned long long t0 = ~0ull;
		for (uint32_t i = 0; i < n_tasks; ++i) if (ht[i].n_entries) t0 = std::min(t0, h[i * 8]);
		if (--timing_left == 0) {
			for (uint32_t i = 0; i < n_tasks; ++i) {
				if (!ht[i].n_entries) continue;
				fprintf(stderr, "task %3u level %2u entries %6u samples %6u atomic %u: start %7.1f zero %6.1f accumulate %6.1f flush %6.1f us | split %u of %u\n", i, ht[i].level, ht[i].n_entries,
				        ht[i].sample_end - ht[i].sample_begin, ht[i].flush_atomic, (h[i * 8] - t0) * 0.01, (h[i * 8 + 1] - h[i * 8]) * 0.01, (h[i * 8 + 2] - h[i * 8 + 1]) * 0.01,
				        (h[i * 8 + 3] - h[i * 8 + 2]) * 0.01, ht[i].pad & 0xffffu, ht[i].pad >> 16);
				if (i % 37 == 0 && h[i * 8 + 3] > h[i * 8]) fprintf(stderr, "  clock of task %u: %.0f MHz\n", i, (double)(h[i * 8 + 5] - h[i * 8 + 4]) / ((h[i * 8 + 3] - h[i * 8]) * 0.01));
			}
		}
		(void)hipFree(dbg);
	}
#endif
}

template <int D>
void dispatch_lists(hipStream_t s, uint32_t F, bool records, const ScatterListsArgs& a, uint32_t n_tasks) {
#define TCNN_SL(FF, RR) return launch_lists<D, FF, RR>(s, a, n_tasks)
	if (records) {
		if constexpr (D == 2) {
			if (F == 2) TCNN_SL(2, true);
			if (F == 4) TCNN_SL(4, true);
		} else {
			if (F == 2) TCNN_SL(2, true);
		}
		throw std::runtime_error{"grid_backward_lists: scatter records need 4 D + 2 F <= 16"};
	}
	switch (F) {
		case 2: TCNN_SL(2, false);
		case 4: TCNN_SL(4, false);
		case 8: TCNN_SL(8, false);
		default: throw std::runtime_error{"grid_backward_lists: needs n_features_per_level in {2, 4, 8}"};
	}
#undef TCNN_SL
}

} // namespace

uint32_t grid_scatter_lists_lds_bytes() { return SL_LDS_BYTES; }

// Tasks of the list-fed kernel, in launch order.  No measured tuning.
//   * a level cut into chunks: one task per chunk, its single owner -- or, where a chunk's list is long (a level that could not be cut
//     finer: no records), several tasks sharing the chunk, each a range of the list, merged through the scratch table;
//   * a level that is one chunk: tasks over sample ranges of >= 8192 samples, merged through the scratch table;
//   * XCD = b % 8 for block b: the tasks that GATHER from one record plane (a level; a level pair where two levels share a record) all go to
//     one XCD -- the 4 MB plane is pulled into that L2 once and stays: with two or three planes per XCD (8 - 12 MB against 4 MB of L2) every
//     record was fetched ~4 times and a task took 25 - 28 us instead of 22 - 24, with four planes per XCD 38 - 44 -- the tasks that stream
//     their samples in order fill the XCDs up, longest first, and the eight lists are interleaved (padded with empty tasks).
void grid_scatter_lists_plan(const GridMeta& meta, uint32_t n, bool paired_records, std::vector<GridScatterTask>& tasks, std::vector<GridScatterRange>& shared_ranges, size_t& scratch_elems) {
	const uint32_t F = meta.n_features_per_level;
	const uint32_t rows = meta.interpolation == (uint32_t)InterpolationType::Nearest ? 1u : (1u << (meta.n_pos_dims - 1));
	tasks.clear();
	shared_ranges.clear();
	scratch_elems = 0;
	std::vector<std::pair<double, GridScatterTask>> q[8]; // (estimated cost, task)
	double load[8] = {};
	std::vector<std::pair<double, GridScatterTask>> loose;
	uint32_t dbg_lo = 0, dbg_hi = meta.n_levels;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_SCATTER_LEVELS")) sscanf(e, "%u,%u", &dbg_lo, &dbg_hi); // profiling aid: levels lo..hi only (results are then incomplete!)
#endif
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		const GridLevel& lv = meta.levels[l];
		if (lv.scatter_binned || l < dbg_lo || l > dbg_hi) continue; // (binned: grid_backward_binned serves this level)
		const uint32_t n_chunks = lv.scatter_n_chunks, per_chunk = lv.scatter_per_chunk;
		const bool listed = n_chunks > 1;
		// elements (listed: one gather each, ~2 corners) or samples (streamed, all corners) per chunk, and how many tasks share them
		const double per_chunk_work = listed ? (double)n * rows / n_chunks : (double)n;
		// A chunk shared by several tasks is flushed through 64-bit global atomics, one per value and task (~27 G/s chip-wide: 16 384 values
		// are ~10 us per task, and at 2^20 samples two tasks per fine chunk were 31 M atomics -- 0.53 ms against 0.26 for single owners whose
		// tasks simply take longer).  So only small chunks are ever shared.
		const bool cheap_flush = per_chunk * F <= 2048;
		const uint32_t splits = listed ? (cheap_flush ? (uint32_t)std::min(std::max(per_chunk_work / 16384.0 + 0.5, 1.0), 32.0) : 1u) : std::min(std::max(n / 8192u, 1u), 64u);
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
				if (!shared_ranges.empty() && shared_ranges.back().grad_begin + shared_ranges.back().n_elems == grad_begin && shared_ranges.back().scratch_begin + shared_ranges.back().n_elems == scratch_begin) {
					shared_ranges.back().n_elems += cnt * F;
				} else {
					shared_ranges.push_back(GridScatterRange{grad_begin, cnt * F, scratch_begin, 0});
				}
				scratch_elems += (size_t)cnt * F;
			}
			for (uint32_t s = 0; s < splits; ++s) {
				const uint32_t sb = s * samples_per_split;
				if (sb >= n) break;
				const GridScatterTask t{l, begin, cnt, sb, std::min(n, sb + samples_per_split), shared ? 1u : 0u, scratch_begin, s | splits << 16};
				// cost in "gathered or streamed lanes": a gathered element ~3x a streamed sample's load, plus the corners' adds and the chunk's zeroing / flush
				const double cost = (listed ? per_chunk_work / splits * 3.0 : (double)samples_per_split * (1.0 + rows)) + cnt * F * (shared ? 1.0 : 0.25);
				if (listed) {
					const uint32_t xcd = (paired_records ? l / 2 : l) % 8;
					q[xcd].emplace_back(cost, t);
					load[xcd] += cost;
				} else {
					loose.emplace_back(cost, t);
				}
			}
		}
	}
	std::stable_sort(loose.begin(), loose.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
	for (const auto& t : loose) {
		size_t best = 0;
		for (size_t b = 1; b < 8; ++b) if (load[b] < load[best]) best = b;
		q[best].push_back(t);
		load[best] += t.first;
	}
	size_t longest = 0;
	for (auto& v : q) {
		std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
		longest = std::max(longest, v.size());
	}
	for (size_t j = 0; j < longest; ++j) {
		for (size_t b = 0; b < 8; ++b) tasks.push_back(j < q[b].size() ? q[b][j].second : GridScatterTask{0, 0, 0, 0, 0, 0, 0, 0});
	}
	while (!tasks.empty() && tasks.back().n_entries == 0) tasks.pop_back();
}

void grid_backward_lists(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, const GridScatterTask* dev_tasks, uint32_t n_tasks,
                         const GridScatterRange* dev_ranges, uint32_t n_ranges, uint64_t* scratch, uint32_t n, MatView x,
                         const void* dL_dy, uint32_t dy_stride_sample, uint32_t dy_stride_level, void* grad, const GridHitLists& lists, bool accumulate, bool dy_records,
                         const MlpReduceJob* reduce_job, uint32_t* fallback_count, const AdamInFlush* adam) {
	if (n_tasks == 0) return;
	CHECK_THROW(lists.elems != nullptr && lists.heads != nullptr && lists.stragglers != nullptr && lists.counts != nullptr && lists.n_items > 0);
	CHECK_THROW(!dy_records || grid_scatter_records_supported(meta));
	CHECK_THROW(n <= grid_hit_max_samples(meta) && meta.hash_type != (uint32_t)HashType::Rng);
	// every chunk must fit the 64 KiB of packed accumulators (the plan's chunks are cut for 128 KiB of 64-bit ones: the same entry count)
	for (uint32_t l = 0; l < meta.n_levels; ++l) CHECK_THROW(meta.levels[l].scatter_binned || meta.levels[l].scatter_per_chunk * meta.n_features_per_level * 4 <= SL_ACC_BYTES);
	ScatterListsArgs a{};
	a.meta = dev_meta;
	a.tasks = dev_tasks;
	a.n = n;
	a.x = x;
	a.dL_dy = (const half_t*)dL_dy;
	a.dy_stride_sample = dy_stride_sample;
	a.dy_stride_level = dy_stride_level;
	a.grad = (half_t*)grad;
	a.lists = lists;
	a.scratch = (unsigned long long*)scratch;
	a.accumulate_mode = accumulate ? 1 : 0;
	a.force_wide = switches().scatter_wide ? 1 : 0; // tests: every task through the 64-bit passes
	a.dbg_times = nullptr;
	a.fallback_count = fallback_count;
	if (adam) a.adam = *adam;
	switch (meta.n_pos_dims) {
		case 2: dispatch_lists<2>(stream, meta.n_features_per_level, dy_records, a, n_tasks); break;
		case 3: dispatch_lists<3>(stream, meta.n_features_per_level, dy_records, a, n_tasks); break;
		default: throw std::runtime_error{"grid_backward_lists: 2 or 3 input dims"};
	}
	grid_scatter_finalize(stream, dev_ranges, n_ranges, scratch, grad, accumulate, reduce_job);
}

} // namespace tcnn_amd
