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
//     no compaction queue.  Round 5: an element carries the row's two entries (relative to the chunk) and its two half weights, and the
//     gradient arrives as a stream too: k_grid_list_gradients (below, the launch in front of the owners') copies dL/dy into list order --
//     per item it loads the item's 2 KB slice of the level's gradient plane into LDS and writes every element's F halves at the
//     element's position -- so the owner reads elements and gradients side by side with dense loads: no gather, no coordinates, no
//     pos_fract, no hash.  (Round 4's owner gathered a 16-byte record per element: every such gather pulls a 128-byte line from the L2
//     into the CU, ~3 clocks per lane whatever it carries -- 40 us of the kernel's 59 - 66, and only while the record plane was resident
//     in the XCD's L2, i.e. for batches of 2^17 .. 2^19 samples.  profiles/r05_scatter_timeline.txt);
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
#ifndef TCNN_SL_THREADS
#define TCNN_SL_THREADS 512
#endif
#ifndef TCNN_SL_WINDOW
#define TCNN_SL_WINDOW 1536
#endif
constexpr uint32_t SL_THREADS = TCNN_SL_THREADS;
constexpr uint32_t SL_WAVES = SL_THREADS / 64;
constexpr uint32_t SL_WINDOW = TCNN_SL_WINDOW;        // elements of a wave's runs looked up through one fill of its owner table
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

template <int D, int F>
struct ScatterCtx {
	typedef typename VecOf<half_t, F>::type vecF;
	GridLevel lv;
	uint32_t primes[D];
	uint32_t hash_type, interpolation;
	uint32_t level, n;
	MatView x;
	const half_t* dy;           // dL_dy + level * dy_stride_level
	uint32_t dy_stride_sample;
};

// coordinates and dL/dy of one sample: the levels that are one chunk (streamed in sample order) and the stragglers
template <int D, int F>
__device__ inline void sl_fetch(const ScatterCtx<D, F>& c, const uint32_t i, float (&xin)[D], typename ScatterCtx<D, F>::vecF& gv) {
	typedef typename ScatterCtx<D, F>::vecF vecF;
	load_coords<D>(c.x, i, xin);
	gv = *(const vecF*)&c.dy[(size_t)i * c.dy_stride_sample];
}

// The cell of one sample: positions and weights' factors, once per sample (grid.h:147-160, common_device.h:856-868)
template <int D>
struct CellPos {
	uint32_t cell[D];
	float pos[D];
};
template <int D, int F>
__device__ inline CellPos<D> sl_cell(const ScatterCtx<D, F>& c, const float (&xin)[D]) {
	CellPos<D> p;
	float unused;
#pragma unroll
	for (int d = 0; d < D; ++d) p.cell[d] = pos_fract(xin[d], c.lv.scale, c.interpolation, &p.pos[d], &unused);
	return p;
}

// One contribution: (GRAD_T) weight * grad in fp16 (grid.h:254), all F features, into entry `index` of the accumulators.
// PACKED: accumulators are uint64 [entry][F / 2] holding two int32 sums; else int64 [entry][F].
template <int F, bool PACKED, typename vecF>
__device__ inline void sl_add_entry(lds_u64* acc, const uint32_t index, const half_t w, const vecF& gv, float& bound) {
	if constexpr (PACKED) {
#pragma unroll
		for (int j = 0; j < F / 2; ++j) {
			const float p0 = (float)(half_t)(w * gv[2 * j]), p1 = (float)(half_t)(w * gv[2 * j + 1]);
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
}

// Both corners of a listed element into packed accumulators, all F features: the same products and the same integers as sl_add_entry's,
// in half the vector instructions -- the owners are bound by their vector instructions once nothing gathers (5 - 6 us of a task's 8 with
// every load and LDS add taken out, profiles/r05_scatter_dev.txt) --: the two features of a pair in one packed multiply (weight A is the
// low half of `weights`, weight B the high one), half -> scaled float in one v_fma_mix_f32 (exact: a half times 2^24), and the task's
// bound from the ELEMENT: |g| (weight A + weight B) bounds the two products' sum within the half roundings the limit's margin covers -- one
// instruction per element and two per pair instead of four per pair.  (With |g| alone, the weights' sum taken as 1, a sample's 2^(D-1) row
// elements each charged the whole |g|: the bound came out 2x (2-D) / 4x (3-D) the sum it stands for, and 838 of a 3-D grid's ~1000 tasks went
// through the 64-bit passes for nothing, profiles/r05_shape_sweep.txt.)  No test per corner: an element without a corner B names A's entry
// twice, the second time with weight +0 (GridHitLists).
template <int F, typename vecF>
__device__ inline void sl_add_element_packed(lds_u64* acc, const uint32_t entries, const uint32_t weights, const vecF& gv, float& bound) {
	typedef _Float16 h2 __attribute__((ext_vector_type(2)));
	const float two24 = 16777216.0f;
	lds_u64* pa = (lds_u64*)((__attribute__((address_space(3))) char*)acc + (entries & 0xffffu) * (F * 4u));
	lds_u64* pb = (lds_u64*)((__attribute__((address_space(3))) char*)acc + (entries >> 16) * (F * 4u));
	float ws; // weight A + weight B
	asm("v_fma_mix_f32 %0, %1, 1.0, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(ws) : "v"(weights));
#pragma unroll
	for (int j = 0; j < F / 2; ++j) {
		const uint32_t g2 = __builtin_bit_cast(uint32_t, (h2{gv[2 * j], gv[2 * j + 1]}));
		uint32_t qa, qb;
		asm("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(qa) : "v"(g2), "v"(weights));               // (GRAD_T) weight * grad in fp16, grid.h:254
		asm("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(qb) : "v"(g2), "v"(weights));
		asm("v_fma_mix_f32 %0, |%1|, %2, %0 op_sel_hi:[1,0,0]" : "+v"(bound) : "v"(g2), "v"(ws));
		asm("v_fma_mix_f32 %0, |%1|, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(bound) : "v"(g2), "v"(ws));
		auto add = [&](lds_u64* at, const uint32_t q) {
			float s0, s1;
			asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(s0) : "v"(q), "v"(two24));
			asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(s1) : "v"(q), "v"(two24));
			const int f0 = (int)s0, f1 = (int)s1;                                 // exact below 128; beyond it the bound has failed anyway
			const uint32_t lo = (uint32_t)f0, hi = (uint32_t)(f1 + (f0 >> 31));   // f0 + f1 2^32 as one 64-bit integer
			__hip_atomic_fetch_add(at + j, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		};
		add(pa, qa);
		add(pb, qb);
	}
}

// One cell row of one sample from its coordinates: corners A (cell_0) and B (cell_0 + 1) of row `row` (bit d - 1: cell_d + 1; per lane), each
// added if its flag is set and its entry lies in [e0, e0 + n_sub).  The streamed levels and the stragglers; listed elements bring entries and weights along.
template <int D, int F, bool PACKED>
__device__ inline void sl_add_row(const ScatterCtx<D, F>& c, lds_u64* acc, const uint32_t e0, const uint32_t n_sub, const CellPos<D>& p, const typename ScatterCtx<D, F>::vecF& gv,
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
	if (want_a && ia < n_sub) sl_add_entry<F, PACKED>(acc, ia, (half_t)wa, gv, bound);
	if (want_b && ib < n_sub) sl_add_entry<F, PACKED>(acc, ib, (half_t)wb, gv, bound);
}

// Where a listed task's elements are: the items' runs of its chunk (GridHitLists, tcnn_common.h)
struct ScatterRuns {
	uint32_t dev_flags;         // laboratory build (TCNN_AMD_SCATTER_DEV): 1 linear element addresses, 2 no LDS adds, 4 no element / gradient loads (timing only: wrong results)
	const uint32_t* elems;      // this level's pool: [n_items][item_capacity][GRID_HIT_WORDS]
	const half_t* gvals;        // dL/dy of the elements' samples, same positions: [n_items][item_capacity][F]
	const uint32_t* heads;      // this level's offsets, already advanced to the task's chunk: heads[item] = where the run starts, heads[n_items + item] = where it ends
	const uint32_t* stragglers; // this level's {sample | corner bit, chunk} pairs
	uint32_t n_stragglers, n_items, item_capacity, chunk;
};

// The accumulation pass of one task: over the runs of its chunk in the items [begin, end) of a level cut into chunks, or over the samples
// [begin, end) of a level that is one chunk.  Entries [e0, e0 + n_sub) of the level are this pass's (the whole chunk, or a part of it in the 64-bit passes).
template <int D, int F, bool PACKED>
__device__ inline float sl_pass(const ScatterCtx<D, F>& c, lds_u64* acc, const uint32_t e0, const uint32_t n_sub, const bool listed, const ScatterRuns& runs, const bool with_stragglers,
                                const uint32_t begin, const uint32_t end, const uint32_t wave, const uint32_t lane, char* wave_lds, const uint32_t first_h0, const uint32_t first_h1, const uint32_t group) {
	typedef typename ScatterCtx<D, F>::vecF vecF;
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
			const CellPos<D> p = sl_cell<D, F>(c, xin);
#pragma unroll
			for (uint32_t row = 0; row < R; ++row) {
				if (row > 0 && nearest) break; // grid.h:232-246: the cell's own entry, weight 1
				sl_add_row<D, F, PACKED>(c, acc, e0, n_sub, p, gv, row, true, !nearest, bound);
			}
		};
		for (uint32_t i0 = w_begin; i0 < w_end; i0 += 64) {
			const uint32_t i = i0 + lane;
			sl_fetch<D, F>(c, min(i, w_end - 1), qx, qg);
			if (p_valid) whole(px, pg);
#pragma unroll
			for (int d = 0; d < D; ++d) px[d] = qx[d];
			pg = qg;
			p_valid = i < w_end;
		}
		if (p_valid) whole(px, pg);
		return bound;
	}
	// Listed.  A wave takes `group` items at a time (64, or fewer where the task has fewer than 64 items per wave), one per lane: the lane reads where its item's run of this chunk starts and ends, a scan
	// over the lanes numbers the runs' elements 0 .. T - 1, and every lane writes its own number into a wave-private byte table at its
	// elements' positions (~16 each) -- element e is then TWO LDS reads away: table[e] names the lane, positions[lane] where its run
	// lies.  (First form: a 6-step search over the lanes' running totals with ds_bpermute per element -- 7 cross-lane reads and ~20 vector
	// instructions per 64 elements, and the kernel is bound by its vector instructions: 28 us per task against 21 with plain lists.)
	//
	// Then a software pipeline over steps of 64 SL_SB elements in FOUR stages, each working on what the stage before asked for a step
	// earlier, so that nothing waits for what it has just requested: (A) the owner bytes of step j + LEAD + 2, (B) with those the runs'
	// positions of step j + LEAD + 1, (C) with those the loads of step j + LEAD -- 8 + 2 F bytes per element, elements and gradients at the
	// same position, dense along a run --, (D) step j is added.  NS = LEAD + 3 register sets rotate STATICALLY (the loop is unrolled NS
	// times; a set is never copied -- a move of a register whose load is still in flight is a wait for it).  LDS operations and loads each
	// return in order, so the compiler's counted waits leave everything younger in flight.  All loads are raw buffer loads without a
	// branch around them: a lane past the last element reads from beyond the descriptor's range (no memory access, zeros) and adds
	// nothing.  (With `if (element) load` the compiler waited with vmcnt(0) right behind every load -- the loaded value has to be merged
	// with the default at the end of the branch -- and nothing was ever in flight.  Round 5's first form looked the position up right in
	// front of the loads: two dependent LDS reads queued behind the step's LDS adds, 7 of a task's 17 us.)
	constexpr int NS = SL_LEAD + 3;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	constexpr uint32_t OOB = 0xffffffe0u; // beyond every descriptor below (their sizes are below 2^31: grid_backward_lists checks)
	const auto rs_pool = __builtin_amdgcn_make_buffer_rsrc((void*)runs.elems, 0, (int)(runs.n_items * runs.item_capacity * (GRID_HIT_WORDS * 4u)), 0x00020000);
	const auto rs_gv = __builtin_amdgcn_make_buffer_rsrc((void*)runs.gvals, 0, (int)(runs.n_items * runs.item_capacity * (F * 2u)), 0x00020000);
	const uint32_t rel0 = e0 - runs.chunk * c.lv.scatter_per_chunk; // the elements' entries are relative to the chunk's first one (the packed pass owns the whole chunk: 0)
	constexpr uint32_t BATCH = 64 * SL_SB;
	const uint32_t n_groups = (end - begin + group - 1) / group;
	for (uint32_t g = wave; g < n_groups; g += SL_WAVES) { // wave-uniform
		const uint32_t item0 = begin + g * group, item = item0 + lane;
		uint32_t h0 = first_h0, h1 = first_h1; // the wave's first group: requested when the task began, under the zeroing of the accumulators
		if (g != wave) {
			const uint32_t* hd = runs.heads + min(item, end - 1);
			h0 = hd[0];
			h1 = hd[runs.n_items];
		}
		const uint32_t cnt = (lane < group && item < end) ? h1 - h0 : 0u;
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
		uint32_t ob[NS][SL_SB], ps[NS][SL_SB];
		u32x2 el[NS][SL_SB];
		vecF gs[NS][SL_SB];
		auto stage_a = [&](const int set, const uint32_t b0) {
#pragma unroll
			for (int s = 0; s < SL_SB; ++s) {
				const uint32_t e = b0 + s * 64 + lane;
				ob[set][s] = owner[e < w1 ? e - w0 : 0u];
			}
		};
		auto stage_b = [&](const int set, const uint32_t b0) {
#pragma unroll
			for (int s = 0; s < SL_SB; ++s) ps[set][s] = run_pos[ob[set][s]] + (b0 + s * 64 + lane);
		};
		auto stage_c = [&](const int set, const uint32_t b0) {
#pragma unroll
			for (int s = 0; s < SL_SB; ++s) {
				const uint32_t e = b0 + s * 64 + lane;
				const bool have = e < w1;
				uint32_t at = ps[set][s];
#ifdef TCNN_AMD_DEV
				if (runs.dev_flags & 1u) at = item0 * runs.item_capacity + e;
				if (runs.dev_flags & 8u) at = item0 * runs.item_capacity + e + (at >> 30);
				if (runs.dev_flags & 4u) { el[set][s] = u32x2{(e * 2654435761u >> 19) | (e * 40503u >> 3) << 16, 0x38003800u}; for (int f = 0; f < F; ++f) gs[set][s][f] = (half_t)0.00390625f; continue; }
#endif
				el[set][s] = __builtin_amdgcn_raw_buffer_load_b64(rs_pool, have ? at * (GRID_HIT_WORDS * 4u) : OOB, 0, 0);
				const uint32_t goff = have ? at * (F * 2u) : OOB;
				if constexpr (F == 2) gs[set][s] = __builtin_bit_cast(vecF, __builtin_amdgcn_raw_buffer_load_b32(rs_gv, goff, 0, 0));
				else if constexpr (F == 4) gs[set][s] = __builtin_bit_cast(vecF, __builtin_amdgcn_raw_buffer_load_b64(rs_gv, goff, 0, 0));
				else gs[set][s] = __builtin_bit_cast(vecF, __builtin_amdgcn_raw_buffer_load_b128(rs_gv, goff, 0, 0));
			}
		};
		auto stage_d = [&](const int set, const uint32_t b0) {
#pragma unroll
			for (int s = 0; s < SL_SB; ++s) {
				const u32x2 ev = el[set][s];
				uint32_t word0 = ev.x;
				const uint32_t word1 = ev.y; // (never ev[i] with a loop variable inside a bit_cast: the compiler read element 0 for every i)
				const bool have = b0 + s * 64 + lane < w1;
#ifdef TCNN_AMD_DEV
				if (runs.dev_flags & 2u) { bound += (float)(word0) * (float)word1 * (float)gs[set][s][0]; continue; }
#endif
				if constexpr (PACKED) {
					// no branch: a lane without an element has loaded zeros -- weights +0, gradient +0 -- and adds them to an entry of its own
					// (entry `lane`: every chunk has more than 64 entries) instead of all such lanes queueing at entry 0
					word0 = have ? word0 : lane * 0x10001u;
					sl_add_element_packed<F>(acc, word0, word1, gs[set][s], bound);
				} else {
					const uint32_t ia = (word0 & 0xffffu) - rel0, ib = (word0 >> 16) - rel0;
					const half_t wa = __builtin_bit_cast(half_t, (uint16_t)word1), wb = __builtin_bit_cast(half_t, (uint16_t)(word1 >> 16));
					if (have && ia < n_sub) sl_add_entry<F, false>(acc, ia, wa, gs[set][s], bound);
					if (have && ib < n_sub) sl_add_entry<F, false>(acc, ib, wb, gs[set][s], bound);
				}
			}
		};
		// fill: A for steps 0 .. LEAD + 1, B for 0 .. LEAD, C for 0 .. LEAD - 1
#pragma unroll
		for (int b = 0; b < SL_LEAD + 2; ++b) stage_a(b, w0 + b * BATCH);
#pragma unroll
		for (int b = 0; b < SL_LEAD + 1; ++b) stage_b(b, w0 + b * BATCH);
#pragma unroll
		for (int b = 0; b < SL_LEAD; ++b) stage_c(b, w0 + b * BATCH);
		const uint32_t quarter = max((w1 - w0) / 4u, 1u);
		for (uint32_t b0 = w0; b0 < w1; b0 += NS * BATCH) {
			// The two workgroups of a CU share its address path, and the arbiter serves the older wave first: left alone the older workgroup
			// runs as if it had the CU to itself and the younger one takes the rest.  Priority by progress instead: whoever is further from
			// the end of its run goes first.
			{
				const uint32_t done = (b0 - w0) / quarter; // wave-uniform
				if (done == 0) __builtin_amdgcn_s_setprio(3);
				else if (done == 1) __builtin_amdgcn_s_setprio(2);
				else if (done == 2) __builtin_amdgcn_s_setprio(1);
				else __builtin_amdgcn_s_setprio(0);
			}
#pragma unroll
			for (int u = 0; u < NS; ++u) { // step j = (b0 - w0) / BATCH + u lives in set u (j is a multiple of NS at u = 0)
				if (b0 + u * BATCH >= w1) break; // wave-uniform
				stage_a((u + SL_LEAD + 2) % NS, b0 + (u + SL_LEAD + 2) * BATCH);
				stage_b((u + SL_LEAD + 1) % NS, b0 + (u + SL_LEAD + 1) * BATCH);
				stage_c((u + SL_LEAD) % NS, b0 + (u + SL_LEAD) * BATCH);
				stage_d(u, b0 + u * BATCH);
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
				sl_fetch<D, F>(c, e & HIT_ID_MASK, xin, gv);
				const CellPos<D> cp = sl_cell<D, F>(c, xin);
				uint32_t m = e >> HIT_SHIFT;
				while (m != 0) {
					const uint32_t row = (uint32_t)__builtin_ctz(m) >> 1;
					const uint32_t two = (m >> (2 * row)) & 3u;
					m &= ~(3u << (2 * row));
					sl_add_row<D, F, PACKED>(c, acc, e0, n_sub, cp, gv, row, (two & 1u) != 0, (two & 2u) != 0, bound);
				}
			}
		}
	}
	return bound;
}

// ---------------------------------------------------------------------------------------------------------------- dL/dy into list order
// One workgroup per (item, level) of a listed level: the item's slice of the level's gradient plane (item_samples x F halves: 2 KB) goes
// into LDS with dense loads, then every element of the item -- all chunks' runs, front to back -- takes its sample's F halves from there
// and stores them at the element's own position in `gvals`.  Dense loads, dense stores, 2 + 2 F bytes in and 2 F bytes out per element; what
// the owners' gathers cost (a 128-byte line from the L2 per element) is paid here once per 32 samples instead.
constexpr uint32_t LG_THREADS = 256;
struct ListGradArgs {
	const GridMeta* meta;
	GridHitLists lists;
	const half_t* dL_dy;
	uint32_t dy_stride_sample, dy_stride_level, n;
	half_t* gvals; // [n_levels][n_items][item_capacity][F]
};

constexpr uint32_t LG_ITEMS = 2; // items per workgroup: the loads of both are in flight together (one item each: 12.5 us for 58 MB at 2^18 samples -- latency, not bytes)
template <int F>
__global__ void __launch_bounds__(LG_THREADS) k_grid_list_gradients(const ListGradArgs a) {
	typedef typename VecOf<half_t, F>::type vecF;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t level = blockIdx.y;
	const uint32_t n_chunks = a.meta->levels[level].scatter_n_chunks;
	if (n_chunks <= 1 || n_chunks > GRID_FILTER_MAX_CHUNKS || a.meta->levels[level].scatter_binned) return; // (workgroup-uniform) not a listed level
	const GridHitLists& hl = a.lists;
	const half_t* dy = a.dL_dy + (size_t)level * a.dy_stride_level;
	// Everything this workgroup reads is requested before anything is waited for: the items' sample numbers, four elements per thread and
	// step (one 8-byte load; the forward kernel wrote whole quads of them -- what lies beyond an item's last element is stale but readable:
	// clamped, and stored into positions nobody reads), then the items' slices of the gradient plane on their way into LDS.
	typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
	constexpr uint32_t PRE = 4; // steps whose sample numbers are in registers before the barrier (4 Ki elements per item: every shape in use)
	u16x4 sv[LG_ITEMS][PRE];
	uint32_t total[LG_ITEMS];
	size_t region[LG_ITEMS];
#pragma unroll
	for (uint32_t it = 0; it < LG_ITEMS; ++it) {
		const uint32_t item = blockIdx.x * LG_ITEMS + it;
		total[it] = item < hl.n_items ? hl.heads[((size_t)level * GRID_HIT_HEADS + GRID_FILTER_MAX_CHUNKS) * hl.n_items + item] : 0u;
		region[it] = ((size_t)level * hl.n_items + min(item, hl.n_items - 1)) * hl.item_capacity;
	}
#pragma unroll
	for (uint32_t it = 0; it < LG_ITEMS; ++it) {
		const uint16_t* sidx = hl.sidx + region[it];
#pragma unroll
		for (uint32_t k = 0; k < PRE; ++k) {
			const uint32_t p = (k * LG_THREADS + threadIdx.x) * 4;
			sv[it][k] = p < total[it] ? *(const u16x4*)(sidx + p) : u16x4{0, 0, 0, 0};
		}
	}
#pragma unroll
	for (uint32_t it = 0; it < LG_ITEMS; ++it) {
		vecF* g = (vecF*)smem + it * hl.item_samples;
		const uint32_t first = (blockIdx.x * LG_ITEMS + it) * hl.item_samples;
		if (total[it] == 0) continue; // (workgroup-uniform)
		for (uint32_t s = threadIdx.x; s < hl.item_samples; s += LG_THREADS) {
			const uint32_t i = min(first + s, a.n - 1);
			g[s] = *(const vecF*)&dy[(size_t)i * a.dy_stride_sample];
		}
	}
	__syncthreads();
	const uint32_t last = hl.item_samples - 1;
#pragma unroll
	for (uint32_t it = 0; it < LG_ITEMS; ++it) {
		const vecF* g = (const vecF*)smem + it * hl.item_samples;
		vecF* out = (vecF*)a.gvals + region[it];
		const uint16_t* sidx = hl.sidx + region[it];
		auto place = [&](const uint32_t p, const u16x4 v) {
			const vecF v0 = g[min((uint32_t)v[0], last)], v1 = g[min((uint32_t)v[1], last)], v2 = g[min((uint32_t)v[2], last)], v3 = g[min((uint32_t)v[3], last)];
			if constexpr (F == 2) {
				typedef uint32_t u4 __attribute__((ext_vector_type(4)));
				*(u4*)(out + p) = u4{__builtin_bit_cast(uint32_t, v0), __builtin_bit_cast(uint32_t, v1), __builtin_bit_cast(uint32_t, v2), __builtin_bit_cast(uint32_t, v3)};
			} else {
				out[p] = v0; out[p + 1] = v1; out[p + 2] = v2; out[p + 3] = v3;
			}
		};
#pragma unroll
		for (uint32_t k = 0; k < PRE; ++k) {
			const uint32_t p = (k * LG_THREADS + threadIdx.x) * 4;
			if (p < total[it]) place(p, sv[it][k]);
		}
		for (uint32_t p = (PRE * LG_THREADS + threadIdx.x) * 4; p < total[it]; p += LG_THREADS * 4) place(p, *(const u16x4*)(sidx + p));
	}
}

template <int F>
void launch_list_gradients(hipStream_t s, const ListGradArgs& a, uint32_t n_levels) {
	hipLaunchKernelGGL((k_grid_list_gradients<F>), dim3(div_round_up(a.lists.n_items, LG_ITEMS), n_levels), dim3(LG_THREADS), LG_ITEMS * a.lists.item_samples * F * 2, s, a);
	HIP_CHECK_THROW(hipGetLastError());
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
	uint32_t dev_flags;
	const half_t* gvals;        // dL/dy in list order (k_grid_list_gradients): [n_levels][n_items][item_capacity][F]
	unsigned long long* scratch;
	int accumulate_mode, force_wide;
	unsigned long long* dbg_times;
	uint32_t* fallback_count;
};

template <int D, int F>
__device__ inline void sl_run_task(const ScatterListsArgs& a, const uint32_t task_index, char* smem) {
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

	ScatterCtx<D, F> c;
	c.lv = meta->levels[task.level];
#pragma unroll
	for (int d = 0; d < D; ++d) c.primes[d] = meta->primes[d];
	c.hash_type = meta->hash_type;
	c.interpolation = meta->interpolation;
	c.level = task.level;
	c.n = n;
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
		runs.dev_flags = a.dev_flags;
		runs.elems = hl.elems + (size_t)task.level * hl.n_items * hl.item_capacity * GRID_HIT_WORDS;
		runs.gvals = a.gvals + (size_t)task.level * hl.n_items * hl.item_capacity * F;
		runs.heads = hl.heads + ((size_t)task.level * GRID_HIT_HEADS + runs.chunk) * hl.n_items;
		runs.stragglers = hl.stragglers + (size_t)task.level * hl.straggler_capacity * 2;
		runs.n_stragglers = min(hl.counts[task.level * GRID_HIT_COUNT_STRIDE], hl.straggler_capacity);
		runs.n_items = hl.n_items;
		runs.item_capacity = hl.item_capacity;
		with_stragglers = s == 0;
	}

	uint32_t first_h0 = 0, first_h1 = 0; // where this lane's item (of the wave's first group) keeps its run of the chunk
	// items per wave and turn: all eight waves busy also where the forward kernel's work items are large and a task has few of them
	const uint32_t group = min(64u, max((end - begin + SL_WAVES - 1) / SL_WAVES, 1u));
	if (listed && begin < end) {
		const uint32_t* hd = runs.heads + min(begin + wave * group + lane, end - 1);
		first_h0 = hd[0];
		first_h1 = hd[runs.n_items];
	}
	typedef uint32_t u4 __attribute__((ext_vector_type(4)));
	auto zero_acc = [&](const uint32_t bytes) {
		u4* a4 = (u4*)smem;
		for (uint32_t i = tid; i < (bytes + 15) / 16; i += SL_THREADS) a4[i] = u4{0, 0, 0, 0};
	};

	// Every task tries the packed sums first (round 4 gave chunks small enough for 64-bit accumulators in one pass those at once: half the
	// rate on the LDS atomics -- 23 - 30 us per coarse task beside 20 for a fine one, profiles/r05_scatter_timeline.txt).  Its bound is over
	// EVERYTHING the task adds, so tasks are cut for the same number of contributions whatever the level (grid_scatter_lists_plan).
	const uint32_t wide_parts = (n_vals * 8 + SL_ACC_BYTES - 1) / SL_ACC_BYTES;
	bool packed_ok = !force_wide;
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
		float bound = sl_pass<D, F, true>(c, acc, task.entry_begin, task.n_entries, listed, runs, with_stragglers, begin, end, wave, lane, smem + SL_ACC_BYTES + 256 + wave * SL_WAVE_LDS, first_h0, first_h1, group);
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
	} else {
		// ------------------------------------------------------------------------------------------------ wide passes: 64-bit accumulators, half of the entries at a time
		if (fallback_count && tid == 0 && !force_wide) atomicAdd(fallback_count, 1u);
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
			(void)sl_pass<D, F, false>(c, acc, task.entry_begin + sub_lo, n_sub, listed, runs, with_stragglers, begin, end, wave, lane, smem + SL_ACC_BYTES + 256 + wave * SL_WAVE_LDS, first_h0, first_h1, group);
			__syncthreads();
			if (task.flush_atomic) {
				unsigned long long* sc = scratch + (size_t)task.scratch_begin + (size_t)sub_lo * F;
				for (uint32_t i = tid; i < sub_vals; i += SL_THREADS) {
					const long long v = acc64[i];
					if (v != 0) atomicAdd(sc + i, (unsigned long long)v);
				}
			} else {
				typedef _Float16 h2 __attribute__((ext_vector_type(2)));
				for (uint32_t i = tid; i < sub_vals / 2; i += SL_THREADS) ((h2*)gs)[i] = h2{fixed_to_half_fast(acc64[2 * i]), fixed_to_half_fast(acc64[2 * i + 1])};
			}
		}
		if (dbg_times && tid == 0 && force_wide) dbg_times[1] = dbg_times[2] = __builtin_amdgcn_s_memrealtime();
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
template <int D, int F>
__global__ void __launch_bounds__(SL_THREADS, SL_THREADS / 128) k_grid_scatter_lists(const ScatterListsArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	sl_run_task<D, F>(a, blockIdx.x, smem);
}

template <int D, int F>
void launch_lists(hipStream_t s, ScatterListsArgs a, uint32_t n_tasks) {
	static bool configured = false;
	if (!configured) { // more than 64 KiB of dynamic LDS has to be opted into once per kernel
		HIP_CHECK_THROW(hipFuncSetAttribute((const void*)k_grid_scatter_lists<D, F>, hipFuncAttributeMaxDynamicSharedMemorySize, SL_LDS_BYTES));
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
	hipLaunchKernelGGL((k_grid_scatter_lists<D, F>), dim3(n_tasks), dim3(SL_THREADS), SL_LDS_BYTES, s, a);
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
void dispatch_lists(hipStream_t s, uint32_t F, const ScatterListsArgs& a, uint32_t n_tasks) {
	switch (F) {
		case 2: return launch_lists<D, 2>(s, a, n_tasks);
		case 4: return launch_lists<D, 4>(s, a, n_tasks);
		case 8: return launch_lists<D, 8>(s, a, n_tasks);
		default: throw std::runtime_error{"grid_backward_lists: needs n_features_per_level in {2, 4, 8}"};
	}
}

} // namespace

uint32_t grid_scatter_lists_lds_bytes() { return SL_LDS_BYTES; }

size_t grid_list_gradients_bytes(const GridMeta& meta, const GridHitLists& lists) {
	return (size_t)meta.n_levels * lists.n_items * lists.item_capacity * meta.n_features_per_level * 2;
}

// Tasks of the list-fed kernel, in launch order.  No measured tuning.
//   * a level cut into chunks: one task per chunk, its single owner -- or, where a chunk is small (cheap to merge), several tasks sharing
//     the chunk, each a range of the items, merged through the scratch table;
//   * a level that is one chunk: tasks over sample ranges of >= 8192 samples, merged through the scratch table;
//   * nothing here gathers any more (round 5), so no task is tied to an XCD: the tasks are launched longest first and the dispatcher
//     packs them (round 4 kept all tasks of a record plane on the XCD whose L2 held it: the ten fine levels of BASELINE config 3a ran on
//     five XCDs in two rounds while three XCDs idled from 38 - 55 us on).
void grid_scatter_lists_plan(const GridMeta& meta, uint32_t n, std::vector<GridScatterTask>& tasks, std::vector<GridScatterRange>& shared_ranges, size_t& scratch_elems) {
	const uint32_t F = meta.n_features_per_level;
	const uint32_t rows = meta.interpolation == (uint32_t)InterpolationType::Nearest ? 1u : (1u << (meta.n_pos_dims - 1));
	tasks.clear();
	shared_ranges.clear();
	scratch_elems = 0;
	std::vector<std::pair<double, GridScatterTask>> listed, loose; // (estimated cost, task)
	uint32_t dbg_lo = 0, dbg_hi = meta.n_levels;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_SCATTER_LEVELS")) sscanf(e, "%u,%u", &dbg_lo, &dbg_hi); // profiling aid: levels lo..hi only (results are then incomplete!)
#endif
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		const GridLevel& lv = meta.levels[l];
		if (lv.scatter_binned || l < dbg_lo || l > dbg_hi) continue; // (binned: grid_backward_binned serves this level)
		const uint32_t n_chunks = lv.scatter_n_chunks, per_chunk = lv.scatter_per_chunk;
		const bool is_listed = n_chunks > 1;
		// elements (listed: one gather each, two corners) or samples (streamed, all corners) per chunk, and how many tasks share them
		const double per_chunk_work = is_listed ? (double)n * rows / n_chunks : (double)n;
		// A chunk shared by several tasks is flushed through 64-bit global atomics, one per value and task (~27 G/s chip-wide: 16 384 values
		// are ~10 us per task, and at 2^20 samples two tasks per fine chunk were 31 M atomics -- 0.53 ms against 0.26 for single owners whose
		// tasks simply take longer).  So only small chunks are ever shared.
		const bool cheap_flush = per_chunk * F <= 2048;
		// (streamed: 4096 samples x 2^D corners are as many contributions as a listed task's 8192 elements x 2 have -- the packed sums' bound is over all of them)
		const uint32_t splits = is_listed ? (cheap_flush ? (uint32_t)std::min(std::max(per_chunk_work / 16384.0 + 0.5, 1.0), 32.0) : 1u) : std::min(std::max(n / 4096u, 1u), 64u);
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
				// ... and the fewer entries a chunk has, the more of a wave's LDS adds meet at one address (measured: 18 us per task for chunks of 256 entries,
				// 14 for chunks of 8192): the slow ones first, so that they do not end the launch
				const double cost = (is_listed ? per_chunk_work / splits * 3.0 * (1.0 + std::min(1.0, 256.0 / std::max(cnt, 1u))) : (double)samples_per_split * (1.0 + rows)) + cnt * F * (shared ? 1.0 : 0.25);
				(is_listed ? listed : loose).emplace_back(cost, t);
			}
		}
	}
	listed.insert(listed.end(), loose.begin(), loose.end());
	std::stable_sort(listed.begin(), listed.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
	for (const auto& t : listed) tasks.push_back(t.second);
	while (!tasks.empty() && tasks.back().n_entries == 0) tasks.pop_back();
}

void grid_backward_lists(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, const GridScatterTask* dev_tasks, uint32_t n_tasks,
                         const GridScatterRange* dev_ranges, uint32_t n_ranges, uint64_t* scratch, uint32_t n, MatView x,
                         const void* dL_dy, uint32_t dy_stride_sample, uint32_t dy_stride_level, void* grad, const GridHitLists& lists, void* gvals, bool accumulate,
                         const MlpReduceJob* reduce_job, uint32_t* fallback_count) {
	if (n_tasks == 0) return;
	CHECK_THROW(lists.elems != nullptr && lists.sidx != nullptr && lists.heads != nullptr && lists.stragglers != nullptr && lists.counts != nullptr && lists.n_items > 0 && gvals != nullptr);
	CHECK_THROW(lists.item_capacity % 8 == 0 && lists.item_samples <= 65536 && lists.item_samples * meta.n_features_per_level * 2 * 2 <= 64 * 1024); // (k_grid_list_gradients: the slices of two items in LDS)
	CHECK_THROW(n > 0 && n <= grid_hit_max_samples(meta) && meta.hash_type != (uint32_t)HashType::Rng);
	// 32-bit byte offsets into a level's pool and into its dL/dy plane
	CHECK_THROW((uint64_t)lists.n_items * lists.item_capacity * std::max<uint64_t>(GRID_HIT_WORDS * 4u, meta.n_features_per_level * 2u) < (1ull << 31));
	// every chunk must fit the 64 KiB of packed accumulators (the plan's chunks are cut for 128 KiB of 64-bit ones: the same entry count), and the elements' 16-bit entries
	for (uint32_t l = 0; l < meta.n_levels; ++l) {
		CHECK_THROW(meta.levels[l].scatter_binned || (meta.levels[l].scatter_per_chunk * meta.n_features_per_level * 4 <= SL_ACC_BYTES && meta.levels[l].scatter_per_chunk < 0xffffu));
	}
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
	a.gvals = (const half_t*)gvals;
	a.scratch = (unsigned long long*)scratch;
	a.accumulate_mode = accumulate ? 1 : 0;
	a.force_wide = switches().scatter_wide ? 1 : 0; // tests: every task through the 64-bit passes
	a.dbg_times = nullptr;
	a.fallback_count = fallback_count;
#ifdef TCNN_AMD_DEV
	if (const char* e = getenv("TCNN_AMD_SCATTER_DEV")) a.dev_flags = (uint32_t)atoi(e);
#endif
	{ // dL/dy into list order, then the owners
		ListGradArgs lg{dev_meta, lists, (const half_t*)dL_dy, dy_stride_sample, dy_stride_level, n, (half_t*)gvals};
		switch (meta.n_features_per_level) {
			case 2: launch_list_gradients<2>(stream, lg, meta.n_levels); break;
			case 4: launch_list_gradients<4>(stream, lg, meta.n_levels); break;
			case 8: launch_list_gradients<8>(stream, lg, meta.n_levels); break;
			default: throw std::runtime_error{"grid_backward_lists: needs n_features_per_level in {2, 4, 8}"};
		}
	}
	switch (meta.n_pos_dims) {
		case 2: dispatch_lists<2>(stream, meta.n_features_per_level, a, n_tasks); break;
		case 3: dispatch_lists<3>(stream, meta.n_features_per_level, a, n_tasks); break;
		default: throw std::runtime_error{"grid_backward_lists: 2 or 3 input dims"};
	}
	grid_scatter_finalize(stream, dev_ranges, n_ranges, scratch, grad, accumulate, reduce_job);
}

} // namespace tcnn_amd
