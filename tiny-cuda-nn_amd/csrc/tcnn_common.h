// tcnn_common.h -- shared enums, error macros and kernel-launcher declarations of libtcnn_amd (gfx950 only).
//
// Names follow the reference's vocabulary (include/tiny-cuda-nn/common.h:112-170) so that the host object model in
// module.cpp reads like the reference's; the numbering is ours.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace tcnn_amd {

enum class Activation : uint32_t { None = 0, ReLU = 1, LeakyReLU = 2, Exponential = 3, Sine = 4, Sigmoid = 5, Squareplus = 6, Softplus = 7, Tanh = 8 };
enum class GridType : uint32_t { Hash = 0, Dense = 1, Tiled = 2 };
enum class HashType : uint32_t { Prime = 0, CoherentPrime = 1, ReversedPrime = 2, Rng = 3 };
enum class InterpolationType : uint32_t { Nearest = 0, Linear = 1, Smoothstep = 2 };
enum class LossType : uint32_t { L2 = 0, RelativeL2 = 1, L1 = 2, RelativeL1 = 3, Mape = 4, Smape = 5, CrossEntropy = 6, Variance = 7, RelativeL2Luminance = 8 }; // src/loss.cu:57-65
enum class Precision : uint32_t { Fp32 = 0, Fp16 = 1 };      // cpp_api.h:69-72
enum class GradientMode : uint32_t { Ignore = 0, Overwrite = 1, Accumulate = 2 }; // common.h GradientMode

constexpr uint32_t BATCH_SIZE_GRANULARITY = 256; // common.h:235
constexpr float LOSS_SCALE_FP16 = 128.0f;        // common.h:232
constexpr uint32_t MAX_N_LEVELS = 128;           // grid_interface.h:84
constexpr uint32_t MAX_MLP_LAYERS = 16;

#define TCNN_STR2(x) #x
#define TCNN_STR(x) TCNN_STR2(x)
#define CHECK_THROW(x) \
	do { if (!(x)) throw std::runtime_error{std::string{__FILE__ ":" TCNN_STR(__LINE__) " check failed: " #x}}; } while (0)
#define HIP_CHECK_THROW(x) \
	do { hipError_t _e = (x); if (_e != hipSuccess) throw std::runtime_error{std::string{__FILE__ ":" TCNN_STR(__LINE__) " " #x " failed: "} + hipGetErrorString(_e)}; } while (0)

inline uint32_t div_round_up(uint32_t v, uint32_t d) { return (v + d - 1) / d; }

// The A/B switches of the training step and of inference (DESIGN.md "Switches"): every one defaults to the fast path, none changes a result
// beyond what is stated there.  They are read from the environment when create_from_config / a module constructor runs (switches_reload,
// capi.cpp) -- not per step -- and the set is PROCESS-WIDE: creating a model under another environment changes the kernels of every live
// model from its next call on (reads and the reload are serialised by a lock).  A process that wants another setting sets the variable and
// creates its models afterwards; tests that compare two settings build one model at a time.
struct Switches {
	bool grid_planes = true;      // TCNN_AMD_GRID_PLANES=0: the AoS forward kernel inside the fused training step
	bool grid_rows_planes = true; // TCNN_AMD_GRID_ROWS_PLANES=0: callers that want the encoded batch as a matrix get k_grid_fwd (AoS) instead of the plane kernel + a transposition
	bool grid_scatter_lds = true; // TCNN_AMD_GRID_SCATTER=atomic: the reference-shaped global-atomic gradient kernel
	bool scatter_records = true;  // TCNN_AMD_SCATTER_RECORDS=0: gradient planes instead of {coordinates, gradient} records
	bool scatter_tune = true;     // TCNN_AMD_SCATTER_TUNE=0: the untuned task list of k_grid_scatter
	int scatter_lists = -1;       // TCNN_AMD_SCATTER_LISTS=0 / 1: never / wherever possible (unset: where it pays, grid_scatter_prefers_lists)
	bool scatter_wide = false;    // TCNN_AMD_SCATTER_WIDE=1: every task of k_grid_scatter_lists through its 64-bit passes (tests)
	bool fused_step = true;       // TCNN_AMD_FUSED_STEP=0: forward / loss / backward / wgrad kernels instead of the fused step
	bool side_jobs = true;        // TCNN_AMD_SIDE_JOBS=0: k_mlp_prep and the slab reduction as launches of their own
	bool live_image = true;       // TCNN_AMD_LIVE_IMAGE=0: k_mlp_prep every step
	bool adam_steps32 = false;    // TCNN_AMD_ADAM_STEPS32=1: uint32 update counts from the start
	bool adam_in_flush = false;   // laboratory build only (TCNN_AMD_ADAM_IN_FLUSH=1): Adam applied by k_grid_scatter's chunk owners -- measured slower, not in the product
	bool adam_in_reduce = true;   // TCNN_AMD_ADAM_IN_REDUCE=0: k_adam as a launch of its own for models without encoding parameters
	bool adam_prologue = true;    // TCNN_AMD_ADAM_PROLOGUE=0: the scatter's finalize pass (+ slab reduction) as a launch of its own in front of k_adam
	bool adam_prologue_refused = false; // TCNN_AMD_ADAM_PROLOGUE=refuse (tests): the optimizer is offered the prologue and turns it down, as it does for shapes its launch does not take
	bool mlp_r32 = true;          // TCNN_AMD_MLP_R32=0: k_mlp_train_regs / k_mlp_train instead of the 32x32x16 kernels
	int mlp_r32a = -1;            // TCNN_AMD_MLP_R32A=0 / 1: k_mlp_train_r32 / k_mlp_train_r32a whatever the batch size
	bool mlp_regs = true;         // TCNN_AMD_MLP_REGS=0: the LDS-image kernels of k_train.hip
	bool mlp_fast = true;         // TCNN_AMD_MLP_FAST=0: k_mlp_train_regs with run-time formats
	uint32_t mlp_prio = 1;        // TCNN_AMD_MLP_PRIO: wave priorities of the MLP kernels (0 none, 1 alternating per trip, 2, 3)
};
Switches switches(); // a copy of the process-wide set, taken under its lock
void switches_reload();
inline uint32_t next_multiple(uint32_t v, uint32_t d) { return div_round_up(v, d) * d; }

// ------------------------------------------------------------------------------------------------------------------
// Grid encoding: per-level table precomputed on the HOST (so host and device agree bit-for-bit on scale/resolution,
// SURVEY 7 "hard parts"); the encoding keeps one device copy (6 KiB, too large for the kernarg segment).
// ------------------------------------------------------------------------------------------------------------------
struct GridLevel {
	uint32_t offset;      // first entry of the level (in entries, grid.h:714)
	uint32_t size;        // hashmap_size = entries in the level
	float    scale;       // grid_scale(level) (common_device.h:709-714)
	uint32_t hashed;      // 1: index = hash(cell) ; 0: index = sum cell_d * stride[d]   (common_device.h:690-707)
	uint32_t stride[4];   // per-dim stride of the dense index INCLUDING the uint32 wrap-around / early-exit behaviour
	uint32_t size_mask;   // size-1 if size is a power of two, else 0 (then a real modulo is used)
	// LDS owner-computes scatter (k_grid_bwd_lds): the level's table is cut into scatter_n_chunks chunks of scatter_per_chunk entries
	uint32_t scatter_per_chunk;
	uint32_t scatter_shift;    // log2(scatter_per_chunk) if it is a power of two, else 0xffffffff
	uint32_t scatter_n_chunks;
	uint32_t scatter_binned;   // 1: too many chunks for the sample filter -- the level's gradients go through k_grid_bin.hip
};

struct GridMeta {
	uint32_t n_pos_dims;
	uint32_t n_features_per_level;
	uint32_t n_levels;
	uint32_t grid_type;
	uint32_t hash_type;
	uint32_t interpolation;
	uint32_t primes[4];
	GridLevel levels[MAX_N_LEVELS];
};

// input matrix addressing: element (dim d, sample i) at data[i * stride_sample + d * stride_dim]
struct MatView {
	const float* data;
	uint32_t stride_sample;
	uint32_t stride_dim;
};
struct MatViewMut {
	float* data;
	uint32_t stride_sample;
	uint32_t stride_dim;
};

// half data travels as void* on the host side
// chunk_mask (optional, uint64 [n_levels][n][GRID_FILTER_MAX_CHUNKS / 64]): bit c set <=> the sample touches scatter chunk c of that level (filter for k_grid_scatter)
void grid_forward(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, bool fp32, uint32_t n, MatView x, const void* grid, void* out, uint32_t out_stride, float* dy_dx,
                  uint64_t* chunk_mask);
// ---- training-step forward (k_grid_planes.hip): half, F >= 2, D in {2, 3}; level-major and XCD-aware.
// out_planes: half [n_levels][n][F]; chunk_bits (optional): uint64 [n_levels][GRID_FILTER_MAX_CHUNKS][n / 64], written for levels with 2 .. GRID_FILTER_MAX_CHUNKS scatter chunks.
bool grid_planes_supported(const GridMeta& meta, uint32_t n);
bool grid_planes_to_rows_supported(const GridMeta& meta, uint32_t n, uint32_t width);
// planes [width / F][n][F] -> the first `width` columns of rows [n][row_stride] and back (halves; the callers that want the encoded batch as a
// matrix, k_grid_planes.hip; width: the levels' features, or the padded width with its planes of zeros)
void grid_planes_to_rows(hipStream_t stream, const GridMeta& meta, uint32_t n, uint32_t width, const void* planes, void* rows, uint32_t row_stride);
void grid_rows_to_planes(hipStream_t stream, const GridMeta& meta, uint32_t n, uint32_t width, const void* rows, uint32_t row_stride, void* planes);
uint32_t grid_planes_spt(const GridMeta& meta);        // samples per thread of the kernel shape used for this grid
void grid_planes_plan(const GridMeta& meta, uint32_t n, bool hit_lists, std::vector<uint32_t>& work, uint32_t& max_items, uint32_t& blocks_per_xcd); // hit_lists: the shape of the kernel that writes them (larger work items)
// prep_job (optional, mlp_side_jobs.h; passed to the kernel by value): the kernel also builds the MLP's fragment images
struct MlpPrepJob;
// ---- hit lists (round 4; round 5: elements that carry their entries and weights, gradients transposed into list order).
// The scatter's sample filter as a STREAM instead of bit planes to be scanned.  For every level cut into 2 .. GRID_FILTER_MAX_CHUNKS chunks
// the forward kernel writes, per work item (grid_hit_item_samples() consecutive samples of one level), the item's elements SORTED BY
// CHUNK into the item's own region of the level's pool, plus 65 offsets: where each chunk's run starts inside the region ([64]: how
// many elements the item has).  An element is one CELL ROW of one sample -- the two corners that differ in dimension 0 only -- and
// carries everything the chunk's owner needs but the gradient:
//     elems [..][0]: entry of corner A (cell_0) | entry of corner B (cell_0 + 1) << 16, both relative to the chunk's first entry;
//     elems [..][1]: (half) weight of A | (half) weight of B << 16 -- the fp32 products of grid.h:147-160 rounded to half, as grid.h:254 uses them
//                    (no corner B here -- Nearest interpolation, or B lies in another chunk --: A's entry again with weight +0, so that
//                    the owner adds both corners of every element without a test: a product with +0 adds nothing);
//     sidx  [..]   : the sample, relative to the item's first one (16 bits).
// After the MLP kernel a streaming pass (k_grid_list_gradients, k_grid_scatter_lists.hip) brings dL/dy into the same order: per item it
// loads the item's slice of the level's gradient plane into LDS (2 KB) and writes gvals [..] = dL/dy of element's sample, position
// for position.  The chunk's owner then walks the items' runs of its chunk reading elems and gvals side by side -- dense loads, no
// gather, no coordinates, no pos_fract, no hash.  (Round 4: 4-byte elements {sample, corner bits}; the owner gathered a 16-byte record
// per element from a 4 MB plane per level pair.  Each such gather costs the CU a whole 128-byte line from its L2, ~3 clocks per lane
// whatever the bytes used -- 40 us of the kernel's 59 - 66 at 2^18 samples, profiles/r05_scatter_timeline.txt -- and needs the plane
// resident in the XCD's L2, which tied the form to batches of 2^17 .. 2^19.)  No counters, no atomics: a region's place is the item's
// number.  The one exception: a row whose two corners fall into different chunks (one in ~8000 on hashed levels) sends corner B to the
// level's straggler list {sample | corner bit << grid_hit_mask_shift, chunk}, which every owner of the level scans and evaluates from the coordinates.
inline constexpr uint32_t grid_hit_mask_shift(uint32_t n_pos_dims) { return 32u - (1u << n_pos_dims); } // 28 (2-D), 24 (3-D): stragglers only
constexpr uint32_t GRID_HIT_COUNT_STRIDE = 64;   // uint32 per level between the straggler counts (one memory channel each)
constexpr uint32_t GRID_HIT_HEADS = 65;          // offsets per item: GRID_FILTER_MAX_CHUNKS + 1
constexpr uint32_t GRID_HIT_WORDS = 2;           // uint32 per element
struct GridHitLists {
	uint32_t* elems = nullptr;       // [n_levels][n_items][item_capacity][GRID_HIT_WORDS]
	uint16_t* sidx = nullptr;        // [n_levels][n_items][item_capacity]
	uint32_t* heads = nullptr;       // [n_levels][GRID_HIT_HEADS][n_items]: chunk-major, so that an owner reads its chunk's offsets of consecutive items with dense loads
	uint32_t* stragglers = nullptr;  // [n_levels][straggler_capacity][2]: {sample | corner bit, chunk}
	uint32_t* counts = nullptr;      // [n_levels][GRID_HIT_COUNT_STRIDE]: stragglers per level; all zero when the forward kernel starts
	uint32_t* zero_counts = nullptr; // the counter set of the NEXT forward launch on this stream: zeroed by this one
	uint32_t n_items = 0, item_samples = 0, item_capacity = 0, straggler_capacity = 0;
	uint32_t dev_flags = 0;          // laboratory build only (TCNN_AMD_FWD_LISTS_DEV): timing-only variants of the list output
};
uint32_t grid_hit_item_samples(const GridMeta& meta); // samples per work item of the forward kernel shape used for this grid (k_grid_planes.hip)
// the kernels address the pool (12 bytes x 2^(D-1) rows per sample and level) and the straggler masks with 32 bits
inline uint32_t grid_hit_max_samples(const GridMeta& meta) { return meta.n_pos_dims <= 2 ? (1u << 24) : (1u << 22); }
// hit_lists (optional, instead of chunk_bits): see above
void grid_forward_planes(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, const uint32_t* dev_work, uint32_t max_items, uint32_t blocks_per_xcd, uint32_t n,
                         MatView x, const void* grid, void* out_planes, uint64_t* chunk_bits, const MlpPrepJob* prep_job = nullptr, const GridHitLists* hit_lists = nullptr);
// reference-shaped gradient scatter with global float atomics (fp32 grids, F == 1, tables too large for the LDS scheme).
// grad: T[n_params] accumulated in place (caller zeroes it).  For F == 1 && !fp32 the caller passes an fp32 scratch as `grad`.
void grid_backward(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, bool fp32_grad, uint32_t n, MatView x, const void* dL_dy, bool dy_fp32, uint32_t dy_stride, void* grad);

// ---- LDS owner-computes scatter (k_grid_scatter.hip): exact 64-bit fixed-point accumulation, one task per workgroup
struct GridScatterTask {
	uint32_t level;
	uint32_t entry_begin, n_entries;      // owned chunk of the level's table (n_entries == 0: padding task)
	uint32_t sample_begin, sample_end;    // samples examined by this task
	uint32_t flush_atomic;                // 1: several tasks share the chunk -> merge into the scratch table, finalize rounds
	uint32_t scratch_begin;               // first element of the chunk inside the scratch table (shared chunks only)
	uint32_t pad;
};
struct GridScatterRange { size_t grad_begin; uint32_t n_elems; uint32_t scratch_begin; uint32_t pad; }; // shared chunks, for the finalize pass
struct AdamInFlush;                                          // below, with the optimizer kernels
struct MlpReduceJob;                                         // mlp_side_jobs.h
typedef std::vector<std::pair<size_t, size_t>> ParamRanges; // sorted, disjoint [begin, end) of the parameter vector
constexpr uint32_t GRID_FILTER_MAX_CHUNKS = 64;     // chunks per level the sample filter can describe (bit planes per level)
uint32_t grid_scatter_max_chunks();                 // = GRID_FILTER_MAX_CHUNKS
void grid_scatter_setup_levels(GridMeta& meta);     // fills GridLevel::scatter_* (how each level's table is cut into chunks)
bool grid_scatter_prefers_lists(const GridMeta& meta); // the grid has levels of many chunks: hit lists (k_grid_scatter_lists.hip) instead of bit planes
// Plans the task list for a batch of n samples (half gradients, F >= 2).
// measured_level_us (optional): per-level workgroup time of a first launch (grid_scatter_level_costs) -> tuned task sizes
void grid_scatter_plan(const GridMeta& meta, uint32_t n, std::vector<GridScatterTask>& tasks, std::vector<GridScatterRange>& shared_ranges, size_t& scratch_elems,
                       const std::vector<float>* measured_level_us = nullptr);
// times: uint64[tasks.size()][8] copied back from grid_backward_lds(task_times)
std::vector<float> grid_scatter_level_costs(const GridMeta& meta, const std::vector<GridScatterTask>& tasks, const std::vector<uint64_t>& times);
// chunk_mask [n_levels][n][GRID_FILTER_MAX_CHUNKS / 64] uint64 -> chunk_bits [n_levels][GRID_FILTER_MAX_CHUNKS][n / 64] uint64 (one ballot word per 64 samples per (level, chunk))
void grid_mask_to_bits(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, uint32_t n, const uint64_t* chunk_mask, uint64_t* chunk_bits);
// dL_dy element (sample i, level l, feature f) at dL_dy[i * dy_stride_sample + l * dy_stride_level + f].
// chunk_bits: optional filter derived from grid_forward's masks for the SAME batch (n samples); nullptr -> every sample is examined in full.
// scratch: uint64[scratch_elems], zero on entry, zero again on return.  Writes EVERY gradient element (no memset needed);
// accumulate = GradientMode::Accumulate.
void grid_backward_lds(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, const GridScatterTask* dev_tasks, uint32_t n_tasks,
                       const GridScatterRange* dev_ranges, uint32_t n_ranges, uint64_t* scratch, uint32_t n, MatView x,
                       const void* dL_dy, uint32_t dy_stride_sample, uint32_t dy_stride_level, void* grad, const uint64_t* chunk_bits, bool accumulate, bool dy_records = false,
                       uint64_t* task_times = nullptr, // task_times (optional): device uint64[n_tasks][8], per-task timestamps for the plan tuner
                       const AdamInFlush* adam = nullptr, // adam (optional, record form only): arrays indexed like grad; see grid_scatter_adam_ranges
                       const MlpReduceJob* reduce_job = nullptr); // (optional) carried by the finalize launch when there is one; ->taken says so
// ---- the same scatter fed by hit lists (k_grid_scatter_lists.hip): two workgroups per CU with 64 KiB of accumulators each -- both features
// of an entry in ONE 64-bit LDS add as 2 x int32 while a per-task bound proves that no half can overflow, 64-bit accumulators in two
// passes otherwise -- tasks of its own plan (GridScatterTask::pad = split s | n_splits << 16: which share of a chunk's list), same scratch, same results.
uint32_t grid_scatter_lists_lds_bytes();
// the kernel's tasks in launch order (one workgroup each; block b runs on XCD b % 8: k_grid_scatter_lists.hip)
void grid_scatter_lists_plan(const GridMeta& meta, uint32_t n, std::vector<GridScatterTask>& tasks, std::vector<GridScatterRange>& shared_ranges, size_t& scratch_elems);
void grid_backward_lists(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, const GridScatterTask* dev_tasks, uint32_t n_tasks,
                         const GridScatterRange* dev_ranges, uint32_t n_ranges, uint64_t* scratch, uint32_t n, MatView x,
                         const void* dL_dy, uint32_t dy_stride_sample, uint32_t dy_stride_level, void* grad, const GridHitLists& lists, void* gvals, bool accumulate,
                         const MlpReduceJob* reduce_job, uint32_t* fallback_count = nullptr);
// gvals: workspace of grid_list_gradients_bytes(): dL/dy in list order, half [n_levels][n_items][item_capacity][F], written by the launch's first kernel
size_t grid_list_gradients_bytes(const GridMeta& meta, const GridHitLists& lists);
// the finalize pass of the shared chunks (+ the MLP's slab reduction), shared by both scatter kernels
void grid_scatter_finalize(hipStream_t stream, const GridScatterRange* dev_ranges, uint32_t n_ranges, uint64_t* scratch, void* grad, bool accumulate, const MlpReduceJob* reduce_job);
// the parameter ranges (relative to grad) a launch of `tasks` with `adam` updates itself; empty = this plan cannot carry the optimizer step
ParamRanges grid_scatter_adam_ranges(const GridMeta& meta, const std::vector<GridScatterTask>& tasks, bool dy_records);
// dy_records: dL_dy is float4 [grid_scatter_record_planes()][n] scatter records {coordinates, gradient halves} (mlp_device.h
// store_dx_record; D = 2 with F = 2 packs two levels into one record); x is then not read
bool grid_scatter_records_supported(const GridMeta& meta);
uint32_t grid_scatter_record_planes(const GridMeta& meta);

// ---- PPNG1 (k_ppng.hip; encodings/ppng_1.h): features half [F][2][3][C][Q][R]; out / dL_dy AoS with row stride out_stride;
// scratch: uint64[n_params], zero on entry, zero again on return (exact integer sums of the fp16 products, rounded once)
void ppng1_forward(hipStream_t stream, bool fp32_out, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min_freq, int32_t log2_max_freq, MatView x,
                   const void* features, void* out, uint32_t out_stride);
void ppng1_backward(hipStream_t stream, bool fp32_dy, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min_freq, int32_t log2_max_freq, MatView x,
                    const void* features, const void* dL_dy, uint32_t dy_stride, uint64_t* scratch, void* grad, bool accumulate);

// PPNG2 (encodings/ppng_2.h): features half [F][2][3][C][Q][Q][R], otherwise as PPNG1
void ppng2_forward(hipStream_t stream, bool fp32_out, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min_freq, int32_t log2_max_freq, MatView x,
                   const void* features, void* out, uint32_t out_stride);
void ppng2_backward(hipStream_t stream, bool fp32_dy, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, uint32_t R, int32_t log2_min_freq, int32_t log2_max_freq, MatView x,
                    const void* features, const void* dL_dy, uint32_t dy_stride, uint64_t* scratch, void* grad, bool accumulate);
// PPNG3 (encodings/ppng_3.h): features half [F][2][Q^3][C] (cell = p_0 + Q p_1 + Q^2 p_2), C in {2, 4, 8}; dL_dx is written (not accumulated)
void ppng3_forward(hipStream_t stream, bool fp32_out, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min_freq, int32_t log2_max_freq, MatView x, const void* features, void* out,
                   uint32_t out_stride);
size_t ppng3_backward_workspace_bytes(uint32_t n, uint32_t F, uint32_t Q, uint32_t C); // the third coordinate's bins of every (f, s, sample); 0 when the layers do not fit the LDS
void ppng3_backward(hipStream_t stream, bool fp32_dy, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min_freq, int32_t log2_max_freq, MatView x, const void* dL_dy,
                    uint32_t dy_stride, void* workspace, uint64_t* scratch, void* grad, bool accumulate);
void ppng3_backward_input(hipStream_t stream, bool fp32_dy, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min_freq, int32_t log2_max_freq, MatView x, const void* features,
                          const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx);
// second order (ppng_3.h:609-676): grad (nullable; scratch as above), dL_ddLdy (nullable, [n][dy_stride]) and dL_dx (nullable, written)
void ppng3_backward_backward_input(hipStream_t stream, bool fp32_dy, uint32_t n, uint32_t F, uint32_t Q, uint32_t C, int32_t log2_min_freq, int32_t log2_max_freq, MatView x, MatView dL_ddLdx,
                                   const void* features, const void* dL_dy, uint32_t dy_stride, uint64_t* scratch, void* grad, bool accumulate, void* dL_ddLdy, MatViewMut* dL_dx);

// ---- binned form for levels cut into more than 64 chunks (k_grid_bin.hip; GridLevel::scatter_binned): no filter, no gathers.
// Same exact result as grid_backward_lds; writes every gradient element of the binned levels.  workspace: grid_bin_workspace_bytes().
bool grid_bin_supported(const GridMeta& meta); // F in {2, 4}
uint32_t grid_bin_max_chunks();                // chunks per level (4096)
uint32_t grid_bin_acc_bytes();                 // LDS accumulators per workgroup = chunk size of binned levels
size_t grid_bin_workspace_bytes(const GridMeta& meta, uint32_t n);
void grid_backward_binned(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, uint32_t n, MatView x, const void* dL_dy, uint32_t dy_stride_sample,
                          uint32_t dy_stride_level, void* grad, bool accumulate, void* workspace,
                          uint32_t* fallback_count = nullptr); // (optional, device) chunks whose packed 32-bit sums could not be proven and were added again in 64 bits
void grid_backward_input(hipStream_t stream, const GridMeta& meta, bool fp32, uint32_t n, const void* dL_dy, uint32_t dy_stride, const float* dy_dx, MatViewMut dL_dx);
// second-order input gradients (k_grid_bwdbwd.hip; grid.h:352-650): each of grad (accumulated in place, GT = float if fp32_grad),
// dL_ddLdy (T [n][dy_stride], needs dy_dx) and dL_dx (overwritten, needs grid) is optional
void grid_backward_backward_input(hipStream_t stream, const GridMeta& meta, const GridMeta* dev_meta, bool fp32, bool fp32_grad, uint32_t n, MatView x, MatView dL_ddLdx, const void* dL_dy,
                                  uint32_t dy_stride, const void* grid, const float* dy_dx, void* grad, void* dL_ddLdy, MatViewMut* dL_dx);

// OneBlob / Identity (AoS output, T = half or float)
void oneblob_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, uint32_t n_bins, MatView x, void* out, uint32_t out_stride);
void oneblob_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, uint32_t n_bins, MatView x, const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx);
void identity_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, float scale, float offset, MatView x, void* out, uint32_t out_stride);
void identity_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, float scale, const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx);
// Frequency / TriangleWave (k_encodings.hip): dy_dx (optional) float [n][n_dims * outputs_per_input], consumed by the backward pass
void periodic_forward(hipStream_t stream, bool triangle, bool fp32, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, MatView x, void* out, uint32_t out_stride, float* dy_dx);
void periodic_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t n_dims, uint32_t outputs_per_input, const void* dL_dy, uint32_t dy_stride, const float* dy_dx, MatViewMut dL_dx);
// Composite encoding reductions (composite.h:47-133): `in` holds the nested outputs [n_nested][n_elems], T = float if fp32 else half
void composite_reduce_forward(hipStream_t stream, bool fp32, bool product, size_t n_elems, uint32_t n_nested, const void* in, void* out);
void composite_reduce_backward(hipStream_t stream, bool fp32, bool product, size_t n_elems, uint32_t n_nested, const void* in, const void* dL_dout, void* dL_din);
// SphericalHarmonics: degree^2 outputs, the padding columns FIRST (spherical_harmonics.h:58-64)
void sh_forward(hipStream_t stream, bool fp32, uint32_t n, uint32_t degree, MatView x, void* out, uint32_t out_stride);
void sh_backward_input(hipStream_t stream, bool fp32, uint32_t n, uint32_t degree, MatView x, const void* dL_dy, uint32_t dy_stride, MatViewMut dL_dx);

// ------------------------------------------------------------------------------------------------------------------
// Fully fused MLP.  Weight matrices are row-major [fan_out][fan_in] half, contiguous (fully_fused_mlp.cu:656-671).
// The kernels consume "fragment images": the weights pre-permuted into MFMA A-operand order (see k_mlp.hip).
// ------------------------------------------------------------------------------------------------------------------
struct MlpLayer {
	uint32_t rows, cols;    // fan_out, fan_in
	uint32_t w_off;         // element offset of the matrix inside the parameter vector
	uint32_t fwd_off;       // first fragment of the forward image (A = W)
	uint32_t bwd_off;       // first fragment of the backward image (A = W^T)
	uint32_t ks_fwd;        // k-steps (of 32) of the forward product = ceil(cols / 32)
	uint32_t ks_bwd;        // k-steps of the backward product = ceil(rows / 32)
	uint32_t natural_k;     // 1: forward fragments use natural k order (layer 0, B operand loaded from memory)
};

struct MlpDesc {
	uint32_t in_width, width, out_width, n_hidden, n_layers;
	uint32_t activation, output_activation;
	uint32_t n_frags_fwd, n_frags_bwd;
	uint32_t n_frags_r32;   // third section of the image: fragments for v_mfma_f32_32x32x16_f16 (k_train_r32.hip; mlp_side_jobs.h R32Frags), 0: none
	MlpLayer layers[MAX_MLP_LAYERS];
};

size_t mlp_image_bytes(const MlpDesc& d);   // bytes of the fwd + bwd (+ r32) images
// params (half, row-major matrices) -> images.  image = [fwd frags][bwd frags], 1 KiB per fragment.
void mlp_prepare_weights(hipStream_t stream, const MlpDesc& d, const void* params, void* image, bool want_bwd);
// x: [n][in_width] half AoS; out: [n][out_width] half; hidden (optional): [n_hidden][n][width] half post-activation
void mlp_forward(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, void* out, void* hidden);
// The same kernel with its input / output conversions fused in (inference path):
//   input : x_half AoS [n][in_width], or level planes [in_width / F][n][F] (x_plane_features = F), or -- x_f32.data != nullptr --
//           the float matrix itself with the Identity encoding applied on the fly ((half)(x * scale + offset), padding = 1), or
//           with the OneBlob encoding applied on the fly (x_oneblob_bins);
//   output: out_half [n][out_width] and / or out_f32: the first out_f32_dims outputs as floats (trim_and_cast, object.cu:61-67).
struct MlpOneBlobInput { MatView x; uint32_t n_dims, n_bins; }; // coordinates [n][n_dims] (any layout), n_bins a power of two >= 32
struct MlpIo {
	const void* x_half;
	uint32_t x_plane_features;
	MatView x_f32;
	uint32_t x_f32_dims;
	float x_scale, x_offset;
	uint32_t x_oneblob_bins; // > 0 (a power of two >= 32): x_f32 holds coordinates and the network's input is their OneBlob encoding, evaluated in the kernel
	void* out_half;
	MatViewMut out_f32;
	uint32_t out_f32_dims;
};
void mlp_forward_io(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const MlpIo& io, void* hidden);
// Reference-shaped backward: dL_dout [n][out_width]; hidden from mlp_forward; writes dhidden [n_hidden][n][width] and (optional) dL_dx [n][in_width]
// dx_plane_features = 0: dL_dx is AoS [n][in_width]; = F > 0: "level planes" [in_width / F][n][F] (what the grid scatter reads)
void mlp_backward(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* dL_dout, const void* out, const void* hidden, void* dhidden, void* dL_dx,
                  uint32_t dx_plane_features);
// ---- the Trainer's fused step (k_train.hip): forward + loss + backward + weight gradients in one kernel.
// Supported when out_width <= 32, width in {64, 128} and the activations of one trip fit in LDS; else use the pieces above.
bool mlp_train_fused_supported(const MlpDesc& d, uint32_t n);
uint32_t mlp_train_fused_grid(const MlpDesc& d, uint32_t n, uint32_t oneblob_bins = 0, uint32_t oneblob_dims = 0); // workgroups = number of weight-gradient slabs (oneblob_*: the encoding is evaluated inside the kernel)
// x [n][in_width] half (x_plane_features = 0) or level planes [in_width / F][n][F] (x_plane_features = F in {2, 4, 8}).
// target / data_pdf [n][dims] float or external_dL_dy [n][out_width] half (loss-scaled).
// compact_context (only where mlp_train_regs_supported() and slabs != nullptr): dL_dout and L are the COMPACT matrices [n][dims]
// (the live columns of the padded ones; mlp_expand_context pads them), else the padded [n][out_width] ones.
// Writes out, dL_dout, L ([n][out_width]; dL_dout and L only without external_dL_dy), dL_dx (optional; AoS or level planes),
// and -- if slabs != nullptr -- one fp32 slab of partial weight gradients per workgroup: slabs[grid][n_params].
// dx_record_x != nullptr (with dx_plane_features = F): dL_dx is written as 16-byte scatter records {coordinates (dx_record_dims
// floats, read from dx_record_x [n][dims]), gradient halves}, float4 [in_width / F][n] or, for 2 dims and F = 2, [in_width / 4][n]
// with two levels per record (mlp_device.h store_dx_record); needs 4 dims + 2 F <= 16.
void mlp_train_fused(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, uint32_t x_plane_features, const float* target, const float* data_pdf,
                     const void* external_dL_dy, uint32_t dims, LossType loss, float loss_scale, void* out, void* dL_dout, float* L, bool compact_context, void* dL_dx,
                     uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims, float* slabs, uint32_t n_params, const MlpOneBlobInput* oneblob = nullptr);
// oneblob (optional; x is then not read): the network's input is the OneBlob encoding of these coordinates, evaluated inside the kernel
// (mlp_train_fused_oneblob_supported says whether this network / batch has such a kernel)
bool mlp_train_fused_oneblob_supported(const MlpDesc& d, uint32_t n, uint32_t n_bins);
// ---- the same step for (16 | 32) -> 64 -> [64 ->] 16 networks with everything in registers (k_train_regs.hip): no LDS images, no
// barriers, transposes on the matrix cores.  mlp_train_fused* dispatch to it when it applies (TCNN_AMD_MLP_REGS=0: never).
// Writes dL_dout and L as compact matrices [n][dims] (96 of the 256 bytes per sample the padded ones would add to the kernel's
// stores are zeros); mlp_expand_context produces the reference's [n][16] matrices from them.  Requires slabs != nullptr.
bool mlp_train_regs_supported(const MlpDesc& d, uint32_t n);
uint32_t mlp_train_regs_grid(const MlpDesc& d, uint32_t n);
void mlp_train_regs(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, uint32_t x_plane_features, const float* target, const float* data_pdf,
                    const void* external_dL_dy, uint32_t dims, LossType loss, float loss_scale, void* out, void* compact_dL_dout, float* compact_L, void* dL_dx,
                    uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims, float* slabs, uint32_t n_params);
// ---- the same step for 32 -> 64 -> 64 -> 16 networks fed by a 2-D grid encoding with 2 features per level, on the 32x32x16 matrix
// instruction (k_train_r32.hip): 32 samples per wave and trip, operands of the weight-gradient products transposed through wave-private
// LDS images.  mlp_train_regs dispatches to it (TCNN_AMD_MLP_R32=0: never); `grid` workgroups write one slab each.
bool mlp_train_r32_applies(const MlpDesc& d, uint32_t n, uint32_t x_plane_features, const float* data_pdf, const void* external_dL_dy, uint32_t dims, LossType loss, const void* out,
                           const void* dL_dx, uint32_t dx_plane_features, const float* dx_record_x, uint32_t dx_record_dims);
uint32_t mlp_train_r32_grid(uint32_t n); // workgroups (= slabs) of the launch: the caller sizes `slabs` with it
void mlp_train_r32(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, const float* target, uint32_t dims, LossType loss, float loss_scale, void* out,
                   void* compact_dL_dout, float* compact_L, void* dL_dx, const float* dx_record_x, float* slabs, uint32_t n_params, uint32_t grid);
// ---- BASELINE config 2's step, OneBlob(64 bins, 2 dims) -> 64 -> 64 -> 16 with the encoding evaluated in the kernel, on the 32x32x16
// matrix instruction (k_train_r32ob.hip): one wave per SIMD, weight-gradient accumulators in AGPRs.  mlp_train_fused dispatches to it.
bool mlp_train_r32ob_shape(const MlpDesc& d, uint32_t n, uint32_t n_bins, uint32_t n_dims);
uint32_t mlp_train_r32ob_grid(uint32_t n);
bool mlp_train_r32ob_applies(const MlpDesc& d, uint32_t n, const MlpOneBlobInput* oneblob, const float* data_pdf, const void* external_dL_dy, uint32_t dims, LossType loss, const void* out,
                             const void* dL_dx, const float* slabs);
void mlp_train_r32ob(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const MlpOneBlobInput& oneblob, const float* target, uint32_t dims, LossType loss, float loss_scale,
                     void* out, void* dL_dout, float* L, float* slabs, uint32_t n_params);
// ---- BASELINE config 5's MLP part, 64 -> 128 -> 128 -> 16 fed by level planes of 4 features, on the 32x32x16 matrix instruction
// (k_train_r32w.hip): the weight-gradient tiles shared out over a workgroup's four waves, fragments from L2.  mlp_train_fused dispatches
// to it with the grid of mlp_train_fused_grid.
bool mlp_train_r32w_applies(const MlpDesc& d, uint32_t n, uint32_t x_plane_features, const float* data_pdf, const void* external_dL_dy, uint32_t dims, LossType loss, const void* out,
                            const void* dL_dx, uint32_t dx_plane_features, const float* dx_record_x, const float* slabs, bool oneblob);
void mlp_train_r32w(hipStream_t stream, const MlpDesc& d, const void* image, uint32_t n, const void* x, const float* target, uint32_t dims, LossType loss, float loss_scale, void* out,
                    void* dL_dout, float* L, void* dL_dx, float* slabs, uint32_t n_params, uint32_t grid);
void mlp_expand_context(hipStream_t stream, uint32_t n, uint32_t dims, const void* compact_dL_dout, const float* compact_L, void* dL_dout, float* L);
// grad[i] (=|+=) sum_k slabs[k][i], fixed order, rounded to half once
// adam (optional, not with accumulate): the optimizer's update of these (matrix) weights is applied behind the reduction, bit-identical to adam_step run afterwards
struct AdamInFlush;
void mlp_reduce_slabs(hipStream_t stream, uint32_t n_params, uint32_t n_slabs, const float* slabs, void* grad_half, bool accumulate, const AdamInFlush* adam = nullptr);
// fully_fused_mlp.cu:757-762: result = dL_dout * act'(out), elementwise over n_elems halfs
void mlp_activation_backward_output(hipStream_t stream, uint32_t n_elems, uint32_t activation, const void* dL_dout, const void* out, void* result);
// dW[rows x cols] = sum_i dO[i][rows]^T In[i][cols]; result written as half into grad (overwrite or accumulate). workspace: float[wgrad_workspace_floats()]
size_t wgrad_workspace_floats(uint32_t rows, uint32_t cols, uint32_t n);
// several such products over the same n samples (a network's layers): same results as one mlp_wgrad() each, fewer launches
// (dO_tiled / In_tiled: the operand is a hidden layer's stored activations or gradients in k_mlp_fwd / k_mlp_bwd's tiled form, ld = the full
// width of that matrix and a panel's first column c0 absorbed by the pointer as + (c0 / 16) * 256 halves; k_mlp.hip hidden_tile_off)
struct WgradPanel { const void* dO; uint32_t ldo, rows; const void* In; uint32_t ldi, cols; void* grad; uint32_t ldg; bool dO_tiled, In_tiled; };
size_t wgrad_panels_workspace_floats(const WgradPanel* panels, uint32_t count, uint32_t n);
void mlp_wgrad_panels(hipStream_t stream, uint32_t n, const WgradPanel* panels, uint32_t count, bool accumulate, float* workspace);
void mlp_wgrad(hipStream_t stream, uint32_t n, const void* dO, uint32_t ldo, uint32_t rows, const void* In, uint32_t ldi, uint32_t cols,
               void* grad_half, uint32_t ldg, bool accumulate, float* workspace);

// ------------------------------------------------------------------------------------------------------------------
// loss / reduction / optimizer / init plumbing
// ------------------------------------------------------------------------------------------------------------------
void loss_evaluate(hipStream_t stream, LossType type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale,
                   const void* pred_half, const float* target, float* values, void* grads_half, const float* data_pdf);
// sum of n floats -> *result_dev (device float, overwritten). workspace-free: uses a two-stage reduction through `partials` (>= 1024 floats)
void reduce_sum(hipStream_t stream, size_t n, const float* values, float* partials, float* result_dev);

struct AdamHyper {
	float learning_rate = 1e-3f, beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-8f, l2_reg = 1e-8f;
	float relative_decay = 0.0f, absolute_decay = 0.0f, clipping_magnitude = 0.0f, non_matrix_learning_rate_factor = 1.0f;
	bool adabound = false, optimize_matrix_params = true, optimize_non_matrix_params = true;
};
// kernel-side form of the hyperparameters of one step (k_misc.hip make_adam_args)
struct AdamArgs {
	float relative_weight_decay, absolute_weight_decay, weight_clipping_magnitude, loss_scale, learning_rate, non_matrix_learning_rate_factor;
	float beta1, beta2, epsilon, lower_lr_bound, upper_lr_bound, l2_reg;
	uint32_t optimize_matrix_params, optimize_non_matrix_params;
	float inv_loss_scale;
	uint32_t inv_loss_scale_exact;
	uint32_t common_step; // the optimizer's own step count: the per-parameter count of every parameter that was updated in every step
};
AdamArgs make_adam_args(const AdamHyper& h, float loss_scale, uint32_t current_step);
// Adam applied by a gradient kernel: the owner of a chunk of the gradient has its final value in LDS when it flushes and updates
// those parameters on the spot (k_grid_scatter.hip) -- the 34 B/param of optimizer state stream under the latency-bound phases of
// the other workgroups instead of in a kernel of their own.  Arrays are indexed like the gradient array the kernel writes.
constexpr uint32_t IMAGE_INV_WIDTH = 4; // a weight sits in at most 4 image elements: forward and transposed fragments of the 16x16x32 and of the 32x32x16 sections
struct AdamInFlush {
	AdamArgs args;
	float* w_fp = nullptr;
	void* w_half = nullptr;
	float* m1 = nullptr;
	float* m2 = nullptr;
	void* steps = nullptr; // uint32, or uint16 if steps16
	uint32_t steps16 = 0;
	const float* debias_table = nullptr;
	// k_wgrad_reduce_adam only (Network::live_image): the network's fragment images and, per parameter, the IMAGE_INV_WIDTH image elements that
	// hold it (0xffffffff: none) -- the kernel writes an updated weight there too, so that the next step needs no k_mlp_prep launch
	void* image = nullptr;
	const uint32_t* image_inv = nullptr;
	AdamInFlush advanced(size_t n) const { // the same arrays seen from parameter n on
		AdamInFlush r = *this;
		r.w_fp += n; r.w_half = (char*)w_half + 2 * n; r.m1 += n; r.m2 += n; r.steps = (char*)steps + (steps16 ? 2 : 4) * n;
		return r;
	}
};
// What the backward pass of a fused step leaves for the optimizer's launch to finish (k_adam_prologue, k_misc.hip): the scatter's finalize
// pass -- the shared chunks' exact sums in `scratch` are rounded into the gradient, the scratch left zero -- and the fixed-order sum of the
// MLP's weight-gradient slabs.  Both are the last writes of the gradients the optimizer reads next: in ONE launch the workgroup that
// finishes a gradient updates its parameters at once, every other workgroup does what k_adam does, and the step has one ~5 us launch
// (k_grid_scatter_finalize) and one kernel boundary less.  Same rounding, same adam_one: gradients, weights, moments, counts bit-identical.
struct AdamPrologue {
	bool offered = false;                      // the trainer will run an optimizer that can take it (set by the trainer)
	bool pending = false;                      // the backward pass left its finalize pass here instead of launching it
	const GridScatterRange* dev_ranges = nullptr;
	std::vector<GridScatterRange> ranges;      // host copy; grad_begin relative to grad_base
	uint64_t* scratch = nullptr;
	void* grad_base = nullptr;                 // half: the gradient array the ranges index (the encoding's part of the gradient vector)
	bool accumulate = false;
	bool has_reduce = false;
	uint32_t reduce_elems = 0, reduce_slabs = 0; // the MLP's weights (the first reduce_elems parameters) and its slabs
	const float* slabs = nullptr;
	int reduce_accumulate = 0;
};
// adam_step with the prologue in the same launch.  false: the shapes do not allow it (alignment) -- nothing was launched, the caller
// runs grid_scatter_finalize / mlp_reduce_slabs and adam_step itself.
bool adam_step_with_prologue(hipStream_t stream, const AdamHyper& h, size_t n, size_t n_matrix, float loss_scale, uint32_t current_step,
                             float* w_fp, void* w_half, void* g_half, float* m1, float* m2, void* steps, bool steps16, const float* debias_table, const AdamPrologue& p);
// steps: the per-parameter update counts, uint32 or -- steps16 -- uint16 (what the optimizer keeps while every count fits:
// 4 of the 36 bytes per parameter the kernel moves are the counts' upper halves otherwise)
void adam_step(hipStream_t stream, const AdamHyper& h, size_t n, size_t n_matrix, float loss_scale, uint32_t current_step,
               float* w_fp, void* w_half, const void* g_half, float* m1, float* m2, void* steps, bool steps16, const float* debias_table);
void adam_widen_steps(hipStream_t stream, size_t n, const void* steps16, void* steps32); // uint16 -> uint32
// debias_table[t] = sqrtf(1 - powf(beta2, t)) / (1 - powf(beta1, t)) (adam.h:97-98), evaluated on the device, for t in [from, to)
void adam_fill_debias_table(hipStream_t stream, float beta1, float beta2, uint32_t from, uint32_t to, float* table);
// dst[i][dst_col + j] = src[i][src_col + j] for j < width; elements of 2 or 4 bytes (Composite encoding)
void copy_columns(hipStream_t stream, size_t elem_bytes, uint32_t n, const void* src, uint32_t src_stride, uint32_t src_col, void* dst, uint32_t dst_stride, uint32_t dst_col, uint32_t width);
// optimizers/sgd.h:44-72 and optimizers/ema.h:44-78 (half parameters)
void sgd_step(hipStream_t stream, size_t n, float loss_scale, float learning_rate, float l2_reg, float* weights_full_precision, void* weights, const void* gradients);
void ema_step(hipStream_t stream, size_t n, float decay, float debias_old, float debias_new, const void* weights, void* weights_ema, float* tmp);
// optimizers/average.h:44-60, batched.h:44-61, lookahead.h:44-59 (half parameters)
void average_step(hipStream_t stream, size_t n, uint32_t n_samples, const void* weights, void* current_sample, void* average);
void batched_accumulate(hipStream_t stream, size_t n, bool first, uint32_t multiplier, const void* gradients, float* pool);
void lookahead_step(hipStream_t stream, size_t n, float alpha, float* weights_full_precision, void* weights, void* weights_lookahead);
// optimizers/novograd.h:44-94 for ONE layer of n weights: the layer's second moment from the sum of its squared gradients, then the step
void novograd_layer_step(hipStream_t stream, size_t n, float relative_decay, float absolute_decay, float loss_scale, float learning_rate, float beta1, float beta2, float epsilon,
                         float* weights_full_precision, void* weights, const void* gradients, float* first_moments, float* layer_second_moment);

// random.h:40-70: strided uniform fill from a pcg32 state; advances (state, inc) on the host copy by n
void generate_random_uniform(hipStream_t stream, uint64_t* state_inc_host, size_t n, float* out, float lower, float upper);
void cast_float_to_half(hipStream_t stream, size_t n, const float* in, void* out);
void cast_half_to_float(hipStream_t stream, size_t n, const void* in, float* out);
// object.cu:61-67: [n][in_stride] T -> float out(dim, sample) = out.data[i*stride_sample + d*stride_dim], d < dims
void trim_and_cast(hipStream_t stream, bool fp32, uint32_t n, uint32_t in_stride, uint32_t dims, const void* in, MatViewMut out);
void fill_half(hipStream_t stream, size_t n, void* out, float value);

} // namespace tcnn_amd
