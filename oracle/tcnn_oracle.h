/*
 * tcnn_oracle.h -- C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain-CPU restatement of the reference
 * algorithm (leejaeyong7/tiny-cuda-nn) for the hot path
 *     encoding (HashGrid | OneBlob | Identity) -> MLP (fp16) -> loss -> Adam.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (tiny-cuda-nn_amd/) never links, imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Conventions:
 *   - "half" values travel as uint16_t bit patterns (IEEE binary16).
 *   - batch matrices are "AoS": element (dim j, sample i) at  ptr[i * width + j]
 *     (memory-identical to the reference's column-major GPUMatrix<T>(width, n),
 *     gpu_matrix.h:417), unless a function says SoA ([dim][sample]).
 *   - weight matrices are row-major [fan_out][fan_in] (fully_fused_mlp.cu:659-671).
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums shared with the tests (numbering is ours; names follow common.h:112-160) ---- */
enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_LEAKY_RELU = 2, ORC_ACT_EXPONENTIAL = 3, ORC_ACT_SINE = 4,
       ORC_ACT_SIGMOID = 5, ORC_ACT_SQUAREPLUS = 6, ORC_ACT_SOFTPLUS = 7, ORC_ACT_TANH = 8 };
enum { ORC_GRID_HASH = 0, ORC_GRID_DENSE = 1, ORC_GRID_TILED = 2 };
enum { ORC_HASH_PRIME = 0, ORC_HASH_COHERENT_PRIME = 1, ORC_HASH_REVERSED_PRIME = 2, ORC_HASH_RNG = 3 };
enum { ORC_INTERP_NEAREST = 0, ORC_INTERP_LINEAR = 1, ORC_INTERP_SMOOTHSTEP = 2 };
enum { ORC_LOSS_L2 = 0, ORC_LOSS_RELATIVE_L2 = 1, ORC_LOSS_L1 = 2, ORC_LOSS_RELATIVE_L1 = 3, ORC_LOSS_MAPE = 4, ORC_LOSS_SMAPE = 5, ORC_LOSS_CROSS_ENTROPY = 6,
       ORC_LOSS_VARIANCE = 7, ORC_LOSS_RELATIVE_L2_LUMINANCE = 8 }; /* src/loss.cu:57-65 */
enum { ORC_ACC_FP32 = 0, ORC_ACC_FP16 = 1 };

#define ORC_MAX_LEVELS 128

/* ---- half helpers ---- */
uint16_t orc_float_to_half(float f);
float    orc_half_to_float(uint16_t h);
uint16_t orc_double_to_half(double d);
void     orc_cast_float_to_half(size_t n, const float* in, uint16_t* out);   /* trainer.h:83-85 */
void     orc_cast_half_to_float(size_t n, const uint16_t* in, float* out);

/* ---- pcg32 (dependencies/pcg32/pcg32.h:40-166). state[0]=state, state[1]=inc ---- */
void     orc_pcg32_seed(uint64_t* st, uint64_t initstate, uint64_t initseq);
uint32_t orc_pcg32_next_uint(uint64_t* st);
float    orc_pcg32_next_float(uint64_t* st);
void     orc_pcg32_advance(uint64_t* st, int64_t delta);
/* std::seed_seq{seed}.generate(2 words) as used by Trainer (trainer.h:52-55) */
void     orc_seed_seq2(uint32_t seed, uint32_t* out2);

/* gpu_matrix.h:284-299  (host-side xavier fill, row-major [rows][cols]) */
void orc_xavier_uniform(uint64_t* st, float* out, uint32_t rows, uint32_t cols, float scale);
/* random.h:40-70 (device-side strided fill; advances st by n afterwards) */
void orc_generate_random_uniform(uint64_t* st, size_t n, float* out, float lower, float upper);

/* ---- grid encoding ---- */
typedef struct {
	uint32_t n_pos_dims;          /* D: 2..4 */
	uint32_t n_features_per_level;/* F: 1,2,4,8 */
	uint32_t n_levels;
	uint32_t log2_hashmap_size;
	uint32_t base_resolution;
	float    per_level_scale;
	uint32_t grid_type;
	uint32_t hash_type;
	uint32_t interpolation;
	uint32_t stochastic_interpolation;
	/* derived by orc_grid_setup */
	uint32_t offsets[ORC_MAX_LEVELS + 1]; /* in entries */
	float    scales[ORC_MAX_LEVELS];
	uint32_t resolutions[ORC_MAX_LEVELS];
	uint32_t n_params;
} orc_grid_t;

/* grid.h:668-730 + common_device.h:709-718.  returns 0 on success */
int      orc_grid_setup(orc_grid_t* g);
/* common_device.h:631-707 */
uint32_t orc_grid_hash(uint32_t n_dims, uint32_t hash_type, const uint32_t* pos_grid);
uint32_t orc_grid_index(uint32_t n_dims, uint32_t hash_type, uint32_t grid_type, uint32_t hashmap_size, uint32_t resolution, const uint32_t* pos_grid);
/* common_device.h:856-868: returns cell, writes frac (after the interpolation function) */
uint32_t orc_pos_fract(float input, float scale, uint32_t interpolation, float* frac, float* frac_derivative);
/* grid.h:49-212.  x: [n][D] float.  out: AoS [n][out_stride] half (features l*F+f; columns >= L*F are written 0,
 * grid.h:749-758).  Optional: indices [n][L][2^D] uint32 (entry index inside the level), dy_dx [n][L*F][D] float. */
void orc_grid_forward(const orc_grid_t* g, uint32_t n, const float* x, const uint16_t* grid, uint16_t* out, uint32_t out_stride,
                      uint32_t* indices, float* dy_dx);
/* grid.h:215-320 with fp16 accumulation in sample order (the reference uses packed-fp16 atomics in arbitrary order).
 * dL_dy AoS [n][dy_stride] half.  grad: half[n_params] accumulated in place (caller zeroes for Overwrite, grid.h:858).
 * grad_f32 (optional): fp32 accumulation of the same products, for tolerance checks. */
void orc_grid_backward(const orc_grid_t* g, uint32_t n, const float* x, const uint16_t* dL_dy, uint32_t dy_stride,
                       uint16_t* grad, float* grad_f32);
/* grid.h:215-320 with the order-independent limit of the fp16 atomic accumulation: contributions are the reference's fp16
 * products, summed exactly and rounded to fp16 once.  grad is overwritten (or, accumulate != 0, the old value joins the exact sum). */
void orc_grid_backward_exact(const orc_grid_t* g, uint32_t n, const float* x, const uint16_t* dL_dy, uint32_t dy_stride, uint16_t* grad, int accumulate);
/* grid.h:323-349 */
void orc_grid_backward_input(const orc_grid_t* g, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride, const float* dy_dx, float* dL_dx);
/* grid.h:352-650, 902-1026 (backward_backward_input): second-order terms of dL_dx = sum_k dL_dy_k dy_k/dx.
 * dL_ddLdx [n][D] float = gradient arriving at dL_dx.  Optional outputs: grad half[n_params] / grad_f32 (accumulated in place like
 * orc_grid_backward), dL_ddLdy half [n][dy_stride] (needs dy_dx [n][L*F][D] of the forward pass; padded columns 0),
 * dL_dx float [n][D] (needs grid; overwritten). */
void orc_grid_backward_backward_input(const orc_grid_t* g, uint32_t n, const float* x, const float* dL_ddLdx, const uint16_t* dL_dy, uint32_t dy_stride,
                                      const uint16_t* grid, const float* dy_dx, uint16_t* grad, float* grad_f32, uint16_t* dL_ddLdy, float* dL_dx);

/* ---- OneBlob (oneblob.h:47-164, definition form) and Identity (identity.h:46-85) ---- */
void orc_oneblob_forward(uint32_t n, uint32_t n_dims, uint32_t n_bins, const float* x, uint16_t* out, uint32_t out_stride);
void orc_oneblob_backward_input(uint32_t n, uint32_t n_dims, uint32_t n_bins, const float* x, const uint16_t* dL_dy, uint32_t dy_stride, float* dL_dx);
void orc_identity_forward(uint32_t n, uint32_t n_dims, float scale, float offset, const float* x, uint16_t* out, uint32_t out_stride);
void orc_identity_backward_input(uint32_t n, uint32_t n_dims, float scale, const uint16_t* dL_dy, uint32_t dy_stride, float* dL_dx);

/* ---- Frequency (frequency.h:44-101), TriangleWave (triangle_wave.h:44-107): out AoS [n][out_stride] half, pad dims = 1;
 * dy_dx (optional) float [n][n_dims * outputs_per_input].  orc_periodic_backward_input serves both backward kernels. ---- */
void orc_frequency_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* x, uint16_t* out, uint32_t out_stride, float* dy_dx);
void orc_trianglewave_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* x, uint16_t* out, uint32_t out_stride, float* dy_dx);
void orc_periodic_backward_input(uint32_t n, uint32_t n_dims, uint32_t outputs_per_input, const uint16_t* dL_dy, uint32_t dy_stride, const float* dy_dx, float* dL_dx);
/* ---- SphericalHarmonics (spherical_harmonics.h:44-108, common_device.h:339-700): x [n][3] in [0,1]^3 -> directions 2x-1;
 * degree^2 outputs, the PADDING COLUMNS FIRST; degree <= 8. ---- */
void orc_sh_forward(uint32_t n, uint32_t degree, const float* x, uint16_t* out, uint32_t out_stride);
void orc_sh_backward_input(uint32_t n, uint32_t degree, const float* x, const uint16_t* dL_dy, uint32_t dy_stride, float* dL_dx);

/* ---- MLP (fully_fused_mlp.cu:500-557,151-259,736-836; cutlass_mlp.cu:39-315) ---- */
typedef struct {
	uint32_t in_width;        /* padded input width */
	uint32_t width;           /* neurons */
	uint32_t out_width;       /* padded output width */
	uint32_t n_hidden_layers; /* 0 allowed (CutlassMLP) */
	uint32_t activation;
	uint32_t output_activation;
	uint32_t acc_mode;        /* ORC_ACC_FP32 (what MFMA does) or ORC_ACC_FP16 (what the reference's wmma does) */
} orc_mlp_t;
size_t orc_mlp_n_params(const orc_mlp_t* m);
/* fully_fused_mlp.cu:866-891 */
void orc_mlp_init_params(const orc_mlp_t* m, uint64_t* st, float* params, float scale);
/* hidden: optional [n_hidden_layers][n][width] half post-activation; out: [n][out_width] half */
void orc_mlp_forward(const orc_mlp_t* m, uint32_t n, const uint16_t* x, const uint16_t* params, uint16_t* hidden, uint16_t* out);
/* hidden/out from orc_mlp_forward; dL_dout [n][out_width] (already loss-scaled).
 * grads: half[n_params] (overwritten, or accumulated if accumulate != 0); grads_f32 optional float[n_params] (always overwritten).
 * dL_dx optional [n][in_width] half. */
void orc_mlp_backward(const orc_mlp_t* m, uint32_t n, const uint16_t* x, const uint16_t* params, const uint16_t* hidden, const uint16_t* out,
                      const uint16_t* dL_dout, uint16_t* dL_dx, uint16_t* grads, float* grads_f32, int accumulate);

/* ---- losses (l2.h:40-74, relative_l2.h:40-75) ---- */
void orc_loss(uint32_t type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale, const uint16_t* pred, const float* target,
              float* values, uint16_t* grads, const float* data_pdf);
/* reduce_sum.h:117-157 (order is not defined by the reference: double accumulation here) */
double orc_reduce_sum(size_t n, const float* values);

/* ---- Adam (adam.h:48-119,150-188) ---- */
typedef struct {
	float learning_rate, beta1, beta2, epsilon, l2_reg;
	float relative_decay, absolute_decay, clipping_magnitude, non_matrix_learning_rate_factor;
	uint32_t adabound, optimize_matrix_params, optimize_non_matrix_params;
} orc_adam_t;
void orc_adam_defaults(orc_adam_t* a);
/* current_step = the optimizer's step counter AFTER increment (adam.h:151), only used by adabound */
void orc_adam_step(const orc_adam_t* a, size_t n, size_t n_matrix, float loss_scale, uint32_t current_step,
                   float* w_fp, uint16_t* w, const uint16_t* g, float* m1, float* m2, uint32_t* steps);

/* activations, exposed for tests (common_device.h:102-160, 241-297) */
uint16_t orc_activation(uint32_t act, uint16_t pre);
uint16_t orc_activation_backward(uint32_t act, uint16_t grad, uint16_t forward_out);

#ifdef __cplusplus
}
#endif
