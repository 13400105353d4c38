/*
 * tcnn_amd.h -- C ABI of libtcnn_amd.so, the MI355X (gfx950) drop-in for tiny-cuda-nn's hot path:
 *     encoding (HashGrid | OneBlob | Identity) -> fully fused fp16 MLP -> loss -> Adam.
 *
 * Every entry point below is what a binding of the reference would bind; the reference interface it replaces is cited
 * as file:line relative to the reference tree (leejaeyong7/tiny-cuda-nn).  Plain pointers and sizes only:
 *   - all data pointers are DEVICE pointers owned by the caller unless stated otherwise;
 *   - "half" buffers are IEEE binary16, passed as void*;
 *   - stream is a hipStream_t passed as void* (NULL = the null stream);
 *   - JSON travels as NUL-terminated UTF-8 text (the reference passes nlohmann::json objects, cpp_api.h:60);
 *   - errors: every int-returning function returns TCNN_OK (0) or TCNN_ERROR and records a message retrievable with
 *     tcnn_last_error() (thread-local).  This replaces the reference's C++ exceptions (common_host.h:71-110).
 *
 * Matrix convention (same memory as the reference): a batch matrix with `w` dims and `n` samples is column-major
 * w x n (gpu_matrix.h:417), i.e. element (dim j, sample i) at ptr[i * w + j]  -- identical to a row-major torch tensor
 * of shape [n, w].  Batch sizes must be multiples of tcnn_batch_size_granularity() = 256 (object.h:130).
 */
#ifndef TCNN_AMD_H
#define TCNN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCNN_OK 0
#define TCNN_ERROR 1

/* cpp_api.h:69-72 */
#define TCNN_PRECISION_FP32 0
#define TCNN_PRECISION_FP16 1

/* common.h GradientMode */
#define TCNN_GRADIENT_IGNORE 0
#define TCNN_GRADIENT_OVERWRITE 1
#define TCNN_GRADIENT_ACCUMULATE 2

/* common.h:157-162 MatrixLayout: RowMajor == SoA == 0 ([dim][sample]), ColumnMajor == AoS == 1 ([sample][dim]) */
#define TCNN_LAYOUT_SOA 0
#define TCNN_LAYOUT_AOS 1

/* cpp_api.h:52-58 LogSeverity */
#define TCNN_LOG_INFO 0
#define TCNN_LOG_DEBUG 1
#define TCNN_LOG_WARNING 2
#define TCNN_LOG_ERROR 3
#define TCNN_LOG_SUCCESS 4

#define TCNN_MEMCPY_HOST_TO_DEVICE 1
#define TCNN_MEMCPY_DEVICE_TO_HOST 2
#define TCNN_MEMCPY_DEVICE_TO_DEVICE 3

typedef struct tcnn_module_s* tcnn_module_t;       /* tcnn::cpp::Module            cpp_api.h:86-111 */
typedef struct tcnn_context_s* tcnn_context_t;     /* tcnn::cpp::Context           cpp_api.h:82-84  */
typedef struct tcnn_trainer_s* tcnn_trainer_t;     /* tcnn::TrainableModel         config.h:46-51   */
typedef struct tcnn_train_ctx_s* tcnn_train_ctx_t; /* Trainer::ForwardContext      trainer.h:89-95  */
typedef void* tcnn_stream_t;                       /* hipStream_t */

const char* tcnn_last_error(void);
const char* tcnn_version(void);

/* ---- free functions, cpp_api.h:62-80 / cpp_api.cu:41-63 ---- */
uint32_t tcnn_batch_size_granularity(void);             /* cpp_api.cu:41  */
int      tcnn_device(int* device_out);                  /* cpp_api.cu:43  cuda_device() */
int      tcnn_set_device(int device);                   /* cpp_api.cu:44  set_cuda_device() */
void     tcnn_free_temporary_memory(void);              /* cpp_api.cu:45  */
int      tcnn_has_networks(void);                       /* cpp_api.cu:47  */
float    tcnn_default_loss_scale(int precision);        /* cpp_api.cu:55  */
int      tcnn_preferred_precision(void);                /* cpp_api.cu:60  */
void     tcnn_set_log_callback(void (*callback)(int severity, const char* message, void* user), void* user); /* cpp_api.cu:61 */

/* ---- device memory and stream helpers for callers that do not link the HIP runtime themselves: what GPUMemory<T>
 * (gpu_memory.h:60-392) and GPUMatrix<T> (gpu_matrix.h:417-470) need from a runtime; used by include/tiny-cuda-nn/ ---- */
int tcnn_gpu_malloc(size_t bytes, void** out);                              /* gpu_memory.h:93-112  allocate_memory */
int tcnn_gpu_free(void* ptr);                                               /* gpu_memory.h:114-130 free_memory */
int tcnn_gpu_memcpy(void* dst, const void* src, size_t bytes, int kind);    /* gpu_memory.h:183-260 copy_from_host / copy_to_host */
int tcnn_gpu_memset(void* ptr, int value, size_t bytes);                    /* gpu_memory.h:171-181 memset */
int tcnn_stream_synchronize(tcnn_stream_t stream);
/* random.h:58-70 generate_random_uniform<float>(stream, rng, n, out, lower, upper): thread i of the reference's kernel advances a
 * copy of the pcg32 by 4 i and writes elements i + n_threads j (j < 4); the caller's generator is then advanced by n.
 * rng_state_inc = {state, inc} of a pcg32 (include/tiny-cuda-nn/random.h keeps them), updated in place. */
int tcnn_generate_random_uniform(tcnn_stream_t stream, uint64_t rng_state_inc[2], size_t n, float* out, float lower, float upper);

/* ---- factories, cpp_api.h:113-115 / cpp_api.cu:146-165.  The caller owns the returned module. ---- */
int  tcnn_create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const char* encoding_json, const char* network_json, tcnn_module_t* out);
int  tcnn_create_network(uint32_t n_input_dims, uint32_t n_output_dims, const char* network_json, tcnn_module_t* out);
int  tcnn_create_encoding(uint32_t n_input_dims, const char* encoding_json, int precision, tcnn_module_t* out);
void tcnn_module_destroy(tcnn_module_t module);

/* ---- Module methods, cpp_api.h:91-106 / cpp_api.cu:72-139 ----
 * input [n][n_input_dims] float; output / dL_doutput [n][n_output_dims()] in output_precision (n_output_dims() is the PADDED
 * width, cpp_api.cu:130); params: ONE flat vector in param_precision, network weights first, then encoding parameters
 * (network_with_input_encoding.h:115-122), re-bound on every call.  dL_dparams == NULL => GradientMode::Ignore, else Overwrite. */
int  tcnn_module_inference(tcnn_module_t m, tcnn_stream_t stream, uint32_t n_elements, const float* input, void* output, void* params);
int  tcnn_module_forward(tcnn_module_t m, tcnn_stream_t stream, uint32_t n_elements, const float* input, void* output, void* params,
                         int prepare_input_gradients, tcnn_context_t* ctx_out);
int  tcnn_module_backward(tcnn_module_t m, tcnn_stream_t stream, tcnn_context_t ctx, uint32_t n_elements, float* dL_dinput,
                          const void* dL_doutput, void* dL_dparams, const float* input, const void* output, const void* params);
/* second-order input gradients (cpp_api.h:94, cpp_api.cu:111-127; grid.h:902-1026): dL_ddLdinput [n][n_input_dims] float is the
 * gradient arriving at dL_dinput; optional results dL_dparams (Overwrite), dL_ddLdoutput [n][n_output_dims], dL_dinput [n][n_input_dims]
 * (overwritten; the reference adds into a zeroed buffer).  The context must come from tcnn_module_forward(prepare_input_gradients = 1).
 * Like in the reference only grid encodings implement it; other modules report "DifferentiableObject::backward_backward_input_impl:
 * not implemented error" (object.h:288). */
int  tcnn_module_backward_backward_input(tcnn_module_t m, tcnn_stream_t stream, tcnn_context_t ctx, uint32_t n_elements, const float* dL_ddLdinput,
                                         const float* input, const void* dL_doutput, void* dL_dparams, void* dL_ddLdoutput, float* dL_dinput, const void* params);
void tcnn_context_destroy(tcnn_context_t ctx);

uint32_t    tcnn_module_n_input_dims(tcnn_module_t m);     /* cpp_api.cu:129 */
uint32_t    tcnn_module_n_output_dims(tcnn_module_t m);    /* cpp_api.cu:130 (padded) */
size_t      tcnn_module_n_params(tcnn_module_t m);         /* cpp_api.cu:131 */
int         tcnn_module_param_precision(tcnn_module_t m);  /* cpp_api.h:103 */
int         tcnn_module_output_precision(tcnn_module_t m); /* cpp_api.h:100 */
/* cpp_api.cu:133-136: pcg32 rng{seed}; fills params_full_precision (device, float[n_params]) */
int         tcnn_module_initialize_params(tcnn_module_t m, uint64_t seed, float* params_full_precision, float scale);
const char* tcnn_module_hyperparams(tcnn_module_t m);      /* JSON text owned by the module; cpp_api.cu:138 */
const char* tcnn_module_name(tcnn_module_t m);             /* cpp_api.cu:139 */

/* ---- boundary A: create_from_config / Trainer (config.h:53-63, trainer.h:48-363) ---- */
int  tcnn_create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const char* config_json, tcnn_trainer_t* out);          /* seed 1337, trainer.h:50 */
int  tcnn_create_from_config_seeded(uint32_t n_input_dims, uint32_t n_output_dims, const char* config_json, uint32_t seed, tcnn_trainer_t* out);
void tcnn_trainer_destroy(tcnn_trainer_t t);

/* Trainer::training_step (trainer.h:163-190).  input: float, n_input_dims x n in `input_layout`; target: float [n][n_output_dims];
 * data_pdf (optional) like target; dL_dinput (optional) float in input_layout; external_dL_dy (optional) half [n][padded_out].
 * Returns the forward context (caller destroys it) -- loss is fetched with tcnn_trainer_loss like trainer->loss(stream, *ctx). */
int  tcnn_trainer_training_step(tcnn_trainer_t t, tcnn_stream_t stream, uint32_t n_elements, const float* input, int input_layout,
                                const float* target, const float* data_pdf, int run_optimizer, float* dL_dinput, int use_inference_params,
                                int gradient_mode, const void* external_dL_dy, tcnn_train_ctx_t* ctx_out);
/* Trainer::loss (trainer.h:205-207): sum of the per-element loss values; synchronises the stream */
int  tcnn_trainer_loss(tcnn_trainer_t t, tcnn_stream_t stream, tcnn_train_ctx_t ctx, float* loss_out);
/* the three pieces of a step, trainer.h:97-157 */
int  tcnn_trainer_forward(tcnn_trainer_t t, tcnn_stream_t stream, float loss_scale, uint32_t n_elements, const float* input, int input_layout, const float* target,
                          const float* data_pdf, int use_inference_params, int prepare_input_gradients, const void* external_dL_dy, tcnn_train_ctx_t* ctx_out);
int  tcnn_trainer_backward(tcnn_trainer_t t, tcnn_stream_t stream, tcnn_train_ctx_t ctx, uint32_t n_elements, const float* input, int input_layout,
                           float* dL_dinput, int use_inference_params, int gradient_mode);
int  tcnn_trainer_optimizer_step(tcnn_trainer_t t, tcnn_stream_t stream, float loss_scale);
void tcnn_train_ctx_destroy(tcnn_train_ctx_t ctx);
/* ForwardContext members (trainer.h:89-95): device pointers valid until the context is destroyed.
 * The fused training step keeps dL_doutput and L as dense [n][n_output_dims] matrices (the padding columns are zeros) and forms
 * the padded ones on the FIRST call of their accessor, on the step's stream. */
const void*  tcnn_train_ctx_output(tcnn_train_ctx_t ctx);      /* half  [n][padded_out] */
const void*  tcnn_train_ctx_dL_doutput(tcnn_train_ctx_t ctx);  /* half  [n][padded_out]; NULL + tcnn_last_error() on failure */
const float* tcnn_train_ctx_L(tcnn_train_ctx_t ctx);           /* float [n][padded_out]; NULL + tcnn_last_error() on failure */

/* network->inference(stream, input, output) (object.h:147-176): float in, float out [n_output_dims x n] in output_layout */
int  tcnn_trainer_inference(tcnn_trainer_t t, tcnn_stream_t stream, uint32_t n_elements, const float* input, int input_layout,
                            float* output, int output_layout, int use_inference_params);
/* network->inference_mixed_precision(stream, input, output) (object.h:133-145): the network's own output, half
 * [n_elements][padded_output_width] -- what inference() casts to float; half the bytes for a caller that gathers row shards */
int  tcnn_trainer_inference_mixed_precision(tcnn_trainer_t t, tcnn_stream_t stream, uint32_t n_elements, const float* input, int input_layout,
                                            void* output_half, int use_inference_params);

size_t   tcnn_trainer_n_params(tcnn_trainer_t t);                 /* trainer.h:338 */
uint32_t tcnn_trainer_padded_output_width(tcnn_trainer_t t);
float*   tcnn_trainer_params_full_precision(tcnn_trainer_t t);    /* trainer.h:226 */
void*    tcnn_trainer_params(tcnn_trainer_t t);                   /* trainer.h:230 (half) */
void*    tcnn_trainer_params_inference(tcnn_trainer_t t);         /* trainer.h:234 */
void*    tcnn_trainer_param_gradients(tcnn_trainer_t t);          /* trainer.h:238 (half) */
int      tcnn_trainer_set_params_full_precision(tcnn_trainer_t t, const float* params, size_t n_params, int device_ptr); /* trainer.h:242 */
int      tcnn_trainer_set_params(tcnn_trainer_t t, const void* params_half, size_t n_params, int device_ptr);            /* trainer.h:256 */
int      tcnn_trainer_initialize_params(tcnn_trainer_t t);        /* trainer.h:68-87 (re-initialise, continues the rng stream) */
int      tcnn_trainer_update_hyperparams(tcnn_trainer_t t, const char* json); /* trainer.h:213 */
const char* tcnn_trainer_hyperparams(tcnn_trainer_t t);           /* trainer.h:218; JSON text owned by the trainer */
/* Snapshot wire format, trainer.h:275-315 (json serialize(bool) / deserialize(const json&)) with adam.h:278-299 and
 * gpu_memory_json.h:36-71: the object {"n_params", "params_type": "__half", "params_binary": <bin>, ["optimizer":
 * {"current_step", "base_learning_rate", "first_moments_binary", "second_moments_binary", "param_steps_binary"}]} as
 * MessagePack bytes, encoded the way nlohmann::json::to_msgpack encodes it (what instant-ngp-style callers write to disk).
 * serialize: the buffer belongs to the trainer and stays valid until its next serialize call or its destruction.
 * deserialize: accepts "params_type" "__half" or "float", binary values or their text form {"bytes": [...]}. */
int      tcnn_trainer_serialize(tcnn_trainer_t t, int serialize_optimizer, const void** out_bytes, size_t* out_size);
int      tcnn_trainer_deserialize(tcnn_trainer_t t, const void* bytes, size_t size);
const char* tcnn_trainer_network_hyperparams(tcnn_trainer_t t);   /* network->hyperparams() */
uint32_t tcnn_trainer_optimizer_step_count(tcnn_trainer_t t);     /* optimizer->step() adam.h:198 */
/* Measurement hook (no counterpart in the reference; bench.py): the NEXT fused training_step() records HIP events on
 * its stream around its pieces -- [0] encoding forward, [1] the fused MLP kernel (forward + loss + backward + weight gradients),
 * [2] encoding backward, [3] optimizer.  _collect synchronises the stream and returns the milliseconds per piece summed over the
 * steps profiled since the last call, and their number. */
int  tcnn_trainer_profile_next_step(tcnn_trainer_t t);
int  tcnn_trainer_profile_collect(tcnn_trainer_t t, tcnn_stream_t stream, float* ms_per_piece /* [4] */, uint32_t* n_steps);
/* Introspection (no counterpart in the reference): how many parameters of the last training_step() had their optimizer update
 * applied by the gradient kernels themselves (k_grid_scatter's flush) instead of by the optimizer kernel; 0 when the step ran
 * the optimizer the usual way (the default; TCNN_AMD_ADAM_IN_FLUSH=1 asks for the fused form, which needs plain Adam, GradientMode
 * Overwrite, run_optimizer=true and a grid whose scatter runs in record form). */
size_t tcnn_trainer_params_updated_in_flush(tcnn_trainer_t t);
/* Introspection (no counterpart in the reference): how many times training_step() calls of this trainer have launched the kernel
 * that rearranges the network's weights into matrix-instruction fragments.  A model without encoding parameters trained with plain
 * Adam keeps those fragments current from inside its optimizer kernel, so the count stops growing after the first step -- until
 * the parameters are set, restored, stepped by optimizer_step(), or a pointer to them has been handed out
 * (tcnn_trainer_params / tcnn_trainer_params_full_precision: from then on every step rearranges them again). */
size_t tcnn_trainer_image_preps(tcnn_trainer_t t);
/* Introspection (no counterpart in the reference): the grid encoding's gradient kernel (replacing kernel_grid_backward, grid.h:215-320)
 * sums a chunk's fp16 products as exact integers, two 32-bit sums per 64-bit LDS add while a per-task bound on sum |product| proves
 * that neither can overflow; a task whose bound fails runs again with 64-bit sums (same result, slower).  This counts those tasks since
 * the trainer was built; (size_t)-1 on error.  Synchronises with the device. */
size_t tcnn_trainer_scatter_wide_fallbacks(tcnn_trainer_t t);
/* Introspection (no counterpart in the reference): training steps of this trainer whose optimizer launch (adam_step, adam.h:150-188)
 * also ran the last two reductions of the backward pass -- the rounding of the grid gradient's shared chunks and the sum of the network's
 * weight-gradient slabs -- instead of a launch of their own in front of it (same results; TCNN_AMD_ADAM_PROLOGUE=0 separates them). */
size_t tcnn_trainer_optimizer_prologue_steps(tcnn_trainer_t t);
/* Introspection (no counterpart in the reference): backward passes of this trainer's grid encoding that ran the list-fed gradient
 * kernel (k_grid_scatter_lists: hit lists written by the forward kernel) rather than the bit-plane, binned or atomic forms -- tests assert
 * which kernel produced the gradients they compare with the oracle (replaces kernel_grid_backward, grid.h:215-320). */
size_t tcnn_trainer_list_scatters(tcnn_trainer_t t);
size_t tcnn_module_list_scatters(tcnn_module_t m); /* the same count for a module's grid encoding(s) (callers with their own network) */
/* Introspection: 1 when this context owns the network's weight-gradient slabs because their reduction was left to the optimizer's
 * launch (tcnn_trainer_optimizer_prologue_steps): they must outlive training_step()'s own scope, until that launch is enqueued. */
int tcnn_train_ctx_keeps_weight_gradient_slabs(tcnn_trainer_t t, tcnn_train_ctx_t ctx);

#ifdef __cplusplus
}
#endif
#endif /* TCNN_AMD_H */
