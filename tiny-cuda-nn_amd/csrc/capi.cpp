// capi.cpp -- the extern "C" boundary of libtcnn_amd.so (declared in include/tcnn_amd.h).
//
// Mirrors the reference's plugin interface tcnn::cpp (include/tiny-cuda-nn/cpp_api.h:50-115, src/cpp_api.cu:39-167) and its
// create_from_config / Trainer surface (config.h:46-63, trainer.h:48-363): same names, same argument meaning, and C++
// exceptions translated into TCNN_ERROR + tcnn_last_error().
#include "../../include/tcnn_amd.h"

#include "model.h"

#include <mutex>
#include <string>

using namespace tcnn_amd;

namespace {
thread_local std::string g_last_error;
tcnn_amd::Switches g_switches;
std::mutex g_switches_mutex; // the switches are PROCESS-WIDE: every tcnn_create_* call re-reads the environment for all live models (tcnn_common.h)

bool env_is(const char* name, char c) {
	const char* e = getenv(name);
	return e && e[0] == c;
}
int env_01(const char* name) { // -1: unset or neither 0 nor 1
	const char* e = getenv(name);
	return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1;
}

template <typename F>
int guarded(F&& f) {
	try {
		f();
		return TCNN_OK;
	} catch (const std::exception& e) {
		g_last_error = e.what();
		log_message(TCNN_LOG_ERROR, g_last_error);
		return TCNN_ERROR;
	} catch (...) {
		g_last_error = "unknown error";
		return TCNN_ERROR;
	}
}

Json parse_or_empty(const char* text) {
	if (!text || !*text) return Json::object();
	return Json::parse(text);
}
} // namespace

struct tcnn_module_s {
	std::unique_ptr<Model> model;
	std::string hyperparams_text;
	std::string name_text;
};

struct tcnn_context_s {
	std::unique_ptr<ModelContext> ctx;
};

struct tcnn_trainer_s {
	std::unique_ptr<Trainer> trainer;
	std::string hyperparams_text;
	std::string network_hyperparams_text;
	std::vector<uint8_t> snapshot; // tcnn_trainer_serialize's result
};

struct tcnn_train_ctx_s {
	std::unique_ptr<TrainContext> ctx;
};

namespace tcnn_amd {
Switches switches() {
	std::lock_guard<std::mutex> lock{g_switches_mutex};
	return g_switches;
}
void switches_reload() {
	Switches w;
	w.grid_planes = !env_is("TCNN_AMD_GRID_PLANES", '0');
	w.grid_rows_planes = !env_is("TCNN_AMD_GRID_ROWS_PLANES", '0');
	{ const char* e = getenv("TCNN_AMD_GRID_SCATTER"); w.grid_scatter_lds = !(e && std::string{e} == "atomic"); }
	w.scatter_records = !env_is("TCNN_AMD_SCATTER_RECORDS", '0');
	w.scatter_tune = !env_is("TCNN_AMD_SCATTER_TUNE", '0');
	w.scatter_lists = env_01("TCNN_AMD_SCATTER_LISTS");
	w.scatter_wide = env_is("TCNN_AMD_SCATTER_WIDE", '1');
	w.fused_step = !env_is("TCNN_AMD_FUSED_STEP", '0');
	w.side_jobs = !env_is("TCNN_AMD_SIDE_JOBS", '0');
	w.live_image = !env_is("TCNN_AMD_LIVE_IMAGE", '0');
	w.adam_steps32 = env_is("TCNN_AMD_ADAM_STEPS32", '1');
#ifdef TCNN_AMD_DEV
	w.adam_in_flush = env_is("TCNN_AMD_ADAM_IN_FLUSH", '1'); // laboratory build only: a measured dead end (20 % slower), DESIGN.md "Dead ends measured in round 4"
#endif
	w.adam_in_reduce = !env_is("TCNN_AMD_ADAM_IN_REDUCE", '0');
	w.adam_prologue = !env_is("TCNN_AMD_ADAM_PROLOGUE", '0');
	w.adam_prologue_refused = env_is("TCNN_AMD_ADAM_PROLOGUE", 'r');
	w.mlp_r32 = !env_is("TCNN_AMD_MLP_R32", '0');
	w.mlp_r32a = env_01("TCNN_AMD_MLP_R32A");
	w.mlp_regs = !env_is("TCNN_AMD_MLP_REGS", '0');
	w.mlp_fast = !env_is("TCNN_AMD_MLP_FAST", '0');
	if (const char* e = getenv("TCNN_AMD_MLP_PRIO")) w.mlp_prio = (uint32_t)atoi(e);
	std::lock_guard<std::mutex> lock{g_switches_mutex};
	g_switches = w;
}
} // namespace tcnn_amd

extern "C" {

const char* tcnn_last_error(void) { return g_last_error.c_str(); }
const char* tcnn_version(void) { return "tcnn_amd 0.1 (gfx950)"; }

uint32_t tcnn_batch_size_granularity(void) { return BATCH_SIZE_GRANULARITY; }

int tcnn_device(int* device_out) {
	return guarded([&] { HIP_CHECK_THROW(hipGetDevice(device_out)); });
}

int tcnn_set_device(int device) {
	return guarded([&] { HIP_CHECK_THROW(hipSetDevice(device)); });
}

void tcnn_free_temporary_memory(void) { Arena::instance().release_all(); }

/* device memory for callers that do not link the HIP runtime themselves (GPUMemory<T>, gpu_memory.h:60-392) */
int tcnn_gpu_malloc(size_t bytes, void** out) {
	return guarded([&] {
		CHECK_THROW(out != nullptr);
		*out = nullptr;
		if (bytes) HIP_CHECK_THROW(hipMalloc(out, bytes));
	});
}
int tcnn_gpu_free(void* ptr) {
	return guarded([&] { if (ptr) HIP_CHECK_THROW(hipFree(ptr)); });
}
int tcnn_gpu_memcpy(void* dst, const void* src, size_t bytes, int kind) {
	return guarded([&] {
		const hipMemcpyKind k = kind == TCNN_MEMCPY_HOST_TO_DEVICE ? hipMemcpyHostToDevice : kind == TCNN_MEMCPY_DEVICE_TO_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
		if (bytes) HIP_CHECK_THROW(hipMemcpy(dst, src, bytes, k));
	});
}
int tcnn_gpu_memset(void* ptr, int value, size_t bytes) {
	return guarded([&] { if (bytes) HIP_CHECK_THROW(hipMemset(ptr, value, bytes)); });
}
int tcnn_stream_synchronize(tcnn_stream_t stream) {
	return guarded([&] { HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)stream)); });
}
int tcnn_generate_random_uniform(tcnn_stream_t stream, uint64_t rng_state_inc[2], size_t n, float* out, float lower, float upper) {
	return guarded([&] {
		CHECK_THROW(rng_state_inc != nullptr && (out != nullptr || n == 0));
		generate_random_uniform((hipStream_t)stream, rng_state_inc, n, out, lower, upper);
		HIP_CHECK_THROW(hipGetLastError());
	});
}
int tcnn_has_networks(void) { return 1; }
float tcnn_default_loss_scale(int precision) { return precision == TCNN_PRECISION_FP32 ? 1.0f : LOSS_SCALE_FP16; }
int tcnn_preferred_precision(void) { return TCNN_PRECISION_FP16; }

void tcnn_set_log_callback(void (*callback)(int, const char*, void*), void* user) {
	log_sink().callback = callback;
	log_sink().user = user;
}

int tcnn_create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const char* encoding_json, const char* network_json, tcnn_module_t* out) {
	switches_reload(); // the A/B switches are read once per model (tcnn_common.h: Switches)
	return guarded([&] {
		CHECK_THROW(out != nullptr);
		auto m = std::make_unique<tcnn_module_s>();
		m->model.reset(new NetworkWithInputEncoding{n_input_dims, n_output_dims, parse_or_empty(encoding_json), parse_or_empty(network_json)});
		*out = m.release();
	});
}

int tcnn_create_network(uint32_t n_input_dims, uint32_t n_output_dims, const char* network_json, tcnn_module_t* out) {
	// cpp_api.cu:151-153: Identity encoding + network
	return tcnn_create_network_with_input_encoding(n_input_dims, n_output_dims, "{\"otype\": \"Identity\"}", network_json, out);
}

int tcnn_create_encoding(uint32_t n_input_dims, const char* encoding_json, int precision, tcnn_module_t* out) {
	switches_reload(); // the A/B switches are read once per model (tcnn_common.h: Switches)
	return guarded([&] {
		CHECK_THROW(out != nullptr);
		auto m = std::make_unique<tcnn_module_s>();
		m->model.reset(new EncodingModel{n_input_dims, parse_or_empty(encoding_json), precision == TCNN_PRECISION_FP32 ? Precision::Fp32 : Precision::Fp16});
		*out = m.release();
	});
}

void tcnn_module_destroy(tcnn_module_t module) { delete module; }

int tcnn_module_inference(tcnn_module_t m, tcnn_stream_t stream, uint32_t n, const float* input, void* output, void* params) {
	return guarded([&] {
		CHECK_THROW(m && m->model);
		m->model->inference((hipStream_t)stream, n, MatView{input, m->model->input_width(), 1u}, output, params);
	});
}

int tcnn_module_forward(tcnn_module_t m, tcnn_stream_t stream, uint32_t n, const float* input, void* output, void* params, int prepare_input_gradients, tcnn_context_t* ctx_out) {
	return guarded([&] {
		CHECK_THROW(m && m->model && ctx_out);
		auto c = std::make_unique<tcnn_context_s>();
		c->ctx = m->model->forward((hipStream_t)stream, n, MatView{input, m->model->input_width(), 1u}, output, params, prepare_input_gradients != 0);
		*ctx_out = c.release();
	});
}

int tcnn_module_backward(tcnn_module_t m, tcnn_stream_t stream, tcnn_context_t ctx, uint32_t n, float* dL_dinput, const void* dL_doutput, void* dL_dparams,
                         const float* input, const void* output, const void* params) {
	return guarded([&] {
		CHECK_THROW(m && m->model);
		if (!ctx || !ctx->ctx) throw std::runtime_error{"Module::bwd: called with invalid context. fwd likely (mistakenly) ran in inference mode."};
		MatViewMut dx{dL_dinput, m->model->input_width(), 1u};
		m->model->backward((hipStream_t)stream, *ctx->ctx, n, MatView{input, m->model->input_width(), 1u}, output, dL_doutput, dL_dinput ? &dx : nullptr, params,
		                   dL_dparams, dL_dparams ? GradientMode::Overwrite : GradientMode::Ignore); // cpp_api.cu:108
	});
}

int tcnn_module_backward_backward_input(tcnn_module_t m, tcnn_stream_t stream, tcnn_context_t ctx, uint32_t n, const float* dL_ddLdinput, const float* input, const void* dL_doutput,
                                        void* dL_dparams, void* dL_ddLdoutput, float* dL_dinput, const void* params) {
	return guarded([&] { // cpp_api.cu:111-127
		CHECK_THROW(m && m->model);
		if (!ctx || !ctx->ctx) throw std::runtime_error{"Module::bwd_bwd_input: called with invalid context. fwd likely (mistakenly) ran in inference mode."};
		CHECK_THROW(dL_ddLdinput != nullptr && input != nullptr);
		const uint32_t w = m->model->input_width();
		MatViewMut dx{dL_dinput, w, 1u};
		m->model->backward_backward_input((hipStream_t)stream, *ctx->ctx, n, MatView{input, w, 1u}, MatView{dL_ddLdinput, w, 1u}, dL_doutput, dL_ddLdoutput, dL_dinput ? &dx : nullptr, params,
		                                  dL_dparams, dL_dparams ? GradientMode::Overwrite : GradientMode::Ignore);
	});
}

void tcnn_context_destroy(tcnn_context_t ctx) { delete ctx; }

uint32_t tcnn_module_n_input_dims(tcnn_module_t m) { return m->model->input_width(); }
uint32_t tcnn_module_n_output_dims(tcnn_module_t m) { return m->model->padded_output_width(); }
size_t tcnn_module_n_params(tcnn_module_t m) { return m->model->n_params(); }
size_t tcnn_module_list_scatters(tcnn_module_t m) { return (size_t)m->model->list_scatters(); }
int tcnn_module_param_precision(tcnn_module_t m) { return (int)m->model->precision(); }
int tcnn_module_output_precision(tcnn_module_t m) { return (int)m->model->precision(); }

int tcnn_module_initialize_params(tcnn_module_t m, uint64_t seed, float* params_full_precision, float scale) {
	return guarded([&] {
		CHECK_THROW(m && m->model);
		Pcg32 rng{seed}; // cpp_api.cu:134
		m->model->initialize_params(rng, params_full_precision, scale);
		HIP_CHECK_THROW(hipDeviceSynchronize());
	});
}

const char* tcnn_module_hyperparams(tcnn_module_t m) {
	m->hyperparams_text = m->model->hyperparams().dump();
	return m->hyperparams_text.c_str();
}

const char* tcnn_module_name(tcnn_module_t m) {
	m->name_text = m->model->name();
	return m->name_text.c_str();
}

// ---------------------------------------------------------------------------------------------------------- trainer
int tcnn_create_from_config_seeded(uint32_t n_input_dims, uint32_t n_output_dims, const char* config_json, uint32_t seed, tcnn_trainer_t* out) {
	switches_reload(); // the A/B switches are read once per model (tcnn_common.h: Switches)
	return guarded([&] {
		CHECK_THROW(out != nullptr);
		auto t = std::make_unique<tcnn_trainer_s>();
		t->trainer.reset(new Trainer{n_input_dims, n_output_dims, parse_or_empty(config_json), seed});
		*out = t.release();
	});
}

int tcnn_create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const char* config_json, tcnn_trainer_t* out) {
	return tcnn_create_from_config_seeded(n_input_dims, n_output_dims, config_json, 1337, out);
}

void tcnn_trainer_destroy(tcnn_trainer_t t) { delete t; }

int tcnn_trainer_training_step(tcnn_trainer_t t, tcnn_stream_t stream, uint32_t n, const float* input, int input_layout, const float* target, const float* data_pdf,
                               int run_optimizer, float* dL_dinput, int use_inference_params, int gradient_mode, const void* external_dL_dy, tcnn_train_ctx_t* ctx_out) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer && ctx_out);
		Trainer& tr = *t->trainer;
		const uint32_t w = tr.model().input_width();
		MatViewMut dx = make_view_mut(dL_dinput, w, n, input_layout);
		auto c = std::make_unique<tcnn_train_ctx_s>();
		c->ctx = tr.training_step((hipStream_t)stream, n, make_view(input, w, n, input_layout), target, data_pdf, run_optimizer != 0, dL_dinput ? &dx : nullptr,
		                          use_inference_params != 0, (GradientMode)gradient_mode, external_dL_dy);
		*ctx_out = c.release();
	});
}

int tcnn_trainer_loss(tcnn_trainer_t t, tcnn_stream_t stream, tcnn_train_ctx_t ctx, float* loss_out) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer && ctx && ctx->ctx && loss_out);
		*loss_out = t->trainer->loss((hipStream_t)stream, *ctx->ctx);
	});
}

int tcnn_trainer_forward(tcnn_trainer_t t, tcnn_stream_t stream, float loss_scale, uint32_t n, const float* input, int input_layout, const float* target, const float* data_pdf,
                         int use_inference_params, int prepare_input_gradients, const void* external_dL_dy, tcnn_train_ctx_t* ctx_out) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer && ctx_out);
		Trainer& tr = *t->trainer;
		auto c = std::make_unique<tcnn_train_ctx_s>();
		c->ctx = tr.forward((hipStream_t)stream, loss_scale, n, make_view(input, tr.model().input_width(), n, input_layout), target, data_pdf, use_inference_params != 0,
		                    prepare_input_gradients != 0, external_dL_dy);
		*ctx_out = c.release();
	});
}

int tcnn_trainer_backward(tcnn_trainer_t t, tcnn_stream_t stream, tcnn_train_ctx_t ctx, uint32_t n, const float* input, int input_layout, float* dL_dinput,
                          int use_inference_params, int gradient_mode) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer && ctx && ctx->ctx);
		Trainer& tr = *t->trainer;
		const uint32_t w = tr.model().input_width();
		MatViewMut dx = make_view_mut(dL_dinput, w, n, input_layout);
		tr.backward((hipStream_t)stream, *ctx->ctx, n, make_view(input, w, n, input_layout), dL_dinput ? &dx : nullptr, use_inference_params != 0, (GradientMode)gradient_mode);
	});
}

int tcnn_trainer_optimizer_step(tcnn_trainer_t t, tcnn_stream_t stream, float loss_scale) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer);
		t->trainer->optimizer_step((hipStream_t)stream, loss_scale);
	});
}

void tcnn_train_ctx_destroy(tcnn_train_ctx_t ctx) { delete ctx; }
const void* tcnn_train_ctx_output(tcnn_train_ctx_t ctx) { return ctx->ctx->output.data(); }
// compact contexts (model.h TrainContext::compact) produce the two padded matrices on first access
const void* tcnn_train_ctx_dL_doutput(tcnn_train_ctx_t ctx) {
	if (guarded([&] { ctx->ctx->materialize(); }) != 0) return nullptr;
	return ctx->ctx->dL_doutput_ptr;
}
const float* tcnn_train_ctx_L(tcnn_train_ctx_t ctx) {
	if (guarded([&] { ctx->ctx->materialize(); }) != 0) return nullptr;
	return ctx->ctx->L.as<float>();
}

int tcnn_trainer_inference(tcnn_trainer_t t, tcnn_stream_t stream, uint32_t n, const float* input, int input_layout, float* output, int output_layout, int use_inference_params) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer);
		Trainer& tr = *t->trainer;
		tr.inference((hipStream_t)stream, n, make_view(input, tr.model().input_width(), n, input_layout), make_view_mut(output, tr.model().output_width(), n, output_layout),
		             use_inference_params != 0);
	});
}

int tcnn_trainer_inference_mixed_precision(tcnn_trainer_t t, tcnn_stream_t stream, uint32_t n, const float* input, int input_layout, void* output_half, int use_inference_params) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer && (output_half || n == 0));
		Trainer& tr = *t->trainer;
		tr.inference_mixed_precision((hipStream_t)stream, n, make_view(input, tr.model().input_width(), n, input_layout), output_half, use_inference_params != 0);
	});
}

size_t tcnn_trainer_n_params(tcnn_trainer_t t) { return t->trainer->n_params(); }
size_t tcnn_trainer_params_updated_in_flush(tcnn_trainer_t t) { return t->trainer->params_updated_in_flush(); }
size_t tcnn_trainer_image_preps(tcnn_trainer_t t) { return t->trainer->image_preps(); }
size_t tcnn_trainer_optimizer_prologue_steps(tcnn_trainer_t t) {
	try { return (size_t)t->trainer->prologue_steps(); } catch (const std::exception& e) { g_last_error = e.what(); return (size_t)-1; }
}
size_t tcnn_trainer_scatter_wide_fallbacks(tcnn_trainer_t t) {
	try { return (size_t)t->trainer->scatter_wide_fallbacks(); } catch (const std::exception& e) { g_last_error = e.what(); return (size_t)-1; }
}
size_t tcnn_trainer_list_scatters(tcnn_trainer_t t) { return (size_t)t->trainer->list_scatters(); }
int tcnn_train_ctx_keeps_weight_gradient_slabs(tcnn_trainer_t t, tcnn_train_ctx_t ctx) {
	return (t && ctx && ctx->ctx && ctx->ctx->model_ctx && t->trainer->model().context_keeps_slabs(*ctx->ctx->model_ctx)) ? 1 : 0;
}
uint32_t tcnn_trainer_padded_output_width(tcnn_trainer_t t) { return t->trainer->model().padded_output_width(); }
float* tcnn_trainer_params_full_precision(tcnn_trainer_t t) { return t->trainer->params_full_precision(); }
void* tcnn_trainer_params(tcnn_trainer_t t) { return t->trainer->params(); }
void* tcnn_trainer_params_inference(tcnn_trainer_t t) {
	void* p = t->trainer->params_inference();
	return p == t->trainer->params_unexposed() ? t->trainer->params() : p; // the training parameters themselves: handed out like tcnn_trainer_params
}
void* tcnn_trainer_param_gradients(tcnn_trainer_t t) { return t->trainer->param_gradients(); }

int tcnn_trainer_set_params_full_precision(tcnn_trainer_t t, const float* params, size_t n_params, int device_ptr) {
	return guarded([&] { t->trainer->set_params_full_precision(params, n_params, device_ptr != 0); });
}

int tcnn_trainer_set_params(tcnn_trainer_t t, const void* params_half, size_t n_params, int device_ptr) {
	return guarded([&] { t->trainer->set_params(params_half, n_params, device_ptr != 0); });
}

int tcnn_trainer_initialize_params(tcnn_trainer_t t) {
	return guarded([&] { t->trainer->initialize_params(); });
}

int tcnn_trainer_update_hyperparams(tcnn_trainer_t t, const char* json) {
	return guarded([&] { t->trainer->update_hyperparams(parse_or_empty(json)); });
}

const char* tcnn_trainer_hyperparams(tcnn_trainer_t t) {
	t->hyperparams_text = t->trainer->hyperparams().dump();
	return t->hyperparams_text.c_str();
}

const char* tcnn_trainer_network_hyperparams(tcnn_trainer_t t) {
	t->network_hyperparams_text = t->trainer->model().hyperparams().dump();
	return t->network_hyperparams_text.c_str();
}

uint32_t tcnn_trainer_optimizer_step_count(tcnn_trainer_t t) { return t->trainer->optimizer().step_count(); }

int tcnn_trainer_profile_next_step(tcnn_trainer_t t) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer);
		t->trainer->profile().arm();
	});
}

int tcnn_trainer_profile_collect(tcnn_trainer_t t, tcnn_stream_t stream, float* ms_per_piece, uint32_t* n_steps) {
	return guarded([&] {
		CHECK_THROW(t && t->trainer && ms_per_piece && n_steps);
		*n_steps = t->trainer->profile().collect((hipStream_t)stream, ms_per_piece);
	});
}

int tcnn_trainer_serialize(tcnn_trainer_t t, int serialize_optimizer, const void** out_bytes, size_t* out_size) {
	return guarded([&] {
		CHECK_THROW(out_bytes != nullptr && out_size != nullptr);
		t->snapshot = Json::to_msgpack(t->trainer->serialize(serialize_optimizer != 0));
		*out_bytes = t->snapshot.data();
		*out_size = t->snapshot.size();
	});
}

int tcnn_trainer_deserialize(tcnn_trainer_t t, const void* bytes, size_t size) {
	return guarded([&] {
		CHECK_THROW(bytes != nullptr && size > 0);
		t->trainer->deserialize(Json::from_msgpack((const uint8_t*)bytes, size));
	});
}

} // extern "C"
