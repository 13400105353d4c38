// cpp_api.h -- the reference's plugin interface tcnn::cpp (include/tiny-cuda-nn/cpp_api.h:50-115, src/cpp_api.cu:39-167),
// header-only over libtcnn_amd.so's C ABI.  This is what the reference's torch extension (bindings/torch/tinycudann/
// bindings.cpp) and any other tcnn::cpp::Module client compile against; see INTEGRATION.md section 1.
#pragma once

#include "../tcnn_amd.h"
#include "json_select.h"

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>

namespace tcnn { namespace cpp {

using json = tcnn::json;
typedef void* stream_t; // hipStream_t

enum class LogSeverity { Info = TCNN_LOG_INFO, Debug = TCNN_LOG_DEBUG, Warning = TCNN_LOG_WARNING, Error = TCNN_LOG_ERROR, Success = TCNN_LOG_SUCCESS }; // cpp_api.h:52-58
enum class Precision { Fp32 = TCNN_PRECISION_FP32, Fp16 = TCNN_PRECISION_FP16 };                                                                        // cpp_api.h:69-72

namespace detail {
inline void ok(int rc) { if (rc != TCNN_OK) throw std::runtime_error{tcnn_last_error()}; }
}

inline uint32_t batch_size_granularity() { return tcnn_batch_size_granularity(); }
inline int cuda_device() { int d = 0; detail::ok(tcnn_device(&d)); return d; }
inline void set_cuda_device(int device) { detail::ok(tcnn_set_device(device)); }
inline void free_temporary_memory() { tcnn_free_temporary_memory(); }
inline bool has_networks() { return tcnn_has_networks() != 0; }
inline float default_loss_scale(Precision p) { return tcnn_default_loss_scale((int)p); }
inline Precision preferred_precision() { return (Precision)tcnn_preferred_precision(); }

// cpp_api.h:80, cpp_api.cu:61-63 (bound by the torch extension, bindings.cpp:304).  The C ABI takes a function pointer and a user
// pointer; the std::function is kept here.  An empty callback restores the library's default sink (stderr).
namespace detail {
inline std::function<void(LogSeverity, const std::string&)>& log_callback() {
	static std::function<void(LogSeverity, const std::string&)> f;
	return f;
}
inline void log_trampoline(int severity, const char* message, void*) {
	if (log_callback()) log_callback()((LogSeverity)severity, std::string{message ? message : ""});
}
}
inline void set_log_callback(const std::function<void(LogSeverity, const std::string&)>& callback) {
	detail::log_callback() = callback;
	tcnn_set_log_callback(callback ? detail::log_trampoline : nullptr, nullptr);
}

struct Context { // cpp_api.h:82-84
	std::shared_ptr<void> ctx;
};

class Module { // cpp_api.h:86-111: all pointers are device pointers owned by the caller
public:
	explicit Module(tcnn_module_t handle) : m_handle{handle} {}
	Module(const Module&) = delete;
	Module& operator=(const Module&) = delete;
	~Module() { tcnn_module_destroy(m_handle); }

	void inference(stream_t stream, uint32_t n_elements, const float* input, void* output, void* params) { detail::ok(tcnn_module_inference(m_handle, stream, n_elements, input, output, params)); }
	Context forward(stream_t stream, uint32_t n_elements, const float* input, void* output, void* params, bool prepare_input_gradients) {
		tcnn_context_t c = nullptr;
		detail::ok(tcnn_module_forward(m_handle, stream, n_elements, input, output, params, prepare_input_gradients ? 1 : 0, &c));
		return Context{std::shared_ptr<void>{(void*)c, [](void* p) { tcnn_context_destroy((tcnn_context_t)p); }}};
	}
	void backward(stream_t stream, const Context& ctx, uint32_t n_elements, float* dL_dinput, const void* dL_doutput, void* dL_dparams, const float* input, const void* output, const void* params) {
		if (!ctx.ctx) throw std::runtime_error{"Module::backward: called with invalid context. forward likely (mistakenly) ran in inference mode."};
		detail::ok(tcnn_module_backward(m_handle, stream, (tcnn_context_t)ctx.ctx.get(), n_elements, dL_dinput, dL_doutput, dL_dparams, input, output, params));
	}
	void backward_backward_input(stream_t stream, const Context& ctx, uint32_t n_elements, const float* dL_ddLdinput, const float* input, const void* dL_doutput, void* dL_dparams, void* dL_ddLdoutput,
	                             float* dL_dinput, const void* params) {
		detail::ok(tcnn_module_backward_backward_input(m_handle, stream, (tcnn_context_t)ctx.ctx.get(), n_elements, dL_ddLdinput, input, dL_doutput, dL_dparams, dL_ddLdoutput, dL_dinput, params));
	}

	uint32_t n_input_dims() const { return tcnn_module_n_input_dims(m_handle); }
	uint32_t n_output_dims() const { return tcnn_module_n_output_dims(m_handle); } // padded width, cpp_api.cu:130
	size_t n_params() const { return tcnn_module_n_params(m_handle); }
	Precision param_precision() const { return (Precision)tcnn_module_param_precision(m_handle); }
	Precision output_precision() const { return (Precision)tcnn_module_output_precision(m_handle); }
	void initialize_params(size_t seed, float* params_full_precision, float scale = 1.0f) { detail::ok(tcnn_module_initialize_params(m_handle, seed, params_full_precision, scale)); }
	json hyperparams() const { return json::parse(tcnn_module_hyperparams(m_handle)); }
	std::string name() const { return tcnn_module_name(m_handle); }
	tcnn_module_t handle() const { return m_handle; }

private:
	tcnn_module_t m_handle;
};

// caller owns the result (the reference's bindings wrap it in a unique_ptr, bindings.cpp:265)
inline Module* create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const json& encoding, const json& network) {
	tcnn_module_t h = nullptr;
	detail::ok(tcnn_create_network_with_input_encoding(n_input_dims, n_output_dims, encoding.dump().c_str(), network.dump().c_str(), &h));
	return new Module{h};
}
inline Module* create_network(uint32_t n_input_dims, uint32_t n_output_dims, const json& network) {
	tcnn_module_t h = nullptr;
	detail::ok(tcnn_create_network(n_input_dims, n_output_dims, network.dump().c_str(), &h));
	return new Module{h};
}
inline Module* create_encoding(uint32_t n_input_dims, const json& encoding, Precision requested_precision) {
	tcnn_module_t h = nullptr;
	detail::ok(tcnn_create_encoding(n_input_dims, encoding.dump().c_str(), (int)requested_precision, &h));
	return new Module{h};
}

}} // namespace tcnn::cpp
