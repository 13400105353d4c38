// cpp_api_amd.cpp -- the reference's plugin interface tcnn::cpp (include/tiny-cuda-nn/cpp_api.h:50-115) implemented over
// libtcnn_amd.so.  A maintainer of the reference compiles this file INSTEAD OF src/cpp_api.cu and links -ltcnn_amd; every
// client of tcnn::cpp::Module -- first of all the torch extension, bindings/torch/tinycudann/bindings.cpp -- then runs on
// MI355X unchanged.  Each method forwards to one entry point of include/tcnn_amd.h; C status codes become the exceptions the
// reference throws.  (cudaStream_t is whatever <cuda_runtime.h> of the build provides: a hipStream_t on ROCm, passed through
// as void*.)  tests/test_integration_shim.py compiles this file against the reference's own header.
#include <tiny-cuda-nn/cpp_api.h>

#include <tcnn_amd.h>

#include <functional>
#include <stdexcept>
#include <string>

namespace tcnn { namespace cpp {

static void ok(int rc) {
	if (rc != TCNN_OK) throw std::runtime_error{tcnn_last_error()};
}

namespace {
struct AmdContext : public tcnn::Context { // cpp_api.h:40-48: the context a forward pass hands to its backward passes
	tcnn_context_t h = nullptr;
	~AmdContext() override { tcnn_context_destroy(h); }
};

tcnn_context_t handle_of(const Context& ctx) {
	auto* c = dynamic_cast<AmdContext*>(ctx.ctx.get());
	if (!c) throw std::runtime_error{"Module::bwd: called with invalid context. fwd likely (mistakenly) ran in inference mode."};
	return c->h;
}

class AmdModule : public Module {
public:
	explicit AmdModule(tcnn_module_t h) : Module{(Precision)tcnn_module_param_precision(h), (Precision)tcnn_module_output_precision(h)}, m{h} {}
	~AmdModule() override { tcnn_module_destroy(m); }

	void inference(cudaStream_t stream, uint32_t n_elements, const float* input, void* output, void* params) override { // cpp_api.cu:68-82
		ok(tcnn_module_inference(m, (tcnn_stream_t)stream, n_elements, input, output, params));
	}

	Context forward(cudaStream_t stream, uint32_t n_elements, const float* input, void* output, void* params, bool prepare_input_gradients) override { // :84-95
		std::unique_ptr<AmdContext> c{new AmdContext};
		ok(tcnn_module_forward(m, (tcnn_stream_t)stream, n_elements, input, output, params, prepare_input_gradients ? 1 : 0, &c->h));
		Context result;
		result.ctx = std::move(c);
		return result;
	}

	void backward(cudaStream_t stream, const Context& ctx, uint32_t n_elements, float* dL_dinput, const void* dL_doutput, void* dL_dparams, const float* input,
	              const void* output, const void* params) override { // :97-109
		ok(tcnn_module_backward(m, (tcnn_stream_t)stream, handle_of(ctx), n_elements, dL_dinput, dL_doutput, dL_dparams, input, output, params));
	}

	void backward_backward_input(cudaStream_t stream, const Context& ctx, uint32_t n_elements, const float* dL_ddLdinput, const float* input, const void* dL_doutput,
	                             void* dL_dparams, void* dL_ddLdoutput, float* dL_dinput, const void* params) override { // :111-127
		ok(tcnn_module_backward_backward_input(m, (tcnn_stream_t)stream, handle_of(ctx), n_elements, dL_ddLdinput, input, dL_doutput, dL_dparams, dL_ddLdoutput, dL_dinput, params));
	}

	uint32_t n_input_dims() const override { return tcnn_module_n_input_dims(m); }
	uint32_t n_output_dims() const override { return tcnn_module_n_output_dims(m); } // the PADDED width, cpp_api.cu:130
	size_t n_params() const override { return tcnn_module_n_params(m); }

	void initialize_params(size_t seed, float* params_full_precision, float scale) override { // :133-136: pcg32{seed}, no seed_seq
		ok(tcnn_module_initialize_params(m, (uint64_t)seed, params_full_precision, scale));
	}

	json hyperparams() const override { return json::parse(tcnn_module_hyperparams(m)); }
	std::string name() const override { return tcnn_module_name(m); }

private:
	tcnn_module_t m;
};

// set_log_callback: the C ABI takes a function pointer + user pointer; the std::function lives here
std::function<void(LogSeverity, const std::string&)>& log_callback() {
	static std::function<void(LogSeverity, const std::string&)> f;
	return f;
}
void log_trampoline(int severity, const char* message, void*) {
	if (log_callback()) log_callback()((LogSeverity)severity, std::string{message ? message : ""});
}
} // namespace

uint32_t batch_size_granularity() { return tcnn_batch_size_granularity(); }

int cuda_device() {
	int device = 0;
	ok(tcnn_device(&device));
	return device;
}
void set_cuda_device(int device) { ok(tcnn_set_device(device)); }

void free_temporary_memory() { tcnn_free_temporary_memory(); }
bool has_networks() { return tcnn_has_networks() != 0; }
float default_loss_scale(Precision p) { return tcnn_default_loss_scale((int)p); }
Precision preferred_precision() { return (Precision)tcnn_preferred_precision(); }

void set_log_callback(const std::function<void(LogSeverity, const std::string&)>& callback) { // cpp_api.cu:61-63, bound at bindings.cpp:304
	log_callback() = callback;
	tcnn_set_log_callback(callback ? log_trampoline : nullptr, nullptr);
}

Module* create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const json& encoding, const json& network) { // cpp_api.cu:147-149
	tcnn_module_t h = nullptr;
	ok(tcnn_create_network_with_input_encoding(n_input_dims, n_output_dims, encoding.dump().c_str(), network.dump().c_str(), &h));
	return new AmdModule{h};
}

Module* create_network(uint32_t n_input_dims, uint32_t n_output_dims, const json& network) { // :151-153
	tcnn_module_t h = nullptr;
	ok(tcnn_create_network(n_input_dims, n_output_dims, network.dump().c_str(), &h));
	return new AmdModule{h};
}

Module* create_encoding(uint32_t n_input_dims, const json& encoding, Precision requested_precision) { // :156-165
	tcnn_module_t h = nullptr;
	ok(tcnn_create_encoding(n_input_dims, encoding.dump().c_str(), (int)requested_precision, &h));
	return new AmdModule{h};
}

}} // namespace tcnn::cpp
