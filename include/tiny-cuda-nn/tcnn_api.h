// tcnn_api.h -- the C++ header surface that callers of the reference compile against (samples/mlp_learning_an_image.cu,
// instant-ngp style code), provided over libtcnn_amd.so's C ABI (include/tcnn_amd.h).  Header-only, no HIP or torch types:
// any C++14 compiler, link with -ltcnn_amd.
//
// What is mirrored (reference file:line):
//   tcnn::json                                     nlohmann::json as used by config.h:53 / the samples' config literals (the real one when <json/json.hpp> is on the include path)
//   tcnn::GPUMemory<T>                             gpu_memory.h:60-392   (allocation, host <-> device copies, memset)
//   tcnn::GPUMatrixDynamic<T>, GPUMatrix<T, L>     gpu_matrix.h:115-470  (m x n, column-major by default = [n][m] in memory)
//   tcnn::Loss<T>, create_loss<T>                  loss.h:48-77, src/loss.cu:54-68          (L2, RelativeL2)
//   tcnn::Optimizer<T>, create_optimizer<T>        optimizer.h:44-95, src/optimizer.cu:50-82 (Adam)
//   tcnn::NetworkWithInputEncoding<T>              network_with_input_encoding.h:40-180
//   tcnn::Trainer<T, PARAMS_T, COMPUTE_T>          trainer.h:48-363  (training_step, loss, inference via the network, params)
//   tcnn::TrainableModel, create_from_config       config.h:46-63
//   tcnn::free_all_gpu_memory_arenas               gpu_memory.h:751
// Errors are std::runtime_error carrying the library's message, as in the reference (common_host.h:71-110).
// The objects are thin: a Trainer owns the native trainer handle; the NetworkWithInputEncoding, Loss and Optimizer objects
// carry configuration until a Trainer binds them (the reference's Trainer likewise takes ownership of the parameters,
// trainer.h:322-336).  Like the reference: one GPU per process, not thread-safe.
#pragma once

#include "../tcnn_amd.h"
#include "json_select.h" // tcnn::json: nlohmann::json where <json/json.hpp> is on the include path, else json_lite.h's
#include "random.h" // trainer.h includes random.h in the reference: callers get default_rng_t / generate_random_uniform from config.h

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace tcnn {

typedef void* stream_t; // hipStream_t (cudaStream_t in the reference)

static constexpr uint32_t BATCH_SIZE_GRANULARITY = 256; // common.h:235

struct half { uint16_t bits; }; // IEEE binary16 storage; arithmetic happens on the device
typedef half network_precision_t; // common.h:86-90 (TCNN_HALF_PRECISION)

enum class MatrixLayout { RowMajor = 0, SoA = 0, ColumnMajor = 1, AoS = 1 }; // common.h:157-162
static constexpr MatrixLayout RM = MatrixLayout::RowMajor;
static constexpr MatrixLayout CM = MatrixLayout::ColumnMajor;

enum class GradientMode { Ignore = 0, Overwrite = 1, Accumulate = 2 };

namespace detail {
inline void check(int rc) {
	if (rc != TCNN_OK) throw std::runtime_error{tcnn_last_error()};
}
inline std::string to_text(const json& j) { return j.dump(); }
inline std::string to_text(const std::string& s) { return s; }
inline std::string to_text(const char* s) { return s; }
template <typename J> auto to_text(const J& j) -> decltype(j.dump()) { return j.dump(); } // e.g. nlohmann::json
} // namespace detail

inline void free_all_gpu_memory_arenas() { tcnn_free_temporary_memory(); }

// ------------------------------------------------------------------------------------------------------------ GPUMemory
template <typename T>
class GPUMemory {
public:
	GPUMemory() = default;
	explicit GPUMemory(size_t size) { resize(size); }
	GPUMemory(const GPUMemory&) = delete;
	GPUMemory& operator=(const GPUMemory&) = delete;
	GPUMemory(GPUMemory&& o) noexcept { *this = std::move(o); }
	GPUMemory& operator=(GPUMemory&& o) noexcept {
		std::swap(m_data, o.m_data);
		std::swap(m_size, o.m_size);
		return *this;
	}
	~GPUMemory() { if (m_data) tcnn_gpu_free(m_data); }

	void resize(size_t size) {
		if (size == m_size) return;
		if (m_data) { detail::check(tcnn_gpu_free(m_data)); m_data = nullptr; }
		m_size = size;
		if (size) { void* p = nullptr; detail::check(tcnn_gpu_malloc(size * sizeof(T), &p)); m_data = (T*)p; }
	}
	void enlarge(size_t size) { if (size > m_size) resize(size); }
	void memset(int value) { detail::check(tcnn_gpu_memset(m_data, value, m_size * sizeof(T))); }
	void copy_from_host(const T* host, size_t n) { detail::check(tcnn_gpu_memcpy(m_data, host, n * sizeof(T), TCNN_MEMCPY_HOST_TO_DEVICE)); }
	void copy_from_host(const T* host) { copy_from_host(host, m_size); }
	void copy_from_host(const std::vector<T>& v) {
		if (v.size() < m_size) throw std::runtime_error{"Trying to copy " + std::to_string(m_size) + " elements, but vector size is only " + std::to_string(v.size()) + "."};
		copy_from_host(v.data(), m_size);
	}
	void resize_and_copy_from_host(const std::vector<T>& v) { resize(v.size()); copy_from_host(v.data(), v.size()); }
	void copy_to_host(T* host, size_t n) const { detail::check(tcnn_gpu_memcpy(host, m_data, n * sizeof(T), TCNN_MEMCPY_DEVICE_TO_HOST)); }
	void copy_to_host(T* host) const { copy_to_host(host, m_size); }
	void copy_to_host(std::vector<T>& v) const { v.resize(m_size); copy_to_host(v.data(), m_size); }
	void copy_from_device(const GPUMemory<T>& other) {
		if (other.m_size > m_size) resize(other.m_size);
		detail::check(tcnn_gpu_memcpy(m_data, other.m_data, other.m_size * sizeof(T), TCNN_MEMCPY_DEVICE_TO_DEVICE));
	}
	T* data() const { return m_data; }
	size_t size() const { return m_size; }
	size_t get_num_elements() const { return m_size; }
	size_t get_bytes() const { return m_size * sizeof(T); }
	size_t bytes() const { return get_bytes(); }

private:
	T* m_data = nullptr;
	size_t m_size = 0;
};

// ------------------------------------------------------------------------------------------------------------ GPUMatrix
// m rows (dims) x n columns (batch).  ColumnMajor (default): element (i, j) at data[i + j * m] = [n][m] in memory.
template <typename T>
class GPUMatrixDynamic {
public:
	GPUMatrixDynamic() = default;
	GPUMatrixDynamic(uint32_t m, uint32_t n, MatrixLayout layout = CM) : m_rows{m}, m_cols{n}, m_layout{layout}, m_owned{std::make_shared<GPUMemory<T>>((size_t)m * n)} { m_data = m_owned->data(); }
	GPUMatrixDynamic(T* data, uint32_t m, uint32_t n, MatrixLayout layout = CM) : m_rows{m}, m_cols{n}, m_data{data}, m_layout{layout} {}
	T* data() const { return m_data; }
	uint32_t m() const { return m_rows; }
	uint32_t n() const { return m_cols; }
	uint32_t rows() const { return m_rows; }
	uint32_t cols() const { return m_cols; }
	size_t n_elements() const { return (size_t)m_rows * m_cols; }
	size_t n_bytes() const { return n_elements() * sizeof(T); }
	MatrixLayout layout() const { return m_layout; }
	void memset(int value) { detail::check(tcnn_gpu_memset(m_data, value, n_bytes())); }
	std::vector<T> to_cpu_vector() const {
		std::vector<T> v(n_elements());
		detail::check(tcnn_gpu_memcpy(v.data(), m_data, n_bytes(), TCNN_MEMCPY_DEVICE_TO_HOST));
		return v;
	}

protected:
	uint32_t m_rows = 0, m_cols = 0;
	T* m_data = nullptr;
	MatrixLayout m_layout = CM;
	std::shared_ptr<GPUMemory<T>> m_owned;
};

template <typename T, MatrixLayout LAYOUT = MatrixLayout::ColumnMajor>
class GPUMatrix : public GPUMatrixDynamic<T> {
public:
	GPUMatrix() = default;
	GPUMatrix(uint32_t m, uint32_t n) : GPUMatrixDynamic<T>{m, n, LAYOUT} {}
	GPUMatrix(T* data, uint32_t m, uint32_t n) : GPUMatrixDynamic<T>{data, m, n, LAYOUT} {}
};

// ------------------------------------------------------------------------------------------ loss / optimizer / network
template <typename T>
class Loss {
public:
	explicit Loss(const json& params) : m_params(params) { // parentheses: braces would select json's initializer-list constructor
		const std::string otype = params.value("otype", "RelativeL2");
		std::string lower;
		for (char ch : otype) lower.push_back((char)((ch >= 'A' && ch <= 'Z') ? ch - 'A' + 'a' : ch));
		static const char* known[] = {"l2", "relativel2", "relativel2luminance", "l1", "relativel1", "mape", "smape", "crossentropy", "variance"}; // src/loss.cu:57-65
		bool ok = false;
		for (const char* k : known) ok = ok || lower == k;
		if (!ok) throw std::runtime_error{"Invalid loss type: " + otype}; // src/loss.cu:85-93
	}
	json hyperparams() const { return m_params; }

private:
	json m_params;
};
template <typename T> Loss<T>* create_loss(const json& params) { return new Loss<T>{params}; }

template <typename T>
class Optimizer {
public:
	explicit Optimizer(const json& params) : m_params(params) {}
	json hyperparams() const { return m_params; }
	void update_hyperparams(const json& params) { m_params = params; }

private:
	json m_params;
};
template <typename T> Optimizer<T>* create_optimizer(const json& params) { return new Optimizer<T>{params}; }

template <typename T, typename PARAMS_T, typename COMPUTE_T> class Trainer;

template <typename T>
class NetworkWithInputEncoding {
public:
	NetworkWithInputEncoding(uint32_t n_dims_to_encode, uint32_t n_output_dims, const json& encoding, const json& network)
	: m_n_input_dims{n_dims_to_encode}, m_n_output_dims{n_output_dims}, m_encoding(encoding), m_network(network) {}

	// object.h:147-176: float in, float out (unpadded width); needs a Trainer to have bound the parameters
	void inference(stream_t stream, const GPUMatrixDynamic<float>& input, GPUMatrixDynamic<float>& output, bool use_inference_params = true) {
		if (!m_trainer) throw std::runtime_error{"NetworkWithInputEncoding::inference: the network has no parameters yet (construct a Trainer first)"};
		if (input.m() != m_n_input_dims || output.m() != m_n_output_dims || input.n() != output.n()) throw std::runtime_error{"inference: matrix shapes do not match the network"};
		detail::check(tcnn_trainer_inference(m_trainer, stream, input.n(), input.data(), (int)input.layout(), output.data(), (int)output.layout(), use_inference_params ? 1 : 0));
	}
	void inference(const GPUMatrixDynamic<float>& input, GPUMatrixDynamic<float>& output) { inference(nullptr, input, output); }

	uint32_t input_width() const { return m_n_input_dims; }
	uint32_t output_width() const { return m_n_output_dims; }
	uint32_t padded_output_width() const { return m_trainer ? tcnn_trainer_padded_output_width(m_trainer) : (m_n_output_dims + 15) / 16 * 16; }
	size_t n_params() const { return m_trainer ? tcnn_trainer_n_params(m_trainer) : 0; }
	json hyperparams() const { return m_trainer ? json::parse(tcnn_trainer_network_hyperparams(m_trainer)) : json{{"otype", "NetworkWithInputEncoding"}, {"encoding", m_encoding}, {"network", m_network}}; }
	const json& encoding_config() const { return m_encoding; }
	const json& network_config() const { return m_network; }

private:
	template <typename A, typename B, typename C> friend class Trainer;
	uint32_t m_n_input_dims, m_n_output_dims;
	json m_encoding, m_network;
	tcnn_trainer_t m_trainer = nullptr; // borrowed from the Trainer that bound this network
};

// ---------------------------------------------------------------------------------------------------------------- Trainer
template <typename T, typename PARAMS_T, typename COMPUTE_T = T>
class Trainer {
public:
	struct ForwardContext { // trainer.h:89-95
		tcnn_train_ctx_t handle = nullptr;
		uint32_t n = 0, padded_output_width = 0;
		~ForwardContext() { if (handle) tcnn_train_ctx_destroy(handle); }
		const COMPUTE_T* output() const { return (const COMPUTE_T*)tcnn_train_ctx_output(handle); }         // [n][padded_output_width]
		const COMPUTE_T* dL_doutput() const { return (const COMPUTE_T*)tcnn_train_ctx_dL_doutput(handle); }
		const float* L() const { return tcnn_train_ctx_L(handle); }
	};

	Trainer(std::shared_ptr<NetworkWithInputEncoding<COMPUTE_T>> model, std::shared_ptr<Optimizer<PARAMS_T>> optimizer, std::shared_ptr<Loss<COMPUTE_T>> loss, uint32_t seed = 1337)
	: m_model{std::move(model)}, m_optimizer{std::move(optimizer)}, m_loss{std::move(loss)} {
		json config = json::object();
		config["encoding"] = m_model->encoding_config();
		config["network"] = m_model->network_config();
		config["optimizer"] = m_optimizer->hyperparams();
		config["loss"] = m_loss->hyperparams();
		detail::check(tcnn_create_from_config_seeded(m_model->input_width(), m_model->output_width(), config.dump().c_str(), seed, &m_handle));
		m_model->m_trainer = m_handle;
	}
	Trainer(const Trainer&) = delete;
	Trainer& operator=(const Trainer&) = delete;
	~Trainer() {
		if (m_model && m_model->m_trainer == m_handle) m_model->m_trainer = nullptr;
		if (m_handle) tcnn_trainer_destroy(m_handle);
	}

	// trainer.h:163-190
	std::unique_ptr<ForwardContext> training_step(stream_t stream, const GPUMatrixDynamic<T>& input, const GPUMatrix<float>& target, const GPUMatrix<float>* data_pdf = nullptr,
	                                              bool run_optimizer = true, GPUMatrixDynamic<T>* dL_dinput = nullptr, bool use_inference_params = false,
	                                              GradientMode gradient_mode = GradientMode::Overwrite, const GPUMatrix<COMPUTE_T>* external_dL_dy = nullptr) {
		if (input.n() != target.n()) throw std::runtime_error{"training_step: input and target batch sizes differ"};
		auto ctx = std::make_unique<ForwardContext>();
		detail::check(tcnn_trainer_training_step(m_handle, stream, input.n(), input.data(), (int)input.layout(), target.data(), data_pdf ? data_pdf->data() : nullptr, run_optimizer ? 1 : 0,
		                                         dL_dinput ? dL_dinput->data() : nullptr, use_inference_params ? 1 : 0, (int)gradient_mode, external_dL_dy ? (const void*)external_dL_dy->data() : nullptr,
		                                         &ctx->handle));
		ctx->n = input.n();
		ctx->padded_output_width = tcnn_trainer_padded_output_width(m_handle);
		return ctx;
	}
	std::unique_ptr<ForwardContext> training_step(const GPUMatrixDynamic<T>& input, const GPUMatrix<float>& target) { return training_step(nullptr, input, target); } // trainer.h:192

	float loss(stream_t stream, const ForwardContext& ctx) { // trainer.h:205-207
		float value = 0;
		detail::check(tcnn_trainer_loss(m_handle, stream, ctx.handle, &value));
		return value;
	}
	void optimizer_step(stream_t stream, float loss_scale) { detail::check(tcnn_trainer_optimizer_step(m_handle, stream, loss_scale)); } // trainer.h:155-157

	size_t n_params() const { return tcnn_trainer_n_params(m_handle); }                                              // trainer.h:338
	float* params_full_precision() const { return tcnn_trainer_params_full_precision(m_handle); }                    // trainer.h:226
	PARAMS_T* params() const { return (PARAMS_T*)tcnn_trainer_params(m_handle); }                                    // trainer.h:230
	PARAMS_T* params_inference() const { return (PARAMS_T*)tcnn_trainer_params_inference(m_handle); }                // trainer.h:234
	PARAMS_T* param_gradients() const { return (PARAMS_T*)tcnn_trainer_param_gradients(m_handle); }                  // trainer.h:238
	void set_params_full_precision(const float* params, size_t n, bool device_ptr = false) { detail::check(tcnn_trainer_set_params_full_precision(m_handle, params, n, device_ptr ? 1 : 0)); }
	void set_params(const PARAMS_T* params, size_t n, bool device_ptr = false) { detail::check(tcnn_trainer_set_params(m_handle, params, n, device_ptr ? 1 : 0)); }
	void initialize_params() { detail::check(tcnn_trainer_initialize_params(m_handle)); }                             // trainer.h:68-87
	void update_hyperparams(const json& params) { detail::check(tcnn_trainer_update_hyperparams(m_handle, params.dump().c_str())); } // trainer.h:213
	json hyperparams() const { return json::parse(tcnn_trainer_hyperparams(m_handle)); }
	uint32_t optimizer_step_count() const { return tcnn_trainer_optimizer_step_count(m_handle); }

	// trainer.h:275-315: the snapshot object (binary values inside); store it with json::to_msgpack like callers of the reference do
	json serialize(bool serialize_optimizer = false) {
		const void* bytes = nullptr;
		size_t size = 0;
		detail::check(tcnn_trainer_serialize(m_handle, serialize_optimizer ? 1 : 0, &bytes, &size));
		return json::from_msgpack(std::vector<uint8_t>((const uint8_t*)bytes, (const uint8_t*)bytes + size));
	}
	void deserialize(const json& data) {
		const std::vector<uint8_t> bytes = json::to_msgpack(data);
		detail::check(tcnn_trainer_deserialize(m_handle, bytes.data(), bytes.size()));
	}

	std::shared_ptr<NetworkWithInputEncoding<COMPUTE_T>> model() const { return m_model; }
	tcnn_trainer_t handle() const { return m_handle; }

private:
	std::shared_ptr<NetworkWithInputEncoding<COMPUTE_T>> m_model;
	std::shared_ptr<Optimizer<PARAMS_T>> m_optimizer;
	std::shared_ptr<Loss<COMPUTE_T>> m_loss;
	tcnn_trainer_t m_handle = nullptr;
};

// ----------------------------------------------------------------------------------------------------- config.h:46-63
struct TrainableModel {
	std::shared_ptr<Loss<network_precision_t>> loss;
	std::shared_ptr<Optimizer<network_precision_t>> optimizer;
	std::shared_ptr<NetworkWithInputEncoding<network_precision_t>> network;
	std::shared_ptr<Trainer<float, network_precision_t, network_precision_t>> trainer;
};

inline TrainableModel create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, json config) {
	const json encoding_opts = config.value("encoding", json::object());
	const json loss_opts = config.value("loss", json::object());
	const json optimizer_opts = config.value("optimizer", json::object());
	const json network_opts = config.value("network", json::object());
	std::shared_ptr<Loss<network_precision_t>> loss{create_loss<network_precision_t>(loss_opts)};
	std::shared_ptr<Optimizer<network_precision_t>> optimizer{create_optimizer<network_precision_t>(optimizer_opts)};
	auto network = std::make_shared<NetworkWithInputEncoding<network_precision_t>>(n_input_dims, n_output_dims, encoding_opts, network_opts);
	auto trainer = std::make_shared<Trainer<float, network_precision_t, network_precision_t>>(network, optimizer, loss);
	return {loss, optimizer, network, trainer};
}
template <typename J>
inline TrainableModel create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const J& config) { return create_from_config(n_input_dims, n_output_dims, json::parse(detail::to_text(config))); }

} // namespace tcnn
