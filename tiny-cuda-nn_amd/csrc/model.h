// model.h -- host-side object model of libtcnn_amd: the MI355X counterpart of the reference's
// Encoding / Network / NetworkWithInputEncoding / Loss / Optimizer / Trainer classes, reduced to the hot path
// (SURVEY.md section 8).  Everything here is orchestration: which kernel runs on which buffer in which order.
//
// Reference files restated (relative to /root/reference):
//   include/tiny-cuda-nn/encodings/grid.h:653-1208, oneblob.h:167-307, identity.h:88-190, src/encoding.cu:144-158
//   src/fully_fused_mlp.cu:636-891, src/network.cu:48-137
//   include/tiny-cuda-nn/network_with_input_encoding.h:40-192
//   include/tiny-cuda-nn/losses/{l2,relative_l2}.h, src/loss.cu:85-93
//   include/tiny-cuda-nn/optimizers/adam.h:122-327, src/optimizer.cu:50-82
//   include/tiny-cuda-nn/trainer.h:48-363, config.h:46-63, object.h:120-270
#pragma once

#include "json_lite.h"
#include "tcnn_common.h"
#include "mlp_side_jobs.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <unordered_map>
#include <vector>

namespace tcnn_amd {

// ------------------------------------------------------------------------------------------------------------------
// logging (common_host.h:46-69)
// ------------------------------------------------------------------------------------------------------------------
struct LogSink {
	void (*callback)(int, const char*, void*) = nullptr;
	void* user = nullptr;
};
inline LogSink& log_sink() { static LogSink s; return s; }
inline void log_message(int severity, const std::string& msg) {
	LogSink& s = log_sink();
	if (s.callback) s.callback(severity, msg.c_str(), s.user);
}

inline std::string to_lower(std::string s) {
	for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);
	return s;
}
inline bool equals_case_insensitive(const std::string& a, const std::string& b) { return to_lower(a) == to_lower(b); } // common_host.h:238-246

// ------------------------------------------------------------------------------------------------------------------
// Stream-ordered caching arena (the role of GPUMemoryArena, gpu_memory.h:426-720): blocks are handed out per stream and
// returned to that stream's free list when their owner dies, so reuse is ordered by the stream itself.
// ------------------------------------------------------------------------------------------------------------------
class Arena {
public:
	static Arena& instance() { static Arena a; return a; }

	void* alloc(hipStream_t stream, size_t bytes, size_t* capacity_out) {
		bytes = std::max<size_t>((bytes + 255) / 256 * 256, 256);
		std::lock_guard<std::mutex> lock{m_mutex};
		int dev = 0;
		HIP_CHECK_THROW(hipGetDevice(&dev));
		auto& fl = m_free[key(dev, stream)];
		auto it = fl.lower_bound(bytes);
		if (it != fl.end() && it->first <= bytes * 2 + (1u << 20)) {
			void* p = it->second;
			*capacity_out = it->first;
			fl.erase(it);
			return p;
		}
		void* p = nullptr;
		hipError_t err = hipMalloc(&p, bytes);
		if (err != hipSuccess) {
			release_locked();
			HIP_CHECK_THROW(hipMalloc(&p, bytes));
		}
		m_total += bytes;
		*capacity_out = bytes;
		return p;
	}

	void free(hipStream_t stream, int dev, void* p, size_t capacity) {
		std::lock_guard<std::mutex> lock{m_mutex};
		m_free[key(dev, stream)].emplace(capacity, p);
	}

	void release_all() { // free_all_gpu_memory_arenas, gpu_memory.h:751
		std::lock_guard<std::mutex> lock{m_mutex};
		release_locked();
	}

	size_t total_bytes() const { return m_total; }

private:
	static std::pair<int, hipStream_t> key(int dev, hipStream_t s) { return {dev, s}; }
	void release_locked() {
		(void)hipDeviceSynchronize();
		for (auto& kv : m_free) {
			for (auto& blk : kv.second) {
				(void)hipFree(blk.second);
				m_total -= blk.first;
			}
			kv.second.clear();
		}
	}
	std::mutex m_mutex;
	std::map<std::pair<int, hipStream_t>, std::multimap<size_t, void*>> m_free;
	size_t m_total = 0;
};

class ArenaBuf {
public:
	ArenaBuf() = default;
	ArenaBuf(hipStream_t stream, size_t bytes) : m_stream{stream}, m_bytes{bytes} {
		if (bytes == 0) return;
		HIP_CHECK_THROW(hipGetDevice(&m_dev));
		m_ptr = Arena::instance().alloc(stream, bytes, &m_capacity);
	}
	~ArenaBuf() { reset(); }
	ArenaBuf(const ArenaBuf&) = delete;
	ArenaBuf& operator=(const ArenaBuf&) = delete;
	ArenaBuf(ArenaBuf&& o) noexcept { *this = std::move(o); }
	ArenaBuf& operator=(ArenaBuf&& o) noexcept {
		if (this != &o) {
			reset();
			m_ptr = o.m_ptr; m_stream = o.m_stream; m_bytes = o.m_bytes; m_capacity = o.m_capacity; m_dev = o.m_dev;
			o.m_ptr = nullptr; o.m_bytes = 0;
		}
		return *this;
	}
	void reset() {
		if (m_ptr) Arena::instance().free(m_stream, m_dev, m_ptr, m_capacity);
		m_ptr = nullptr;
		m_bytes = 0;
	}
	void* data() const { return m_ptr; }
	template <typename T> T* as() const { return (T*)m_ptr; }
	size_t bytes() const { return m_bytes; }
	explicit operator bool() const { return m_ptr != nullptr; }
private:
	void* m_ptr = nullptr;
	hipStream_t m_stream = nullptr;
	size_t m_bytes = 0, m_capacity = 0;
	int m_dev = 0;
};

// Persistent device allocation (GPUMemory<T>, gpu_memory.h:60-392)
class DeviceBuf {
public:
	DeviceBuf() = default;
	explicit DeviceBuf(size_t bytes) { resize(bytes); }
	~DeviceBuf() { if (m_ptr) (void)hipFree(m_ptr); }
	DeviceBuf(const DeviceBuf&) = delete;
	DeviceBuf& operator=(const DeviceBuf&) = delete;
	DeviceBuf(DeviceBuf&& o) noexcept : m_ptr{o.m_ptr}, m_bytes{o.m_bytes} { o.m_ptr = nullptr; o.m_bytes = 0; }
	DeviceBuf& operator=(DeviceBuf&& o) noexcept {
		if (this != &o) {
			if (m_ptr) (void)hipFree(m_ptr);
			m_ptr = o.m_ptr; m_bytes = o.m_bytes;
			o.m_ptr = nullptr; o.m_bytes = 0;
		}
		return *this;
	}
	void resize(size_t bytes) {
		if (bytes == m_bytes) return;
		if (m_ptr) { (void)hipFree(m_ptr); m_ptr = nullptr; }
		m_bytes = bytes;
		if (bytes) HIP_CHECK_THROW(hipMalloc(&m_ptr, bytes));
	}
	void memset(int v) { if (m_bytes) HIP_CHECK_THROW(hipMemset(m_ptr, v, m_bytes)); }
	void* data() const { return m_ptr; }
	template <typename T> T* as() const { return (T*)m_ptr; }
	size_t bytes() const { return m_bytes; }
private:
	void* m_ptr = nullptr;
	size_t m_bytes = 0;
};

// ------------------------------------------------------------------------------------------------------------------
// pcg32 (dependencies/pcg32/pcg32.h:40-166, the published PCG-XSH-RR 64/32 generator)
// ------------------------------------------------------------------------------------------------------------------
struct Pcg32 {
	uint64_t st[2] = {0x853c49e6748fea9bULL, 0xda3e39cb94b95bdbULL};
	static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;
	Pcg32() = default;
	explicit Pcg32(uint64_t initstate, uint64_t initseq = 1) {
		st[0] = 0;
		st[1] = (initseq << 1u) | 1u;
		next_uint();
		st[0] += initstate;
		next_uint();
	}
	uint32_t next_uint() {
		const uint64_t old = st[0];
		st[0] = old * MULT + st[1];
		const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		const uint32_t rot = (uint32_t)(old >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	float next_float() {
		const uint32_t u = (next_uint() >> 9) | 0x3f800000u;
		float f;
		memcpy(&f, &u, 4);
		return f - 1.0f;
	}
};

// ------------------------------------------------------------------------------------------------------------------
// Encodings
// ------------------------------------------------------------------------------------------------------------------
struct EncodingContext {
	ArenaBuf to_reduce; // Composite encoding with a Sum / Product reduction: the nested outputs [nested][n][width]
	ArenaBuf dy_dx;       // grid only: float [n][L*F][D]
	ArenaBuf chunk_mask;  // grid only: uint64 [L][32][n/64] bit planes, which samples touch which scatter chunk (filter for the LDS scatter)
	ArenaBuf hit_elems;   // grid only, instead of chunk_mask: the hit lists of this batch (GridHitLists::elems); counters and validity below
	GridHitLists hit_lists;
	uint64_t hit_generation = 0; // the encoding's list counters are reused by later forward passes: lists are valid while this is the latest one on its stream
	const void* hit_stream = nullptr;
	uint32_t n = 0;
	mutable bool dy_records = false; // the level planes handed to backward() hold 16-byte scatter records {coordinates, gradients} (mlp_train_fused); set by the caller of backward()
	// Set by the caller of backward() for a step whose optimizer update may ride on the gradient kernel (AdamInFlush, arrays
	// indexed from this encoding's first parameter).  An encoding that takes the offer reports in adam_done which of its
	// parameters it has updated; whatever is not listed there is still the optimizer's to do.
	mutable const AdamInFlush* adam = nullptr;
	mutable ParamRanges adam_done;
	// likewise: a small job the backward pass may carry on one of its launches (MlpReduceJob::taken reports it)
	mutable const MlpReduceJob* reduce_job = nullptr;
	// likewise: the optimizer's launch offers to run the scatter's finalize pass (and the reduce job) as its prologue (AdamPrologue,
	// tcnn_common.h).  A backward pass that takes the offer fills it in, sets ->pending and launches no finalize pass of its own.
	mutable AdamPrologue* prologue = nullptr;
	std::vector<EncodingContext> nested; // Composite: one context per nested encoding
};

class Encoding {
public:
	virtual ~Encoding() {}
	virtual uint32_t input_width() const = 0;
	virtual uint32_t output_width() const = 0; // unpadded
	uint32_t padded_output_width() const { return output_width() + m_n_to_pad; }
	virtual uint32_t required_output_alignment() const { return 1; }
	virtual void set_padded_output_width(uint32_t padded) { // encoding.h: each encoding pads its own output with ones
		CHECK_THROW(padded >= output_width());
		m_n_to_pad = padded - output_width();
	}
	void set_alignment(uint32_t alignment) { // encoding.h:70-72
		const uint32_t a = std::lcm(alignment, required_output_alignment());
		set_padded_output_width(next_multiple(output_width(), a));
	}
	virtual size_t n_params() const { return 0; }
	virtual void initialize_params(Pcg32& rng, float* params_full_precision, float scale) {}
	// out: [n][padded_output_width] T (T = float if fp32 else half)
	// prepare_param_gradients: a backward pass with parameter gradients will follow (lets the grid record its scatter filter)
	virtual EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) = 0;
	// dL_dy [n][padded] T; grads: T[n_params] or nullptr (Ignore)
	// dy_planes: dL_dy is laid out as level planes [padded / F][n][F] (only if level_plane_features() allowed it), else AoS
	virtual void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) = 0;
	// second-order input gradients (object.h:278-288); the grid encoding (grid.h:902-1026) and PPNG3 (ppng_3.h:609-676) provide them
	virtual void backward_backward_input(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, MatView dL_ddLdx, const void* dL_dy, void* dL_ddLdy,
	                                     MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode) {
		throw std::runtime_error{"DifferentiableObject::backward_backward_input_impl: not implemented error"};
	}
	// > 0: this encoding's backward prefers dL_dy in level planes with that many features per plane (see k_grid_bwd_lds)
	virtual uint32_t level_plane_features(bool need_dL_dx, GradientMode mode) const { return 0; }
	// true: the encoding is half(x * scale + offset) padded with ones -- cheap enough to apply inside the consumer's load
	virtual bool as_identity(float& scale, float& offset) const { return false; }
	// n_bins if this is a half-precision OneBlob encoding the MLP kernels can evaluate inside their input load (MlpIo::x_oneblob_bins), else 0
	virtual uint32_t as_oneblob() const { return 0; }
	// true: backward() (with level planes allowed) prefers 16-byte records {coordinates, gradients} per (level, sample)
	virtual bool scatter_records_usable(MatView x) const { return false; }
	virtual uint32_t scatter_record_planes() const { return 0; } // 16-byte records per sample when scatter_records_usable()
	virtual uint64_t scatter_wide_fallbacks() { return 0; }      // grid only: tasks of the list-fed scatter that had to take the 64-bit passes
	virtual uint64_t list_scatters() const { return 0; }         // grid only: backward passes that ran the list-fed scatter (k_grid_scatter_lists)
	// > 0: forward_planes() can write the encoded batch as level planes [padded / F][n][F] (no input gradients in that form)
	virtual uint32_t forward_plane_features(uint32_t n) { return 0; }
	// prep_job (optional): a side job the forward kernel carries along -- the fragment images of the network behind the encoding
	virtual EncodingContext forward_planes(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out_planes, bool prepare_param_gradients,
	                                       const MlpPrepJob* prep_job = nullptr) {
		throw std::runtime_error{"Encoding: level-plane output is not available"};
	}
	virtual Json hyperparams() const = 0;
	bool fp32() const { return m_fp32; }
protected:
	explicit Encoding(bool fp32) : m_fp32{fp32} {}
	bool m_fp32;
	uint32_t m_n_to_pad = 0;
};

inline GridType string_to_grid_type(const std::string& s) {
	if (equals_case_insensitive(s, "Hash")) return GridType::Hash;
	if (equals_case_insensitive(s, "Dense")) return GridType::Dense;
	if (equals_case_insensitive(s, "Tiled") || equals_case_insensitive(s, "Tile")) return GridType::Tiled;
	throw std::runtime_error{"Invalid grid type: " + s};
}
inline HashType string_to_hash_type(const std::string& s) {
	if (equals_case_insensitive(s, "Prime")) return HashType::Prime;
	if (equals_case_insensitive(s, "CoherentPrime")) return HashType::CoherentPrime;
	if (equals_case_insensitive(s, "ReversedPrime")) return HashType::ReversedPrime;
	if (equals_case_insensitive(s, "Rng")) return HashType::Rng;
	throw std::runtime_error{"Invalid hash type: " + s};
}
inline InterpolationType string_to_interpolation_type(const std::string& s) {
	if (equals_case_insensitive(s, "Nearest")) return InterpolationType::Nearest;
	if (equals_case_insensitive(s, "Linear")) return InterpolationType::Linear;
	if (equals_case_insensitive(s, "Smoothstep")) return InterpolationType::Smoothstep;
	throw std::runtime_error{"Invalid interpolation type: " + s};
}

class GridEncoding : public Encoding {
public:
	GridEncoding(uint32_t n_dims_to_encode, const Json& enc, bool fp32) : Encoding{fp32} { // grid.h:1143-1208 + :668-730
		const uint32_t F = enc.value("n_features_per_level", 2u);
		if (F != 1 && F != 2 && F != 4 && F != 8) throw std::runtime_error{"GridEncoding: n_features_per_level must be 1, 2, 4, or 8."};
		const uint32_t log2_hashmap_size = enc.value("log2_hashmap_size", 19u);
		const std::string encoding_type = enc.value("otype", "Grid");
		const std::string default_type = equals_case_insensitive(encoding_type, "TiledGrid") ? "Tiled" : (equals_case_insensitive(encoding_type, "DenseGrid") ? "Dense" : "Hash");
		uint32_t n_features;
		if (enc.contains("n_features") || enc.contains("n_grid_features")) {
			n_features = enc.contains("n_features") ? enc.value("n_features", 0u) : enc.value("n_grid_features", 0u);
			if (enc.contains("n_levels")) throw std::runtime_error{"GridEncoding: may not specify n_features and n_levels simultaneously (one determines the other)"};
		} else {
			n_features = F * enc.value("n_levels", 16u);
		}
		const uint32_t n_levels = n_features / F;
		const GridType grid_type = string_to_grid_type(enc.value("type", default_type));
		const uint32_t base_resolution = enc.value("base_resolution", 16u);
		const float per_level_scale = enc.value("per_level_scale", grid_type == GridType::Dense ? std::exp(std::log(256.0f / (float)base_resolution) / (n_levels - 1)) : 2.0f);
		const HashType hash_type = string_to_hash_type(enc.value("hash", "CoherentPrime"));
		m_stochastic_interpolation = enc.value("stochastic_interpolation", false);
		if (m_stochastic_interpolation) throw std::runtime_error{"GridEncoding: stochastic_interpolation is not supported by this build"};
		if (n_dims_to_encode < 2 || n_dims_to_encode > 4) throw std::runtime_error{"GridEncoding: number of input dims must be 2 or 3."};
		if (n_levels > MAX_N_LEVELS) throw std::runtime_error{"GridEncoding: m_n_levels must be at most MAX_N_LEVELS=128"};
		if (n_features % F != 0) throw std::runtime_error{"GridEncoding: n_features must be a multiple of N_FEATURES_PER_LEVEL"};

		m_n_features = n_features;
		m_log2_hashmap_size = log2_hashmap_size;
		m_base_resolution = base_resolution;
		m_per_level_scale = per_level_scale;

		memset(&m_meta, 0, sizeof(m_meta));
		m_meta.n_pos_dims = n_dims_to_encode;
		m_meta.n_features_per_level = F;
		m_meta.n_levels = n_levels;
		m_meta.grid_type = (uint32_t)grid_type;
		m_meta.hash_type = (uint32_t)hash_type;
		m_meta.interpolation = (uint32_t)string_to_interpolation_type(enc.value("interpolation", "Linear"));
		static const uint32_t prime[7] = {1958374283u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
		static const uint32_t coherent[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
		static const uint32_t reversed[7] = {2165219737u, 1434869437u, 2097192037u, 3674653429u, 805459861u, 2654435761u, 1958374283u};
		const uint32_t* primes = hash_type == HashType::Prime ? prime : (hash_type == HashType::ReversedPrime ? reversed : coherent);
		for (uint32_t d = 0; d < 4; ++d) m_meta.primes[d] = primes[d]; // common_device.h:645-661

		const float log2_scale = std::log2(per_level_scale); // float overload (grid.h:694)
		uint32_t offset = 0;
		for (uint32_t i = 0; i < n_levels; ++i) {
			const float scale = exp2f(i * log2_scale) * base_resolution - 1.0f; // common_device.h:709-714
			const uint32_t resolution = (uint32_t)ceilf(scale) + 1;             // :716-718
			const uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
			uint32_t params_in_level = std::pow((float)resolution, n_dims_to_encode) > (float)max_params ? max_params : powi(resolution, n_dims_to_encode);
			params_in_level = next_multiple(params_in_level, 8u);
			if (grid_type == GridType::Tiled) params_in_level = std::min(params_in_level, powi(base_resolution, n_dims_to_encode));
			else if (grid_type == GridType::Hash) params_in_level = std::min(params_in_level, 1u << log2_hashmap_size);

			GridLevel& lv = m_meta.levels[i];
			lv.offset = offset;
			lv.size = params_in_level;
			lv.scale = scale;
			// grid_index's stride loop (common_device.h:692-704), evaluated once here, uint32 wrap-around included
			uint32_t stride = 1;
			for (uint32_t d = 0; d < 4; ++d) lv.stride[d] = 0;
			for (uint32_t d = 0; d < n_dims_to_encode && stride <= params_in_level; ++d) {
				lv.stride[d] = stride;
				stride *= resolution;
			}
			lv.hashed = (grid_type == GridType::Hash && params_in_level < stride) ? 1u : 0u;
			lv.size_mask = (params_in_level & (params_in_level - 1)) == 0 ? params_in_level - 1 : 0;
			m_resolutions.push_back(resolution);
			offset += params_in_level;
		}
		m_n_entries = offset;
		m_n_params = (size_t)offset * F;
		if (F >= 2) {
			grid_scatter_setup_levels(m_meta);
			for (uint32_t i = 0; i < n_levels; ++i) {
				// the sample filter describes 64 chunks per level; levels cut finer go through the binned kernels (up to 4096 chunks)
				m_scatter_levels_ok &= m_meta.levels[i].scatter_binned || m_meta.levels[i].scatter_n_chunks <= grid_scatter_max_chunks();
				m_any_binned |= m_meta.levels[i].scatter_binned != 0;
			}
		}
	}

	// device copy of the level table, uploaded on first use (construction itself never touches the GPU)
	const GridMeta* dev_meta() {
		if (!m_dev_meta.data()) {
			m_dev_meta.resize(sizeof(GridMeta));
			HIP_CHECK_THROW(hipMemcpy(m_dev_meta.data(), &m_meta, sizeof(GridMeta), hipMemcpyHostToDevice));
		}
		return (const GridMeta*)m_dev_meta.data();
	}

	static uint32_t powi(uint32_t base, uint32_t exponent) {
		uint32_t result = 1;
		for (uint32_t i = 0; i < exponent; ++i) result *= base;
		return result;
	}

	uint32_t input_width() const override { return m_meta.n_pos_dims; }
	uint32_t output_width() const override { return m_n_features; }
	uint32_t required_output_alignment() const override { return m_meta.n_features_per_level; }
	size_t n_params() const override { return m_n_params; }
	const GridMeta& meta() const { return m_meta; }
	const std::vector<uint32_t>& resolutions() const { return m_resolutions; }

	void initialize_params(Pcg32& rng, float* params_full_precision, float scale) override { // grid.h:1059-1062
		generate_random_uniform(nullptr, rng.st, n_params(), params_full_precision, -1e-4f * scale, 1e-4f * scale);
	}

	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		EncodingContext ctx;
		if ((!out && !prepare_input_gradients) || padded_output_width() == 0 || n == 0) return ctx;
		// The encoded batch as a matrix (callers with their own network: the PyTorch Encoding module, a grid nested in a Composite): the level-plane
		// kernel -- XCD-aware, and the one that writes the hit lists the fast gradient kernel reads -- and a transposition behind it, instead of
		// the AoS kernel and the bit-plane gradient kernel (3-D, F = 2, 2^18 samples through the PyTorch module: 1.74 -> 0.42 ms forward + backward; a Composite of such a grid and spherical harmonics in front of a 64x2 network: 1.89 -> 0.49 ms per training step).
		if (out && !prepare_input_gradients && switches().grid_rows_planes && forward_plane_features(n) > 0 && grid_planes_to_rows_supported(m_meta, n, padded_output_width())) {
			ArenaBuf planes{stream, (size_t)n * padded_output_width() * sizeof(uint16_t)};
			ctx = forward_planes(stream, n, x, params, planes.data(), prepare_param_gradients);
			grid_planes_to_rows(stream, m_meta, n, padded_output_width(), planes.data(), out, padded_output_width()); // (the planes of zeros behind the levels' included)
			return ctx;
		}
		if (prepare_input_gradients) ctx.dy_dx = ArenaBuf{stream, (size_t)n * m_n_features * m_meta.n_pos_dims * sizeof(float)};
		const bool want_filter = prepare_param_gradients && lds_scatter_usable() && n % 64 == 0;
		ArenaBuf mask;
		if (want_filter) mask = ArenaBuf{stream, (size_t)m_meta.n_levels * n * (GRID_FILTER_MAX_CHUNKS / 64) * sizeof(uint64_t)};
		grid_forward(stream, m_meta, dev_meta(), m_fp32, n, x, params, out, padded_output_width(), ctx.dy_dx.as<float>(), mask.as<uint64_t>());
		if (want_filter) {
			ctx.chunk_mask = ArenaBuf{stream, (size_t)m_meta.n_levels * grid_scatter_max_chunks() * (n / 64) * sizeof(uint64_t)};
			ctx.n = n;
			grid_mask_to_bits(stream, m_meta, dev_meta(), n, mask.as<uint64_t>(), ctx.chunk_mask.as<uint64_t>());
		}
		return ctx;
	}

	// TCNN_AMD_GRID_PLANES=0 keeps the AoS forward kernel inside the fused training step (A/B runs)
	static bool use_planes() { return switches().grid_planes; }
	uint32_t forward_plane_features(uint32_t n) override {
		// (a padded encoding -- 12 levels x 2 features in front of a 16-aligned network -- has whole planes of zeros behind its levels' planes)
		return (!m_fp32 && use_planes() && m_n_to_pad % m_meta.n_features_per_level == 0 && grid_planes_supported(m_meta, n)) ? m_meta.n_features_per_level : 0;
	}

	struct PlanesPlan {
		DeviceBuf dev_work;
		uint32_t max_items = 0, blocks_per_xcd = 0;
	};
	PlanesPlan& planes_plan(uint32_t n_, bool hit_lists) {
		const uint32_t n = n_ | (hit_lists ? 0x80000000u : 0u); // (key; n itself is below 2^24 + 1)
		auto it = m_planes_plans.find(n);
		if (it != m_planes_plans.end()) return *it->second;
		auto plan = std::make_unique<PlanesPlan>();
		std::vector<uint32_t> work;
		grid_planes_plan(m_meta, n_, hit_lists, work, plan->max_items, plan->blocks_per_xcd);
		plan->dev_work.resize(work.size() * sizeof(uint32_t));
		HIP_CHECK_THROW(hipMemcpy(plan->dev_work.data(), work.data(), work.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		return *(m_planes_plans[n] = std::move(plan));
	}

	EncodingContext forward_planes(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out_planes, bool prepare_param_gradients,
	                               const MlpPrepJob* prep_job = nullptr) override {
		EncodingContext ctx;
		CHECK_THROW(forward_plane_features(n) > 0);
		const bool want_filter = prepare_param_gradients && lds_scatter_usable();
		const bool want_lists = want_filter && hit_lists_usable(n);
		if (want_lists) {
			// Hit lists (k_grid_scatter_lists.hip): the storage is this pass's own; the stragglers' counts live in one of the stream's two
			// counter sets -- this launch counts in one and zeroes the other for the next forward pass, so no memset sits between the steps.
			HitCounters& hc = hit_counters(stream);
			GridHitLists& hl = ctx.hit_lists;
			hl.item_samples = grid_hit_item_samples(m_meta);
			hl.n_items = div_round_up(n, hl.item_samples);
			hl.item_capacity = hl.item_samples << (m_meta.n_pos_dims - 1);
			hl.straggler_capacity = n << (m_meta.n_pos_dims - 1);
			const size_t L = m_meta.n_levels;
			const size_t elems_bytes = L * hl.n_items * hl.item_capacity * GRID_HIT_WORDS * sizeof(uint32_t), sidx_bytes = next_multiple_sz(L * hl.n_items * hl.item_capacity * sizeof(uint16_t), 256);
			const size_t heads_bytes = next_multiple_sz(L * hl.n_items * GRID_HIT_HEADS * sizeof(uint32_t), 256);
			ctx.hit_elems = ArenaBuf{stream, elems_bytes + sidx_bytes + heads_bytes + L * hl.straggler_capacity * 2 * sizeof(uint32_t)};
			hl.elems = ctx.hit_elems.as<uint32_t>();
			hl.sidx = (uint16_t*)((char*)ctx.hit_elems.data() + elems_bytes);
			hl.heads = (uint32_t*)((char*)ctx.hit_elems.data() + elems_bytes + sidx_bytes);
			hl.stragglers = (uint32_t*)((char*)ctx.hit_elems.data() + elems_bytes + sidx_bytes + heads_bytes);
			hl.counts = hc.sets[hc.next].as<uint32_t>();
			hl.zero_counts = hc.sets[hc.next ^ 1].as<uint32_t>();
			hc.next ^= 1;
#ifdef TCNN_AMD_DEV
			if (const char* e = getenv("TCNN_AMD_FWD_LISTS_DEV")) hl.dev_flags = (uint32_t)atoi(e); // 1: no copy-out, 2: plain instead of streamed stores, 4: no offsets (1 and 4: wrong results)
#endif
			ctx.hit_generation = ++hc.generation;
			ctx.hit_stream = (const void*)stream;
			ctx.n = n;
		} else if (want_filter) {
			ctx.chunk_mask = ArenaBuf{stream, (size_t)m_meta.n_levels * grid_scatter_max_chunks() * (n / 64) * sizeof(uint64_t)};
			ctx.n = n;
		}
		PlanesPlan& plan = planes_plan(n, want_lists);
		grid_forward_planes(stream, m_meta, dev_meta(), plan.dev_work.as<uint32_t>(), plan.max_items, plan.blocks_per_xcd, n, x, params, out_planes, ctx.chunk_mask.as<uint64_t>(), prep_job,
		                    want_lists ? &ctx.hit_lists : nullptr);
		if (m_n_to_pad) HIP_CHECK_THROW(hipMemsetAsync((uint16_t*)out_planes + (size_t)n * m_n_features, 0, (size_t)n * m_n_to_pad * sizeof(uint16_t), stream)); // grid.h:749-759: the grid pads with zeros
		return ctx;
	}

	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		if ((!dL_dx && mode == GradientMode::Ignore) || n == 0) return;
		const size_t elem = m_fp32 ? 4 : 2;
		if (mode != GradientMode::Ignore) {
			CHECK_THROW(grads != nullptr);
			const bool scratch32 = !m_fp32 && m_meta.n_features_per_level == 1; // grid.h:660: F == 1 accumulates in fp32
			if (scratch32) {
				ArenaBuf tmp{stream, n_params() * sizeof(float)};
				if (mode == GradientMode::Overwrite) HIP_CHECK_THROW(hipMemsetAsync(tmp.data(), 0, n_params() * sizeof(float), stream));
				else cast_half_to_float(stream, n_params(), grads, tmp.as<float>());
				grid_backward(stream, m_meta, dev_meta(), true, n, x, dL_dy, false, padded_output_width(), tmp.data());
				cast_float_to_half(stream, n_params(), tmp.as<float>(), grads);
			} else if (lds_scatter_usable() && n % 64 == 0) {
				// MI355X path: LDS owner-computes scatter with exact integer accumulation; writes every element (k_grid_scatter.hip)
				// hit lists of this very batch, their counters not yet handed to a later forward pass of this stream
				const bool lists = ctx.hit_elems && ctx.n == n && ctx.hit_stream == (const void*)stream && hit_counters(stream).generation == ctx.hit_generation;
				if (lists) { // k_grid_scatter_lists.hip: a static plan, nothing to tune
					CHECK_THROW(!ctx.dy_records); // the listed elements bring entries and weights along: dL/dy is gathered from plain level planes (or rows)
					ListsPlan& lp = lists_plan(n, stream);
					const uint32_t F = m_meta.n_features_per_level;
					// dL/dy as rows (a caller's own network): into level planes first -- the streamed tasks read a level's gradients sample after
					// sample, 4 bytes out of every row's 64 otherwise (2^18 samples: the owners 82 us from rows, 42 from planes; the transposition 12)
					ArenaBuf dy_as_planes;
					if (!dy_planes && grid_planes_to_rows_supported(m_meta, n, m_n_features)) {
						dy_as_planes = ArenaBuf{stream, (size_t)n * m_n_features * sizeof(uint16_t)};
						grid_rows_to_planes(stream, m_meta, n, m_n_features, dL_dy, padded_output_width(), dy_as_planes.data());
					}
					const bool planes_now = dy_planes || dy_as_planes;
					const void* dy_src = dy_as_planes ? dy_as_planes.data() : dL_dy;
					const uint32_t dy_stride_sample = planes_now ? F : padded_output_width(), dy_stride_level = planes_now ? n * F : F;
					ctx.adam_done.clear(); // (this kernel does not carry the optimizer's update: measured 20 % slower in round 4)
					++m_list_scatters;
					// the finalize pass (and the reduce job with it) may be left to the optimizer's launch: no ranges, no job -> no launch here
					const bool defer = take_prologue(ctx, lp.dev_ranges.as<GridScatterRange>(), lp.host_ranges, lp.scratch.as<uint64_t>(), grads, mode);
					ArenaBuf gvals{stream, grid_list_gradients_bytes(m_meta, ctx.hit_lists)}; // dL/dy in list order: written and read by the two kernels of this call
					grid_backward_lists(stream, m_meta, dev_meta(), lp.dev_tasks.as<GridScatterTask>(), lp.n_tasks,
					                    lp.dev_ranges.as<GridScatterRange>(), defer ? 0u : lp.n_ranges, lp.scratch.as<uint64_t>(), n, x, dy_src, dy_stride_sample, dy_stride_level, grads, ctx.hit_lists,
					                    gvals.data(), mode == GradientMode::Accumulate, defer ? nullptr : ctx.reduce_job, hit_counters(stream).fallbacks.as<uint32_t>());
					if (dL_dx) {
						CHECK_THROW(ctx.dy_dx);
						CHECK_THROW(!dy_planes);
						grid_backward_input(stream, m_meta, m_fp32, n, dL_dy, padded_output_width(), ctx.dy_dx.as<float>(), *dL_dx);
					}
					return;
				}
				ScatterPlan& plan = scatter_plan(n, stream);
				const uint32_t F = m_meta.n_features_per_level;
				const uint64_t* mask = (ctx.chunk_mask && ctx.n == n) ? ctx.chunk_mask.as<uint64_t>() : nullptr;
				// The second filtered launch at this batch size is timed per task and the plan re-cut from the measured
				// per-level work (grid_scatter_plan): one stream synchronisation, once per (encoding, batch size).
				const bool tune = !plan.tuned && mask && plan.n_tasks > 0 && scatter_tuning_enabled() && ++plan.launches == 2;
				DeviceBuf times;
				if (tune) {
					times.resize((size_t)plan.n_tasks * 8 * sizeof(uint64_t));
					times.memset(0);
				}
				const uint32_t dy_stride_sample = dy_planes ? F : padded_output_width(), dy_stride_level = dy_planes ? n * F : F;
				const AdamInFlush* adam = nullptr;
				ctx.adam_done.clear();
				if (ctx.adam && mode == GradientMode::Overwrite) {
					if (dy_planes && ctx.dy_records) ctx.adam_done = plan.adam_ranges;
					if (!ctx.adam_done.empty()) adam = ctx.adam;
				}
				// (not in the step whose launch is timed for the tuner: the plan, ranges and scratch included, is rebuilt right after it)
				const bool defer = !tune && take_prologue(ctx, plan.dev_ranges.as<GridScatterRange>(), plan.host_ranges, plan.scratch.as<uint64_t>(), grads, mode);
				grid_backward_lds(stream, m_meta, dev_meta(), plan.dev_tasks.as<GridScatterTask>(), plan.n_tasks, plan.dev_ranges.as<GridScatterRange>(), defer ? 0u : plan.n_ranges,
				                  plan.scratch.as<uint64_t>(), n, x, dL_dy, dy_stride_sample, dy_stride_level, grads, mask,
				                  mode == GradientMode::Accumulate, dy_planes && ctx.dy_records, tune ? times.as<uint64_t>() : nullptr, adam, defer ? nullptr : ctx.reduce_job);
				if (m_any_binned) { // levels cut into more than 64 chunks (k_grid_bin.hip)
					CHECK_THROW(!(dy_planes && ctx.dy_records));
					ArenaBuf workspace{stream, grid_bin_workspace_bytes(m_meta, n)};
					grid_backward_binned(stream, m_meta, dev_meta(), n, x, dL_dy, dy_stride_sample, dy_stride_level, grads, mode == GradientMode::Accumulate, workspace.data(),
					                     hit_counters(stream).fallbacks.as<uint32_t>());
				}
				if (tune) {
					HIP_CHECK_THROW(hipStreamSynchronize(stream));
					std::vector<uint64_t> h((size_t)plan.n_tasks * 8);
					HIP_CHECK_THROW(hipMemcpy(h.data(), times.data(), h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
					const std::vector<float> level_us = grid_scatter_level_costs(m_meta, plan.host_tasks, h);
					build_scatter_plan(plan, n, &level_us);
					plan.tuned = true;
				}
			} else {
				CHECK_THROW(!dy_planes);
				if (mode == GradientMode::Overwrite) HIP_CHECK_THROW(hipMemsetAsync(grads, 0, n_params() * elem, stream)); // grid.h:858
				grid_backward(stream, m_meta, dev_meta(), m_fp32, n, x, dL_dy, m_fp32, padded_output_width(), grads);
			}
		}
		if (dL_dx) {
			CHECK_THROW(ctx.dy_dx);
			CHECK_THROW(!dy_planes);
			grid_backward_input(stream, m_meta, m_fp32, n, dL_dy, padded_output_width(), ctx.dy_dx.as<float>(), *dL_dx);
		}
	}

	// grid.h:902-1026
	void backward_backward_input(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, MatView dL_ddLdx, const void* dL_dy, void* dL_ddLdy,
	                             MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode) override {
		if ((!dL_ddLdy && mode == GradientMode::Ignore && !dL_dx) || padded_output_width() == 0 || n == 0) return;
		const size_t elem = m_fp32 ? 4 : 2;
		if (dL_ddLdy) CHECK_THROW(ctx.dy_dx); // needs the forward pass to have run with prepare_input_gradients
		const float* dy_dx = ctx.dy_dx.as<float>();
		if (mode != GradientMode::Ignore) {
			CHECK_THROW(grads != nullptr);
			const bool scratch32 = !m_fp32 && m_meta.n_features_per_level == 1; // grid.h:660: F == 1 accumulates in fp32 and casts (:934-941, :971-975)
			if (scratch32) {
				ArenaBuf tmp{stream, n_params() * sizeof(float)};
				if (mode == GradientMode::Overwrite) HIP_CHECK_THROW(hipMemsetAsync(tmp.data(), 0, n_params() * sizeof(float), stream));
				else cast_half_to_float(stream, n_params(), grads, tmp.as<float>());
				grid_backward_backward_input(stream, m_meta, dev_meta(), false, true, n, x, dL_ddLdx, dL_dy, padded_output_width(), params, dy_dx, tmp.data(), nullptr, nullptr);
				cast_float_to_half(stream, n_params(), tmp.as<float>(), grads);
			} else {
				if (mode == GradientMode::Overwrite) HIP_CHECK_THROW(hipMemsetAsync(grads, 0, n_params() * elem, stream)); // grid.h:943-945
				grid_backward_backward_input(stream, m_meta, dev_meta(), m_fp32, m_fp32, n, x, dL_ddLdx, dL_dy, padded_output_width(), params, dy_dx, grads, nullptr, nullptr);
			}
		}
		if (dL_ddLdy || dL_dx) {
			grid_backward_backward_input(stream, m_meta, dev_meta(), m_fp32, m_fp32, n, x, dL_ddLdx, dL_dy, padded_output_width(), params, dy_dx, nullptr, dL_ddLdy, dL_dx);
		}
	}

	uint32_t level_plane_features(bool need_dL_dx, GradientMode mode) const override {
		const uint32_t F = m_meta.n_features_per_level;
		return (lds_scatter_usable() && !need_dL_dx && mode != GradientMode::Ignore) ? F : 0;
	}

	// The MLP kernel writes {coordinates, gradient} records and the scatter does one gather per hit instead of two: measured on
	// C3a the scatter gains 12 us and the MLP kernel loses 6 us (4x the dX bytes).  TCNN_AMD_SCATTER_RECORDS=0 turns it off.
	bool scatter_records_usable(MatView x) const override {
		return switches().scatter_records && lds_scatter_usable() && !m_any_binned && grid_scatter_records_supported(m_meta) && x.stride_dim == 1 && x.stride_sample == m_meta.n_pos_dims;
	}

	uint32_t scatter_record_planes() const override { return grid_scatter_record_planes(m_meta); }

	// half precision, F >= 2, and every level's table either cut into at most 64 chunks (the sample filter) or binned
	bool lds_scatter_usable() const { return !m_fp32 && m_meta.n_features_per_level >= 2 && use_lds_scatter() && m_scatter_levels_ok; }

	// TCNN_AMD_GRID_SCATTER=atomic selects the reference-shaped global-atomic kernel (kept for A/B runs and as the fp32 / F==1 path)
	static bool use_lds_scatter() { return switches().grid_scatter_lds; }

	// The optimizer's launch has offered to run this backward pass's finalize pass (ctx.prologue): hand it the shared ranges, their scratch
	// and the reduce job.  Only where the gradient kernel does not apply the optimizer step itself (AdamInFlush) and a finalize pass exists.
	bool take_prologue(const EncodingContext& ctx, const GridScatterRange* dev_ranges, const std::vector<GridScatterRange>& ranges, uint64_t* scratch, void* grads, GradientMode mode) const {
		AdamPrologue* p = ctx.prologue;
		if (!p || !p->offered || p->pending || ctx.adam || (ranges.empty() && !ctx.reduce_job) || m_any_binned) return false;
		p->dev_ranges = dev_ranges;
		p->ranges = ranges;
		p->scratch = scratch;
		p->grad_base = grads;
		p->accumulate = mode == GradientMode::Accumulate;
		p->has_reduce = false;
		if (ctx.reduce_job) {
			p->has_reduce = true;
			p->reduce_elems = ctx.reduce_job->n_elems;
			p->reduce_slabs = ctx.reduce_job->n_slabs;
			p->slabs = ctx.reduce_job->slabs;
			p->reduce_accumulate = ctx.reduce_job->accumulate;
			ctx.reduce_job->taken = true;
		}
		p->pending = true;
		return true;
	}

	struct ScatterPlan {
		DeviceBuf dev_tasks, dev_ranges, scratch;
		std::vector<GridScatterTask> host_tasks;
		std::vector<GridScatterRange> host_ranges;
		uint32_t n_tasks = 0, n_ranges = 0;
		uint32_t launches = 0;
		bool tuned = false;
		ParamRanges adam_ranges; // what a launch of this plan in record form updates itself when it is handed an AdamInFlush
	};
	// TCNN_AMD_SCATTER_TUNE=0 keeps the untuned task list (A/B runs)
	static bool scatter_tuning_enabled() { return switches().scatter_tune; }
	// (re)builds the device-side plan; the caller guarantees that no launch using the old one is still running
	void build_scatter_plan(ScatterPlan& plan, uint32_t n, const std::vector<float>* measured_level_us) {
		std::vector<GridScatterRange> ranges;
		size_t scratch_elems = 0;
		grid_scatter_plan(m_meta, n, plan.host_tasks, ranges, scratch_elems, measured_level_us);
		plan.n_tasks = (uint32_t)plan.host_tasks.size();
		plan.n_ranges = (uint32_t)ranges.size();
		plan.host_ranges = ranges;
		plan.adam_ranges = grid_scatter_adam_ranges(m_meta, plan.host_tasks, true);
		plan.dev_tasks.resize(plan.host_tasks.size() * sizeof(GridScatterTask));
		if (!plan.host_tasks.empty()) HIP_CHECK_THROW(hipMemcpy(plan.dev_tasks.data(), plan.host_tasks.data(), plan.host_tasks.size() * sizeof(GridScatterTask), hipMemcpyHostToDevice));
		plan.dev_ranges.resize(ranges.size() * sizeof(GridScatterRange));
		if (!ranges.empty()) HIP_CHECK_THROW(hipMemcpy(plan.dev_ranges.data(), ranges.data(), ranges.size() * sizeof(GridScatterRange), hipMemcpyHostToDevice));
		plan.scratch.resize(0);
		plan.scratch.resize(scratch_elems * sizeof(uint64_t));
		plan.scratch.memset(0); // the finalize pass leaves it zeroed again after every step
	}
	// One plan per (batch size, stream): a plan owns mutable device state -- the scratch table of the shared chunks, which a step
	// leaves zeroed for the next one, and the task list the tuner replaces after a stream synchronisation -- so two streams that
	// run backward passes of this encoding at the same time must not share one.  Steps on ONE stream are ordered and may.
	ScatterPlan& scatter_plan(uint32_t n, hipStream_t stream) {
		const auto key = std::make_pair(n, (const void*)stream);
		auto it = m_scatter_plans.find(key);
		if (it != m_scatter_plans.end()) return *it->second;
		auto plan = std::make_unique<ScatterPlan>();
		build_scatter_plan(*plan, n, nullptr);
		return *(m_scatter_plans[key] = std::move(plan));
	}

	// the list-fed kernel's tasks (grid_scatter_lists_plan), per (batch size, stream, record form): like ScatterPlan it owns device state a
	// launch leaves behind for the next one on the same stream (the zeroed scratch table)
	struct ListsPlan {
		DeviceBuf dev_tasks, dev_ranges, scratch;
		std::vector<GridScatterRange> host_ranges;
		uint32_t n_tasks = 0, n_ranges = 0;
	};
	ListsPlan& lists_plan(uint32_t n, hipStream_t stream) {
		const auto key = std::make_pair(n, (const void*)stream);
		auto it = m_lists_plans.find(key);
		if (it != m_lists_plans.end()) return *it->second;
		auto plan = std::make_unique<ListsPlan>();
		std::vector<GridScatterTask> tasks;
		std::vector<GridScatterRange> ranges;
		size_t scratch_elems = 0;
		grid_scatter_lists_plan(m_meta, n, tasks, ranges, scratch_elems);
		plan->n_tasks = (uint32_t)tasks.size();
		plan->n_ranges = (uint32_t)ranges.size();
		plan->host_ranges = ranges;
		auto upload = [](DeviceBuf& b, const void* src, size_t bytes) {
			b.resize(bytes);
			if (bytes) HIP_CHECK_THROW(hipMemcpy(b.data(), src, bytes, hipMemcpyHostToDevice));
		};
		upload(plan->dev_tasks, tasks.data(), tasks.size() * sizeof(GridScatterTask));
		upload(plan->dev_ranges, ranges.data(), ranges.size() * sizeof(GridScatterRange));
		plan->scratch.resize(scratch_elems * sizeof(uint64_t));
		plan->scratch.memset(0); // the finalize pass leaves it zeroed again after every step
		return *(m_lists_plans[key] = std::move(plan));
	}

	// Hit lists: TCNN_AMD_SCATTER_LISTS=0 keeps the bit planes (A/B runs, tests; read per step so that one process can cover both)
	bool hit_lists_usable(uint32_t n) const {
		const int want = switches().scatter_lists; // 0: never; 1: wherever the kernel can take the grid (tests); -1: where it pays (grid_scatter_prefers_lists)
		if (want == 0) return false;
		if (m_any_binned || n > grid_hit_max_samples(m_meta) || m_meta.n_pos_dims > 3 || m_meta.hash_type == (uint32_t)HashType::Rng) return false; // (Rng: its hash is a loop)
		if (want == 1) return true;
		// Where it pays: grids with levels of many chunks, at every batch size -- nothing in the list-fed form depends on what an L2 holds
		// (round 4's gathered 16-byte records from one plane per XCD and was kept to 2^17 .. 2^19 samples).  C3a, step with lists / with
		// bit planes in ms (profiles/r05_sweep.txt): 2^14 0.086 / 0.088, 2^16 0.120 / 0.128, 2^18 0.194 / 0.222, 2^20 0.474 / 0.584, 2^21 0.862 / 1.123.
		// Below 2^16 samples a 2-D grid's bit planes are level with them or ahead -- two launches and 64 KiB of accumulators per task for a hundred
		// elements; other 2-D grids at 2^12 / 2^14 / 2^16 (profiles/r05_shape_sweep.txt): T = 2^17 0.057 / 0.047, 0.056 / 0.051, 0.079 / 0.079;
		// F = 4 0.063 / 0.034, 0.050 / 0.040, 0.060 / 0.058.  In 3-D the bit-plane form tests four rows per sample and chunk and loses everywhere
		// (2^12: 0.098 / 0.147, 2^18: 0.33 / 1.44).
		if (m_meta.n_pos_dims == 2 && n < (1u << 16)) return false;
		return grid_scatter_prefers_lists(m_meta);
	}
	static size_t next_multiple_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }
	struct HitCounters {
		DeviceBuf sets[2], fallbacks; // two sets of list tails [n_levels][GRID_HIT_COUNT_STRIDE]; how many tasks took the 64-bit passes
		int next = 0;
		uint64_t generation = 0;
	};
	HitCounters& hit_counters(hipStream_t stream) {
		auto it = m_hit_counters.find((const void*)stream);
		if (it != m_hit_counters.end()) return *it->second;
		auto hc = std::make_unique<HitCounters>();
		for (auto& set : hc->sets) {
			set.resize((size_t)m_meta.n_levels * GRID_HIT_COUNT_STRIDE * sizeof(uint32_t));
			set.memset(0);
		}
		hc->fallbacks.resize(sizeof(uint32_t));
		hc->fallbacks.memset(0);
		return *(m_hit_counters[(const void*)stream] = std::move(hc));
	}
public:
	// tasks of the list-fed scatter that found their packed 32-bit sums unprovable and ran the 64-bit passes, since this encoding was built (all streams)
	uint64_t scatter_wide_fallbacks() override {
		uint64_t total = 0;
		for (auto& kv : m_hit_counters) {
			uint32_t v = 0;
			HIP_CHECK_THROW(hipMemcpy(&v, kv.second->fallbacks.data(), sizeof(v), hipMemcpyDeviceToHost));
			total += v;
		}
		return total;
	}

	uint64_t list_scatters() const override { return m_list_scatters; }

	Json hyperparams() const override { // grid.h:1098-1115
		static const char* types[] = {"Hash", "Dense", "Tiled"};
		static const char* interps[] = {"Nearest", "Linear", "Smoothstep"};
		static const char* hashes[] = {"Prime", "CoherentPrime", "ReversedPrime", "Rng"};
		Json j = Json::object();
		j["otype"] = "Grid";
		j["type"] = types[m_meta.grid_type];
		j["n_levels"] = m_meta.n_levels;
		j["n_features_per_level"] = m_meta.n_features_per_level;
		j["base_resolution"] = m_base_resolution;
		j["per_level_scale"] = m_per_level_scale;
		j["interpolation"] = interps[m_meta.interpolation];
		j["hash"] = hashes[m_meta.hash_type];
		if (m_meta.grid_type == (uint32_t)GridType::Hash) j["log2_hashmap_size"] = m_log2_hashmap_size;
		return j;
	}

private:
	GridMeta m_meta;
	DeviceBuf m_dev_meta;
	std::map<std::pair<uint32_t, const void*>, std::unique_ptr<ScatterPlan>> m_scatter_plans;
	std::map<const void*, std::unique_ptr<HitCounters>> m_hit_counters;
	std::map<std::pair<uint32_t, const void*>, std::unique_ptr<ListsPlan>> m_lists_plans;
	std::map<uint32_t, std::unique_ptr<PlanesPlan>> m_planes_plans;
	bool m_scatter_levels_ok = true;
	bool m_any_binned = false;
	uint64_t m_list_scatters = 0;
	std::vector<uint32_t> m_resolutions;
	uint32_t m_n_features, m_log2_hashmap_size, m_base_resolution, m_n_entries;
	float m_per_level_scale;
	bool m_stochastic_interpolation;
	size_t m_n_params;
};

class OneBlobEncoding : public Encoding {
public:
	OneBlobEncoding(uint32_t n_bins, uint32_t n_dims_to_encode, bool fp32) : Encoding{fp32}, m_n_bins{n_bins}, m_n_dims{n_dims_to_encode} {
		if ((n_bins & (n_bins - 1)) != 0 || n_bins == 0) throw std::runtime_error{"Number of bins must be a power of 2"}; // oneblob.h:173-177
	}
	uint32_t input_width() const override { return m_n_dims; }
	uint32_t output_width() const override { return m_n_dims * m_n_bins; }
	uint32_t as_oneblob() const override { // TCNN_AMD_FUSE_ONEBLOB=0: always the encoding's own kernel (A/B runs, tests)
		const char* e = getenv("TCNN_AMD_FUSE_ONEBLOB");
		return (!m_fp32 && m_n_bins >= 32 && !(e && e[0] == '0')) ? m_n_bins : 0;
	}
	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		if (out && padded_output_width() > 0) oneblob_forward(stream, m_fp32, n, m_n_dims, m_n_bins, x, out, padded_output_width());
		return {};
	}
	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		if (!dL_dx) return;
		oneblob_backward_input(stream, m_fp32, n, m_n_dims, m_n_bins, x, dL_dy, padded_output_width(), *dL_dx);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "OneBlob";
		j["n_bins"] = m_n_bins;
		return j;
	}
private:
	uint32_t m_n_bins, m_n_dims;
};

// dL_dx = 0 for n samples of n_dims input dims, whatever the view's layout
inline void zero_input_gradient(hipStream_t stream, uint32_t n, uint32_t n_dims, const MatViewMut& dL_dx) {
	if (n == 0 || n_dims == 0) return;
	const bool aos = dL_dx.stride_dim == 1 && dL_dx.stride_sample == n_dims, soa = dL_dx.stride_sample == 1 && dL_dx.stride_dim == n;
	if (aos || soa) {
		HIP_CHECK_THROW(hipMemsetAsync(dL_dx.data, 0, (size_t)n * n_dims * sizeof(float), stream));
	} else { // any other view: one strided column per input dim
		for (uint32_t d = 0; d < n_dims; ++d) {
			HIP_CHECK_THROW(hipMemset2DAsync(dL_dx.data + (size_t)d * dL_dx.stride_dim, (size_t)dL_dx.stride_sample * sizeof(float), 0, sizeof(float), n, stream));
		}
	}
}

// encodings/empty.h:58-150: encodes nothing -- its output is its padding, ones (e.g. as a placeholder inside a Composite); zero
// gradient towards its inputs.  (There output_width() reports the padded width; here it is 0 and the padding is the padding.)
class EmptyEncoding : public Encoding {
public:
	EmptyEncoding(uint32_t n_dims_to_encode, bool fp32) : Encoding{fp32}, m_n_dims{n_dims_to_encode} {}
	uint32_t input_width() const override { return m_n_dims; }
	uint32_t output_width() const override { return 0; }
	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		if (out && padded_output_width() > 0) identity_forward(stream, m_fp32, n, 0, 1.0f, 0.0f, x, out, padded_output_width()); // no live columns: all padding
		return {};
	}
	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		if (dL_dx) zero_input_gradient(stream, n, m_n_dims, *dL_dx);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "Empty";
		return j;
	}
private:
	uint32_t m_n_dims;
};

// encodings/ppng.h:30-119 + ppng_1.h:215-379 (this fork's PPNG1: per frequency and phase a rank-R product of three 1-D tables
// looked up at sin(freq (x - 0.5) + phase); k_ppng.hip).  3 input dims, half precision; parameter gradients only, as there.
// PPNG2 (ppng_2.h:345-506) differs in the tables: three Q x Q planes per (frequency, phase, feature, rank) instead of three rows.
class PpngEncoding : public Encoding {
public:
	PpngEncoding(uint32_t variant, uint32_t n_dims_to_encode, const Json& enc, bool fp32) : Encoding{fp32}, m_variant{variant} {
		const std::string PPNG1 = variant == 3 ? std::string{"PPNG"} : "PPNG" + std::to_string(variant); // the messages name the variant (ppng_3.h:710, :721 say "PPNG")
		if (n_dims_to_encode != 3) throw std::runtime_error{PPNG1 + (variant == 3 ? ": number of input dims must be 2,3 or 4." : ": number of input dims must be 2 or 3")}; // ppng_1.h:370-376, ppng_3.h:717-722 (3 only)
		if (fp32) throw std::runtime_error{PPNG1 + ": this build provides the half-precision form"};
		m_log2_min_freq = (int32_t)enc.value("log2_min_freq", 0);
		m_log2_max_freq = (int32_t)enc.value("log2_max_freq", 6);
		m_n_quants = enc.value("n_quants", 64u);
		m_n_frequencies = enc.value("n_frequencies", 6u);
		m_rank = variant == 3 ? 1u : enc.value("rank", 4u);
		m_n_features = enc.value("n_features", 4u);
		if (variant != 3 && m_rank != 2 && m_rank != 4 && m_rank != 8 && m_rank != 16) throw std::runtime_error{PPNG1 + ": rank must be 1, 2, 4, 8 or 16"};
		if (variant == 3 && m_n_features == 1) throw std::runtime_error{"PPNG: this build provides 2, 4 or 8 features (the single-feature form sums fp32 products in arbitrary order there)"};
		if (m_n_features != 2 && m_n_features != 4 && m_n_features != 8) throw std::runtime_error{PPNG1 + ": number of features must be 1, 2, 4 or 8"};
		if (m_n_frequencies < 2 || m_n_quants < 2) throw std::runtime_error{PPNG1 + ": needs at least 2 frequencies and 2 quantisation bins"};
		if (variant == 3) { // ppng_3.h:489-494
			m_n_params = (size_t)m_n_frequencies * 2 * m_n_quants * m_n_quants * m_n_quants * m_n_features;
			if (m_n_params >= (1ull << 32)) throw std::runtime_error{"PPNG: the volume does not fit 32-bit offsets"};
		} else {
			m_n_params = (size_t)m_n_frequencies * 2 * 3 * m_n_features * m_n_quants * (variant == 2 ? m_n_quants : 1u) * m_rank;
		}
	}
	uint32_t input_width() const override { return 3; }
	uint32_t output_width() const override { return m_n_frequencies * 2 * m_n_features; }
	size_t n_params() const override { return m_n_params; }
	void initialize_params(Pcg32& rng, float* params_full_precision, float scale) override { // ppng_1.h:325-328; PPNG3 keeps the base class' range (ppng.h:66-69)
		const float range = m_variant == 3 ? 1e-4f : 0.7f;
		generate_random_uniform(nullptr, rng.st, n_params(), params_full_precision, -range * scale, range * scale);
	}
	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		if (out && padded_output_width() > 0 && n > 0) {
			if (m_variant == 3) ppng3_forward(stream, false, n, m_n_frequencies, m_n_quants, m_n_features, m_log2_min_freq, m_log2_max_freq, x, params, out, padded_output_width());
			else (m_variant == 2 ? ppng2_forward : ppng1_forward)(stream, false, n, m_n_frequencies, m_n_quants, m_n_features, m_rank, m_log2_min_freq, m_log2_max_freq, x, params, out, padded_output_width());
		}
		return {};
	}
	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		CHECK_THROW(!dy_planes);
		if (dL_dx) {
			// PPNG1 / PPNG2: the reference leaves dL_dinput untouched (ppng_1.h:268-321 never writes it); PPNG3: ppng_3.h:586-607
			if (m_variant == 3 && padded_output_width() > 0) ppng3_backward_input(stream, false, n, m_n_frequencies, m_n_quants, m_n_features, m_log2_min_freq, m_log2_max_freq, x, params, dL_dy, padded_output_width(), *dL_dx);
			else zero_input_gradient(stream, n, 3, *dL_dx);
		}
		if (mode == GradientMode::Ignore || padded_output_width() == 0) return;
		CHECK_THROW(grads != nullptr);
		uint64_t* const scratch = scratch_for(stream);
		if (m_variant == 3) {
			const size_t ws_bytes = ppng3_backward_workspace_bytes(n, m_n_frequencies, m_n_quants, m_n_features);
			ArenaBuf ws = ws_bytes ? ArenaBuf{stream, ws_bytes} : ArenaBuf{};
			ppng3_backward(stream, false, n, m_n_frequencies, m_n_quants, m_n_features, m_log2_min_freq, m_log2_max_freq, x, dL_dy, padded_output_width(), ws.data(),
			               scratch, grads, mode == GradientMode::Accumulate);
			return;
		}
		(m_variant == 2 ? ppng2_backward : ppng1_backward)(stream, false, n, m_n_frequencies, m_n_quants, m_n_features, m_rank, m_log2_min_freq, m_log2_max_freq, x, params, dL_dy,
		                                                   padded_output_width(), scratch, grads, mode == GradientMode::Accumulate);
	}
	// ppng_3.h:609-676 (PPNG1 / PPNG2 have none)
	void backward_backward_input(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, MatView dL_ddLdx, const void* dL_dy, void* dL_ddLdy,
	                             MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode) override {
		if (m_variant != 3) return Encoding::backward_backward_input(stream, ctx, n, x, dL_ddLdx, dL_dy, dL_ddLdy, dL_dx, params, grads, mode);
		if ((!dL_ddLdy && mode == GradientMode::Ignore && !dL_dx) || padded_output_width() == 0 || n == 0) return;
		uint64_t* scratch = nullptr;
		if (mode != GradientMode::Ignore) {
			CHECK_THROW(grads != nullptr);
			scratch = scratch_for(stream);
		}
		ppng3_backward_backward_input(stream, false, n, m_n_frequencies, m_n_quants, m_n_features, m_log2_min_freq, m_log2_max_freq, x, dL_ddLdx, params, dL_dy, padded_output_width(), scratch,
		                              mode != GradientMode::Ignore ? grads : nullptr, mode == GradientMode::Accumulate, dL_ddLdy, dL_dx);
	}
	Json hyperparams() const override { // ppng.h:92-103
		Json j = Json::object();
		j["otype"] = "PPNG" + std::to_string(m_variant);
		j["n_frequencies"] = m_n_frequencies;
		j["log2_min_freq"] = m_log2_min_freq;
		j["log2_max_freq"] = m_log2_max_freq;
		j["n_quants"] = m_n_quants;
		j["n_features_per_level"] = m_n_features;
		j["rank"] = m_rank;
		return j;
	}
private:
	uint64_t* scratch_for(hipStream_t stream) { // per stream: zero between passes (k_ppng_finalize)
		std::unique_ptr<DeviceBuf>& scratch = m_scratch[(const void*)stream];
		if (!scratch) {
			scratch = std::make_unique<DeviceBuf>(m_n_params * sizeof(uint64_t));
			scratch->memset(0);
		}
		return scratch->as<uint64_t>();
	}
	uint32_t m_variant;
	int32_t m_log2_min_freq, m_log2_max_freq;
	uint32_t m_n_quants, m_n_frequencies, m_rank, m_n_features;
	size_t m_n_params;
	std::map<const void*, std::unique_ptr<DeviceBuf>> m_scratch;
};

class IdentityEncoding : public Encoding {
public:
	IdentityEncoding(uint32_t n_dims_to_encode, float scale, float offset, bool fp32) : Encoding{fp32}, m_n_dims{n_dims_to_encode}, m_scale{scale}, m_offset{offset} {}
	uint32_t input_width() const override { return m_n_dims; }
	uint32_t output_width() const override { return m_n_dims; }
	bool as_identity(float& scale, float& offset) const override {
		scale = m_scale;
		offset = m_offset;
		return !m_fp32;
	}
	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		if (out && padded_output_width() > 0) identity_forward(stream, m_fp32, n, m_n_dims, m_scale, m_offset, x, out, padded_output_width());
		return {};
	}
	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		if (!dL_dx) return;
		identity_backward_input(stream, m_fp32, n, m_n_dims, m_scale, dL_dy, padded_output_width(), *dL_dx);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "Identity";
		j["scale"] = m_scale;
		j["offset"] = m_offset;
		return j;
	}
private:
	uint32_t m_n_dims;
	float m_scale, m_offset;
};

// src/encoding.cu:144-158 (case-insensitive registry :48-54; default otype OneBlob)
// encodings/frequency.h:104-220 (two outputs per frequency: sin, cos) and encodings/triangle_wave.h:110-220 (one)
class PeriodicEncoding : public Encoding {
public:
	PeriodicEncoding(bool triangle, uint32_t n_frequencies, uint32_t n_dims_to_encode, bool fp32) : Encoding{fp32}, m_triangle{triangle}, m_n_frequencies{n_frequencies}, m_n_dims{n_dims_to_encode} {}
	uint32_t input_width() const override { return m_n_dims; }
	uint32_t outputs_per_input() const { return m_n_frequencies * (m_triangle ? 1u : 2u); }
	uint32_t output_width() const override { return m_n_dims * outputs_per_input(); }
	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		EncodingContext ctx;
		if (!out || padded_output_width() == 0 || n == 0) return ctx;
		if (prepare_input_gradients) ctx.dy_dx = ArenaBuf{stream, (size_t)n * output_width() * sizeof(float)};
		periodic_forward(stream, m_triangle, m_fp32, n, m_n_dims, m_n_frequencies, x, out, padded_output_width(), ctx.dy_dx.as<float>());
		return ctx;
	}
	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		if (!dL_dx || n == 0) return;
		CHECK_THROW(ctx.dy_dx); // frequency.h:150-152: needs a forward pass with prepare_input_gradients
		periodic_backward_input(stream, m_fp32, n, m_n_dims, outputs_per_input(), dL_dy, padded_output_width(), ctx.dy_dx.as<float>(), *dL_dx);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = m_triangle ? "TriangleWave" : "Frequency";
		j["n_frequencies"] = m_n_frequencies;
		return j;
	}
private:
	bool m_triangle;
	uint32_t m_n_frequencies, m_n_dims;
};

// encodings/spherical_harmonics.h:110-230
class SphericalHarmonicsEncoding : public Encoding {
public:
	SphericalHarmonicsEncoding(uint32_t degree, uint32_t n_dims_to_encode, bool fp32) : Encoding{fp32}, m_degree{degree} {
		if (n_dims_to_encode != 3) throw std::runtime_error{"Can only encode 3D directions in spherical harmonics."};
		if (degree == 0) throw std::runtime_error{"Spherical harmonics must have positive degree."};
		if (degree > 8) throw std::runtime_error{"Spherical harmonics are only implemented up to degree 8."};
	}
	uint32_t input_width() const override { return 3; }
	uint32_t output_width() const override { return m_degree * m_degree; }
	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		if (out && padded_output_width() > 0) sh_forward(stream, m_fp32, n, m_degree, x, out, padded_output_width());
		return {};
	}
	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		if (!dL_dx) return;
		sh_backward_input(stream, m_fp32, n, m_degree, x, dL_dy, padded_output_width(), *dL_dx);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "SphericalHarmonics";
		j["degree"] = m_degree;
		return j;
	}
private:
	uint32_t m_degree;
};

inline std::unique_ptr<Encoding> create_encoding(uint32_t n_dims_to_encode, const Json& enc, uint32_t alignment, bool fp32);

// encodings/composite.h:126-420, reduction "Concatenation": nested encodings over slices of the input dims, their (padded)
// outputs side by side, their parameters one after the other.  Every nested encoding writes its own AoS block, which is then
// copied into its column range of the composite row (and dL_dy copied out the same way): a few extra bytes per sample instead
// of a width / stride distinction in every encoding kernel.
class CompositeEncoding : public Encoding {
public:
	CompositeEncoding(uint32_t n_dims_to_encode, const Json& params, bool fp32) : Encoding{fp32}, m_n_dims{n_dims_to_encode} {
		if (!params.contains("nested") || !params["nested"].is_array()) throw std::runtime_error{"Must provide an array of nested encodings to CompositeEncoding."};
		const std::string reduction = params.value("reduction", "Concatenation"); // common.h string_to_reduction_type
		if (equals_case_insensitive(reduction, "Concatenation")) m_reduction = Reduction::Concatenation;
		else if (equals_case_insensitive(reduction, "Sum")) m_reduction = Reduction::Sum;
		else if (equals_case_insensitive(reduction, "Product")) m_reduction = Reduction::Product;
		else throw std::runtime_error{"Invalid reduction type: " + reduction};
		const Json& nested = params["nested"];
		m_config = params;
		uint32_t total = 0;
		for (size_t i = 0; i < nested.size(); ++i) { // composite.h:147-158
			total += nested.at(i).value("n_dims_to_encode", 0u);
			if (nested.at(i).contains("dims_to_encode_begin")) { total = 0xFFFFFFFFu; break; }
		}
		if (total != 0xFFFFFFFFu && total > n_dims_to_encode) throw std::runtime_error{"CompositeEncoding: nested encodings must not encode more dims " + std::to_string(total) + " than composite " + std::to_string(n_dims_to_encode)};
		uint32_t unspecified = total == 0xFFFFFFFFu ? 0xFFFFFFFFu : (n_dims_to_encode - total);
		uint32_t offset = 0;
		for (size_t i = 0; i < nested.size(); ++i) { // composite.h:163-187
			const Json& cfg = nested.at(i);
			uint32_t dims;
			if (cfg.contains("n_dims_to_encode")) {
				if (cfg.contains("dims_to_encode_begin")) offset = cfg.value("dims_to_encode_begin", 0u);
				dims = cfg.value("n_dims_to_encode", 0u);
			} else {
				if (unspecified == 0xFFFFFFFFu) throw std::runtime_error{"CompositeEncoding: may only leave 'n_dims_to_encode' unspecified for a single nested encoding"};
				dims = unspecified;
				unspecified = 0xFFFFFFFFu;
			}
			if (dims > 0) {
				if (offset + dims > n_dims_to_encode) throw std::runtime_error{"CompositeEncoding: nested encoding reaches past the input dims"};
				m_nested.emplace_back(create_encoding(dims, cfg, 1, fp32));
				m_begin.push_back(offset);
			}
			offset += dims;
		}
		if (m_nested.empty()) throw std::runtime_error{"CompositeEncoding: no nested encoding encodes anything"};
		if (m_reduction == Reduction::Concatenation) {
			// composite.h:189-200: pad each nested output so that the next one starts at a multiple of ITS required alignment
			uint32_t so_far = 0;
			for (size_t i = 0; i + 1 < m_nested.size(); ++i) {
				const uint32_t desired = m_nested[i + 1]->required_output_alignment();
				m_nested[i]->set_padded_output_width(next_multiple(so_far + m_nested[i]->output_width(), desired) - so_far);
				so_far += m_nested[i]->padded_output_width();
			}
		} else {
			// composite.h:199-211: every nested encoding at the common alignment, and all of the same width
			const uint32_t alignment = required_output_alignment();
			for (auto& e : m_nested) e->set_alignment(alignment);
			for (auto& e : m_nested) {
				if (e->output_width() != m_nested.front()->output_width()) throw std::runtime_error{"CompositeEncoding: nested encodings of a Sum / Product reduction must have the same output width"};
			}
		}
	}
	uint32_t input_width() const override { return m_n_dims; }
	uint32_t output_width() const override { // composite.h:352-366: the sum of the nested PADDED widths; reductions: the (common) padded width
		if (m_reduction != Reduction::Concatenation) return m_nested.front()->padded_output_width();
		uint32_t total = 0;
		for (const auto& e : m_nested) total += e->padded_output_width();
		return total;
	}
	uint32_t required_output_alignment() const override {
		uint32_t a = 1;
		for (const auto& e : m_nested) a = std::lcm(a, e->required_output_alignment());
		return a;
	}
	void set_padded_output_width(uint32_t padded) override { // composite.h:375-385: the last nested encoding absorbs the padding
		if (m_reduction != Reduction::Concatenation) { // :381-384: every nested encoding is padded to the same width
			for (auto& e : m_nested) e->set_padded_output_width(padded);
			return;
		}
		const uint32_t prev = output_width() - m_nested.back()->padded_output_width();
		CHECK_THROW(padded >= prev + m_nested.back()->output_width());
		m_nested.back()->set_padded_output_width(padded - prev);
	}
	size_t n_params() const override {
		size_t n = 0;
		for (const auto& e : m_nested) n += e->n_params();
		return n;
	}
	void initialize_params(Pcg32& rng, float* params_full_precision, float scale) override { // composite.h:422-428
		size_t offset = 0;
		for (auto& e : m_nested) {
			e->initialize_params(rng, params_full_precision + offset, scale);
			offset += e->n_params();
		}
	}
	EncodingContext forward(hipStream_t stream, uint32_t n, MatView x, const void* params, void* out, bool prepare_input_gradients, bool prepare_param_gradients) override {
		EncodingContext ctx;
		if (!out || n == 0) return ctx;
		const size_t elem = m_fp32 ? 4 : 2;
		ctx.nested.resize(m_nested.size());
		uint32_t col = 0;
		size_t p_off = 0;
		if (m_reduction != Reduction::Concatenation) { // composite.h:259-300
			const uint32_t w = reduction_width();
			const size_t block_bytes = (size_t)n * w * elem;
			ctx.to_reduce = ArenaBuf{stream, block_bytes * m_nested.size()}; // [nested][n][w]: kept for the product's backward pass
			for (size_t i = 0; i < m_nested.size(); ++i) {
				Encoding& e = *m_nested[i];
				const MatView xs{x.data + (size_t)m_begin[i] * x.stride_dim, x.stride_sample, x.stride_dim};
				ctx.nested[i] = e.forward(stream, n, xs, (const char*)params + p_off * elem, (char*)ctx.to_reduce.data() + i * block_bytes, prepare_input_gradients, prepare_param_gradients);
				p_off += e.n_params();
			}
			composite_reduce_forward(stream, m_fp32, m_reduction == Reduction::Product, (size_t)n * w, (uint32_t)m_nested.size(), ctx.to_reduce.data(), out);
			return ctx;
		}
		for (size_t i = 0; i < m_nested.size(); ++i) {
			Encoding& e = *m_nested[i];
			const uint32_t w = e.padded_output_width();
			ArenaBuf block{stream, (size_t)n * w * elem};
			const MatView xs{x.data + (size_t)m_begin[i] * x.stride_dim, x.stride_sample, x.stride_dim};
			ctx.nested[i] = e.forward(stream, n, xs, (const char*)params + p_off * elem, block.data(), prepare_input_gradients, prepare_param_gradients);
			copy_columns(stream, elem, n, block.data(), w, 0, out, padded_output_width(), col, w);
			col += w;
			p_off += e.n_params();
		}
		return ctx;
	}
	void backward(hipStream_t stream, const EncodingContext& ctx, uint32_t n, MatView x, const void* dL_dy, MatViewMut* dL_dx, const void* params, void* grads, GradientMode mode, bool dy_planes) override {
		if (n == 0) return;
		CHECK_THROW(!dy_planes && ctx.nested.size() == m_nested.size());
		const size_t elem = m_fp32 ? 4 : 2;
		// input dims no nested encoding looks at have zero gradient
		if (dL_dx) zero_input_gradient(stream, n, m_n_dims, *dL_dx);
		uint32_t col = 0;
		size_t p_off = 0;
		if (m_reduction != Reduction::Concatenation) { // composite.h:302-330: dL/d(nested outputs) from dL/d(reduced output), then the nested passes
			const uint32_t w = reduction_width();
			const size_t block_bytes = (size_t)n * w * elem;
			CHECK_THROW(ctx.to_reduce);
			ArenaBuf dnested{stream, block_bytes * m_nested.size()};
			composite_reduce_backward(stream, m_fp32, m_reduction == Reduction::Product, (size_t)n * w, (uint32_t)m_nested.size(), ctx.to_reduce.data(), dL_dy, dnested.data());
			for (size_t i = 0; i < m_nested.size(); ++i) {
				Encoding& e = *m_nested[i];
				const MatView xs{x.data + (size_t)m_begin[i] * x.stride_dim, x.stride_sample, x.stride_dim};
				MatViewMut dxs{};
				if (dL_dx) dxs = MatViewMut{dL_dx->data + (size_t)m_begin[i] * dL_dx->stride_dim, dL_dx->stride_sample, dL_dx->stride_dim};
				e.backward(stream, ctx.nested[i], n, xs, (const char*)dnested.data() + i * block_bytes, dL_dx ? &dxs : nullptr, (const char*)params + p_off * elem,
				           grads ? (char*)grads + p_off * elem : nullptr, mode, false);
				p_off += e.n_params();
			}
			return;
		}
		for (size_t i = 0; i < m_nested.size(); ++i) {
			Encoding& e = *m_nested[i];
			const uint32_t w = e.padded_output_width();
			ArenaBuf block{stream, (size_t)n * w * elem};
			copy_columns(stream, elem, n, dL_dy, padded_output_width(), col, block.data(), w, 0, w);
			const MatView xs{x.data + (size_t)m_begin[i] * x.stride_dim, x.stride_sample, x.stride_dim};
			MatViewMut dxs{};
			if (dL_dx) dxs = MatViewMut{dL_dx->data + (size_t)m_begin[i] * dL_dx->stride_dim, dL_dx->stride_sample, dL_dx->stride_dim};
			e.backward(stream, ctx.nested[i], n, xs, block.data(), dL_dx ? &dxs : nullptr, (const char*)params + p_off * elem, grads ? (char*)grads + p_off * elem : nullptr, mode, false);
			col += w;
			p_off += e.n_params();
		}
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "Composite";
		j["reduction"] = m_reduction == Reduction::Sum ? "Sum" : (m_reduction == Reduction::Product ? "Product" : "Concatenation");
		Json nested = Json::array();
		for (const auto& e : m_nested) nested.push_back(e->hyperparams());
		j["nested"] = nested;
		return j;
	}
private:
	enum class Reduction { Concatenation, Sum, Product };
	// Width of a reduction's operands.  The reference lays the nested outputs out at multiples of their UNPADDED width but
	// reduces at multiples of the PADDED one (composite.h:274 against :261): the two agree -- and the result is defined --
	// only when no nested output needs padding, which is what this build accepts.
	uint32_t reduction_width() const {
		for (const auto& e : m_nested) {
			if (e->padded_output_width() != e->output_width()) {
				throw std::runtime_error{"CompositeEncoding: a Sum / Product reduction needs nested encodings whose output width (" + std::to_string(e->output_width()) +
				                         ") is a multiple of the required alignment (padded to " + std::to_string(e->padded_output_width()) + ")"};
			}
		}
		return m_nested.front()->padded_output_width();
	}
public:
	uint64_t list_scatters() const override {
		uint64_t total = 0;
		for (const auto& e : m_nested) total += e->list_scatters();
		return total;
	}

private:
	uint32_t m_n_dims;
	Json m_config;
	Reduction m_reduction = Reduction::Concatenation;
	std::vector<std::unique_ptr<Encoding>> m_nested;
	std::vector<uint32_t> m_begin;
};

inline std::unique_ptr<Encoding> create_encoding(uint32_t n_dims_to_encode, const Json& enc, uint32_t alignment, bool fp32) {
	const std::string name = to_lower(enc.value("otype", "OneBlob"));
	std::unique_ptr<Encoding> result;
	if (name == "grid" || name == "hashgrid" || name == "tiledgrid" || name == "densegrid") {
		result.reset(new GridEncoding{n_dims_to_encode, enc, fp32});
	} else if (name == "oneblob") {
		result.reset(new OneBlobEncoding{enc.value("n_bins", 16u), n_dims_to_encode, fp32});
	} else if (name == "identity") {
		result.reset(new IdentityEncoding{n_dims_to_encode, enc.value("scale", 1.0f), enc.value("offset", 0.0f), fp32});
	} else if (name == "empty") {
		result.reset(new EmptyEncoding{n_dims_to_encode, fp32});
	} else if (name == "ppng1" || name == "ppng2" || name == "ppng3") {
		result.reset(new PpngEncoding{(uint32_t)(name.back() - '0'), n_dims_to_encode, enc, fp32});
	} else if (name == "frequency") {
		result.reset(new PeriodicEncoding{false, enc.value("n_frequencies", 12u), n_dims_to_encode, fp32});
	} else if (name == "trianglewave") {
		result.reset(new PeriodicEncoding{true, enc.value("n_frequencies", 12u), n_dims_to_encode, fp32});
	} else if (name == "sphericalharmonics") {
		result.reset(new SphericalHarmonicsEncoding{enc.value("degree", 4u), n_dims_to_encode, fp32});
	} else if (name == "composite") {
		result.reset(new CompositeEncoding{n_dims_to_encode, enc, fp32});
	} else if (name == "oneblobfrequency" || name == "nrc") {
		// src/encoding.cu:96-119: the neural radiance caching input layer = TriangleWave on 3 position dims, OneBlob on the next 5,
		// Identity on whatever is left
		Json tri = Json::object(), blob = Json::object(), rest = Json::object();
		tri["n_dims_to_encode"] = 3u;
		tri["otype"] = "TriangleWave";
		tri["n_frequencies"] = enc.value("n_frequencies", 12u);
		blob["n_dims_to_encode"] = 5u;
		blob["otype"] = "OneBlob";
		blob["n_bins"] = enc.value("n_bins", 4u);
		rest["otype"] = "Identity";
		Json nested = Json::array();
		nested.push_back(tri);
		nested.push_back(blob);
		nested.push_back(rest);
		Json composite = Json::object();
		composite["otype"] = "Composite";
		composite["nested"] = nested;
		result.reset(new CompositeEncoding{n_dims_to_encode, composite, fp32});
	} else {
		throw std::runtime_error{"Encoding '" + enc.value("otype", "OneBlob") + "' not found (this build provides Grid/HashGrid/TiledGrid/DenseGrid, OneBlob, Identity, Empty, Frequency, TriangleWave, SphericalHarmonics, Composite, PPNG1, PPNG2)"};
	}
	if (alignment > 0) result->set_alignment(alignment);
	return result;
}

// ------------------------------------------------------------------------------------------------------------------
// Network: FullyFusedMLP (and CutlassMLP configs that the fused kernels cover; identical math)
// ------------------------------------------------------------------------------------------------------------------
inline Activation string_to_activation(const std::string& s) {
	static const std::pair<const char*, Activation> table[] = {
		{"None", Activation::None}, {"ReLU", Activation::ReLU}, {"LeakyReLU", Activation::LeakyReLU}, {"Exponential", Activation::Exponential},
		{"Sine", Activation::Sine}, {"Sigmoid", Activation::Sigmoid}, {"Squareplus", Activation::Squareplus}, {"Softplus", Activation::Softplus},
		{"Tanh", Activation::Tanh}};
	for (const auto& kv : table) if (equals_case_insensitive(s, kv.first)) return kv.second;
	throw std::runtime_error{"Invalid activation name: " + s};
}
inline const char* to_string(Activation a) {
	static const char* names[] = {"None", "ReLU", "LeakyReLU", "Exponential", "Sine", "Sigmoid", "Squareplus", "Softplus", "Tanh"};
	return names[(uint32_t)a];
}

struct NetworkContext {
	ArenaBuf hidden; // half [n_hidden][n][width]
};

class Network {
public:
	static constexpr uint32_t REQUIRED_ALIGNMENT = 16; // fully_fused_mlp.h:108-110, cutlass_mlp.h:115-121

	explicit Network(const Json& net) { // network.cu:48-137
		const std::string otype = net.value("otype", "MLP");
		m_fully_fused = equals_case_insensitive(otype, "MegakernelMLP") || equals_case_insensitive(otype, "FullyFusedMLP");
		const bool cutlass = equals_case_insensitive(otype, "MLP") || equals_case_insensitive(otype, "CutlassMLP");
		if (!m_fully_fused && !cutlass) throw std::runtime_error{"Invalid network type: " + otype};
		if (!net.contains("n_input_dims") || !net.contains("n_output_dims")) throw std::runtime_error{"network config needs n_input_dims and n_output_dims"};
		m_input_width = net.value("n_input_dims", 0u);
		m_output_width = net.value("n_output_dims", 0u);
		m_width = net.value("n_neurons", 128u);
		m_n_hidden = net.value("n_hidden_layers", 5u);
		m_activation = string_to_activation(net.value("activation", "ReLU"));
		m_output_activation = string_to_activation(net.value("output_activation", "None"));
		// fully_fused_mlp.cu:869-881 restricts FullyFusedMLP to these widths; CutlassMLP (cutlass_mlp.cu:39-92) takes any -- here: 256 as well
		const bool width_ok = m_width == 16 || m_width == 32 || m_width == 64 || m_width == 128 || (cutlass && m_width == 256);
		if (!width_ok) {
			throw std::runtime_error{"FullyFusedMLP only supports 16, 32, 64, and 128 neurons, but got " + std::to_string(m_width) + ". (CutlassMLP: also 256; other widths are not provided by this build.)"};
		}
		if (m_n_hidden <= 0) throw std::runtime_error{"FullyFusedMLP requires at least 1 hidden layer (3 layers in total)."};
		if (m_n_hidden + 1 > MAX_MLP_LAYERS) throw std::runtime_error{"MLP: too many layers for this build"};
		if (m_activation == Activation::Sine) throw std::runtime_error{"FullyFusedMLP: Sine activation is not supported in the fused kernels"};
		if (m_input_width % 16 != 0) throw std::runtime_error{"MLP: input width must be a multiple of 16"};
		m_padded_output_width = next_multiple(m_output_width, REQUIRED_ALIGNMENT);

		// matrices: fully_fused_mlp.cu:659-671
		memset(&m_desc, 0, sizeof(m_desc));
		m_desc.in_width = m_input_width;
		m_desc.width = m_width;
		m_desc.out_width = m_padded_output_width;
		m_desc.n_hidden = m_n_hidden;
		m_desc.n_layers = m_n_hidden + 1;
		m_desc.activation = (uint32_t)m_activation;
		m_desc.output_activation = (uint32_t)m_output_activation;
		uint32_t w_off = 0, f_off = 0, b_off = 0;
		for (uint32_t l = 0; l < m_desc.n_layers; ++l) {
			MlpLayer& L = m_desc.layers[l];
			L.rows = l == m_desc.n_layers - 1 ? m_padded_output_width : m_width;
			L.cols = l == 0 ? m_input_width : m_width;
			L.w_off = w_off;
			L.ks_fwd = div_round_up(L.cols, 32);
			L.ks_bwd = div_round_up(L.rows, 32);
			L.natural_k = l == 0 ? 1u : 0u;
			L.fwd_off = f_off;
			L.bwd_off = b_off;
			f_off += (L.rows / 16) * L.ks_fwd;
			b_off += (L.cols / 16) * L.ks_bwd;
			w_off += L.rows * L.cols;
		}
		m_desc.n_frags_fwd = f_off;
		m_desc.n_frags_bwd = b_off;
		m_desc.n_frags_r32 = r32_shape_ok(m_desc) ? R32Frags(m_desc).n_frags() : 0u;
		m_n_params = w_off;
	}

	uint32_t input_width() const { return m_input_width; }
	uint32_t output_width() const { return m_output_width; }
	uint32_t padded_output_width() const { return m_padded_output_width; }
	uint32_t width() const { return m_width; }
	uint32_t n_hidden() const { return m_n_hidden; }
	size_t n_params() const { return m_n_params; }
	const MlpDesc& desc() const { return m_desc; }

	std::vector<std::pair<uint32_t, uint32_t>> layer_sizes() const {
		std::vector<std::pair<uint32_t, uint32_t>> r;
		for (uint32_t l = 0; l < m_desc.n_layers; ++l) r.emplace_back(m_desc.layers[l].rows, m_desc.layers[l].cols);
		return r;
	}

	void initialize_params(Pcg32& rng, float* params_full_precision, float scale) { // fully_fused_mlp.cu:866-891, gpu_matrix.h:284-299
		std::vector<float> host(m_n_params);
		for (uint32_t l = 0; l < m_desc.n_layers; ++l) {
			const MlpLayer& L = m_desc.layers[l];
			const float s = scale * std::sqrt(6.0f / (float)(L.cols + L.rows));
			float* w = host.data() + L.w_off;
			for (size_t i = 0; i < (size_t)L.rows * L.cols; ++i) w[i] = rng.next_float() * 2.0f * s - s;
		}
		HIP_CHECK_THROW(hipMemcpy(params_full_precision, host.data(), m_n_params * sizeof(float), hipMemcpyHostToDevice));
	}

	// ---- A fragment image that lives across training steps, for models whose optimizer step rides on the weight gradients' slab
	// reduction (BASELINE config 2).  k_wgrad_reduce_adam writes every weight it updates into the image elements that hold it
	// (AdamInFlush::image / image_inv), so the next step starts with a current image instead of a k_mlp_prep launch -- 4.5 of config 2's
	// 27 us.  The image is current for the parameter vector `live_params()`; whoever changes those parameters any other way calls
	// invalidate_live_image() (the Trainer does, and stops using the image for good once it has handed out a pointer to its parameters).
	struct LiveImage {
		DeviceBuf image, inverse;       // inverse: uint32 [n_params][IMAGE_INV_WIDTH], the image elements that hold a weight (0xffffffff: none)
		const void* params = nullptr;   // what the image is current for
	};
	// nullptr: a weight of this network sits in more than IMAGE_INV_WIDTH image elements (no supported shape does)
	LiveImage* live_image() const {
		if (m_live_state == 0) {
			const uint32_t total = (m_desc.n_frags_fwd + m_desc.n_frags_bwd + m_desc.n_frags_r32) * 512;
			std::vector<uint32_t> inv(m_n_params * IMAGE_INV_WIDTH, 0xffffffffu), count(m_n_params, 0);
			m_live_state = 1;
			for (uint32_t gid = 0; gid < total && m_live_state == 1; ++gid) {
				const int32_t src = mlp_prep_source(m_desc, gid);
				if (src < 0) continue;
				if ((size_t)src >= m_n_params || count[src] == IMAGE_INV_WIDTH) m_live_state = -1;
				else inv[(size_t)src * IMAGE_INV_WIDTH + count[src]++] = gid;
			}
			if (m_live_state == 1) {
				m_live.image.resize(mlp_image_bytes(m_desc));
				m_live.inverse.resize(inv.size() * sizeof(uint32_t));
				HIP_CHECK_THROW(hipMemcpy(m_live.inverse.data(), inv.data(), inv.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
			}
		}
		return m_live_state == 1 ? &m_live : nullptr;
	}
	void invalidate_live_image() const { m_live.params = nullptr; }

	// images: [fwd][bwd] fragment images of `params`
	ArenaBuf prepare(hipStream_t stream, const void* params, bool want_bwd) const {
		ArenaBuf image{stream, mlp_image_bytes(m_desc)};
		mlp_prepare_weights(stream, m_desc, params, image.data(), want_bwd);
		return image;
	}

	void inference(hipStream_t stream, uint32_t n, const void* input, void* output, const void* params) const {
		ArenaBuf image = prepare(stream, params, false);
		mlp_forward(stream, m_desc, image.data(), n, input, output, nullptr);
	}
	// inference with the input / output conversions fused into the kernel (see MlpIo)
	void inference_io(hipStream_t stream, uint32_t n, const MlpIo& io, const void* params) const {
		ArenaBuf image = prepare(stream, params, false);
		mlp_forward_io(stream, m_desc, image.data(), n, io, nullptr);
	}

	NetworkContext forward(hipStream_t stream, uint32_t n, const void* input, void* output, const void* params) const {
		NetworkContext ctx;
		ctx.hidden = ArenaBuf{stream, (size_t)m_n_hidden * n * m_width * 2};
		ArenaBuf image = prepare(stream, params, false);
		mlp_forward(stream, m_desc, image.data(), n, input, output, ctx.hidden.data());
		return ctx;
	}

	// dL_dinput: optional half [n][in_width]; gradients: half[n_params] or nullptr
	void backward(hipStream_t stream, const NetworkContext& ctx, uint32_t n, const void* input, const void* output, const void* dL_doutput,
	              void* dL_dinput, const void* params, void* gradients, GradientMode mode, uint32_t dx_plane_features = 0) const {
		ArenaBuf image = prepare(stream, params, true);
		ArenaBuf dhidden{stream, (size_t)m_n_hidden * n * m_width * 2};
		// output-activation transfer, computed once up front like the reference (fully_fused_mlp.cu:757-762)
		const void* dY = dL_doutput;
		ArenaBuf dY_tmp;
		MlpDesc desc = m_desc;
		if (m_output_activation != Activation::None) {
			dY_tmp = ArenaBuf{stream, (size_t)n * m_padded_output_width * 2};
			mlp_activation_backward_output(stream, n * m_padded_output_width, (uint32_t)m_output_activation, dL_doutput, output, dY_tmp.data());
			dY = dY_tmp.data();
			desc.output_activation = (uint32_t)Activation::None;
		}
		mlp_backward(stream, desc, image.data(), n, dY, output, ctx.hidden.data(), dhidden.data(), dL_dinput, dx_plane_features);
		if (mode == GradientMode::Ignore) return;
		CHECK_THROW(gradients != nullptr);

		const bool accumulate = mode == GradientMode::Accumulate;
		const size_t hstride = (size_t)n * m_width; // elements per hidden layer
		std::vector<WgradPanel> panels;
		for (uint32_t l = 0; l < m_desc.n_layers; ++l) {
			const MlpLayer& L = m_desc.layers[l];
			const _Float16* dO = l == m_desc.n_layers - 1 ? (const _Float16*)dY : dhidden.as<_Float16>() + hstride * l;
			const uint32_t ldo = l == m_desc.n_layers - 1 ? m_padded_output_width : m_width;
			const _Float16* In = l == 0 ? (const _Float16*)input : ctx.hidden.as<_Float16>() + hstride * (l - 1);
			const uint32_t ldi = l == 0 ? m_input_width : m_width;
			_Float16* g = (_Float16*)gradients + L.w_off;
			for (uint32_t r0 = 0; r0 < L.rows; r0 += 128) {
				const uint32_t rr = std::min(128u, L.rows - r0);
				for (uint32_t c0 = 0; c0 < L.cols; c0 += 128) { // column panels of at most 128
					const uint32_t cc = std::min(128u, L.cols - c0);
					// (the hidden layers' activations and gradients are tiled, k_mlp.hip hidden_tile_off: a panel's first column is whole tiles in)
					const bool dO_tiled = l != m_desc.n_layers - 1, In_tiled = l != 0;
					panels.push_back(WgradPanel{dO_tiled ? dO + (size_t)(r0 / 16) * 256 : dO + r0, ldo, rr, In_tiled ? In + (size_t)(c0 / 16) * 256 : In + c0, ldi, cc,
					                            g + (size_t)r0 * L.cols + c0, L.cols, dO_tiled, In_tiled});
				}
			}
		}
		ArenaBuf ws{stream, wgrad_panels_workspace_floats(panels.data(), (uint32_t)panels.size(), n) * sizeof(float)};
		mlp_wgrad_panels(stream, n, panels.data(), (uint32_t)panels.size(), accumulate, ws.as<float>()); // (layers of one shape share a launch)
	}

	Json hyperparams() const { // fully_fused_mlp.h:137-145
		Json j = Json::object();
		j["otype"] = m_fully_fused ? "FullyFusedMLP" : "CutlassMLP";
		j["activation"] = to_string(m_activation);
		j["output_activation"] = to_string(m_output_activation);
		j["n_neurons"] = m_width;
		j["n_hidden_layers"] = m_n_hidden;
		return j;
	}

private:
	bool m_fully_fused;
	uint32_t m_input_width, m_output_width, m_padded_output_width, m_width, m_n_hidden;
	Activation m_activation, m_output_activation;
	MlpDesc m_desc;
	size_t m_n_params;
	mutable LiveImage m_live;
	mutable int m_live_state = 0; // 0: not built yet, 1: usable, -1: not usable for this network
};

// ------------------------------------------------------------------------------------------------------------------
// Model = DifferentiableObject<float, T, T> (object.h:120-270)
// ------------------------------------------------------------------------------------------------------------------
struct ModelContext {
	virtual ~ModelContext() {}
};

class Model {
public:
	virtual ~Model() {}
	virtual uint32_t input_width() const = 0;
	virtual uint32_t output_width() const = 0;
	virtual uint32_t padded_output_width() const = 0;
	virtual size_t n_params() const = 0;
	virtual Precision precision() const = 0;
	virtual std::vector<std::pair<uint32_t, uint32_t>> layer_sizes() const = 0;
	virtual void initialize_params(Pcg32& rng, float* params_full_precision, float scale) = 0;
	virtual uint64_t list_scatters() const { return 0; } // backward passes of the model's grid encoding(s) that ran the list-fed gradient kernel
	virtual void inference(hipStream_t stream, uint32_t n, MatView input, void* output, const void* params) = 0;
	// float output of the unpadded width (object.h:147-176: inference + trim_and_cast_from); models may fuse the conversion
	virtual void inference_f32(hipStream_t stream, uint32_t n, MatView input, MatViewMut output, const void* params) {
		check_batch(n);
		if (n == 0) return;
		const uint32_t pw = padded_output_width();
		ArenaBuf tmp{stream, (size_t)n * pw * (precision() == Precision::Fp32 ? 4 : 2)};
		inference(stream, n, input, tmp.data(), params);
		trim_and_cast(stream, precision() == Precision::Fp32, n, pw, output_width(), tmp.data(), output);
	}
	virtual std::unique_ptr<ModelContext> forward(hipStream_t stream, uint32_t n, MatView input, void* output, const void* params, bool prepare_input_gradients) = 0;
	virtual void backward(hipStream_t stream, const ModelContext& ctx, uint32_t n, MatView input, const void* output, const void* dL_doutput,
	                      MatViewMut* dL_dinput, const void* params, void* gradients, GradientMode mode) = 0;
	// object.h:278-331: second-order input gradients.  dL_ddLdinput [n][input_width] float arrives at dL_dinput; optional
	// results: dL_ddLdoutput [n][padded_output_width], dL_dinput (overwritten), parameter gradients per `mode`.
	virtual void backward_backward_input(hipStream_t stream, const ModelContext& ctx, uint32_t n, MatView input, MatView dL_ddLdinput, const void* dL_doutput,
	                                     void* dL_ddLdoutput, MatViewMut* dL_dinput, const void* params, void* gradients, GradientMode mode) {
		throw std::runtime_error{"DifferentiableObject::backward_backward_input_impl: not implemented error"}; // object.h:288
	}
	virtual Json hyperparams() const = 0;
	std::string name() const { return hyperparams().value("otype", "<Unknown>"); } // object.h:51-53

	static void check_batch(uint32_t n) { // object.h:130,150,197,246
		if (n % BATCH_SIZE_GRANULARITY != 0) throw std::runtime_error{"batch size " + std::to_string(n) + " is not a multiple of BATCH_SIZE_GRANULARITY = 256"};
	}
};

// Event brackets around the pieces of ONE fused training step, on the step's stream (the measurement hook of bench.py, SURVEY 8d:
// "time the MLP fwd+bwd+wgrad region separately").  Events are created without the system-scope release a default event
// performs (hipEventDisableSystemFence): that write-back made a bracketed kernel 40 % slower than it is inside a step.
struct StepProfile {
	enum Piece { Encode = 0, MlpKernel = 1, EncodingBackward = 2, Optimizer = 3, N_PIECES = 4 };
	struct Sample { hipEvent_t begin[N_PIECES], end[N_PIECES]; bool used[N_PIECES]; };
	std::vector<Sample> samples;
	bool armed = false;
	Sample* current = nullptr;

	void arm() { armed = true; }
	void begin_step() {
		current = nullptr;
		if (!armed) return;
		armed = false;
		samples.emplace_back();
		current = &samples.back();
		for (int i = 0; i < N_PIECES; ++i) {
			current->used[i] = false;
			HIP_CHECK_THROW(hipEventCreateWithFlags(&current->begin[i], 0x20000000u)); // hipEventDisableSystemFence, timing enabled
			HIP_CHECK_THROW(hipEventCreateWithFlags(&current->end[i], 0x20000000u));
		}
	}
	void mark(hipStream_t stream, Piece p, bool end) {
		if (!current) return;
		HIP_CHECK_THROW(hipEventRecord(end ? current->end[p] : current->begin[p], stream));
		if (end) current->used[p] = true;
	}
	void end_step() { current = nullptr; }
	// sums over the profiled steps since the last call (milliseconds per piece) and their number; synchronises the stream
	uint32_t collect(hipStream_t stream, float* ms) {
		HIP_CHECK_THROW(hipStreamSynchronize(stream));
		for (int i = 0; i < N_PIECES; ++i) ms[i] = 0.0f;
		for (Sample& s : samples) {
			for (int i = 0; i < N_PIECES; ++i) {
				float t = 0.0f;
				if (s.used[i]) HIP_CHECK_THROW(hipEventElapsedTime(&t, s.begin[i], s.end[i]));
				ms[i] += t;
				(void)hipEventDestroy(s.begin[i]);
				(void)hipEventDestroy(s.end[i]);
			}
		}
		const uint32_t n = (uint32_t)samples.size();
		samples.clear();
		return n;
	}
	~StepProfile() {
		for (Sample& s : samples)
			for (int i = 0; i < N_PIECES; ++i) {
				(void)hipEventDestroy(s.begin[i]);
				(void)hipEventDestroy(s.end[i]);
			}
	}
};

class NetworkWithInputEncoding : public Model {
public:
	NetworkWithInputEncoding(uint32_t n_dims_to_encode, uint32_t n_output_dims, const Json& encoding, const Json& network) {
		// network_with_input_encoding.h:45-56: encoding aligned to the network's minimum alignment (16), network sized from it
		m_encoding = create_encoding(n_dims_to_encode, encoding, Network::REQUIRED_ALIGNMENT, false);
		Json local = network.is_object() ? network : Json::object();
		local["n_input_dims"] = m_encoding->padded_output_width();
		local["n_output_dims"] = n_output_dims;
		m_network.reset(new Network{local});
	}

	uint32_t input_width() const override { return m_encoding->input_width(); }
	uint32_t output_width() const override { return m_network->output_width(); }
	uint32_t padded_output_width() const override { return m_network->padded_output_width(); }
	size_t n_params() const override { return m_network->n_params() + m_encoding->n_params(); }
	Precision precision() const override { return Precision::Fp16; }
	std::vector<std::pair<uint32_t, uint32_t>> layer_sizes() const override { return m_network->layer_sizes(); }
	Encoding& encoding() { return *m_encoding; }
	Network& network() { return *m_network; }

	void initialize_params(Pcg32& rng, float* params_full_precision, float scale) override { // :124-130, network first
		m_network->initialize_params(rng, params_full_precision, scale);
		m_encoding->initialize_params(rng, params_full_precision + m_network->n_params(), scale);
	}

	void inference(hipStream_t stream, uint32_t n, MatView input, void* output, const void* params) override {
		inference_fused(stream, n, input, output, nullptr, params);
	}
	void inference_f32(hipStream_t stream, uint32_t n, MatView input, MatViewMut output, const void* params) override {
		inference_fused(stream, n, input, nullptr, &output, params);
	}

	// object.h:147-176 for this model, with as few passes over memory as the kernels allow: an Identity encoding is applied
	// inside the MLP kernel's input load, a grid is evaluated into level planes by the XCD-aware kernel, and the float
	// output (trim_and_cast_from) is written by the MLP kernel itself.
	void inference_fused(hipStream_t stream, uint32_t n, MatView input, void* output_half, const MatViewMut* output_f32, const void* params) {
		check_batch(n);
		if (n == 0) return;
		const _Float16* p = (const _Float16*)params;
		MlpIo io{};
		io.out_half = output_half;
		if (output_f32) {
			io.out_f32 = *output_f32;
			io.out_f32_dims = m_network->output_width();
		}
		ArenaBuf network_input;
		float scale = 1, offset = 0;
		if (m_encoding->as_identity(scale, offset)) {
			io.x_f32 = input;
			io.x_f32_dims = m_encoding->input_width();
			io.x_scale = scale;
			io.x_offset = offset;
		} else if (m_encoding->as_oneblob() && m_encoding->padded_output_width() == m_network->input_width() && (m_network->desc().width == 64 || m_network->desc().width == 128)) {
			io.x_f32 = input;
			io.x_f32_dims = m_encoding->input_width();
			io.x_oneblob_bins = m_encoding->as_oneblob();
		} else {
			network_input = ArenaBuf{stream, (size_t)n * m_encoding->padded_output_width() * 2};
			io.x_half = network_input.data();
			io.x_plane_features = m_encoding->forward_plane_features(n);
			if (io.x_plane_features) m_encoding->forward_planes(stream, n, input, p + m_network->n_params(), network_input.data(), false);
			else m_encoding->forward(stream, n, input, p + m_network->n_params(), network_input.data(), false, false);
		}
		m_network->inference_io(stream, n, io, p);
	}

	struct Ctx : public ModelContext {
		ArenaBuf network_input;     // AoS [n][padded], or level planes when x_plane_f > 0
		EncodingContext encoding_ctx;
		NetworkContext network_ctx; // hidden activations; empty for fused contexts
		uint32_t x_plane_f = 0;
		bool fused = false;         // produced by fused_encode(): backward() goes through the fused MLP kernel
		uint32_t oneblob_bins = 0;  // > 0: no encoded batch was written -- the MLP kernels evaluate the OneBlob encoding of the input themselves
		ArenaBuf image;             // the network's fragment images, if the encoding's forward kernel built them on the way (MlpPrepJob)
		// the MLP's weight-gradient slabs once their reduction was handed to a LATER launch (AdamPrologue: the optimizer's, enqueued after
		// fused_step returns): they stay allocated as long as this context does -- the arena's rule is "consumer enqueued before the free"
		mutable ArenaBuf slabs_kept;
	};

	// cpp_api.cu:84-95.  Without input gradients the forward pass keeps nothing but the encoded batch: backward() recomputes the
	// MLP's forward pass inside the fused kernel (k_train.hip) from it, which is cheaper than storing and re-reading the hidden
	// activations (k_mlp_fwd with hidden stores + k_mlp_bwd + 3 x k_wgrad: 158 us; the fused kernel: 78 us on C3a).
	std::unique_ptr<ModelContext> forward(hipStream_t stream, uint32_t n, MatView input, void* output, const void* params, bool prepare_input_gradients) override {
		check_batch(n);
		auto ctx = std::make_unique<Ctx>();
		if (n == 0) return ctx;
		const _Float16* p = (const _Float16*)params;
		if (!prepare_input_gradients && fused_step_supported(n)) {
			fused_encode(stream, *ctx, n, input, params, false, true);
			MlpIo io{};
			if (ctx->oneblob_bins) {
				io.x_f32 = input;
				io.x_f32_dims = m_encoding->input_width();
				io.x_oneblob_bins = ctx->oneblob_bins;
			} else {
				io.x_half = ctx->network_input.data();
				io.x_plane_features = ctx->x_plane_f;
			}
			io.out_half = output;
			m_network->inference_io(stream, n, io, p);
			return ctx;
		}
		ctx->network_input = ArenaBuf{stream, (size_t)n * m_encoding->padded_output_width() * 2};
		ctx->encoding_ctx = m_encoding->forward(stream, n, input, p + m_network->n_params(), ctx->network_input.data(), prepare_input_gradients, true);
		ctx->network_ctx = m_network->forward(stream, n, ctx->network_input.data(), output, p);
		return ctx;
	}

	void backward(hipStream_t stream, const ModelContext& mctx, uint32_t n, MatView input, const void* output, const void* dL_doutput,
	              MatViewMut* dL_dinput, const void* params, void* gradients, GradientMode mode) override {
		check_batch(n);
		if (n == 0) return;
		const Ctx& ctx = dynamic_cast<const Ctx&>(mctx);
		const _Float16* p = (const _Float16*)params;
		_Float16* g = (_Float16*)gradients;
		if (ctx.fused) {
			if (dL_dinput) throw std::runtime_error{"NetworkWithInputEncoding::backward: input gradients were not prepared by forward()"};
			fused_mlp_and_scatter(stream, ctx, n, input, nullptr, nullptr, dL_doutput, LossType::L2, 1.0f, nullptr, nullptr, nullptr, false, nullptr, params, gradients, mode);
			return;
		}
		ArenaBuf dL_dnetwork_input;
		if (m_encoding->n_params() > 0 || dL_dinput) { // :93-96
			dL_dnetwork_input = ArenaBuf{stream, (size_t)n * m_encoding->padded_output_width() * 2};
		}
		// the grid scatter reads dL/d(encoding) with unit stride when the MLP writes it as level planes
		const uint32_t plane_f = dL_dnetwork_input ? m_encoding->level_plane_features(dL_dinput != nullptr, mode) : 0;
		m_network->backward(stream, ctx.network_ctx, n, ctx.network_input.data(), output, dL_doutput, dL_dnetwork_input.data(), p, g, mode, plane_f);
		if (dL_dnetwork_input) {
			m_encoding->backward(stream, ctx.encoding_ctx, n, input, dL_dnetwork_input.data(), dL_dinput, p + m_network->n_params(),
			                     g ? g + m_network->n_params() : nullptr, mode, plane_f > 0);
		}
	}

	// TCNN_AMD_FUSED_STEP=0 selects the reference-shaped kernel sequence (forward / loss / backward / wgrad) for A/B runs
	static bool use_fused_step() { return switches().fused_step; }
	// TCNN_AMD_SIDE_JOBS=0: k_mlp_prep stays a launch of its own (A/B runs; read per step so that tests cover both)
	static bool side_jobs_enabled() { return switches().side_jobs; }
	bool fused_step_supported(uint32_t n) const { return use_fused_step() && mlp_train_fused_supported(m_network->desc(), n); }
	// the fused step of a model without encoding parameters hands its weight gradients to the optimizer inside the slab reduction
	bool optimizer_rides_on_reduce() const { return m_encoding->n_params() == 0; }
	// TCNN_AMD_LIVE_IMAGE=0: every step builds its fragment images with k_mlp_prep again instead of keeping one current (Network::live_image;
	// bit-identical, one ~4.5 us launch more per step of a model without encoding parameters)
	static bool live_image_enabled() { return switches().live_image; }
	bool live_image_kept() const { return m_live_image_kept; }           // the last fused step's optimizer launch left the live image current
	void invalidate_live_image() { m_network->invalidate_live_image(); } // the parameters change(d) some other way
	size_t image_preps() const { return m_image_preps; }                 // k_mlp_prep launches of fused steps so far (a test's view of the above)
	uint64_t scatter_wide_fallbacks() { return m_encoding->scatter_wide_fallbacks(); }
	uint64_t list_scatters() const override { return m_encoding->list_scatters(); }
	bool context_keeps_slabs(const ModelContext& c) const { const Ctx* x = dynamic_cast<const Ctx*>(&c); return x && (bool)x->slabs_kept; }
	// the register-resident fused kernel (k_train_regs.hip) writes dL_doutput / L as compact [n][dims] matrices (TrainContext::compact)
	bool fused_compact_context_supported(uint32_t n) const {
		const bool ok = use_fused_step() && mlp_train_regs_supported(m_network->desc(), n) && m_network->padded_output_width() == 16;
		// the register-resident kernels address [n][...] matrices with 32-bit byte offsets: beyond 2^22 rows the step silently took the
		// much slower LDS-image kernel -- say so once (a caller can split the batch)
		if (!ok && use_fused_step() && n > (1u << 22) && mlp_train_regs_supported(m_network->desc(), 1u << 22)) {
			static bool told = false;
			if (!told) {
				told = true;
				const std::string msg = "training_step: batches of more than 4194304 rows run the general fused MLP kernel (several times slower for this network); split the batch to stay on the fast path";
				log_message(TCNN_LOG_WARNING, msg);
				fprintf(stderr, "tcnn_amd warning: %s\n", msg.c_str());
			}
		}
		return ok;
	}

	// forward + loss + backward of a training step with the MLP part as ONE kernel (k_train.hip): same results as
	// forward() -> loss_evaluate() -> backward(), activations never leave the CU.  out / dL_dout / L: [n][padded_out].
	std::unique_ptr<ModelContext> fused_step(hipStream_t stream, uint32_t n, MatView input, const float* target, const float* data_pdf, const void* external_dL_dy,
	                                         LossType loss, float loss_scale, void* out, void* dL_dout, float* L, bool compact_context, MatViewMut* dL_dinput, const void* params,
	                                         void* gradients, GradientMode mode, StepProfile* profile = nullptr, const AdamInFlush* adam = nullptr, ParamRanges* adam_done = nullptr,
	                                         AdamPrologue* prologue = nullptr) {
		check_batch(n);
		auto ctx = std::make_unique<Ctx>();
		// Everything in order on the caller's stream.  Running the two small kernels around the MLP kernel (k_mlp_prep, k_wgrad_reduce,
		// ~5 us each, independent of the encoding kernels) on a side stream was measured and lost 13-15 us per step on every
		// workload: a cross-stream event dependency costs more here than the kernels it hides (the same happened with Adam).
		if (profile) profile->mark(stream, StepProfile::Encode, false);
		fused_encode(stream, *ctx, n, input, params, dL_dinput != nullptr, mode != GradientMode::Ignore, side_jobs_enabled());
		if (profile) profile->mark(stream, StepProfile::Encode, true);
		fused_mlp_and_scatter(stream, *ctx, n, input, target, data_pdf, external_dL_dy, loss, loss_scale, out, dL_dout, L, compact_context, dL_dinput, params, gradients, mode, profile,
		                      adam, adam_done, prologue);
		return ctx;
	}

	// first half of the fused step: the encoding, as level planes where the encoding can produce them
	void fused_encode(hipStream_t stream, Ctx& ctx, uint32_t n, MatView input, const void* params, bool prepare_input_gradients, bool prepare_param_gradients, bool prep_image = false) {
		const _Float16* p = (const _Float16*)params;
		const uint32_t n_net = (uint32_t)m_network->n_params();
		// a OneBlob encoding is evaluated by the MLP kernels inside their input load where they can (k_mlp.hip, k_train.hip)
		if (!prepare_input_gradients && m_encoding->as_oneblob() && m_encoding->padded_output_width() == m_network->input_width() &&
		    mlp_train_fused_oneblob_supported(m_network->desc(), n, m_encoding->as_oneblob())) {
			ctx.oneblob_bins = m_encoding->as_oneblob();
			ctx.fused = true;
			return;
		}
		ctx.network_input = ArenaBuf{stream, (size_t)n * m_encoding->padded_output_width() * 2};
		// grids hand the encoded batch over as level planes (XCD-aware forward kernel, scatter filter produced on the way)
		ctx.x_plane_f = prepare_input_gradients ? 0 : m_encoding->forward_plane_features(n);
		if (ctx.x_plane_f && prep_image) { // the forward kernel also builds the network's fragment images (k_mlp_prep as a side job)
			ctx.image = ArenaBuf{stream, mlp_image_bytes(m_network->desc())};
			MlpPrepJob job;
			job.desc = m_network->desc();
			job.params = params;
			job.image = ctx.image.data();
			ctx.encoding_ctx = m_encoding->forward_planes(stream, n, input, p + n_net, ctx.network_input.data(), prepare_param_gradients, &job);
		} else if (ctx.x_plane_f) ctx.encoding_ctx = m_encoding->forward_planes(stream, n, input, p + n_net, ctx.network_input.data(), prepare_param_gradients);
		else ctx.encoding_ctx = m_encoding->forward(stream, n, input, p + n_net, ctx.network_input.data(), prepare_input_gradients, prepare_param_gradients);
		ctx.fused = true;
	}

	// second half: ONE MLP kernel (forward recomputed in registers, loss or external dL/doutput, backward, weight gradients),
	// the slab reduction and the encoding's backward pass.  target == nullptr requires external_dL_dy.
	void fused_mlp_and_scatter(hipStream_t stream, const Ctx& ctx, uint32_t n, MatView input, const float* target, const float* data_pdf, const void* external_dL_dy,
	                           LossType loss, float loss_scale, void* out, void* dL_dout, float* L, bool compact_context, MatViewMut* dL_dinput, const void* params, void* gradients,
	                           GradientMode mode, StepProfile* profile = nullptr, const AdamInFlush* adam = nullptr, ParamRanges* adam_done = nullptr, AdamPrologue* prologue = nullptr) {
		const _Float16* p = (const _Float16*)params;
		_Float16* g = (_Float16*)gradients;
		const uint32_t n_net = (uint32_t)m_network->n_params();
		const uint32_t x_plane_f = ctx.x_plane_f;
		const bool need_dx = m_encoding->n_params() > 0 || dL_dinput;
		ArenaBuf dL_dnetwork_input;
		const uint32_t plane_f = need_dx ? m_encoding->level_plane_features(dL_dinput != nullptr, mode) : 0;
		// scatter records: the MLP kernel interleaves the samples' coordinates with dL/d(encoding) so that the grid scatter needs one gather per hit
		// (not with hit lists: their elements carry entries and weights, the scatter gathers dL/dy alone from plain level planes)
		const bool records = plane_f > 0 && !ctx.encoding_ctx.hit_elems && m_encoding->padded_output_width() == m_encoding->output_width() && m_encoding->scatter_records_usable(input);
		if (need_dx) dL_dnetwork_input = ArenaBuf{stream, records ? (size_t)n * m_encoding->scatter_record_planes() * 16 : (size_t)n * m_encoding->padded_output_width() * 2};
		ctx.encoding_ctx.dy_records = records;

		const MlpDesc& d = m_network->desc();
		// a model whose only parameters are the network's: the optimizer's update rides on the slab reduction (k_wgrad_reduce_adam), which
		// then also keeps the network's live image current (Network::live_image)
		const bool with_adam = adam && adam_done && !need_dx && m_encoding->n_params() == 0 && mode == GradientMode::Overwrite;
		Network::LiveImage* live = with_adam && !ctx.image && live_image_enabled() ? m_network->live_image() : nullptr;
		m_live_image_kept = false;
		ArenaBuf prepared;
		const void* image_data;
		if (ctx.image) image_data = ctx.image.data();
		else if (live) {
			if (live->params != params) { // first step, or the parameters were changed behind the image's back
				mlp_prepare_weights(stream, d, params, live->image.data(), true);
				++m_image_preps;
			}
			image_data = live->image.data();
		} else {
			prepared = m_network->prepare(stream, params, true);
			++m_image_preps;
			image_data = prepared.data();
		}
		ArenaBuf slabs;
		uint32_t n_slabs = 0;
		if (mode != GradientMode::Ignore) {
			CHECK_THROW(gradients != nullptr);
			// the 32x32x16 kernels of BASELINE configs 3 have a grid of their own (k_train_r32.hip); the test is the one mlp_train_regs makes
			const bool r32 = !ctx.oneblob_bins && mlp_train_regs_supported(d, n) && mlp_train_r32_applies(d, n, x_plane_f, data_pdf, external_dL_dy, m_network->output_width(), loss, out,
			                                                           need_dx ? dL_dnetwork_input.data() : nullptr, plane_f, records ? input.data : nullptr, records ? m_encoding->input_width() : 0u);
			n_slabs = r32 ? mlp_train_r32_grid(n) : mlp_train_fused_grid(d, n, ctx.oneblob_bins, ctx.oneblob_bins ? m_encoding->input_width() : 0u);
			slabs = ArenaBuf{stream, (size_t)n_slabs * n_net * sizeof(float)};
		}
		const MlpOneBlobInput oneblob_input{input, m_encoding->input_width(), ctx.oneblob_bins};
		if (profile) profile->mark(stream, StepProfile::MlpKernel, false);
		mlp_train_fused(stream, d, image_data, n, ctx.network_input.data(), x_plane_f, target, data_pdf, external_dL_dy, m_network->output_width(), loss, loss_scale, out, dL_dout, L, compact_context,
		                dL_dnetwork_input.data(), plane_f, records ? input.data : nullptr, records ? m_encoding->input_width() : 0u, slabs.as<float>(), n_net,
		                ctx.oneblob_bins ? &oneblob_input : nullptr);
		if (profile) profile->mark(stream, StepProfile::MlpKernel, true);
		// (Carrying the slab reduction on the grid scatter's launch the way k_mlp_prep rides on the forward kernel was built and
		// measured: its workgroups each take a whole CU's LDS slot for a few microseconds, in front of the task list they delay the
		// long coarse-level tasks, behind it they wait for a slot -- the scatter grew by 5.4 / 6.5 us for the 4.4 us launch saved.)
		// The slab reduction rides on the scatter's finalize launch when the encoding's backward pass has one (two ~4.5 us launches
		// become one); otherwise, or with TCNN_AMD_SIDE_JOBS=0, it is a launch of its own.
		MlpReduceJob reduce_job;
		if (mode != GradientMode::Ignore) {
			reduce_job.n_elems = n_net;
			reduce_job.n_slabs = n_slabs;
			reduce_job.slabs = slabs.as<float>();
			reduce_job.grad = g;
			reduce_job.accumulate = mode == GradientMode::Accumulate ? 1 : 0;
			if (!(need_dx && side_jobs_enabled())) {
				AdamInFlush adam_here;
				if (with_adam) {
					adam_here = *adam;
					if (live) {
						adam_here.image = live->image.data();
						adam_here.image_inv = live->inverse.as<uint32_t>();
					}
				}
				mlp_reduce_slabs(stream, n_net, n_slabs, slabs.as<float>(), g, mode == GradientMode::Accumulate, with_adam ? &adam_here : nullptr);
				if (with_adam) {
					adam_done->clear();
					adam_done->emplace_back((size_t)0, (size_t)n_net);
					if (live) {
						live->params = params; // current again once this launch has run
						m_live_image_kept = true;
					}
				}
			} else ctx.encoding_ctx.reduce_job = &reduce_job;
		}
		if (need_dx) {
			if (profile) profile->mark(stream, StepProfile::EncodingBackward, false);
			// the optimizer's offer to have its update applied by the gradient kernel, re-based to the encoding's parameters
			AdamInFlush enc_adam;
			if (adam && adam_done) {
				enc_adam = adam->advanced(n_net);
				ctx.encoding_ctx.adam = &enc_adam;
			}
			ctx.encoding_ctx.prologue = prologue;
			m_encoding->backward(stream, ctx.encoding_ctx, n, input, dL_dnetwork_input.data(), dL_dinput, p + n_net, g ? g + n_net : nullptr, mode, plane_f > 0);
			ctx.encoding_ctx.prologue = nullptr;
			ctx.encoding_ctx.adam = nullptr;
			if (ctx.encoding_ctx.reduce_job) {
				ctx.encoding_ctx.reduce_job = nullptr;
				if (!reduce_job.taken) mlp_reduce_slabs(stream, n_net, n_slabs, slabs.as<float>(), g, mode == GradientMode::Accumulate);
				else if (prologue && prologue->pending && prologue->has_reduce) ctx.slabs_kept = std::move(slabs); // read by the optimizer's launch
			}
			if (adam_done) {
				adam_done->clear();
				for (const auto& r : ctx.encoding_ctx.adam_done) adam_done->emplace_back(r.first + n_net, r.second + n_net);
			}
			if (profile) profile->mark(stream, StepProfile::EncodingBackward, true);
		}
	}

	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "NetworkWithInputEncoding";
		j["encoding"] = m_encoding->hyperparams();
		j["network"] = m_network->hyperparams();
		return j;
	}

private:
	bool m_live_image_kept = false;
	size_t m_image_preps = 0;
	std::unique_ptr<Encoding> m_encoding;
	std::unique_ptr<Network> m_network;
};

// standalone encoding as a Module (cpp_api.cu:156-165: alignment 0 -> no padding)
class EncodingModel : public Model {
public:
	EncodingModel(uint32_t n_dims_to_encode, const Json& encoding, Precision precision) : m_precision{precision} {
		m_encoding = create_encoding(n_dims_to_encode, encoding, 0, precision == Precision::Fp32);
	}
	uint32_t input_width() const override { return m_encoding->input_width(); }
	uint32_t output_width() const override { return m_encoding->padded_output_width(); }
	uint32_t padded_output_width() const override { return m_encoding->padded_output_width(); }
	size_t n_params() const override { return m_encoding->n_params(); }
	Precision precision() const override { return m_precision; }
	std::vector<std::pair<uint32_t, uint32_t>> layer_sizes() const override { return {}; }
	void initialize_params(Pcg32& rng, float* params_full_precision, float scale) override { m_encoding->initialize_params(rng, params_full_precision, scale); }
	uint64_t list_scatters() const override { return m_encoding->list_scatters(); }

	struct Ctx : public ModelContext {
		EncodingContext encoding_ctx;
	};
	void inference(hipStream_t stream, uint32_t n, MatView input, void* output, const void* params) override {
		check_batch(n);
		m_encoding->forward(stream, n, input, params, output, false, false);
	}
	std::unique_ptr<ModelContext> forward(hipStream_t stream, uint32_t n, MatView input, void* output, const void* params, bool prepare_input_gradients) override {
		check_batch(n);
		auto ctx = std::make_unique<Ctx>();
		ctx->encoding_ctx = m_encoding->forward(stream, n, input, params, output, prepare_input_gradients, true);
		return ctx;
	}
	void backward(hipStream_t stream, const ModelContext& mctx, uint32_t n, MatView input, const void* output, const void* dL_doutput,
	              MatViewMut* dL_dinput, const void* params, void* gradients, GradientMode mode) override {
		check_batch(n);
		const Ctx& ctx = dynamic_cast<const Ctx&>(mctx);
		m_encoding->backward(stream, ctx.encoding_ctx, n, input, dL_doutput, dL_dinput, params, gradients, mode, false);
	}
	void backward_backward_input(hipStream_t stream, const ModelContext& mctx, uint32_t n, MatView input, MatView dL_ddLdinput, const void* dL_doutput,
	                             void* dL_ddLdoutput, MatViewMut* dL_dinput, const void* params, void* gradients, GradientMode mode) override {
		check_batch(n);
		const Ctx& ctx = dynamic_cast<const Ctx&>(mctx);
		m_encoding->backward_backward_input(stream, ctx.encoding_ctx, n, input, dL_ddLdinput, dL_doutput, dL_ddLdoutput, dL_dinput, params, gradients, mode);
	}
	Json hyperparams() const override { return m_encoding->hyperparams(); }
private:
	Precision m_precision;
	std::unique_ptr<Encoding> m_encoding;
};

// ------------------------------------------------------------------------------------------------------------------
// Loss (src/loss.cu:85-93) and Adam (optimizers/adam.h:122-327, src/optimizer.cu:50-82)
// ------------------------------------------------------------------------------------------------------------------
inline LossType create_loss(const Json& loss) {
	const std::string otype = loss.value("otype", "RelativeL2");
	static const std::pair<const char*, LossType> table[] = { // src/loss.cu:57-65
		{"L2", LossType::L2}, {"RelativeL2", LossType::RelativeL2}, {"RelativeL2Luminance", LossType::RelativeL2Luminance}, {"L1", LossType::L1},
		{"RelativeL1", LossType::RelativeL1}, {"Mape", LossType::Mape}, {"Smape", LossType::Smape}, {"CrossEntropy", LossType::CrossEntropy}, {"Variance", LossType::Variance}};
	for (const auto& kv : table) if (equals_case_insensitive(otype, kv.first)) return kv.second;
	throw std::runtime_error{"Invalid loss type: " + otype};
}
inline const char* to_string(LossType t) {
	static const char* names[] = {"L2", "RelativeL2", "L1", "RelativeL1", "Mape", "Smape", "CrossEntropy", "Variance", "RelativeL2Luminance"};
	return names[(uint32_t)t];
}
// the fused training kernel evaluates these two inside its loss stage; the others go forward -> k_loss -> backward
inline bool loss_in_fused_kernel(LossType t) { return t == LossType::L2 || t == LossType::RelativeL2; }

// snapshot helpers (gpu_memory_json.h:36-71): device memory <-> json binary value
inline Json device_to_binary(const void* device, size_t n_bytes) {
	std::vector<uint8_t> host(n_bytes);
	if (n_bytes) HIP_CHECK_THROW(hipMemcpy(host.data(), device, n_bytes, hipMemcpyDeviceToHost));
	return Json::binary(std::move(host));
}
// binary value, or nlohmann's text form of one: {"bytes": [...], "subtype": null} (gpu_memory_json.h:55-67)
inline std::vector<uint8_t> binary_of(const Json& j) {
	if (j.is_binary()) return j.get_binary();
	if (j.is_object()) {
		const Json& arr = j["bytes"];
		std::vector<uint8_t> bytes(arr.size());
		for (size_t i = 0; i < bytes.size(); ++i) bytes[i] = (uint8_t)arr.at(i).as_double();
		return bytes;
	}
	throw std::runtime_error{"Invalid json type: must be either binary or object"};
}

// optimizer.h:44-95
class Optimizer {
public:
	virtual ~Optimizer() {}
	virtual void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) = 0;
	virtual void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) = 0;
	// A step in two parts, for gradient kernels that can apply the update to the parameters they own (AdamInFlush): begin_split_step
	// starts the step and describes it (false: this optimizer cannot be split, nothing has happened -- call step()); finish_split_step
	// updates every parameter outside `done`.  Together they equal step() bit for bit.
	virtual bool begin_split_step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, AdamInFlush& out) { return false; }
	virtual void finish_split_step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients, const ParamRanges& done) {
		throw std::runtime_error{"Optimizer: finish_split_step without begin_split_step"};
	}
	// step() with the backward pass's finalize pass as the prologue of the same launch (AdamPrologue).  takes_prologue(): this optimizer
	// can; step_with_prologue returns false when this prologue's shapes do not fit -- nothing has happened then, the caller runs the
	// finalize pass and step() itself.
	virtual bool takes_prologue() const { return false; }
	virtual bool step_with_prologue(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, void* gradients, const AdamPrologue& prologue) { return false; }
	virtual float learning_rate() const = 0;
	virtual void set_learning_rate(float val) = 0;
	virtual uint32_t step_count() const = 0;
	virtual size_t n_weights() const = 0;
	virtual void* custom_weights() const { return nullptr; } // half weights the trainer uses for inference instead of its own (EMA)
	virtual void update_hyperparams(const Json& params) = 0;
	virtual Json hyperparams() const = 0;
	virtual Json serialize() const = 0;
	virtual void deserialize(const Json& data, size_t n_weights) = 0;
	// after a snapshot was loaded into `weights`: optimizers that assemble custom weights from several sources rebuild them
	virtual void weights_restored(hipStream_t stream, const void* weights) {}
};
inline std::unique_ptr<Optimizer> create_optimizer(const Json& params);

class AdamOptimizer : public Optimizer {
public:
	explicit AdamOptimizer(const Json& params) { update_hyperparams(params); }
	float learning_rate() const override { return m_h.learning_rate; }
	void set_learning_rate(float val) override { m_h.learning_rate = val; }
	size_t n_weights() const override { return m_n_weights; }

	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override { // adam.h:128-148
		m_n_weights = n_weights;
		if (n_weights * sizeof(float) > m_first_moments.bytes()) {
			m_first_moments.resize(n_weights * sizeof(float));
			m_second_moments.resize(n_weights * sizeof(float));
		}
		m_steps16 = narrow_steps_enabled();
		m_param_steps.resize(0);
		m_param_steps.resize(n_weights * step_bytes());
		m_first_moments.memset(0);
		m_second_moments.memset(0);
		m_param_steps.memset(0);
		m_n_matrix = 0;
		for (const auto& ls : layer_sizes) m_n_matrix += (size_t)ls.first * ls.second;
	}

	// adam.h:278-299
	Json serialize() const override {
		auto blob = [](const DeviceBuf& buf, size_t n_bytes) { return device_to_binary(buf.data(), n_bytes); };
		Json data = Json::object();
		data["current_step"] = Json((uint32_t)m_current_step);
		data["base_learning_rate"] = Json((float)m_h.learning_rate);
		data["first_moments_binary"] = blob(m_first_moments, m_n_weights * sizeof(float));
		data["second_moments_binary"] = blob(m_second_moments, m_n_weights * sizeof(float));
		if (m_steps16) { // the snapshot carries uint32 counts whatever is kept here
			std::vector<uint16_t> narrow(m_n_weights);
			if (m_n_weights) HIP_CHECK_THROW(hipMemcpy(narrow.data(), m_param_steps.data(), m_n_weights * sizeof(uint16_t), hipMemcpyDeviceToHost));
			std::vector<uint8_t> wide(m_n_weights * sizeof(uint32_t));
			for (size_t i = 0; i < m_n_weights; ++i) {
				const uint32_t v = narrow[i];
				std::memcpy(wide.data() + 4 * i, &v, 4);
			}
			data["param_steps_binary"] = Json::binary(std::move(wide));
		} else {
			data["param_steps_binary"] = blob(m_param_steps, m_n_weights * sizeof(uint32_t));
		}
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		auto load = [&](DeviceBuf& buf, const std::vector<uint8_t>& bytes, size_t elem) {
			if (bytes.size() != n_weights * elem) throw std::runtime_error{"Adam: snapshot state has the wrong size."};
			buf.resize(bytes.size());
			if (!bytes.empty()) HIP_CHECK_THROW(hipMemcpy(buf.data(), bytes.data(), bytes.size(), hipMemcpyHostToDevice));
		};
		m_n_weights = n_weights;
		load(m_first_moments, binary_of(data["first_moments_binary"]), sizeof(float));
		load(m_second_moments, binary_of(data["second_moments_binary"]), sizeof(float));
		m_current_step = (uint32_t)data["current_step"].as_double();
		m_steps16 = narrow_steps_enabled() && m_current_step < NARROW_STEP_LIMIT;
		if (data.contains("param_steps_binary")) {
			const std::vector<uint8_t>& bytes = binary_of(data["param_steps_binary"]);
			if (bytes.size() != n_weights * sizeof(uint32_t)) throw std::runtime_error{"Adam: snapshot state has the wrong size."};
			std::vector<uint16_t> narrow(m_steps16 ? n_weights : 0);
			for (size_t i = 0; m_steps16 && i < n_weights; ++i) {
				uint32_t v;
				std::memcpy(&v, bytes.data() + 4 * i, 4);
				if (v >= NARROW_STEP_LIMIT) m_steps16 = false; // a count the narrow form cannot hold (a foreign snapshot): keep all 32 bits
				else narrow[i] = (uint16_t)v;
			}
			m_param_steps.resize(0);
			m_param_steps.resize(n_weights * step_bytes());
			if (n_weights) HIP_CHECK_THROW(hipMemcpy(m_param_steps.data(), m_steps16 ? (const void*)narrow.data() : (const void*)bytes.data(), n_weights * step_bytes(), hipMemcpyHostToDevice));
		} else {
			m_param_steps.resize(0);
			m_param_steps.resize(n_weights * step_bytes());
			m_param_steps.memset(0);
		}
		m_h.learning_rate = (float)data["base_learning_rate"].as_double();
	}

	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override { // adam.h:150-188
		++m_current_step;
		ensure_debias_table(stream);
		ensure_step_width(stream);
		adam_step(stream, m_h, m_n_weights, m_n_matrix, loss_scale, m_current_step, weights_full_precision, weights, gradients,
		          m_first_moments.as<float>(), m_second_moments.as<float>(), m_param_steps.data(), m_steps16, m_debias.as<float>());
	}

	bool takes_prologue() const override { return true; }
	bool step_with_prologue(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, void* gradients, const AdamPrologue& prologue) override {
		if (switches().adam_prologue_refused) return false; // (tests: the caller's path for a prologue this launch does not take)
		++m_current_step;
		ensure_debias_table(stream);
		ensure_step_width(stream);
		if (adam_step_with_prologue(stream, m_h, m_n_weights, m_n_matrix, loss_scale, m_current_step, weights_full_precision, weights, gradients, m_first_moments.as<float>(),
		                            m_second_moments.as<float>(), m_param_steps.data(), m_steps16, m_debias.as<float>(), prologue)) return true;
		--m_current_step; // (the table and the width of the counts are as step() wants them; it counts the step itself)
		return false;
	}

	// The per-parameter update counts (this fork's adam.h:66-99: a parameter whose gradient is zero in a step is skipped, so it has
	// a count of its own) are 4 of the 18 bytes per parameter a step reads and 4 of the 18 it writes.  No count exceeds the
	// optimizer's own step count, so while that is below 2^16 - 1 the counts are kept as uint16 -- 34 -> 32 bytes moved per parameter
	// by the HBM-bound kernel -- and widened in place of one step's time before one could overflow; snapshots carry uint32 either
	// way.  TCNN_AMD_ADAM_STEPS32=1: uint32 from the start (A/B runs, tests).
	static constexpr uint32_t NARROW_STEP_LIMIT = 65535;
	static bool narrow_steps_enabled() { return !switches().adam_steps32; }
	size_t step_bytes() const { return m_steps16 ? sizeof(uint16_t) : sizeof(uint32_t); }
	void ensure_step_width(hipStream_t stream) { // call with m_current_step = the step about to be applied
		if (!m_steps16 || m_current_step < NARROW_STEP_LIMIT) return;
		DeviceBuf wide;
		wide.resize(m_n_weights * sizeof(uint32_t));
		adam_widen_steps(stream, m_n_weights, m_param_steps.data(), wide.data());
		HIP_CHECK_THROW(hipStreamSynchronize(stream)); // once in 65 535 steps: the narrow array is freed right after
		m_param_steps = std::move(wide);
		m_steps16 = false;
	}

	bool begin_split_step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, AdamInFlush& out) override {
		++m_current_step;
		ensure_debias_table(stream);
		ensure_step_width(stream);
		out.args = make_adam_args(m_h, loss_scale, m_current_step);
		out.w_fp = weights_full_precision;
		out.w_half = weights;
		out.m1 = m_first_moments.as<float>();
		out.m2 = m_second_moments.as<float>();
		out.steps = m_param_steps.data();
		out.steps16 = m_steps16 ? 1u : 0u;
		out.debias_table = m_debias.as<float>();
		return true;
	}
	void finish_split_step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients, const ParamRanges& done) override {
		size_t begin = 0;
		auto update = [&](size_t b, size_t e) { // parameters [b, e) the usual way; the matrix weights are the first m_n_matrix of the vector
			if (e <= b) return;
			const size_t n_matrix = m_n_matrix > b ? std::min(m_n_matrix - b, e - b) : 0;
			adam_step(stream, m_h, e - b, n_matrix, loss_scale, m_current_step, weights_full_precision + b, (char*)weights + 2 * b, (const char*)gradients + 2 * b,
			          m_first_moments.as<float>() + b, m_second_moments.as<float>() + b, (char*)m_param_steps.data() + b * step_bytes(), m_steps16, m_debias.as<float>());
		};
		for (const auto& r : done) {
			CHECK_THROW(r.first >= begin && r.second <= m_n_weights);
			update(begin, r.first);
			begin = r.second;
		}
		update(begin, m_n_weights);
	}

	// debias factors for steps [0, m_debias_filled), computed ahead in blocks of 4096 steps; refilled when the betas change
	void ensure_debias_table(hipStream_t stream) {
		const bool stale = m_debias_beta1 != m_h.beta1 || m_debias_beta2 != m_h.beta2;
		if (!stale && m_current_step < m_debias_filled) return;
		const uint32_t want = next_multiple(m_current_step + 1, 4096u) + 4096u;
		if ((size_t)want * sizeof(float) > m_debias.bytes()) {
			HIP_CHECK_THROW(hipStreamSynchronize(stream)); // the old table may still be in use
			m_debias.resize((size_t)want * 2 * sizeof(float));
			m_debias_filled = 0;
		}
		const uint32_t from = stale ? 0u : m_debias_filled;
		adam_fill_debias_table(stream, m_h.beta1, m_h.beta2, from, want, m_debias.as<float>());
		m_debias_filled = want;
		m_debias_beta1 = m_h.beta1;
		m_debias_beta2 = m_h.beta2;
	}

	void update_hyperparams(const Json& p) override { // adam.h:210-258
		if (!p.is_object()) return;
		if (p.contains("beta1")) m_h.beta1 = (float)p["beta1"].as_double();
		if (p.contains("beta2")) m_h.beta2 = (float)p["beta2"].as_double();
		if (p.contains("epsilon")) m_h.epsilon = (float)p["epsilon"].as_double();
		if (p.contains("learning_rate")) m_h.learning_rate = (float)p["learning_rate"].as_double();
		if (p.contains("l2_reg")) m_h.l2_reg = (float)p["l2_reg"].as_double();
		if (p.contains("adabound")) m_h.adabound = p["adabound"].as_bool();
		if (p.contains("relative_decay")) m_h.relative_decay = (float)p["relative_decay"].as_double();
		if (p.contains("absolute_decay")) m_h.absolute_decay = (float)p["absolute_decay"].as_double();
		if (p.contains("clipping_magnitude")) m_h.clipping_magnitude = (float)p["clipping_magnitude"].as_double();
		if (p.contains("non_matrix_learning_rate_factor")) m_h.non_matrix_learning_rate_factor = (float)p["non_matrix_learning_rate_factor"].as_double();
		if (p.contains("optimize_matrix_params")) m_h.optimize_matrix_params = p["optimize_matrix_params"].as_bool();
		if (p.contains("optimize_non_matrix_params")) m_h.optimize_non_matrix_params = p["optimize_non_matrix_params"].as_bool();
	}

	Json hyperparams() const override { // adam.h:260-276
		Json j = Json::object();
		j["otype"] = "Adam";
		j["beta1"] = m_h.beta1;
		j["beta2"] = m_h.beta2;
		j["epsilon"] = m_h.epsilon;
		j["learning_rate"] = m_h.learning_rate;
		j["l2_reg"] = m_h.l2_reg;
		j["adabound"] = m_h.adabound;
		j["relative_decay"] = m_h.relative_decay;
		j["absolute_decay"] = m_h.absolute_decay;
		j["clipping_magnitude"] = m_h.clipping_magnitude;
		j["non_matrix_learning_rate_factor"] = m_h.non_matrix_learning_rate_factor;
		j["optimize_matrix_params"] = m_h.optimize_matrix_params;
		j["optimize_non_matrix_params"] = m_h.optimize_non_matrix_params;
		return j;
	}

	uint32_t step_count() const override { return m_current_step; }
	const AdamHyper& hyper() const { return m_h; }
	float* first_moments() const { return m_first_moments.as<float>(); }
	float* second_moments() const { return m_second_moments.as<float>(); }
	size_t n_matrix() const { return m_n_matrix; }

private:
	AdamHyper m_h;
	size_t m_n_weights = 0, m_n_matrix = 0;
	DeviceBuf m_first_moments, m_second_moments, m_param_steps, m_debias;
	uint32_t m_debias_filled = 0;
	float m_debias_beta1 = -1.0f, m_debias_beta2 = -1.0f;
	uint32_t m_current_step = 0;
	bool m_steps16 = true; // width of m_param_steps, see ensure_step_width
};

// optimizers/sgd.h:44-150
class SgdOptimizer : public Optimizer {
public:
	explicit SgdOptimizer(const Json& params) { update_hyperparams(params); }
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>&) override { m_n_weights = n_weights; }
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override {
		++m_current_step;
		sgd_step(stream, m_n_weights, loss_scale, m_learning_rate, m_l2_reg, weights_full_precision, weights, gradients);
	}
	float learning_rate() const override { return m_learning_rate; }
	void set_learning_rate(float val) override { m_learning_rate = val; }
	uint32_t step_count() const override { return m_current_step; }
	size_t n_weights() const override { return m_n_weights; }
	void update_hyperparams(const Json& p) override {
		if (!p.is_object()) return;
		if (p.contains("learning_rate")) m_learning_rate = (float)p["learning_rate"].as_double();
		if (p.contains("l2_reg")) m_l2_reg = (float)p["l2_reg"].as_double();
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "SGD";
		j["learning_rate"] = m_learning_rate;
		j["l2_reg"] = m_l2_reg;
		return j;
	}
	Json serialize() const override {
		Json data = Json::object();
		data["current_step"] = Json((uint32_t)m_current_step);
		data["learning_rate"] = Json(m_learning_rate);
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		m_n_weights = n_weights;
		m_current_step = (uint32_t)data["current_step"].as_double();
		m_learning_rate = (float)data["learning_rate"].as_double();
	}
private:
	size_t m_n_weights = 0;
	uint32_t m_current_step = 0;
	float m_learning_rate = 1e-3f, m_l2_reg = 1e-8f; // sgd.h:151-152
};

// optimizers/exponential_decay.h:45-160: scales the nested optimizer's learning rate by decay_base every decay_interval steps
class ExponentialDecayOptimizer : public Optimizer {
public:
	explicit ExponentialDecayOptimizer(const Json& params) {
		m_nested = create_optimizer(params.value("nested", Json::object()));
		update_hyperparams(params);
		m_learning_rate_factor = 1.0f;
		m_base_learning_rate = m_nested->learning_rate();
	}
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override { m_nested->allocate(n_weights, layer_sizes); }
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override { // :60-71
		if (step_count() == 0) m_learning_rate_factor = 1.0f;
		if (step_count() >= m_decay_start && (step_count() - m_decay_start) % m_decay_interval == 0 && step_count() <= m_decay_end) m_learning_rate_factor *= m_decay_base;
		m_nested->set_learning_rate(m_base_learning_rate * m_learning_rate_factor);
		m_nested->step(stream, loss_scale, weights_full_precision, weights, gradients);
	}
	float learning_rate() const override { return m_base_learning_rate * m_learning_rate_factor; }
	void set_learning_rate(float val) override {
		m_base_learning_rate = val / m_learning_rate_factor;
		m_nested->set_learning_rate(m_base_learning_rate * m_learning_rate_factor);
	}
	uint32_t step_count() const override { return m_nested->step_count(); }
	size_t n_weights() const override { return m_nested->n_weights(); }
	void* custom_weights() const override { return m_nested->custom_weights(); }
	void update_hyperparams(const Json& p) override {
		if (!p.is_object()) return;
		if (p.contains("decay_base")) m_decay_base = (float)p["decay_base"].as_double();
		if (p.contains("decay_interval")) m_decay_interval = std::max(1u, (uint32_t)p["decay_interval"].as_double());
		if (p.contains("decay_start")) m_decay_start = (uint32_t)p["decay_start"].as_double();
		if (p.contains("decay_end")) m_decay_end = (uint32_t)p["decay_end"].as_double();
		if (p.contains("nested")) m_nested->update_hyperparams(p["nested"]);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "ExponentialDecay";
		j["nested"] = m_nested->hyperparams();
		j["decay_base"] = m_decay_base;
		j["decay_interval"] = m_decay_interval;
		j["decay_start"] = m_decay_start;
		j["decay_end"] = m_decay_end;
		return j;
	}
	Json serialize() const override {
		Json data = Json::object();
		data["nested"] = m_nested->serialize();
		data["learning_rate"] = Json(m_base_learning_rate);
		data["learning_rate_factor"] = Json(m_learning_rate_factor);
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		m_base_learning_rate = (float)data["learning_rate"].as_double();
		m_learning_rate_factor = data.value("learning_rate_factor", 1.0f);
		m_nested->deserialize(data["nested"], n_weights);
	}
private:
	std::unique_ptr<Optimizer> m_nested;
	float m_learning_rate_factor = 1.0f, m_base_learning_rate = 0.0f;
	float m_decay_base = 0.1f;
	uint32_t m_decay_interval = 10000, m_decay_start = 10000, m_decay_end = 10000000;
};

// optimizers/ema.h:44-230: debiased exponential moving average of the nested optimizer's weights, used for inference
class EmaOptimizer : public Optimizer {
public:
	explicit EmaOptimizer(const Json& params) {
		m_nested = create_optimizer(params.value("nested", Json::object()));
		update_hyperparams(params);
	}
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override {
		m_nested->allocate(n_weights, layer_sizes);
		if (n_weights * 2 <= m_weights_ema.bytes()) return;
		m_weights_ema.resize(n_weights * 2);
		m_weights_ema.memset(0);
		if (m_full_precision) {
			m_tmp.resize(n_weights * sizeof(float));
			m_tmp.memset(0);
		}
	}
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override { // :98-132
		m_nested->step(stream, loss_scale, weights_full_precision, weights, gradients);
		const uint32_t current_step = m_nested->step_count();
		const float ema_debias_old = 1 - (float)std::pow(m_ema_decay, current_step - 1);
		const float ema_debias_new = 1.0f / (1 - (float)std::pow(m_ema_decay, current_step));
		if (void* nested_custom = m_nested->custom_weights()) weights = nested_custom;
		ema_step(stream, n_weights(), m_ema_decay, ema_debias_old, ema_debias_new, weights, m_weights_ema.data(), m_full_precision ? m_tmp.as<float>() : nullptr);
	}
	float learning_rate() const override { return m_nested->learning_rate(); }
	void set_learning_rate(float val) override { m_nested->set_learning_rate(val); }
	uint32_t step_count() const override { return m_nested->step_count(); }
	size_t n_weights() const override { return m_nested->n_weights(); }
	void* custom_weights() const override { return m_weights_ema.data(); }
	void update_hyperparams(const Json& p) override {
		if (!p.is_object()) return;
		if (p.contains("decay")) m_ema_decay = (float)p["decay"].as_double();
		if (p.contains("full_precision")) m_full_precision = p["full_precision"].as_bool();
		if (p.contains("nested")) m_nested->update_hyperparams(p["nested"]);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "EMA";
		j["nested"] = m_nested->hyperparams();
		j["decay"] = m_ema_decay;
		j["full_precision"] = m_full_precision;
		return j;
	}
	Json serialize() const override {
		Json data = Json::object();
		data["nested"] = m_nested->serialize();
		data["weights_ema_binary"] = device_to_binary(m_weights_ema.data(), n_weights() * 2);
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		const std::vector<uint8_t> bytes = binary_of(data["weights_ema_binary"]);
		if (bytes.size() != n_weights * 2) throw std::runtime_error{"EMA: snapshot state has the wrong size."};
		m_weights_ema.resize(bytes.size());
		HIP_CHECK_THROW(hipMemcpy(m_weights_ema.data(), bytes.data(), bytes.size(), hipMemcpyHostToDevice));
		if (m_full_precision) {
			m_tmp.resize(n_weights * sizeof(float));
			cast_half_to_float(nullptr, n_weights, m_weights_ema.data(), m_tmp.as<float>());
			HIP_CHECK_THROW(hipDeviceSynchronize());
		}
		m_nested->deserialize(data["nested"], n_weights);
	}
private:
	float m_ema_decay = 0.99f;
	bool m_full_precision = false;
	std::unique_ptr<Optimizer> m_nested;
	DeviceBuf m_weights_ema, m_tmp;
};

// optimizers/novograd.h:96-261: layer-wise second moments.  Only the weight MATRICES are optimized (the layer list is all it walks,
// :132-166): parameters behind them -- grid entries -- are left alone, as in the reference.
class NovogradOptimizer : public Optimizer {
public:
	explicit NovogradOptimizer(const Json& params) { update_hyperparams(params); }
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override { // :103-124
		m_n_weights = n_weights;
		m_layers.clear();
		for (const auto& ls : layer_sizes) m_layers.push_back((size_t)ls.first * ls.second);
		m_first_moments.resize(0);
		m_first_moments.resize(n_weights * sizeof(float));
		m_first_moments.memset(0);
		m_per_layer_second_moments.resize(0);
		m_per_layer_second_moments.resize(m_layers.size() * sizeof(float));
		m_per_layer_second_moments.memset(0);
	}
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override { // :126-167
		++m_current_step;
		size_t offset = 0;
		for (size_t i = 0; i < m_layers.size(); ++i) {
			novograd_layer_step(stream, m_layers[i], m_relative_decay, m_absolute_decay, loss_scale, m_learning_rate, m_current_step == 1 ? 0.0f : m_beta1, // exact values on the first step
			                    m_current_step == 1 ? 0.0f : m_beta2, m_epsilon, weights_full_precision + offset, (char*)weights + 2 * offset, (const char*)gradients + 2 * offset,
			                    m_first_moments.as<float>() + offset, m_per_layer_second_moments.as<float>() + i);
			offset += m_layers[i];
		}
	}
	float learning_rate() const override { return m_learning_rate; }
	void set_learning_rate(float val) override { m_learning_rate = val; }
	uint32_t step_count() const override { return m_current_step; }
	size_t n_weights() const override { return m_n_weights; }
	void update_hyperparams(const Json& p) override {
		if (!p.is_object()) return;
		if (p.contains("beta1")) m_beta1 = (float)p["beta1"].as_double();
		if (p.contains("beta2")) m_beta2 = (float)p["beta2"].as_double();
		if (p.contains("epsilon")) m_epsilon = (float)p["epsilon"].as_double();
		if (p.contains("learning_rate")) m_learning_rate = (float)p["learning_rate"].as_double();
		if (p.contains("relative_decay")) m_relative_decay = (float)p["relative_decay"].as_double();
		if (p.contains("absolute_decay")) m_absolute_decay = (float)p["absolute_decay"].as_double();
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "Novograd";
		j["beta1"] = m_beta1;
		j["beta2"] = m_beta2;
		j["epsilon"] = m_epsilon;
		j["learning_rate"] = m_learning_rate;
		j["relative_decay"] = m_relative_decay;
		j["absolute_decay"] = m_absolute_decay;
		return j;
	}
	Json serialize() const override {
		Json data = Json::object();
		data["current_step"] = Json((uint32_t)m_current_step);
		data["base_learning_rate"] = Json((float)m_learning_rate);
		data["first_moments_binary"] = device_to_binary(m_first_moments.data(), m_n_weights * sizeof(float));
		data["per_layer_second_moments_binary"] = device_to_binary(m_per_layer_second_moments.data(), m_layers.size() * sizeof(float));
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		const std::vector<uint8_t> first = binary_of(data["first_moments_binary"]), second = binary_of(data["per_layer_second_moments_binary"]);
		if (first.size() != n_weights * sizeof(float) || second.size() != m_layers.size() * sizeof(float)) throw std::runtime_error{"Novograd: snapshot state has the wrong size."};
		m_n_weights = n_weights;
		m_first_moments.resize(first.size());
		m_per_layer_second_moments.resize(second.size());
		if (!first.empty()) HIP_CHECK_THROW(hipMemcpy(m_first_moments.data(), first.data(), first.size(), hipMemcpyHostToDevice));
		if (!second.empty()) HIP_CHECK_THROW(hipMemcpy(m_per_layer_second_moments.data(), second.data(), second.size(), hipMemcpyHostToDevice));
		m_current_step = (uint32_t)data["current_step"].as_double();
		m_learning_rate = (float)data["base_learning_rate"].as_double();
	}
private:
	size_t m_n_weights = 0;
	std::vector<size_t> m_layers;
	DeviceBuf m_first_moments, m_per_layer_second_moments;
	uint32_t m_current_step = 0;
	float m_learning_rate = 1e-3f, m_beta1 = 0.9f, m_beta2 = 0.999f, m_epsilon = 1e-8f, m_relative_decay = 0.0f, m_absolute_decay = 0.0f;
};

// optimizers/average.h:62-174: the mean of the weights after each of the last n_samples steps, as inference weights
class AverageOptimizer : public Optimizer {
public:
	explicit AverageOptimizer(const Json& params) {
		m_nested = create_optimizer(params.value("nested", Json::object()));
		update_hyperparams(params);
	}
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override { // :69-80
		m_n_weights = n_weights;
		m_layer_sizes = layer_sizes;
		m_allocated = true;
		m_nested->allocate(n_weights, layer_sizes);
		m_weights_samples.resize(0);
		m_weights_samples.resize(n_weights * m_n_samples * 2);
		m_weights_samples.memset(0);
		m_weights_average.resize(0);
		m_weights_average.resize(n_weights * 2);
		m_weights_average.memset(0);
	}
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override { // :82-92
		m_nested->step(stream, loss_scale, weights_full_precision, weights, gradients);
		char* current = (char*)m_weights_samples.data() + (size_t)(step_count() % m_n_samples) * m_n_weights * 2; // the slot of the step just taken
		average_step(stream, m_n_weights, m_n_samples, weights, current, m_weights_average.data());
	}
	float learning_rate() const override { return m_nested->learning_rate(); }
	void set_learning_rate(float val) override { m_nested->set_learning_rate(val); }
	uint32_t step_count() const override { return m_nested->step_count(); }
	size_t n_weights() const override { return m_nested->n_weights(); }
	void* custom_weights() const override { return m_weights_average.data(); }
	void update_hyperparams(const Json& p) override { // :131-142: a new window length starts the window over
		if (!p.is_object()) return;
		if (p.contains("n_samples")) {
			m_n_samples = (uint32_t)p["n_samples"].as_double();
			if (m_n_samples == 0) throw std::runtime_error{"AverageOptimizer: n_samples must be positive"};
			if (m_allocated) allocate(m_n_weights, m_layer_sizes);
		}
		if (p.contains("nested")) m_nested->update_hyperparams(p["nested"]);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "Average";
		j["nested"] = m_nested->hyperparams();
		j["n_samples"] = m_n_samples;
		return j;
	}
	Json serialize() const override {
		Json data = Json::object();
		data["nested"] = m_nested->serialize();
		data["weights_samples_binary"] = device_to_binary(m_weights_samples.data(), m_n_weights * m_n_samples * 2);
		data["weights_average_binary"] = device_to_binary(m_weights_average.data(), m_n_weights * 2);
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		const std::vector<uint8_t> samples = binary_of(data["weights_samples_binary"]), average = binary_of(data["weights_average_binary"]);
		if (average.size() != n_weights * 2 || samples.size() != n_weights * 2 * m_n_samples) throw std::runtime_error{"Average: snapshot state has the wrong size."};
		m_n_weights = n_weights;
		m_weights_samples.resize(samples.size());
		m_weights_average.resize(average.size());
		if (!samples.empty()) HIP_CHECK_THROW(hipMemcpy(m_weights_samples.data(), samples.data(), samples.size(), hipMemcpyHostToDevice));
		if (!average.empty()) HIP_CHECK_THROW(hipMemcpy(m_weights_average.data(), average.data(), average.size(), hipMemcpyHostToDevice));
		m_nested->deserialize(data["nested"], n_weights);
	}
private:
	uint32_t m_n_samples = 128;
	size_t m_n_weights = 0;
	bool m_allocated = false;
	std::vector<std::pair<uint32_t, uint32_t>> m_layer_sizes;
	std::unique_ptr<Optimizer> m_nested;
	DeviceBuf m_weights_samples, m_weights_average;
};

// optimizers/batched.h:63-162: the nested optimizer steps once per batch_size_multiplier calls, on the mean of their gradients
class BatchedOptimizer : public Optimizer {
public:
	explicit BatchedOptimizer(const Json& params) {
		m_nested = create_optimizer(params.value("nested", Json::object()));
		update_hyperparams(params);
	}
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override {
		m_nested->allocate(n_weights, layer_sizes);
		m_averaged_gradients.resize(n_weights * sizeof(float));
		m_averaged_gradients_half.resize(n_weights * 2);
		m_averaged_gradients.memset(0);
		m_averaged_gradients_half.memset(0);
	}
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override { // :77-89
		batched_accumulate(stream, n_weights(), m_current_step % m_batch_size_multiplier == 0, m_batch_size_multiplier, gradients, m_averaged_gradients.as<float>());
		++m_current_step;
		if (m_current_step % m_batch_size_multiplier == 0) {
			cast_float_to_half(stream, n_weights(), m_averaged_gradients.as<float>(), m_averaged_gradients_half.data());
			m_nested->step(stream, loss_scale, weights_full_precision, weights, m_averaged_gradients_half.data());
		}
	}
	float learning_rate() const override { return m_nested->learning_rate(); }
	void set_learning_rate(float val) override { m_nested->set_learning_rate(val); }
	uint32_t step_count() const override { return m_current_step; }
	size_t n_weights() const override { return m_nested->n_weights(); }
	void* custom_weights() const override { return m_nested->custom_weights(); }
	void weights_restored(hipStream_t stream, const void* weights) override { m_nested->weights_restored(stream, weights); }
	void update_hyperparams(const Json& p) override {
		if (!p.is_object()) return;
		if (p.contains("batch_size_multiplier")) {
			m_batch_size_multiplier = (uint32_t)p["batch_size_multiplier"].as_double();
			if (m_batch_size_multiplier == 0) throw std::runtime_error{"BatchedOptimizer: batch_size_multiplier must be positive"};
		}
		if (p.contains("nested")) m_nested->update_hyperparams(p["nested"]);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "Batched";
		j["nested"] = m_nested->hyperparams();
		j["batch_size_multiplier"] = m_batch_size_multiplier;
		return j;
	}
	Json serialize() const override {
		Json data = Json::object();
		data["nested"] = m_nested->serialize();
		data["averaged_gradients_binary"] = device_to_binary(m_averaged_gradients.data(), n_weights() * sizeof(float));
		data["averaged_gradients_half_binary"] = device_to_binary(m_averaged_gradients_half.data(), n_weights() * 2);
		data["current_step"] = Json((uint32_t)m_current_step);
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		const std::vector<uint8_t> pool = binary_of(data["averaged_gradients_binary"]), half = binary_of(data["averaged_gradients_half_binary"]);
		if (pool.size() != n_weights * sizeof(float) || half.size() != n_weights * 2) throw std::runtime_error{"Batched: snapshot state has the wrong size."};
		m_current_step = (uint32_t)data["current_step"].as_double();
		m_averaged_gradients.resize(pool.size());
		m_averaged_gradients_half.resize(half.size());
		if (!pool.empty()) HIP_CHECK_THROW(hipMemcpy(m_averaged_gradients.data(), pool.data(), pool.size(), hipMemcpyHostToDevice));
		if (!half.empty()) HIP_CHECK_THROW(hipMemcpy(m_averaged_gradients_half.data(), half.data(), half.size(), hipMemcpyHostToDevice));
		m_nested->deserialize(data["nested"], n_weights);
	}
private:
	uint32_t m_batch_size_multiplier = 16, m_current_step = 0;
	std::unique_ptr<Optimizer> m_nested;
	DeviceBuf m_averaged_gradients, m_averaged_gradients_half;
};

// optimizers/lookahead.h:61-168: every n_steps steps the slow weights move a fraction alpha towards the fast ones, which restart
// from them; the slow weights are the inference weights
class LookaheadOptimizer : public Optimizer {
public:
	explicit LookaheadOptimizer(const Json& params) {
		m_nested = create_optimizer(params.value("nested", Json::object()));
		update_hyperparams(params);
	}
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override {
		m_nested->allocate(n_weights, layer_sizes);
		if (n_weights * 2 <= m_weights_lookahead.bytes()) return;
		m_weights_lookahead.resize(n_weights * 2);
		m_weights_lookahead.memset(0);
	}
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override { // :78-98
		const uint32_t current_step = m_nested->step_count();
		if (current_step == 0) HIP_CHECK_THROW(hipMemcpyAsync(m_weights_lookahead.data(), weights, n_weights() * 2, hipMemcpyDeviceToDevice, stream));
		if (current_step % m_n_steps == 0) lookahead_step(stream, n_weights(), m_alpha, weights_full_precision, weights, m_weights_lookahead.data());
		m_nested->step(stream, loss_scale, weights_full_precision, weights, gradients);
	}
	float learning_rate() const override { return m_nested->learning_rate(); }
	void set_learning_rate(float val) override { m_nested->set_learning_rate(val); }
	uint32_t step_count() const override { return m_nested->step_count(); }
	size_t n_weights() const override { return m_nested->n_weights(); }
	void* custom_weights() const override { return m_weights_lookahead.data(); }
	void update_hyperparams(const Json& p) override {
		if (!p.is_object()) return;
		if (p.contains("alpha")) m_alpha = (float)p["alpha"].as_double();
		if (p.contains("n_steps")) {
			m_n_steps = (uint32_t)p["n_steps"].as_double();
			if (m_n_steps == 0) throw std::runtime_error{"LookaheadOptimizer: n_steps must be positive"};
		}
		if (p.contains("nested")) m_nested->update_hyperparams(p["nested"]);
	}
	Json hyperparams() const override {
		Json j = Json::object();
		j["otype"] = "Lookahead";
		j["nested"] = m_nested->hyperparams();
		j["alpha"] = m_alpha;
		j["n_steps"] = m_n_steps;
		return j;
	}
	Json serialize() const override {
		Json data = Json::object();
		data["nested"] = m_nested->serialize();
		data["weights_lookahead_binary"] = device_to_binary(m_weights_lookahead.data(), n_weights() * 2);
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		const std::vector<uint8_t> bytes = binary_of(data["weights_lookahead_binary"]);
		if (bytes.size() != n_weights * 2) throw std::runtime_error{"Lookahead: snapshot state has the wrong size."};
		m_weights_lookahead.resize(bytes.size());
		if (!bytes.empty()) HIP_CHECK_THROW(hipMemcpy(m_weights_lookahead.data(), bytes.data(), bytes.size(), hipMemcpyHostToDevice));
		m_nested->deserialize(data["nested"], n_weights);
	}
private:
	float m_alpha = 0.5f;
	uint32_t m_n_steps = 16;
	std::unique_ptr<Optimizer> m_nested;
	DeviceBuf m_weights_lookahead;
};

// optimizers/composite.h:44-74 (slice_weights) as it is meant: the layers that start at or after `offset`; a cut inside a layer
// is an error.  (The reference's loop advances the layer index before adding that layer's size: it skips layer 0 and runs one
// past the end of the list for any offset > 0 -- undefined behaviour there, the intended slice here.)
inline std::vector<std::pair<uint32_t, uint32_t>> slice_layer_sizes(const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes, size_t offset) {
	std::vector<std::pair<uint32_t, uint32_t>> out;
	size_t pos = 0;
	for (const auto& ls : layer_sizes) {
		const size_t size = (size_t)ls.first * ls.second;
		if (pos < offset && offset < pos + size) throw std::runtime_error{"Invalid slice. Can't slice within a layer."};
		if (pos >= offset) out.push_back(ls);
		pos += size;
	}
	return out;
}

// optimizers/composite.h:76-173: nested[i] optimizes the next "n_params_to_optimize" weights of the parameter vector
class CompositeOptimizer : public Optimizer {
public:
	explicit CompositeOptimizer(const Json& params) {
		if (!params.contains("nested") || !params["nested"].is_array() || params["nested"].size() == 0) {
			throw std::runtime_error{"Must provide an array of nested encodings to CompositeOptimizer."};
		}
		m_offsets.push_back(0);
		for (size_t i = 0; i < params["nested"].size(); ++i) {
			const Json& nested = params["nested"].at(i);
			m_offsets.push_back(m_offsets.back() + (size_t)nested.value("n_params_to_optimize", 0));
			m_nested.emplace_back(create_optimizer(nested));
			m_base_learning_rates.push_back(m_nested.back()->learning_rate());
		}
		update_hyperparams(params);
	}
	void allocate(size_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes) override {
		CHECK_THROW(n_weights >= m_offsets.back()); // weights past the last nested optimizer stay as they are
		m_need_custom_weights = false;
		for (size_t i = 0; i < m_nested.size(); ++i) {
			m_nested[i]->allocate(m_offsets[i + 1] - m_offsets[i], slice_layer_sizes(layer_sizes, m_offsets[i]));
			m_need_custom_weights |= m_nested[i]->custom_weights() != nullptr;
		}
		m_n_total = n_weights;
		if (m_need_custom_weights) { // sized for the whole vector: the trainer runs inference from it (the reference sizes it to the optimized part only)
			m_custom_weights.resize(n_weights * 2);
			m_custom_weights.memset(0);
		}
	}
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients) override {
		for (size_t i = 0; i < m_nested.size(); ++i) {
			const size_t offset = m_offsets[i];
			m_nested[i]->step(stream, loss_scale, weights_full_precision + offset, (_Float16*)weights + offset, (const _Float16*)gradients + offset);
		}
		weights_restored(stream, weights);
	}
	// composite.h:86-93: the custom weights are the nested optimizers' custom weights where they have any, else the weights
	void weights_restored(hipStream_t stream, const void* weights) override {
		if (!m_need_custom_weights) return;
		for (size_t i = 0; i < m_nested.size(); ++i) {
			const size_t offset = m_offsets[i];
			m_nested[i]->weights_restored(stream, (const _Float16*)weights + offset);
			const void* src = m_nested[i]->custom_weights() ? m_nested[i]->custom_weights() : (const void*)((const _Float16*)weights + offset);
			HIP_CHECK_THROW(hipMemcpyAsync((_Float16*)m_custom_weights.data() + offset, src, m_nested[i]->n_weights() * 2, hipMemcpyDeviceToDevice, stream));
		}
		if (m_n_total > m_offsets.back()) {
			const size_t done = m_offsets.back();
			HIP_CHECK_THROW(hipMemcpyAsync((_Float16*)m_custom_weights.data() + done, (const _Float16*)weights + done, (m_n_total - done) * 2, hipMemcpyDeviceToDevice, stream));
		}
	}
	float learning_rate() const override { return m_learning_rate_factor; }
	void set_learning_rate(float val) override {
		m_learning_rate_factor = val;
		for (size_t i = 0; i < m_nested.size(); ++i) m_nested[i]->set_learning_rate(m_base_learning_rates[i] * m_learning_rate_factor);
	}
	uint32_t step_count() const override { return m_nested[0]->step_count(); }
	size_t n_weights() const override { return m_offsets.back(); }
	void* custom_weights() const override { return m_need_custom_weights ? m_custom_weights.data() : nullptr; }
	void update_hyperparams(const Json& p) override {
		if (!p.is_object() || !p.contains("nested") || !p["nested"].is_array()) return;
		for (size_t i = 0; i < m_nested.size() && i < p["nested"].size(); ++i) m_nested[i]->update_hyperparams(p["nested"].at(i));
	}
	Json hyperparams() const override {
		Json nested = Json::array();
		for (const auto& n : m_nested) nested.push_back(n->hyperparams());
		Json j = Json::object();
		j["otype"] = "Composite";
		j["nested"] = nested;
		return j;
	}
	Json serialize() const override {
		Json nested = Json::array(), rates = Json::array();
		for (const auto& n : m_nested) nested.push_back(n->serialize());
		for (float r : m_base_learning_rates) rates.push_back(Json(r));
		Json data = Json::object();
		data["nested"] = nested;
		data["base_learning_rates"] = rates;
		data["learning_rate_factor"] = m_learning_rate_factor;
		return data;
	}
	void deserialize(const Json& data, size_t n_weights) override {
		CHECK_THROW(n_weights >= m_offsets.back());
		const Json& nested = data["nested"];
		for (size_t i = 0; i < m_nested.size(); ++i) m_nested[i]->deserialize(nested.at(i), m_offsets[i + 1] - m_offsets[i]);
		const Json& rates = data["base_learning_rates"];
		m_base_learning_rates.clear();
		for (size_t i = 0; i < rates.size(); ++i) m_base_learning_rates.push_back((float)rates.at(i).as_double());
		set_learning_rate((float)data["learning_rate_factor"].as_double());
	}
private:
	std::vector<std::unique_ptr<Optimizer>> m_nested;
	std::vector<float> m_base_learning_rates;
	std::vector<size_t> m_offsets;
	size_t m_n_total = 0;
	float m_learning_rate_factor = 1.0f;
	bool m_need_custom_weights = false;
	DeviceBuf m_custom_weights;
};

// src/optimizer.cu:50-82
inline std::unique_ptr<Optimizer> create_optimizer(const Json& params) {
	const std::string otype = params.value("otype", "Adam");
	if (equals_case_insensitive(otype, "Adam")) return std::unique_ptr<Optimizer>{new AdamOptimizer{params}};
	if (equals_case_insensitive(otype, "SGD")) return std::unique_ptr<Optimizer>{new SgdOptimizer{params}};
	if (equals_case_insensitive(otype, "ExponentialDecay")) return std::unique_ptr<Optimizer>{new ExponentialDecayOptimizer{params}};
	if (equals_case_insensitive(otype, "Ema")) return std::unique_ptr<Optimizer>{new EmaOptimizer{params}};
	if (equals_case_insensitive(otype, "Composite")) return std::unique_ptr<Optimizer>{new CompositeOptimizer{params}};
	if (equals_case_insensitive(otype, "Average")) return std::unique_ptr<Optimizer>{new AverageOptimizer{params}};
	if (equals_case_insensitive(otype, "Batched")) return std::unique_ptr<Optimizer>{new BatchedOptimizer{params}};
	if (equals_case_insensitive(otype, "Lookahead")) return std::unique_ptr<Optimizer>{new LookaheadOptimizer{params}};
	if (equals_case_insensitive(otype, "Novograd")) return std::unique_ptr<Optimizer>{new NovogradOptimizer{params}};
	throw std::runtime_error{"Invalid optimizer type: " + otype + " (this build provides Adam, SGD, ExponentialDecay, Ema, Composite, Average, Batched, Lookahead, Novograd)"};
}

// ------------------------------------------------------------------------------------------------------------------
// Trainer (trainer.h:48-363) + create_from_config (config.h:53-63)
// ------------------------------------------------------------------------------------------------------------------
struct TrainContext { // Trainer::ForwardContext, trainer.h:89-95
	ArenaBuf output;      // half  [n][padded_out]
	ArenaBuf dL_doutput;  // half  [n][padded_out]
	const void* dL_doutput_ptr = nullptr; // = dL_doutput or the caller's external_dL_dy
	ArenaBuf L;           // float [n][padded_out]
	std::unique_ptr<ModelContext> model_ctx;
	uint32_t n = 0;

	// Compact form (the register-resident fused MLP kernel, k_train_regs.hip): of the reference's [n][16] matrices dL_doutput and
	// L only the n_output_dims live columns are non-zero, and at 96 bytes per sample they would be more than a third of what that
	// kernel stores.  The step writes them dense as [n][dims]; the padded matrices are produced from those on first access
	// (tcnn_train_ctx_dL_doutput / _L), and loss() sums the compact L.
	bool compact = false;
	uint32_t dims = 0, padded_width = 0;
	hipStream_t stream = nullptr;
	ArenaBuf compact_dL_doutput; // half  [n][dims]
	ArenaBuf compact_L;          // float [n][dims]
	void materialize();
};

inline void TrainContext::materialize() {
	if (!compact || L) return;
	L = ArenaBuf{stream, (size_t)n * padded_width * sizeof(float)};
	dL_doutput = ArenaBuf{stream, (size_t)n * padded_width * 2};
	dL_doutput_ptr = dL_doutput.data();
	CHECK_THROW(padded_width == 16);
	mlp_expand_context(stream, n, dims, compact_dL_doutput.data(), compact_L.as<float>(), dL_doutput.data(), L.as<float>());
}

inline MatView make_view(const float* data, uint32_t width, uint32_t n, int layout) {
	return layout == 1 ? MatView{data, width, 1u} : MatView{data, 1u, n};
}
inline MatViewMut make_view_mut(float* data, uint32_t width, uint32_t n, int layout) {
	return layout == 1 ? MatViewMut{data, width, 1u} : MatViewMut{data, 1u, n};
}

class Trainer {
public:
	Trainer(uint32_t n_input_dims, uint32_t n_output_dims, const Json& config, uint32_t seed) {
		// config.h:53-63
		const Json encoding = config.value("encoding", Json::object());
		const Json loss = config.value("loss", Json::object());
		const Json optimizer = config.value("optimizer", Json::object());
		const Json network = config.value("network", Json::object());
		m_loss = create_loss(loss);
		m_optimizer = create_optimizer(optimizer);
		m_model.reset(new NetworkWithInputEncoding{n_input_dims, n_output_dims, encoding, network});
		// trainer.h:52-55
		std::seed_seq seq{seed};
		std::vector<uint32_t> seeds(2);
		seq.generate(std::begin(seeds), std::end(seeds));
		m_rng = Pcg32{seeds.front()};
		m_scalar.resize(sizeof(float) * 2048);
		initialize_params();
	}

	void initialize_params() { // trainer.h:68-87
		const size_t n = m_model->n_params();
		m_optimizer->allocate(n, m_model->layer_sizes());
		m_params_fp.resize(n * sizeof(float));
		m_params.resize(n * 2);
		m_grads.resize(n * 2);
		m_params_fp.memset(0);
		m_params.memset(0);
		m_grads.memset(0);
		m_model->initialize_params(m_rng, m_params_fp.as<float>(), 1.0f);
		cast_float_to_half(nullptr, n, m_params_fp.as<float>(), m_params.data());
		HIP_CHECK_THROW(hipDeviceSynchronize());
	}

	std::unique_ptr<TrainContext> forward(hipStream_t stream, float loss_scale, uint32_t n, MatView input, const float* target, const float* data_pdf,
	                                      bool use_inference_params, bool prepare_input_gradients, const void* external_dL_dy) { // trainer.h:97-141
		auto ctx = std::make_unique<TrainContext>();
		ctx->n = n;
		const uint32_t pw = m_model->padded_output_width();
		ctx->output = ArenaBuf{stream, (size_t)n * pw * 2};
		ctx->model_ctx = m_model->forward(stream, n, input, ctx->output.data(), use_inference_params ? params_inference() : m_params.data(), prepare_input_gradients);
		ctx->L = ArenaBuf{stream, (size_t)n * pw * sizeof(float)};
		if (external_dL_dy) {
			ctx->dL_doutput_ptr = external_dL_dy;
			// the reference leaves L uninitialised in this case; zero it so that loss() is defined
			HIP_CHECK_THROW(hipMemsetAsync(ctx->L.data(), 0, ctx->L.bytes(), stream));
		} else {
			CHECK_THROW(target != nullptr);
			ctx->dL_doutput = ArenaBuf{stream, (size_t)n * pw * 2};
			ctx->dL_doutput_ptr = ctx->dL_doutput.data();
			loss_evaluate(stream, m_loss, n, pw, m_model->output_width(), loss_scale, ctx->output.data(), target, ctx->L.as<float>(), ctx->dL_doutput.data(), data_pdf);
		}
		return ctx;
	}

	void backward(hipStream_t stream, const TrainContext& ctx, uint32_t n, MatView input, MatViewMut* dL_dinput, bool use_inference_params, GradientMode mode) { // trainer.h:147-149
		const_cast<TrainContext&>(ctx).materialize(); // a context of the fused step keeps dL_doutput compact until somebody asks for it
		m_model->backward(stream, *ctx.model_ctx, n, input, ctx.output.data(), ctx.dL_doutput_ptr, dL_dinput, use_inference_params ? params_inference() : m_params.data(), m_grads.data(), mode);
	}

	size_t params_updated_in_flush() const { return m_params_updated_in_flush; }
	size_t m_prologue_steps = 0;
	// TCNN_AMD_ADAM_IN_FLUSH=1: the optimizer's update is applied by the gradient kernels where they can carry it.  Off by default:
	// bit-identical and measured equal in time on C3a (0.229 / 0.230 vs 0.230 / 0.225 ms per step; DESIGN.md "Adam in the scatter").
	static bool adam_in_flush_enabled() { return switches().adam_in_flush; }
	// TCNN_AMD_ADAM_IN_REDUCE=0: models without encoding parameters (BASELINE config 2) run the optimizer as a launch of its own again
	// instead of behind the weight gradients' slab reduction (k_wgrad_reduce_adam; bit-identical, one ~4 us launch less per step)
	static bool adam_in_reduce_enabled() { return switches().adam_in_reduce; }

	// the finalize pass and the slab reduction an AdamPrologue holds, as the launches they were
	void run_prologue_alone(hipStream_t stream, const AdamPrologue& p) {
		MlpReduceJob job;
		if (p.has_reduce) {
			job.n_elems = p.reduce_elems;
			job.n_slabs = p.reduce_slabs;
			job.slabs = p.slabs;
			job.grad = m_grads.data();
			job.accumulate = p.reduce_accumulate;
		}
		if (!p.ranges.empty()) grid_scatter_finalize(stream, p.dev_ranges, (uint32_t)p.ranges.size(), p.scratch, p.grad_base, p.accumulate, p.has_reduce ? &job : nullptr);
		else if (p.has_reduce) mlp_reduce_slabs(stream, p.reduce_elems, p.reduce_slabs, p.slabs, m_grads.data(), p.reduce_accumulate != 0);
	}

	void optimizer_step(hipStream_t stream, float loss_scale) { // trainer.h:155-157
		m_model->invalidate_live_image();
		m_optimizer->step(stream, loss_scale, m_params_fp.as<float>(), m_params.data(), m_grads.data());
	}

	std::unique_ptr<TrainContext> training_step(hipStream_t stream, uint32_t n, MatView input, const float* target, const float* data_pdf, bool run_optimizer,
	                                            MatViewMut* dL_dinput, bool use_inference_params, GradientMode mode, const void* external_dL_dy) { // trainer.h:163-190
		const float loss_scale = LOSS_SCALE_FP16;
		std::unique_ptr<TrainContext> ctx;
		const bool other_weights = use_inference_params && params_inference() != m_params.data(); // EMA weights requested for this step
		if (m_model->fused_step_supported(n) && (external_dL_dy || loss_in_fused_kernel(m_loss)) && !other_weights) {
			// MI355X path: encoding -> ONE fused MLP kernel (forward + loss + backward + weight gradients) -> grid scatter
			ctx = std::make_unique<TrainContext>();
			ctx->n = n;
			const uint32_t pw = m_model->padded_output_width();
			ctx->output = ArenaBuf{stream, (size_t)n * pw * 2};
			ctx->compact = !external_dL_dy && mode != GradientMode::Ignore && m_model->fused_compact_context_supported(n);
			if (ctx->compact) {
				CHECK_THROW(target != nullptr);
				ctx->dims = m_model->output_width();
				ctx->padded_width = pw;
				ctx->stream = stream;
				ctx->compact_dL_doutput = ArenaBuf{stream, (size_t)n * ctx->dims * 2};
				ctx->compact_L = ArenaBuf{stream, (size_t)n * ctx->dims * sizeof(float)};
			} else {
				ctx->L = ArenaBuf{stream, (size_t)n * pw * sizeof(float)};
				if (external_dL_dy) {
					ctx->dL_doutput_ptr = external_dL_dy;
					HIP_CHECK_THROW(hipMemsetAsync(ctx->L.data(), 0, ctx->L.bytes(), stream));
				} else {
					CHECK_THROW(target != nullptr);
					ctx->dL_doutput = ArenaBuf{stream, (size_t)n * pw * 2};
					ctx->dL_doutput_ptr = ctx->dL_doutput.data();
				}
			}
			m_params_updated_in_flush = 0;
			m_profile.begin_step();
			// TCNN_AMD_ADAM_IN_FLUSH=1: the optimizer's update rides on the gradient kernels where they can carry it (k_grid_scatter:
			// the owner of a chunk updates its parameters as it flushes); what they did not take is done afterwards.
			AdamInFlush adam;
			ParamRanges adam_done;
			const bool split = run_optimizer && mode == GradientMode::Overwrite && (adam_in_flush_enabled() || (adam_in_reduce_enabled() && m_model->optimizer_rides_on_reduce())) &&
			                   m_optimizer->begin_split_step(stream, loss_scale, m_params_fp.as<float>(), m_params.data(), adam);
			if (m_params_exposed) m_model->invalidate_live_image(); // somebody holds a pointer to the parameters: no image outlives a step
			// the optimizer's launch offers to finish the backward pass's gradients itself (AdamPrologue: the scatter's finalize pass and the
			// MLP's slab reduction as the prologue of k_adam's launch -- one launch and one kernel boundary less per step)
			AdamPrologue prologue;
			prologue.offered = run_optimizer && !split && mode != GradientMode::Ignore && switches().adam_prologue && m_optimizer->takes_prologue();
			ctx->model_ctx = m_model->fused_step(stream, n, input, target, data_pdf, external_dL_dy, m_loss, loss_scale, ctx->output.data(),
			                                     ctx->compact ? ctx->compact_dL_doutput.data() : ctx->dL_doutput.data(),
			                                     ctx->compact ? ctx->compact_L.as<float>() : ctx->L.as<float>(), ctx->compact, dL_dinput, m_params.data(), m_grads.data(), mode,
			                                     &m_profile, split ? &adam : nullptr, split ? &adam_done : nullptr, prologue.offered ? &prologue : nullptr);
			if (run_optimizer) {
				m_profile.mark(stream, StepProfile::Optimizer, false);
				m_params_updated_in_flush = 0;
				for (const auto& r : adam_done) m_params_updated_in_flush += r.second - r.first;
				if (split) {
					if (!m_model->live_image_kept()) m_model->invalidate_live_image(); // the update below (or the gradient kernels') changes network weights
					m_optimizer->finish_split_step(stream, loss_scale, m_params_fp.as<float>(), m_params.data(), m_grads.data(), adam_done);
				} else if (prologue.pending) {
					m_model->invalidate_live_image();
					if (m_optimizer->step_with_prologue(stream, loss_scale, m_params_fp.as<float>(), m_params.data(), m_grads.data(), prologue)) ++m_prologue_steps;
					else {
						run_prologue_alone(stream, prologue); // shapes the fused launch does not take: the two launches of before
						m_optimizer->step(stream, loss_scale, m_params_fp.as<float>(), m_params.data(), m_grads.data());
					}
				} else optimizer_step(stream, loss_scale);
				m_profile.mark(stream, StepProfile::Optimizer, true);
			} else if (prologue.pending) run_prologue_alone(stream, prologue); // (never: nothing is offered without an optimizer step)
			m_profile.end_step();
			return ctx;
		} else {
			ctx = forward(stream, loss_scale, n, input, target, data_pdf, use_inference_params, dL_dinput != nullptr, external_dL_dy);
			backward(stream, *ctx, n, input, dL_dinput, use_inference_params, mode);
		}
		m_params_updated_in_flush = 0;
		if (run_optimizer) optimizer_step(stream, loss_scale);
		return ctx;
	}

	float loss(hipStream_t stream, const TrainContext& ctx) { // trainer.h:205-207 + reduce_sum.h:140-151
		float* partials = m_scalar.as<float>();
		float* result = partials + 1024;
		if (ctx.compact && !ctx.L) reduce_sum(stream, (size_t)ctx.n * ctx.dims, ctx.compact_L.as<float>(), partials, result); // the padding columns are zeros
		else reduce_sum(stream, (size_t)ctx.n * m_model->padded_output_width(), ctx.L.as<float>(), partials, result);
		float host = 0;
		HIP_CHECK_THROW(hipMemcpyAsync(&host, result, sizeof(float), hipMemcpyDeviceToHost, stream));
		HIP_CHECK_THROW(hipStreamSynchronize(stream));
		return host;
	}

	void inference(hipStream_t stream, uint32_t n, MatView input, MatViewMut output, bool use_inference_params) { // object.h:147-176
		Model::check_batch(n);
		if (n == 0) return;
		m_model->inference_f32(stream, n, input, output, use_inference_params ? params_inference() : m_params.data());
	}

	// object.h:133-145, inference_mixed_precision: the network's own output, half [n][padded_output_width] (what inference() casts)
	void inference_mixed_precision(hipStream_t stream, uint32_t n, MatView input, void* output_half, bool use_inference_params) {
		Model::check_batch(n);
		if (n == 0) return;
		m_model->inference(stream, n, input, output_half, use_inference_params ? params_inference() : m_params.data());
	}

	void set_params_full_precision(const float* params, size_t n_params, bool device_ptr) { // trainer.h:242-254
		if (n_params != m_model->n_params()) throw std::runtime_error{"Can't set fp params because buffer has the wrong size."};
		m_model->invalidate_live_image();
		HIP_CHECK_THROW(hipMemcpy(m_params_fp.data(), params, sizeof(float) * n_params, device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
		cast_float_to_half(nullptr, n_params, m_params_fp.as<float>(), m_params.data());
		if (params_inference() != m_params.data()) HIP_CHECK_THROW(hipMemcpy(params_inference(), m_params.data(), 2 * n_params, hipMemcpyDeviceToDevice));
		HIP_CHECK_THROW(hipDeviceSynchronize());
	}

	void set_params(const void* params, size_t n_params, bool device_ptr) { // trainer.h:256-269
		if (n_params != m_model->n_params()) throw std::runtime_error{"Can't set params because buffer has the wrong size."};
		m_model->invalidate_live_image();
		HIP_CHECK_THROW(hipMemcpy(m_params.data(), params, 2 * n_params, device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
		cast_half_to_float(nullptr, n_params, m_params.data(), m_params_fp.as<float>());
		if (params_inference() != m_params.data()) HIP_CHECK_THROW(hipMemcpy(params_inference(), m_params.data(), 2 * n_params, hipMemcpyDeviceToDevice));
		HIP_CHECK_THROW(hipDeviceSynchronize());
	}

	void update_hyperparams(const Json& params) { // trainer.h:213-216
		m_optimizer->update_hyperparams(params.value("optimizer", Json::object()));
	}

	// ---- snapshot wire format, trainer.h:275-315 (+ adam.h:278-299, gpu_memory_json.h:36-71): an object holding the half
	// parameters as a binary blob and, optionally, the optimizer's state.  Callers store it as MessagePack (Json::to_msgpack).
	Json serialize(bool serialize_optimizer) {
		HIP_CHECK_THROW(hipDeviceSynchronize());
		const size_t n = m_model->n_params();
		Json data = Json::object();
		data["n_params"] = Json((uint64_t)n);
		data["params_type"] = "__half"; // type_to_string<__half>() of the reference: the name snapshots carry
		data["params_binary"] = device_to_binary(params_inference(), 2 * n); // trainer.h:281: the inference parameters (EMA weights if there are any)
		if (serialize_optimizer) data["optimizer"] = m_optimizer->serialize();
		return data;
	}

	void deserialize(const Json& data) {
		const std::string type = data.value("params_type", "__half");
		const std::vector<uint8_t> bytes = binary_of(data["params_binary"]);
		if (type == "float") {
			set_params_full_precision((const float*)bytes.data(), bytes.size() / sizeof(float), false);
		} else if (type == "__half") {
			set_params(bytes.data(), bytes.size() / 2, false);
		} else {
			throw std::runtime_error{"Trainer: snapshot parameters must be of type float of __half"};
		}
		if (data.contains("optimizer")) {
			m_optimizer->deserialize(data["optimizer"], m_model->n_params());
			m_optimizer->weights_restored(nullptr, m_params.data());
		}
		HIP_CHECK_THROW(hipDeviceSynchronize());
	}

	Json hyperparams() const { // trainer.h:218-224
		Json j = Json::object();
		j["otype"] = "Trainer";
		j["optimizer"] = m_optimizer->hyperparams();
		Json l = Json::object();
		l["otype"] = to_string(m_loss);
		j["loss"] = l;
		return j;
	}

	NetworkWithInputEncoding& model() { return *m_model; }
	Optimizer& optimizer() { return *m_optimizer; }
	StepProfile& profile() { return m_profile; }
	// trainer.h:329-333: the optimizer's own weights (EMA) if it keeps any, else the training parameters
	void* params_inference() const { void* custom = m_optimizer->custom_weights(); return custom ? custom : m_params.data(); }
	size_t n_params() const { return m_model->n_params(); }
	// The reference hands out mutable pointers here (trainer.h:226-232).  A caller may write through them at any later time, so from the
	// first call on no fragment image of the half parameters is trusted beyond the step that built it (Network::live_image)
	float* params_full_precision() { expose_params(); return m_params_fp.as<float>(); }
	void* params() { expose_params(); return m_params.data(); }
	const void* params_unexposed() const { return m_params.data(); } // for comparisons only
	size_t image_preps() const { return m_model->image_preps(); }
	uint64_t scatter_wide_fallbacks() { return m_model->scatter_wide_fallbacks(); }
	uint64_t list_scatters() const { return m_model->list_scatters(); }
	size_t prologue_steps() const { return m_prologue_steps; }
	void* param_gradients() const { return m_grads.data(); }

private:
	std::unique_ptr<NetworkWithInputEncoding> m_model;
	std::unique_ptr<Optimizer> m_optimizer;
	LossType m_loss;
	Pcg32 m_rng;
	DeviceBuf m_params_fp, m_params, m_grads, m_scalar;
	StepProfile m_profile;
	size_t m_params_updated_in_flush = 0;
	bool m_params_exposed = false;
	void expose_params() {
		m_params_exposed = true;
		m_model->invalidate_live_image();
	}
};

} // namespace tcnn_amd
