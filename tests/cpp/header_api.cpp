// header_api.cpp -- a caller written against the reference's C++ header API (the shape of samples/mlp_learning_an_image.cu:
// config literal -> create_from_config -> training loop -> inference), compiled with plain g++ and linked with -ltcnn_amd.
//   header_api --no-gpu : host-only checks (JSON literals, factories' error reporting, plugin interface metadata)
//   header_api          : trains y = f(x) for a few steps on the GPU and checks that the loss falls and inference agrees
#include <tiny-cuda-nn/config.h>
#include <tiny-cuda-nn/cpp_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

using namespace tcnn;

#define REQUIRE(x) do { if (!(x)) { std::printf("FAILED: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

static int host_checks() {
	json config = {
		{"loss", {{"otype", "RelativeL2"}}},
		{"optimizer", {{"otype", "Adam"}, {"learning_rate", 1e-2}, {"beta1", 0.9f}, {"beta2", 0.99f}, {"epsilon", 1e-15}, {"l2_reg", 1e-6}}},
		{"encoding", {{"otype", "HashGrid"}, {"n_levels", 16}, {"n_features_per_level", 2}, {"log2_hashmap_size", 15}, {"base_resolution", 16}, {"per_level_scale", 1.5}}},
		{"network", {{"otype", "FullyFusedMLP"}, {"activation", "ReLU"}, {"output_activation", "None"}, {"n_neurons", 64}, {"n_hidden_layers", 2}}},
	};
	REQUIRE(config.is_object() && config["encoding"].is_object());
	REQUIRE(config["encoding"].value("n_levels", 0u) == 16u);
	REQUIRE(config.value("network", json::object()).value("otype", "") == "FullyFusedMLP");
	const json reparsed = json::parse(config.dump());
	REQUIRE(reparsed["optimizer"].value("beta1", 0.0f) == 0.9f && reparsed["optimizer"].value("learning_rate", 0.0) == 1e-2);
	REQUIRE(reparsed["encoding"].value("per_level_scale", 0.0f) == 1.5f && reparsed["network"].value("activation", "") == "ReLU");

	REQUIRE(cpp::batch_size_granularity() == BATCH_SIZE_GRANULARITY);
	REQUIRE(cpp::preferred_precision() == cpp::Precision::Fp16);
	REQUIRE(cpp::default_loss_scale(cpp::Precision::Fp16) == 128.0f);
	std::unique_ptr<cpp::Module> m{cpp::create_network_with_input_encoding(2, 3, config["encoding"], config["network"])};
	REQUIRE(m->n_input_dims() == 2 && m->n_output_dims() == 16);
	REQUIRE(m->n_params() == 708368u + 7168u);
	REQUIRE(m->hyperparams()["encoding"].value("otype", "") == "Grid");
	bool threw = false;
	try { std::unique_ptr<cpp::Module> bad{cpp::create_network(32, 3, json{{"otype", "NoSuchNetwork"}})}; } catch (const std::runtime_error& e) { threw = std::strstr(e.what(), "Invalid network type") != nullptr; }
	REQUIRE(threw);
	threw = false;
	try { Loss<network_precision_t> l{json{{"otype", "Huber"}}}; } catch (const std::runtime_error&) { threw = true; }
	REQUIRE(threw);
	// cpp_api.h:80 set_log_callback: the library's messages (here: the error it reports for an invalid network) reach the callback
	{
		std::string seen;
		cpp::set_log_callback([&](cpp::LogSeverity severity, const std::string& msg) { if (severity == cpp::LogSeverity::Error) seen = msg; });
		try { std::unique_ptr<cpp::Module> bad{cpp::create_network(32, 3, json{{"otype", "NoSuchNetwork"}})}; } catch (const std::runtime_error&) {}
		cpp::set_log_callback(nullptr);
		REQUIRE(seen.find("Invalid network type") != std::string::npos);
	}

	// <tiny-cuda-nn/random.h>: the stream of default_rng_t{1337} (tests/golden/reference_kat.json "pcg32", computed from the
	// reference's own dependencies/pcg32 at survey time), advance(), comments in configs, pretty-printed dump
	default_rng_t rng{1337};
	REQUIRE(rng.state_inc[1] == 3u);
	const float want[4] = {0.147699356f, 0.471029401f, 0.173984647f, 0.129117727f};
	for (float w : want) REQUIRE(rng.next_float() == w);
	default_rng_t jump{1337};
	jump.advance(2);
	REQUIRE(jump.next_float() == 0.173984647f);
#ifdef EXPECT_NLOHMANN_JSON // the real nlohmann::json (tcnn_api.h found <json/json.hpp>): comments are an argument of its parser, as the reference's callers pass it
	static_assert(std::is_same<json, nlohmann::json>::value, "tcnn::json should be nlohmann::json in this build");
	const json commented = json::parse("{ // line comment\n \"a\": 1, /* block */ \"b\": [1, 2] }", nullptr, true, true);
#else
	const json commented = json::parse("{ // line comment\n \"a\": 1, /* block */ \"b\": [1, 2] }");
#endif
	REQUIRE(commented.value("a", 0) == 1 && json::parse(commented.dump(4)).dump() == commented.dump());

	// MessagePack as nlohmann::json::to_msgpack writes it: sorted keys, smallest integer formats, float32 where exact, bin8/16/32
	{
		json j = json::object();
		j["b"] = json{true, json(), 1.5, "x"};
		j["a"] = 1;
		const std::vector<uint8_t> want = {0x82, 0xa1, 0x61, 0x01, 0xa1, 0x62, 0x94, 0xc3, 0xc0, 0xca, 0x3f, 0xc0, 0x00, 0x00, 0xa1, 0x78};
		REQUIRE(json::to_msgpack(j) == want);
		json k = json::object();
		k["n_params"] = (uint64_t)70000;
		k["lr"] = 0.1; // not exact as float32 -> float64
		k["neg"] = -33;
		k["blob"] = json::binary(std::vector<uint8_t>(300, 0xab));
		const std::vector<uint8_t> bytes = json::to_msgpack(k);
		const std::vector<uint8_t> head = {0x84, 0xa4, 'b', 'l', 'o', 'b', 0xc5, 0x01, 0x2c}; // map of 4, "blob" first, bin16 of 300
		REQUIRE(std::equal(head.begin(), head.end(), bytes.begin()));
		const json back = json::from_msgpack(bytes);
		REQUIRE(back["blob"].is_binary() && back["blob"].get_binary().size() == 300 && back["blob"].get_binary()[299] == 0xab);
		REQUIRE(back.value("n_params", 0u) == 70000u && back.value("lr", 0.0) == 0.1 && back.value("neg", 0) == -33);
		REQUIRE(json::to_msgpack(back) == bytes);
	}
	std::printf("host checks ok\n");
	return 0;
}

static int gpu_checks() {
	json config = {
		{"loss", {{"otype", "RelativeL2"}}},
		{"optimizer", {{"otype", "Adam"}, {"learning_rate", 1e-2}, {"beta1", 0.9f}, {"beta2", 0.99f}, {"epsilon", 1e-15}, {"l2_reg", 1e-6}}},
		{"encoding", {{"otype", "HashGrid"}, {"n_levels", 16}, {"n_features_per_level", 2}, {"log2_hashmap_size", 15}, {"base_resolution", 16}, {"per_level_scale", 1.5}}},
		{"network", {{"otype", "FullyFusedMLP"}, {"activation", "ReLU"}, {"output_activation", "None"}, {"n_neurons", 64}, {"n_hidden_layers", 2}}},
	};
	const uint32_t n_input_dims = 2, n_output_dims = 3, batch_size = 1 << 14;
	auto model = create_from_config(n_input_dims, n_output_dims, config);
	auto trainer = model.trainer;
	auto network = model.network;
	REQUIRE(trainer->n_params() == 708368u + 7168u);
	REQUIRE(network->padded_output_width() == 16);

	// a smooth target on [0,1)^2
	std::vector<float> xs((size_t)batch_size * 2), ts((size_t)batch_size * 3);
	uint32_t state = 12345;
	auto rnd = [&] { state = state * 1664525u + 1013904223u; return (state >> 8) * (1.0f / 16777216.0f); };
	for (uint32_t i = 0; i < batch_size; ++i) {
		const float x = rnd(), y = rnd();
		xs[2 * i] = x; xs[2 * i + 1] = y;
		ts[3 * i] = 0.5f + 0.5f * std::sin(6 * x); ts[3 * i + 1] = x * y; ts[3 * i + 2] = 0.5f + 0.5f * std::cos(4 * y);
	}
	GPUMatrix<float> training_batch(n_input_dims, batch_size), training_target(n_output_dims, batch_size), prediction(n_output_dims, batch_size);
	GPUMemory<float> staging(xs.size());
	staging.copy_from_host(xs);
	tcnn_gpu_memcpy(training_batch.data(), staging.data(), staging.get_bytes(), TCNN_MEMCPY_DEVICE_TO_DEVICE);
	tcnn_gpu_memcpy(training_target.data(), ts.data(), ts.size() * sizeof(float), TCNN_MEMCPY_HOST_TO_DEVICE);

	float first = 0, last = 0;
	for (uint32_t i = 0; i < 100; ++i) {
		auto ctx = trainer->training_step(nullptr, training_batch, training_target);
		if (i == 0 || i == 99) (i == 0 ? first : last) = trainer->loss(nullptr, *ctx);
	}
	std::printf("loss %g -> %g after 100 steps\n", first, last);
	REQUIRE(std::isfinite(last) && last < 0.2f * first);
	REQUIRE(trainer->optimizer_step_count() == 100);

	network->inference(nullptr, training_batch, prediction);
	tcnn_stream_synchronize(nullptr);
	const std::vector<float> p = prediction.to_cpu_vector();
	double err = 0;
	for (size_t i = 0; i < p.size(); ++i) err += std::fabs(p[i] - ts[i]);
	err /= p.size();
	std::printf("mean abs error of inference vs target: %g\n", err);
	REQUIRE(err < 0.1);

	// trainer.h:275-315: snapshot object with binary values; a second trainer restored from it infers the same
	{
		const json snap = trainer->serialize(true);
		REQUIRE(snap.value("n_params", 0u) == 708368u + 7168u && snap.value("params_type", "") == "__half");
		REQUIRE(snap["params_binary"].is_binary() && snap["params_binary"].get_binary().size() == 2 * (708368u + 7168u));
		REQUIRE(snap["optimizer"].value("current_step", 0u) == 100u && snap["optimizer"]["first_moments_binary"].get_binary().size() == 4 * (708368u + 7168u));
		auto other = create_from_config(n_input_dims, n_output_dims, config);
		other.trainer->deserialize(json::from_msgpack(json::to_msgpack(snap)));
		GPUMatrix<float> again(n_output_dims, batch_size);
		other.network->inference(nullptr, training_batch, again);
		tcnn_stream_synchronize(nullptr);
		REQUIRE(again.to_cpu_vector() == p);
		REQUIRE(other.trainer->optimizer_step_count() == 100);
	}

	// row-major (SoA) matrices go through the same entry points
	GPUMatrix<float, RM> soa_in(n_input_dims, 256), soa_out(n_output_dims, 256);
	std::vector<float> soa(512);
	for (uint32_t i = 0; i < 256; ++i) { soa[i] = xs[2 * i]; soa[256 + i] = xs[2 * i + 1]; }
	tcnn_gpu_memcpy(soa_in.data(), soa.data(), soa.size() * sizeof(float), TCNN_MEMCPY_HOST_TO_DEVICE);
	network->inference(nullptr, soa_in, soa_out);
	const std::vector<float> q = soa_out.to_cpu_vector();
	for (uint32_t i = 0; i < 256; ++i)
		for (uint32_t j = 0; j < 3; ++j) REQUIRE(q[j * 256 + i] == p[3 * i + j]);

	// generate_random_uniform (random.h:58-70): element i + n_threads * j is draw j of a copy of the generator advanced by 4 i;
	// afterwards the caller's generator has moved on by n
	{
		const size_t n = 1000; // n_threads = 250
		GPUMemory<float> dev(n);
		default_rng_t gen{1337}, host{1337};
		generate_random_uniform<float>(nullptr, gen, n, dev.data());
		tcnn_stream_synchronize(nullptr);
		std::vector<float> got;
		dev.copy_to_host(got);
		const size_t n_threads = (n + 3) / 4;
		// the reference launches whole blocks of 128 threads: n_threads rounds up to a multiple of 128 for the stride
		const size_t stride = (n_threads + 127) / 128 * 128;
		for (size_t i = 0; i < 5; ++i) {
			default_rng_t t{1337};
			t.advance((int64_t)(4 * i));
			for (size_t j = 0; j < 4; ++j) if (i + stride * j < n) REQUIRE(got[i + stride * j] == t.next_float());
		}
		host.advance((int64_t)n);
		REQUIRE(gen.state_inc[0] == host.state_inc[0]);
	}
	free_all_gpu_memory_arenas();
	std::printf("gpu checks ok\n");
	return 0;
}

int main(int argc, char** argv) {
	try {
		if (host_checks()) return 1;
		if (argc > 1 && std::strcmp(argv[1], "--no-gpu") == 0) return 0;
		return gpu_checks();
	} catch (const std::exception& e) {
		std::printf("exception: %s\n", e.what());
		return 2;
	}
}
