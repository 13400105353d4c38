// mlp_learning_an_image.hip -- the caller harness of the hot path: learn a 2-D image with a NetworkWithInputEncoding, the way
// the reference's sample and image benchmark drive the library, written against include/tiny-cuda-nn/ (libtcnn_amd.so).
//
// Mirrors the PROTOCOLS of (reference, /root/reference):
//   samples/mlp_learning_an_image.cu:84-99,154-300   training loop: random batch -> image lookup -> trainer->training_step,
//                                                    loss every 10/100/1000 steps, periodic full-resolution inference dumps
//   benchmarks/image/bench_ours.cu:92-115,188-335    batch sizes 2^21 .. 2^14, max(1000 * 2^18 / batch, 250) steps, mean training
//                                                    throughput over the second half, then 5x as many inference steps, results
//                                                    in bench_result_ours.json ({"fully_fused": [{"batch_size", "training_throughput",
//                                                    "inference_throughput"}, ...]})
// What is ours: the image lookup.  The reference samples the image through a CUDA texture object (tex2D<float4>, normalized
// coordinates, linear filter, clamp addressing); here k_eval_image does the same arithmetic in a plain HIP kernel:
//   xB = x * W - 0.5, i = floor(xB), alpha = frac(xB) rounded to 8 fractional bits (the texture unit's 9-bit fixed-point weights),
//   value = (1-a)(1-b) T[i][j] + a(1-b) T[i+1][j] + (1-a) b T[i][j+1] + a b T[i+1][j+1], indices clamped to the image.
// Image files: JPEG (baseline and progressive, samples/jpeg_decoder.h -- the reference's data/images/albert.jpg loads as it is) and
// binary PPM (P6) / PGM (P5), 8 bit, all turned into what stbi_loadf hands the reference's samples (value = (v / 255)^2.2, RGBA
// floats; stbi_wrapper.cpp:37-44); "synthetic[:WxH]" renders a deterministic multi-scale test image.  Outputs are written as PPM.
//
// usage:  mlp_learning_an_image <image.jpg|image.ppm|image.pgm|synthetic[:WxH]> [config.json] [n_training_steps] [final_image.ppm]
//         mlp_learning_an_image --bench <image> <config.json> [result.json] [--batches 18,16] [--cooldown SECONDS]
//         mlp_learning_an_image --sample <image> <coords.f32> <out.f32>      (k_eval_image on given coordinates; used by the tests)
#include <tiny-cuda-nn/config.h>
#include <tiny-cuda-nn/random.h>

#include "jpeg_decoder.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <iterator>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

using namespace tcnn;
using precision_t = network_precision_t;

#define HIP_CHECK_THROW(x) \
	do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error{std::string{#x " failed: "} + hipGetErrorString(e_)}; } while (0)

static uint32_t next_multiple(uint32_t v, uint32_t d) { return (v + d - 1) / d * d; }

// ---------------------------------------------------------------------------------------------------------------- kernels
// [n][stride] floats out: RGB from the bilinear lookup, remaining channels 1 (mlp_learning_an_image.cu:84-99)
template <uint32_t STRIDE>
__global__ void __launch_bounds__(256) k_eval_image(const uint32_t n, const float4* __restrict__ image, const int width, const int height, const bool filter,
                                                    const float* __restrict__ xs_and_ys, float* __restrict__ result) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float x = xs_and_ys[2 * i], y = xs_and_ys[2 * i + 1];
	if (!filter) { // bench_ours.cu:101-104: snap to texel centres
		x = (roundf(x * width - 0.5f) + 0.5f) / width;
		y = (roundf(y * height - 0.5f) + 0.5f) / height;
	}
	const float xb = x * width - 0.5f, yb = y * height - 0.5f;
	const float xf = floorf(xb), yf = floorf(yb);
	const float a = floorf((xb - xf) * 256.0f + 0.5f) * (1.0f / 256.0f), b = floorf((yb - yf) * 256.0f + 0.5f) * (1.0f / 256.0f);
	const int i0 = min(max((int)xf, 0), width - 1), i1 = min(max((int)xf + 1, 0), width - 1);
	const int j0 = min(max((int)yf, 0), height - 1), j1 = min(max((int)yf + 1, 0), height - 1);
	const float4 t00 = image[(size_t)j0 * width + i0], t10 = image[(size_t)j0 * width + i1], t01 = image[(size_t)j1 * width + i0], t11 = image[(size_t)j1 * width + i1];
	const float w00 = (1 - a) * (1 - b), w10 = a * (1 - b), w01 = (1 - a) * b, w11 = a * b;
	float* out = result + (size_t)i * STRIDE;
	out[0] = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
	out[1] = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
	out[2] = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
#pragma unroll
	for (uint32_t c = 3; c < STRIDE; ++c) out[c] = 1;
}

// mlp_learning_an_image.cu:60-70
__global__ void __launch_bounds__(256) k_to_ldr(const uint64_t n_elements, const uint32_t n_channels, const uint32_t stride, const float* __restrict__ in, uint8_t* __restrict__ out) {
	const uint64_t i = threadIdx.x + (uint64_t)blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint64_t pixel = i / n_channels;
	const uint32_t channel = (uint32_t)(i - pixel * n_channels);
	out[i] = (uint8_t)(powf(fmaxf(fminf(in[pixel * stride + channel], 1.0f), 0.0f), 1.0f / 2.2f) * 255.0f + 0.5f);
}

template <uint32_t STRIDE>
static void eval_image(hipStream_t stream, uint32_t n, const float* image_rgba, int width, int height, bool filter, const float* xs_and_ys, float* result) {
	if (n == 0) return;
	hipLaunchKernelGGL((k_eval_image<STRIDE>), dim3((n + 255) / 256), dim3(256), 0, stream, n, (const float4*)image_rgba, width, height, filter, xs_and_ys, result);
	HIP_CHECK_THROW(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------ image IO
static std::vector<float> synthetic_image(int width, int height) {
	// smooth shading + a zone plate (all spatial frequencies) + hard-edged discs + fine hashed grain: something for every level of a grid
	std::vector<float> img((size_t)width * height * 4);
	for (int y = 0; y < height; ++y) {
		for (int x = 0; x < width; ++x) {
			const float u = (x + 0.5f) / width, v = (y + 0.5f) / height;
			const float r2 = (u - 0.5f) * (u - 0.5f) + (v - 0.5f) * (v - 0.5f);
			const float zone = 0.5f + 0.5f * std::cos(900.0f * r2);
			float disc = 0;
			for (int k = 0; k < 6; ++k) {
				const float cx = 0.15f + 0.14f * k, cy = 0.2f + 0.1f * ((k * 3) % 5), rad = 0.03f + 0.01f * k;
				if ((u - cx) * (u - cx) + (v - cy) * (v - cy) < rad * rad) disc = 0.3f + 0.1f * k;
			}
			uint32_t h = (uint32_t)x * 73856093u ^ (uint32_t)y * 19349663u;
			h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
			const float grain = (h & 0xffff) / 65535.0f;
			const float base = 0.25f + 0.5f * u * (1 - v) + 0.25f * std::sin(6.2831853f * (2 * u + v));
			float* p = &img[((size_t)y * width + x) * 4];
			p[0] = std::min(std::max(0.55f * base + 0.3f * zone + 0.05f * grain + disc * 0.3f, 0.0f), 1.0f);
			p[1] = std::min(std::max(0.60f * base + 0.2f * zone * (1 - u) + 0.05f * grain + disc * 0.2f, 0.0f), 1.0f);
			p[2] = std::min(std::max(0.50f * base + 0.3f * (1 - zone) * v + 0.05f * grain + disc * 0.4f, 0.0f), 1.0f);
			p[3] = 1.0f;
		}
	}
	return img;
}

// 8-bit samples -> the floats stbi_loadf hands the reference's samples: linear RGBA with gamma 2.2, alpha 1
static std::vector<float> ldr_to_linear_rgba(const uint8_t* raw, int width, int height, int comps) {
	std::vector<float> img((size_t)width * height * 4);
	float lut[256];
	for (int v = 0; v < 256; ++v) lut[v] = std::pow(v / 255.0f, 2.2f); // stbi_loadf: LDR -> linear with gamma 2.2
	for (size_t p = 0; p < (size_t)width * height; ++p) {
		for (int c = 0; c < 3; ++c) img[p * 4 + c] = lut[raw[p * comps + (comps == 3 ? c : 0)]];
		img[p * 4 + 3] = 1.0f;
	}
	return img;
}

static std::vector<float> load_pnm(const std::string& filename, int& width, int& height) {
	std::ifstream f{filename, std::ios::binary};
	if (!f) throw std::runtime_error{"Could not open image file '" + filename + "'."};
	if (f.get() == 0xFF && f.get() == 0xD8) { // JPEG (baseline or progressive; samples/jpeg_decoder.h): e.g. the reference's data/images/albert.jpg
		f.seekg(0);
		const std::vector<uint8_t> bytes{std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()};
		const jpeg_lite::Image img = jpeg_lite::decode(bytes.data(), bytes.size());
		width = img.width;
		height = img.height;
		return ldr_to_linear_rgba(img.pixels.data(), width, height, img.channels);
	}
	f.clear();
	f.seekg(0);
	auto token = [&]() {
		std::string t;
		for (;;) {
			const int c = f.get();
			if (c == EOF) break;
			if (c == '#') { std::string skip; std::getline(f, skip); continue; }
			if (std::isspace(c)) { if (!t.empty()) break; continue; }
			t.push_back((char)c);
		}
		return t;
	};
	const std::string magic = token();
	if (magic != "P5" && magic != "P6") throw std::runtime_error{"'" + filename + "' is neither a JPEG nor a binary PPM (P6) / PGM (P5) file (PNG and others: convert first)."};
	width = std::stoi(token());
	height = std::stoi(token());
	const int maxval = std::stoi(token());
	if (width <= 0 || height <= 0 || maxval != 255) throw std::runtime_error{"'" + filename + "': only 8-bit images are supported."};
	const int comps = magic == "P6" ? 3 : 1;
	std::vector<uint8_t> raw((size_t)width * height * comps);
	f.read((char*)raw.data(), (std::streamsize)raw.size());
	if ((size_t)f.gcount() != raw.size()) throw std::runtime_error{"'" + filename + "' is truncated."};
	return ldr_to_linear_rgba(raw.data(), width, height, comps);
}

static GPUMemory<float> load_image(const std::string& spec, int& width, int& height) {
	std::vector<float> host;
	if (spec.rfind("synthetic", 0) == 0) {
		width = height = 1024;
		const size_t colon = spec.find(':');
		if (colon != std::string::npos && std::sscanf(spec.c_str() + colon + 1, "%dx%d", &width, &height) != 2) throw std::runtime_error{"synthetic image: expected synthetic:WxH"};
		host = synthetic_image(width, height);
	} else {
		host = load_pnm(spec, width, height);
	}
	GPUMemory<float> result(host.size());
	result.copy_from_host(host.data());
	return result;
}

static void save_image(const float* image, int width, int height, int n_channels, int channel_stride, const std::string& filename) {
	const size_t n = (size_t)width * height * n_channels;
	GPUMemory<uint8_t> ldr(n);
	hipLaunchKernelGGL(k_to_ldr, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, nullptr, (uint64_t)n, (uint32_t)n_channels, (uint32_t)channel_stride, image, ldr.data());
	HIP_CHECK_THROW(hipGetLastError());
	std::vector<uint8_t> host(n);
	ldr.copy_to_host(host.data());
	std::ofstream f{filename, std::ios::binary};
	if (!f) throw std::runtime_error{"Could not write '" + filename + "'."};
	f << (n_channels == 3 ? "P6" : "P5") << "\n" << width << " " << height << "\n255\n";
	f.write((const char*)host.data(), (std::streamsize)host.size());
}

static json load_config(const char* path) {
	std::ifstream f{path};
	if (!f) throw std::runtime_error{std::string{"Could not open config '"} + path + "'."};
	std::stringstream ss;
	ss << f.rdbuf();
	return json::parse(ss.str());
}

static json default_config() { // mlp_learning_an_image.cu:112-147
	return json::parse(R"({
		"loss": {"otype": "RelativeL2"},
		"optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "l2_reg": 0.0},
		"encoding": {"otype": "OneBlob", "n_bins": 32},
		"network": {"otype": "FullyFusedMLP", "n_neurons": 64, "n_hidden_layers": 4, "activation": "ReLU", "output_activation": "None"}
	})");
}

// full-resolution pixel-centre coordinates, padded to the batch granularity (mlp_learning_an_image.cu:190-204)
static GPUMemory<float> pixel_centres(int width, int height, uint32_t& n_coords, uint32_t& n_coords_padded) {
	n_coords = (uint32_t)width * height;
	n_coords_padded = next_multiple(n_coords, BATCH_SIZE_GRANULARITY);
	std::vector<float> host((size_t)n_coords_padded * 2, 0.5f);
	for (int y = 0; y < height; ++y) {
		for (int x = 0; x < width; ++x) {
			const size_t idx = ((size_t)y * width + x) * 2;
			host[idx + 0] = (float)(x + 0.5) / (float)width;
			host[idx + 1] = (float)(y + 0.5) / (float)height;
		}
	}
	GPUMemory<float> xs_and_ys(host.size());
	xs_and_ys.copy_from_host(host.data());
	return xs_and_ys;
}

// mean squared error of the learned image against the lookup at the pixel centres, as PSNR (linear colour, peak 1)
static double psnr(const std::vector<float>& prediction, uint32_t pred_stride, const std::vector<float>& reference, uint32_t n_coords) {
	double se = 0;
	for (uint32_t i = 0; i < n_coords; ++i)
		for (uint32_t c = 0; c < 3; ++c) {
			const double d = (double)prediction[(size_t)i * pred_stride + c] - reference[(size_t)i * 3 + c];
			se += d * d;
		}
	return -10.0 * std::log10(se / (3.0 * n_coords) + 1e-30);
}

// ---------------------------------------------------------------------------------------------------------------- the sample
static int run_sample(int argc, char** argv) {
	json config = argc >= 3 ? load_config(argv[2]) : default_config();
	if (argc >= 3) std::cout << "Loading custom json config '" << argv[2] << "'." << std::endl;

	int width, height;
	GPUMemory<float> image = load_image(argv[1], width, height);

	uint32_t n_coords, n_coords_padded;
	GPUMemory<float> xs_and_ys = pixel_centres(width, height, n_coords, n_coords_padded);
	GPUMemory<float> sampled_image((size_t)n_coords * 3);
	eval_image<3>(nullptr, n_coords, image.data(), width, height, true, xs_and_ys.data(), sampled_image.data());
	save_image(sampled_image.data(), width, height, 3, 3, "reference.ppm");
	std::vector<float> reference_host;
	sampled_image.copy_to_host(reference_host);

	const uint32_t batch_size = 1 << 18;
	const uint32_t n_training_steps = argc >= 4 ? (uint32_t)atoi(argv[3]) : 10000000;
	const uint32_t n_input_dims = 2, n_output_dims = 3;

	hipStream_t stream;
	HIP_CHECK_THROW(hipStreamCreate(&stream));
	default_rng_t rng{1337};

	GPUMatrix<float> training_target(n_output_dims, batch_size);
	GPUMatrix<float> training_batch(n_input_dims, batch_size);
	GPUMatrix<float> prediction(n_output_dims, n_coords_padded);
	GPUMatrix<float> inference_batch(xs_and_ys.data(), n_input_dims, n_coords_padded);

	std::shared_ptr<Loss<precision_t>> loss{create_loss<precision_t>(config.value("loss", json::object()))};
	std::shared_ptr<Optimizer<precision_t>> optimizer{create_optimizer<precision_t>(config.value("optimizer", json::object()))};
	auto network = std::make_shared<NetworkWithInputEncoding<precision_t>>(n_input_dims, n_output_dims, config.value("encoding", json::object()), config.value("network", json::object()));
	auto trainer = std::make_shared<Trainer<float, precision_t, precision_t>>(network, optimizer, loss);

	auto begin = std::chrono::steady_clock::now();
	float tmp_loss = 0;
	uint32_t tmp_loss_counter = 0;
	std::cout << "Beginning optimization with " << n_training_steps << " training steps." << std::endl;
	uint32_t interval = 10;
	for (uint32_t i = 0; i < n_training_steps; ++i) {
		const bool print_loss = i % interval == 0;
		const bool visualize_learned_func = argc < 5 && i % interval == 0;

		generate_random_uniform<float>(stream, rng, (size_t)batch_size * n_input_dims, training_batch.data());
		eval_image<n_output_dims>(stream, batch_size, image.data(), width, height, true, training_batch.data(), training_target.data());
		{
			auto ctx = trainer->training_step(stream, training_batch, training_target);
			if (i % std::min(interval, (uint32_t)100) == 0) {
				tmp_loss += trainer->loss(stream, *ctx);
				++tmp_loss_counter;
			}
		}
		if (print_loss) {
			const auto end = std::chrono::steady_clock::now();
			std::cout << "Step#" << i << ": loss=" << tmp_loss / (float)tmp_loss_counter << " time=" << std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count() << "[us]" << std::endl;
			tmp_loss = 0;
			tmp_loss_counter = 0;
		}
		if (visualize_learned_func) {
			network->inference(stream, inference_batch, prediction);
			HIP_CHECK_THROW(hipStreamSynchronize(stream));
			const std::string filename = std::to_string(i) + ".ppm";
			std::cout << "Writing '" << filename << "'... ";
			save_image(prediction.data(), width, height, 3, n_output_dims, filename);
			std::cout << "done." << std::endl;
		}
		if (print_loss) begin = std::chrono::steady_clock::now(); // visualisation is not part of the timing
		if (print_loss && i > 0 && interval < 1000) interval *= 10;
	}

	network->inference(stream, inference_batch, prediction);
	HIP_CHECK_THROW(hipStreamSynchronize(stream));
	if (argc >= 5) save_image(prediction.data(), width, height, 3, n_output_dims, argv[4]);
	const double quality = psnr(prediction.to_cpu_vector(), n_output_dims, reference_host, n_coords);
	std::cout << "PSNR of the learned image against the reference lookup: " << quality << " dB" << std::endl;

	HIP_CHECK_THROW(hipStreamDestroy(stream));
	free_all_gpu_memory_arenas();
	return EXIT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------------------ the benchmark
static int run_bench(int argc, char** argv) {
	if (argc < 4) throw std::runtime_error{"--bench needs <image> <config.json>"};
	std::string result_path = "bench_result_ours.json";
	std::vector<uint32_t> batch_sizes = {1u << 21, 1u << 20, 1u << 19, 1u << 18, 1u << 17, 1u << 16, 1u << 15, 1u << 14}; // bench_ours.cu:188
	int cooldown = 0; // the reference sleeps 10 s between runs to let a desktop GPU cool down
	for (int k = 4; k < argc; ++k) {
		const std::string arg = argv[k];
		if (arg == "--batches" && k + 1 < argc) {
			batch_sizes.clear();
			std::stringstream ss{argv[++k]};
			for (std::string t; std::getline(ss, t, ',');) batch_sizes.push_back(1u << std::stoi(t));
		} else if (arg == "--cooldown" && k + 1 < argc) {
			cooldown = std::stoi(argv[++k]);
		} else {
			result_path = arg;
		}
	}

	int width, height;
	GPUMemory<float> image = load_image(argv[2], width, height);
	uint32_t n_coords, n_coords_padded;
	GPUMemory<float> xs_and_ys = pixel_centres(width, height, n_coords, n_coords_padded);
	GPUMemory<float> sampled_image((size_t)n_coords * 3);
	const bool filter = false; // bench_ours.cu:180
	eval_image<3>(nullptr, n_coords, image.data(), width, height, filter, xs_and_ys.data(), sampled_image.data());
	save_image(sampled_image.data(), width, height, 3, 3, "reference.ppm");
	std::vector<float> reference_host;
	sampled_image.copy_to_host(reference_host);

	const std::vector<std::string> methods = {"fully_fused", "cutlass"}; // both reach the same kernels here (CutlassMLP is the same function)
	json bench_result = json::object();
	default_rng_t rng{1337};
	hipStream_t stream;
	HIP_CHECK_THROW(hipStreamCreate(&stream));

	for (const std::string& method : methods) {
		json rows = json::array();
		for (const uint32_t batch_size : batch_sizes) {
			uint32_t n_iterations = std::max(1000u * (1u << 18) / batch_size, 250u);
			uint32_t n_iterations_warmup = n_iterations / 2;
			const uint32_t n_dims = 2, n_out = 3;

			GPUMemory<float> batch((size_t)batch_size * n_dims);
			GPUMatrix<float> bench_target(n_out, batch_size);
			GPUMatrix<float> prediction(n_out, n_coords_padded);

			json config = load_config(argv[3]);
			json network_opts = config.value("network", json::object());
			network_opts["otype"] = method == "cutlass" ? "CutlassMLP" : "FullyFusedMLP";
			std::shared_ptr<Loss<precision_t>> loss{create_loss<precision_t>(config.value("loss", json::object()))};
			std::shared_ptr<Optimizer<precision_t>> optimizer{create_optimizer<precision_t>(config.value("optimizer", json::object()))};
			auto network = std::make_shared<NetworkWithInputEncoding<precision_t>>(n_dims, n_out, config.value("encoding", json::object()), network_opts);
			auto trainer = std::make_shared<Trainer<float, precision_t, precision_t>>(network, optimizer, loss);

			auto begin = std::chrono::steady_clock::now();
			float tmp_loss = 0;
			uint32_t tmp_loss_counter = 0;
			uint32_t print_interval = n_iterations / 10;
			const uint32_t STEPS_INCREMENT = 5;
			double mean_training_throughput = 0;
			size_t mean_counter = 0;

			for (uint32_t i = 0; i < n_iterations; i += STEPS_INCREMENT) {
				const bool print_loss = i % print_interval == 0;
				for (uint32_t j = 0; j < STEPS_INCREMENT; ++j) {
					generate_random_uniform<float>(stream, rng, (size_t)batch_size * n_dims, batch.data());
					eval_image<n_out>(stream, batch_size, image.data(), width, height, filter, batch.data(), bench_target.data());
					auto ctx = trainer->training_step(stream, GPUMatrix<float>{batch.data(), n_dims, batch_size}, bench_target);
					if (j == STEPS_INCREMENT - 1) {
						tmp_loss += trainer->loss(stream, *ctx);
						++tmp_loss_counter;
					}
				}
				if (print_loss) {
					HIP_CHECK_THROW(hipDeviceSynchronize());
					const auto end = std::chrono::steady_clock::now();
					const auto microseconds = std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count();
					const double throughput = (double)print_interval * batch_size / ((double)microseconds / 1000000.0);
					std::cout << "Iteration#" << i << ": loss=" << tmp_loss / (float)tmp_loss_counter << " time=" << microseconds << "[us] thp=" << throughput << "/s" << std::endl;
					begin = end;
					tmp_loss = 0;
					tmp_loss_counter = 0;
					if (i >= n_iterations_warmup) {
						mean_training_throughput += throughput;
						++mean_counter;
					}
				}
			}
			mean_training_throughput /= (double)std::max<size_t>(mean_counter, 1);

			GPUMatrix<float> inference_batch(xs_and_ys.data(), n_dims, n_coords_padded);
			network->inference(stream, inference_batch, prediction);
			HIP_CHECK_THROW(hipStreamSynchronize(stream));
			save_image(prediction.data(), width, height, 3, n_out, std::to_string(batch_size) + "-after-" + std::to_string(n_iterations) + "-iters-" + method + ".ppm");
			const double quality = psnr(prediction.to_cpu_vector(), n_out, reference_host, n_coords);
			std::cout << "Finished training benchmark. Mean throughput is " << mean_training_throughput << "/s. PSNR " << quality << " dB." << std::endl;
			if (cooldown > 0) std::this_thread::sleep_for(std::chrono::seconds{cooldown});

			double mean_inference_throughput = 0;
			mean_counter = 0;
			print_interval *= 5;
			n_iterations *= 5;
			n_iterations_warmup *= 5;
			begin = std::chrono::steady_clock::now();
			for (uint32_t i = 0; i < n_iterations; ++i) {
				const bool print_loss = i % print_interval == 0;
				generate_random_uniform<float>(stream, rng, (size_t)batch_size * n_dims, batch.data());
				GPUMatrix<float> in{batch.data(), n_dims, batch_size};
				network->inference(stream, in, bench_target);
				if (print_loss) {
					HIP_CHECK_THROW(hipDeviceSynchronize());
					const auto end = std::chrono::steady_clock::now();
					const auto microseconds = std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count();
					const double throughput = (double)print_interval * batch_size / ((double)microseconds / 1000000.0);
					std::cout << "Iteration#" << i << ": time=" << microseconds << "[us] thp=" << throughput << "/s" << std::endl;
					begin = end;
					if (i >= n_iterations_warmup) {
						mean_inference_throughput += throughput;
						++mean_counter;
					}
				}
			}
			mean_inference_throughput /= (double)std::max<size_t>(mean_counter, 1);
			std::cout << "Finished inference benchmark. Mean throughput is " << mean_inference_throughput << "/s." << std::endl;
			if (cooldown > 0) std::this_thread::sleep_for(std::chrono::seconds{cooldown});

			json row = json::object();
			row["batch_size"] = batch_size;
			row["training_throughput"] = mean_training_throughput;
			row["inference_throughput"] = mean_inference_throughput;
			row["psnr"] = quality;
			rows.push_back(row);
		}
		bench_result[method] = rows;
	}
	std::ofstream out{result_path};
	out << bench_result.dump(4);
	HIP_CHECK_THROW(hipStreamDestroy(stream));
	free_all_gpu_memory_arenas();
	return EXIT_SUCCESS;
}

// k_eval_image on caller-provided coordinates: coords.f32 = [n][2] floats, out.f32 = [n][3] floats (filter = linear)
static int run_lookup(int argc, char** argv) {
	if (argc < 5) throw std::runtime_error{"--sample needs <image> <coords.f32> <out.f32>"};
	int width, height;
	GPUMemory<float> image = load_image(argv[2], width, height);
	std::ifstream f{argv[3], std::ios::binary};
	std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	const uint32_t n = (uint32_t)(raw.size() / 8);
	GPUMemory<float> coords((size_t)n * 2), result((size_t)n * 3);
	coords.copy_from_host((const float*)raw.data());
	eval_image<3>(nullptr, n, image.data(), width, height, true, coords.data(), result.data());
	HIP_CHECK_THROW(hipDeviceSynchronize());
	std::vector<float> host;
	result.copy_to_host(host);
	std::ofstream out{argv[4], std::ios::binary};
	out.write((const char*)host.data(), (std::streamsize)(host.size() * sizeof(float)));
	// the image itself, so that the checker needs no image decoder of its own: <out>.image = int32 w, int32 h, [h][w][4] floats
	std::vector<float> img;
	image.copy_to_host(img);
	std::ofstream iout{std::string{argv[4]} + ".image", std::ios::binary};
	const int32_t wh[2] = {width, height};
	iout.write((const char*)wh, 8);
	iout.write((const char*)img.data(), (std::streamsize)(img.size() * sizeof(float)));
	return EXIT_SUCCESS;
}

int main(int argc, char* argv[]) {
	try {
		if (argc < 2) {
			std::cout << "USAGE: " << argv[0] << " path-to-image.ppm|synthetic[:WxH] [path-to-optional-config.json] [n_training_steps] [final-image.ppm]" << std::endl;
			std::cout << "       " << argv[0] << " --bench image config.json [result.json] [--batches 18,16] [--cooldown S]" << std::endl;
			return 0;
		}
		if (std::string{argv[1]} == "--bench") return run_bench(argc, argv);
		if (std::string{argv[1]} == "--sample") return run_lookup(argc, argv);
		return run_sample(argc, argv);
	} catch (const std::exception& e) {
		std::cout << "Uncaught exception: " << e.what() << std::endl;
		return EXIT_FAILURE;
	}
}
