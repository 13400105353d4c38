// jpeg_to_pnm <in.jpg> <out.pgm|ppm>: decodes with samples/jpeg_decoder.h and writes a binary PGM (1 channel) or PPM (3 channels);
// the driver of tests/test_jpeg_decoder.py
#include "../../samples/jpeg_decoder.h"

#include <cstdio>
#include <fstream>
#include <iterator>

int main(int argc, char** argv) {
	if (argc != 3) {
		std::fprintf(stderr, "usage: jpeg_to_pnm in.jpg out.pnm\n");
		return 2;
	}
	try {
		std::ifstream f{argv[1], std::ios::binary};
		if (!f) throw std::runtime_error{"cannot open input"};
		const std::vector<uint8_t> bytes{std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()};
		const jpeg_lite::Image img = jpeg_lite::decode(bytes.data(), bytes.size());
		std::ofstream o{argv[2], std::ios::binary};
		o << (img.channels == 1 ? "P5" : "P6") << "\n" << img.width << " " << img.height << "\n255\n";
		o.write((const char*)img.pixels.data(), (std::streamsize)img.pixels.size());
		return 0;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
}
