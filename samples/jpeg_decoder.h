// jpeg_decoder.h -- a small JPEG reader for the caller harness (samples/mlp_learning_an_image.hip).
//
// The reference's samples load their training image through the vendored stb_image (dependencies/stbi/stbi_wrapper.cpp:37-44,
// used at samples/mlp_learning_an_image.cu:50-52); BASELINE config 3 names data/images/albert.jpg, a PROGRESSIVE 8-bit grayscale
// JPEG.  This is an own implementation of the format from ITU-T T.81: baseline / extended sequential (SOF0, SOF1) and progressive
// (SOF2) Huffman-coded images, 8 bits per sample, 1 or 3 components (YCbCr, JFIF), any 1x / 2x sampling factors, restart
// intervals.  Not supported (reported as errors): arithmetic coding, lossless, 12-bit, CMYK.
// Decoding path: every scan fills coefficient blocks (progressive scans refine them in place), then one pass dequantises,
// inverse-transforms (separable float IDCT), upsamples chroma by replication and converts to RGB.  Decoders differ by +-1 in the
// IDCT rounding and by more at chroma edges (libjpeg interpolates subsampled chroma); tests/test_jpeg_decoder.py pins this one
// against PIL within those bounds.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace jpeg_lite {

struct Image {
	int width = 0, height = 0, channels = 0; // channels: 1 (gray) or 3 (RGB)
	std::vector<uint8_t> pixels;             // [height][width][channels]
};

namespace detail {

static const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman { // canonical code: codes of length l are consecutive, starting at first_code[l]
	bool present = false;
	uint8_t values[256];
	int max_code[18], val_offset[17];
	void build(const uint8_t counts[16], const uint8_t* vals, int n) {
		std::memcpy(values, vals, (size_t)n);
		int code = 0, k = 0;
		for (int l = 1; l <= 16; ++l) {
			val_offset[l] = k - code;
			k += counts[l - 1];
			code += counts[l - 1];
			max_code[l] = counts[l - 1] ? code - 1 : -1;
			code <<= 1;
		}
		max_code[17] = 0x7fffffff;
		present = true;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0;
	int blocks_w = 0, blocks_h = 0; // in whole MCUs (padded)
	int dc_pred = 0;
	std::vector<int16_t> coef;      // [blocks_h][blocks_w][64], natural order
};

struct BitReader {
	const uint8_t* p;
	const uint8_t* end;
	uint32_t bits = 0;
	int n_bits = 0;
	bool hit_marker = false;
	void reset() { bits = 0; n_bits = 0; hit_marker = false; }
	void fill() {
		while (n_bits <= 24) {
			uint32_t byte = 0;
			if (!hit_marker && p < end) {
				byte = *p;
				if (byte == 0xFF) {
					if (p + 1 < end && p[1] == 0x00) p += 2; // stuffed zero
					else { hit_marker = true; byte = 0; }    // a marker: feed zeros, leave p on it
				} else {
					++p;
				}
			}
			bits |= byte << (24 - n_bits);
			n_bits += 8;
		}
	}
	int get_bit() {
		if (n_bits < 1) fill();
		const int b = (int)(bits >> 31);
		bits <<= 1;
		--n_bits;
		return b;
	}
	int get_bits(int n) {
		if (n == 0) return 0;
		if (n_bits < n) fill();
		const int v = (int)(bits >> (32 - n));
		bits <<= n;
		n_bits -= n;
		return v;
	}
	int decode(const Huffman& h) {
		if (!h.present) throw std::runtime_error{"JPEG: scan refers to a Huffman table that was not defined"};
		int code = 0;
		for (int l = 1; l <= 16; ++l) {
			code = (code << 1) | get_bit();
			if (h.max_code[l] >= 0 && code <= h.max_code[l] && code + h.val_offset[l] >= 0) {
				const int idx = code + h.val_offset[l];
				if (idx < 256) return h.values[idx];
			}
		}
		throw std::runtime_error{"JPEG: invalid Huffman code"};
	}
	static int extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); } // T.81 F.2.2.1
	// n comes from a Huffman table of the FILE (0..255): a magnitude category beyond 16 bits is not JPEG (T.81 F.1.2.1.1: DC <= 11,
	// AC <= 10 for 8-bit samples) and would shift by more than the word
	int receive_extend(int n) {
		if (n < 0 || n > 16) throw std::runtime_error{"JPEG: magnitude category out of range"};
		return extend(get_bits(n), n);
	}
};

inline void idct_block(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
	static float c[8][8];
	static bool init = false;
	if (!init) {
		for (int x = 0; x < 8; ++x)
			for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
		init = true;
	}
	float tmp[64], deq[64];
	for (int i = 0; i < 64; ++i) deq[i] = (float)in[i] * (float)q[i];
	for (int y = 0; y < 8; ++y)     // rows: tmp[y][x] = sum_u c[x][u] deq[y][u]
		for (int x = 0; x < 8; ++x) {
			float s = 0;
			for (int u = 0; u < 8; ++u) s += c[x][u] * deq[y * 8 + u];
			tmp[y * 8 + x] = s;
		}
	for (int x = 0; x < 8; ++x)     // columns
		for (int y = 0; y < 8; ++y) {
			float s = 0;
			for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
			const int p = (int)std::floor(s + 128.5f);
			out[y * stride + x] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));
		}
}

} // namespace detail

inline Image decode(const uint8_t* data, size_t size) {
	using namespace detail;
	if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) throw std::runtime_error{"JPEG: missing SOI marker"};
	uint16_t qt[4][64] = {};
	bool qt_present[4] = {false, false, false, false};
	Huffman hdc[4], hac[4];
	std::vector<Component> comps;
	int width = 0, height = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0, restart_interval = 0;
	bool progressive = false, have_frame = false;
	size_t pos = 2;
	auto u16 = [&](size_t at) {
		if (at + 2 > size) throw std::runtime_error{"JPEG: truncated file"};
		return (int)(data[at] << 8 | data[at + 1]);
	};

	for (;;) {
		while (pos < size && data[pos] != 0xFF) ++pos; // tolerate garbage between segments
		while (pos < size && data[pos] == 0xFF) ++pos; // fill bytes
		if (pos >= size) throw std::runtime_error{"JPEG: no EOI marker"};
		const int marker = data[pos++];
		if (marker == 0xD9) break;                                  // EOI
		if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue; // TEM, stray RSTn
		const int len = u16(pos);
		if (len < 2 || pos + (size_t)len > size) throw std::runtime_error{"JPEG: bad segment length"};
		const uint8_t* seg = data + pos + 2;
		const int seg_len = len - 2;
		pos += (size_t)len;

		if (marker == 0xDB) { // DQT
			int i = 0;
			while (i < seg_len) {
				const int pq = seg[i] >> 4, tq = seg[i] & 15;
				++i;
				if (tq > 3 || i + (pq ? 128 : 64) > seg_len) throw std::runtime_error{"JPEG: bad quantisation table"};
				for (int k = 0; k < 64; ++k) {
					qt[tq][ZIGZAG[k]] = pq ? (uint16_t)(seg[i] << 8 | seg[i + 1]) : seg[i];
					i += pq ? 2 : 1;
				}
				qt_present[tq] = true;
			}
		} else if (marker == 0xC4) { // DHT
			int i = 0;
			while (i < seg_len) {
				if (i + 17 > seg_len) throw std::runtime_error{"JPEG: bad Huffman table"};
				const int tc = seg[i] >> 4, th = seg[i] & 15;
				int n = 0;
				for (int k = 0; k < 16; ++k) n += seg[i + 1 + k];
				if (th > 3 || tc > 1 || n > 256 || i + 17 + n > seg_len) throw std::runtime_error{"JPEG: bad Huffman table"};
				(tc ? hac[th] : hdc[th]).build(seg + i + 1, seg + i + 17, n);
				i += 17 + n;
			}
		} else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) { // SOF0 / SOF1 / SOF2
			if (have_frame) throw std::runtime_error{"JPEG: more than one frame"};
			if (seg_len < 6 || seg[0] != 8) throw std::runtime_error{"JPEG: only 8 bits per sample are supported"};
			progressive = marker == 0xC2;
			height = seg[1] << 8 | seg[2];
			width = seg[3] << 8 | seg[4];
			const int nc = seg[5];
			if (width <= 0 || height <= 0) throw std::runtime_error{"JPEG: empty image"};
			// a header can ask for 65535 x 65535: bound what the coefficient and pixel buffers may take (a training image, not a scan of a wall)
			if ((size_t)width * (size_t)height > ((size_t)1 << 28)) throw std::runtime_error{"JPEG: image larger than 2^28 pixels is refused"};
			if ((nc != 1 && nc != 3) || seg_len < 6 + 3 * nc) throw std::runtime_error{"JPEG: only 1 or 3 components are supported"};
			comps.resize((size_t)nc);
			for (int c = 0; c < nc; ++c) {
				comps[c].id = seg[6 + 3 * c];
				comps[c].h = seg[7 + 3 * c] >> 4;
				comps[c].v = seg[7 + 3 * c] & 15;
				comps[c].tq = seg[8 + 3 * c];
				if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) throw std::runtime_error{"JPEG: bad component"};
				hmax = std::max(hmax, comps[c].h);
				vmax = std::max(vmax, comps[c].v);
			}
			mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
			mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
			for (auto& c : comps) {
				c.blocks_w = mcus_x * c.h;
				c.blocks_h = mcus_y * c.v;
				c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
			}
			have_frame = true;
		} else if (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
			throw std::runtime_error{"JPEG: unsupported coding process (lossless, hierarchical or arithmetic)"};
		} else if (marker == 0xDD) { // DRI
			if (seg_len < 2) throw std::runtime_error{"JPEG: bad restart interval"};
			restart_interval = seg[0] << 8 | seg[1];
		} else if (marker == 0xDA) { // SOS + entropy-coded data
			if (!have_frame) throw std::runtime_error{"JPEG: scan before frame header"};
			if (seg_len < 1) throw std::runtime_error{"JPEG: bad scan header"};
			const int ns = seg[0];
			if (ns < 1 || ns > (int)comps.size() || seg_len < 4 + 2 * ns) throw std::runtime_error{"JPEG: bad scan header"};
			Component* sc[3];
			int td[3], ta[3];
			for (int i = 0; i < ns; ++i) {
				sc[i] = nullptr;
				for (auto& c : comps) if (c.id == seg[1 + 2 * i]) sc[i] = &c;
				if (!sc[i]) throw std::runtime_error{"JPEG: scan refers to an unknown component"};
				td[i] = seg[2 + 2 * i] >> 4;
				ta[i] = seg[2 + 2 * i] & 15;
				if (td[i] > 3 || ta[i] > 3) throw std::runtime_error{"JPEG: bad table selector"};
			}
			int ss = seg[1 + 2 * ns], se = seg[2 + 2 * ns];
			const int ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
			if (!progressive) { ss = 0; se = 63; }
			if (ss > se || se > 63 || (progressive && ss == 0 && se != 0) || (progressive && ss > 0 && ns != 1)) throw std::runtime_error{"JPEG: bad spectral selection"};

			BitReader br{data + pos, data + size};
			for (auto& c : comps) c.dc_pred = 0;
			int eobrun = 0;
			int restarts_left = restart_interval;
			int next_rst = 0;

			// one block of one component (T.81 F.2.2, G.1.2)
			auto decode_block = [&](Component& c, int16_t* b, int tdc, int tac) {
				if (!progressive) {
					const int t = br.decode(hdc[tdc]);
					c.dc_pred += t ? br.receive_extend(t) : 0;
					b[0] = (int16_t)c.dc_pred;
					for (int k = 1; k < 64;) {
						const int rs = br.decode(hac[tac]), r = rs >> 4, s = rs & 15;
						if (s == 0) {
							if (r != 15) break; // EOB
							k += 16;
							continue;
						}
						k += r;
						if (k > 63) throw std::runtime_error{"JPEG: coefficient index out of range"};
						b[ZIGZAG[k++]] = (int16_t)br.receive_extend(s);
					}
					return;
				}
				if (ss == 0) { // DC scan
					if (ah == 0) {
						const int t = br.decode(hdc[tdc]);
						c.dc_pred += t ? br.receive_extend(t) : 0;
						b[0] = (int16_t)(c.dc_pred * (1 << al));
					} else if (br.get_bit()) {
						b[0] = (int16_t)(b[0] | (1 << al));
					}
					return;
				}
				if (ah == 0) { // AC first pass (G.1.2.2)
					if (eobrun > 0) { --eobrun; return; }
					for (int k = ss; k <= se;) {
						const int rs = br.decode(hac[tac]), r = rs >> 4, s = rs & 15;
						if (s == 0) {
							if (r < 15) {
								eobrun = (1 << r) - 1;
								if (r) eobrun += br.get_bits(r);
								break;
							}
							k += 16;
							continue;
						}
						k += r;
						if (k > 63) throw std::runtime_error{"JPEG: coefficient index out of range"};
						b[ZIGZAG[k++]] = (int16_t)(br.receive_extend(s) * (1 << al));
					}
					return;
				}
				// AC refinement (G.1.2.3): correction bits for the coefficients that are already non-zero, interleaved with newly
				// non-zero ones (+-1 << al) placed after `r` still-zero coefficients
				const int p1 = 1 << al, m1 = -(1 << al);
				int k = ss;
				if (eobrun == 0) {
					for (; k <= se;) {
						const int rs = br.decode(hac[tac]);
						int r = rs >> 4;
						const int s = rs & 15;
						int value = 0;
						if (s == 0) {
							if (r < 15) {
								eobrun = 1 << r;
								if (r) eobrun += br.get_bits(r);
								break;
							}
						} else {
							if (s != 1) throw std::runtime_error{"JPEG: bad refinement code"};
							value = br.get_bit() ? p1 : m1;
						}
						for (; k <= se; ++k) {
							int16_t& coef = b[ZIGZAG[k]];
							if (coef != 0) {
								if (br.get_bit() && (coef & p1) == 0) coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
							} else {
								if (r == 0) {
									if (value) coef = (int16_t)value;
									++k;
									break;
								}
								--r;
							}
						}
					}
				}
				if (eobrun > 0) { // the rest of the band: correction bits only
					for (; k <= se; ++k) {
						int16_t& coef = b[ZIGZAG[k]];
						if (coef != 0 && br.get_bit() && (coef & p1) == 0) coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
					}
					--eobrun;
				}
			};
			auto handle_restart = [&]() {
				if (!restart_interval) return;
				if (--restarts_left > 0) return;
				// byte-align, expect RSTn
				br.reset();
				const uint8_t* q = br.p;
				while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
					if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break; // another marker: give up on restarts
					++q;
				}
				if (q + 1 < br.end && q[0] == 0xFF && q[1] == 0xD0 + next_rst) q += 2;
				next_rst = (next_rst + 1) & 7;
				br.p = q;
				br.reset();
				for (auto& c : comps) c.dc_pred = 0;
				eobrun = 0;
				restarts_left = restart_interval;
			};

			if (ns == 1) { // non-interleaved: the component's own blocks covering the image (A.2.3)
				Component& c = *sc[0];
				const int bw = (((width * c.h + hmax - 1) / hmax) + 7) / 8, bh = (((height * c.v + vmax - 1) / vmax) + 7) / 8;
				for (int by = 0; by < bh; ++by)
					for (int bx = 0; bx < bw; ++bx) {
						decode_block(c, &c.coef[((size_t)by * c.blocks_w + bx) * 64], td[0], ta[0]);
						handle_restart();
					}
			} else {
				for (int my = 0; my < mcus_y; ++my)
					for (int mx = 0; mx < mcus_x; ++mx) {
						for (int i = 0; i < ns; ++i) {
							Component& c = *sc[i];
							for (int v = 0; v < c.v; ++v)
								for (int h = 0; h < c.h; ++h) decode_block(c, &c.coef[((size_t)(my * c.v + v) * c.blocks_w + mx * c.h + h) * 64], td[i], ta[i]);
						}
						handle_restart();
					}
			}
			// continue after the entropy-coded segment: the reader stops in front of the next marker
			pos = (size_t)(br.p - data);
		}
		// everything else (APPn, COM, DNL, ...) is skipped
	}
	if (!have_frame) throw std::runtime_error{"JPEG: no frame header"};

	// ---- coefficients -> samples -> RGB
	std::vector<std::vector<uint8_t>> planes(comps.size());
	for (size_t ci = 0; ci < comps.size(); ++ci) {
		Component& c = comps[ci];
		if (!qt_present[c.tq]) throw std::runtime_error{"JPEG: frame refers to a quantisation table that was not defined"};
		const int pw = c.blocks_w * 8;
		planes[ci].resize((size_t)pw * c.blocks_h * 8);
		for (int by = 0; by < c.blocks_h; ++by)
			for (int bx = 0; bx < c.blocks_w; ++bx) idct_block(&c.coef[((size_t)by * c.blocks_w + bx) * 64], qt[c.tq], &planes[ci][((size_t)by * 8) * pw + bx * 8], pw);
	}
	Image img;
	img.width = width;
	img.height = height;
	img.channels = (int)comps.size();
	img.pixels.resize((size_t)width * height * img.channels);
	for (int y = 0; y < height; ++y)
		for (int x = 0; x < width; ++x) {
			int s[3] = {0, 0, 0};
			for (size_t ci = 0; ci < comps.size(); ++ci) {
				const Component& c = comps[ci];
				const int sx = x * c.h / hmax, sy = y * c.v / vmax; // chroma by replication
				s[ci] = planes[ci][(size_t)sy * (c.blocks_w * 8) + sx];
			}
			uint8_t* px = &img.pixels[((size_t)y * width + x) * img.channels];
			if (comps.size() == 1) {
				px[0] = (uint8_t)s[0];
			} else { // JFIF YCbCr -> RGB
				const float Y = (float)s[0], cb = (float)s[1] - 128.0f, cr = (float)s[2] - 128.0f;
				const float rgb[3] = {Y + 1.402f * cr, Y - 0.344136f * cb - 0.714136f * cr, Y + 1.772f * cb};
				for (int k = 0; k < 3; ++k) {
					const int v = (int)std::floor(rgb[k] + 0.5f);
					px[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
				}
			}
		}
	return img;
}

} // namespace jpeg_lite
