// json_lite.h -- a small self-contained JSON value (parse / query / dump) for config handling.
//
// The reference passes nlohmann::json objects through its factories (config.h:53, cpp_api.h:113-115).  This library's
// boundary is a C ABI that takes JSON *text*; this header is all the JSON we need behind that boundary:
// objects, arrays, strings, numbers, booleans, null; `value(key, default)`, `contains`, `dump()`.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace tcnn_amd {

class Json {
public:
	enum class Kind { Null, Bool, Number, String, Array, Object, Binary };

	Json() = default;
	Json(bool b) : m_kind{Kind::Bool}, m_bool{b} {}
	Json(double d) : m_kind{Kind::Number}, m_num{d} {}
	Json(float d) : m_kind{Kind::Number}, m_num{d}, m_is_float32{true} {}
	Json(int d) : m_kind{Kind::Number}, m_num{(double)d}, m_is_int{true} {}
	Json(uint32_t d) : m_kind{Kind::Number}, m_num{(double)d}, m_is_int{true} {}
	Json(int64_t d) : m_kind{Kind::Number}, m_num{(double)d}, m_is_int{true} {}
	Json(uint64_t d) : m_kind{Kind::Number}, m_num{(double)d}, m_is_int{true} {}
	Json(const char* s) : m_kind{Kind::String}, m_str{s} {}
	Json(const std::string& s) : m_kind{Kind::String}, m_str{s} {}
	// literal syntax of the configs in the reference's samples: {{"otype", "Adam"}, {"learning_rate", 1e-2}} is an object
	// (every element a [string, value] pair), anything else a list
	Json(std::initializer_list<Json> init) {
		bool pairs = init.size() > 0;
		for (const Json& e : init) pairs = pairs && e.m_kind == Kind::Array && e.m_arr.size() == 2 && e.m_arr[0].m_kind == Kind::String;
		if (pairs) {
			m_kind = Kind::Object;
			for (const Json& e : init) (*this)[e.m_arr[0].m_str] = e.m_arr[1];
		} else {
			m_kind = Kind::Array;
			m_arr.assign(init.begin(), init.end());
		}
	}

	static Json object() { Json j; j.m_kind = Kind::Object; return j; }
	static Json array() { Json j; j.m_kind = Kind::Array; return j; }
	// nlohmann::json::binary_t: what the snapshot format keeps its parameter blobs in (gpu_memory_json.h:36-71)
	static Json binary(std::vector<uint8_t> bytes) { Json j; j.m_kind = Kind::Binary; j.m_bin = std::move(bytes); return j; }
	bool is_binary() const { return m_kind == Kind::Binary; }
	const std::vector<uint8_t>& get_binary() const {
		if (m_kind != Kind::Binary) throw std::runtime_error{"json: value is not binary"};
		return m_bin;
	}

	static Json parse(const std::string& text) {
		Parser p{text};
		p.skip_ws();
		Json j = p.parse_value();
		p.skip_ws();
		if (p.pos != text.size()) p.fail("trailing characters");
		return j;
	}

	Kind kind() const { return m_kind; }
	bool is_null() const { return m_kind == Kind::Null; }
	bool is_object() const { return m_kind == Kind::Object; }
	bool is_array() const { return m_kind == Kind::Array; }
	bool is_string() const { return m_kind == Kind::String; }
	bool is_number() const { return m_kind == Kind::Number; }
	bool is_bool() const { return m_kind == Kind::Bool; }

	bool contains(const std::string& key) const { return m_kind == Kind::Object && find(key) != nullptr; }

	const Json& at(const std::string& key) const {
		const Json* j = find(key);
		if (!j) throw std::runtime_error{"json: key '" + key + "' not found"};
		return *j;
	}
	const Json& operator[](const std::string& key) const { return at(key); }

	Json& operator[](const std::string& key) {
		if (m_kind == Kind::Null) m_kind = Kind::Object;
		if (m_kind != Kind::Object) throw std::runtime_error{"json: not an object"};
		for (auto& kv : m_obj) if (kv.first == key) return kv.second;
		m_obj.emplace_back(key, Json{});
		return m_obj.back().second;
	}

	size_t size() const { return m_kind == Kind::Array ? m_arr.size() : (m_kind == Kind::Object ? m_obj.size() : 0); }
	const Json& at(size_t i) const { return m_arr.at(i); }
	void push_back(Json j) { if (m_kind == Kind::Null) m_kind = Kind::Array; m_arr.push_back(std::move(j)); }
	const std::vector<std::pair<std::string, Json>>& items() const { return m_obj; }

	double as_double() const {
		if (m_kind == Kind::Number) return m_num;
		if (m_kind == Kind::Bool) return m_bool ? 1.0 : 0.0;
		throw std::runtime_error{"json: value is not a number"};
	}
	bool as_bool() const {
		if (m_kind == Kind::Bool) return m_bool;
		if (m_kind == Kind::Number) return m_num != 0.0;
		throw std::runtime_error{"json: value is not a boolean"};
	}
	const std::string& as_string() const {
		if (m_kind != Kind::String) throw std::runtime_error{"json: value is not a string"};
		return m_str;
	}

	// nlohmann-style .value(key, default)
	uint32_t value(const std::string& key, uint32_t def) const { const Json* j = find(key); return j ? (uint32_t)j->as_double() : def; }
	int value(const std::string& key, int def) const { const Json* j = find(key); return j ? (int)j->as_double() : def; }
	float value(const std::string& key, float def) const { const Json* j = find(key); return j ? (float)j->as_double() : def; }
	double value(const std::string& key, double def) const { const Json* j = find(key); return j ? j->as_double() : def; }
	bool value(const std::string& key, bool def) const { const Json* j = find(key); return j ? j->as_bool() : def; }
	std::string value(const std::string& key, const char* def) const { const Json* j = find(key); return j ? j->as_string() : std::string{def}; }
	std::string value(const std::string& key, const std::string& def) const { const Json* j = find(key); return j ? j->as_string() : def; }
	Json value(const std::string& key, const Json& def) const { const Json* j = find(key); return j ? *j : def; }

	// indent < 0: compact; else pretty-printed with that many spaces per level (nlohmann's dump(4))
	std::string dump(int indent = -1) const {
		std::string out;
		dump_to(out, indent, 0);
		return out;
	}

private:
	const Json* find(const std::string& key) const {
		if (m_kind != Kind::Object) return nullptr;
		for (const auto& kv : m_obj) if (kv.first == key) return &kv.second;
		return nullptr;
	}

	static void dump_string(const std::string& s, std::string& out) {
		out += '"';
		for (char ch : s) {
			switch (ch) {
				case '"': out += "\\\""; break;
				case '\\': out += "\\\\"; break;
				case '\n': out += "\\n"; break;
				case '\t': out += "\\t"; break;
				case '\r': out += "\\r"; break;
				default:
					if ((unsigned char)ch < 0x20) { char buf[8]; snprintf(buf, sizeof(buf), "\\u%04x", ch); out += buf; }
					else out += ch;
			}
		}
		out += '"';
	}

	void dump_to(std::string& out, int indent = -1, int depth = 0) const {
		auto newline = [&](int d) { if (indent >= 0) { out += '\n'; out.append((size_t)indent * d, ' '); } };
		switch (m_kind) {
			case Kind::Null: out += "null"; break;
			case Kind::Bool: out += m_bool ? "true" : "false"; break;
			case Kind::Number: {
				char buf[64];
				if (m_is_int || (std::floor(m_num) == m_num && std::fabs(m_num) < 1e15)) {
					snprintf(buf, sizeof(buf), "%lld", (long long)m_num);
					if (!m_is_int) { out += buf; out += ".0"; break; }
				} else if (m_is_float32) {
					snprintf(buf, sizeof(buf), "%.9g", m_num);
				} else {
					snprintf(buf, sizeof(buf), "%.17g", m_num);
				}
				out += buf;
				break;
			}
			case Kind::String: dump_string(m_str, out); break;
			case Kind::Binary: {
				out += "{\"bytes\":[";
				for (size_t i = 0; i < m_bin.size(); ++i) { if (i) out += ','; out += std::to_string((unsigned)m_bin[i]); }
				out += "],\"subtype\":null}";
				break;
			}
			case Kind::Array: {
				out += '[';
				for (size_t i = 0; i < m_arr.size(); ++i) { if (i) out += ','; newline(depth + 1); m_arr[i].dump_to(out, indent, depth + 1); }
				if (!m_arr.empty()) newline(depth);
				out += ']';
				break;
			}
			case Kind::Object: {
				out += '{';
				bool first = true;
				for (const auto& kv : m_obj) {
					if (!first) out += ',';
					first = false;
					newline(depth + 1);
					dump_string(kv.first, out);
					out += indent >= 0 ? ": " : ":";
					kv.second.dump_to(out, indent, depth + 1);
				}
				if (!m_obj.empty()) newline(depth);
				out += '}';
				break;
			}
		}
	}

	struct Parser {
		const std::string& s;
		size_t pos = 0;
		explicit Parser(const std::string& text) : s{text} {}

		[[noreturn]] void fail(const char* what) const {
			throw std::runtime_error{std::string{"json parse error at offset "} + std::to_string(pos) + ": " + what};
		}
		// white space and comments (// ... and /* ... */): the reference's samples parse their configs with skip_comments = true
		void skip_ws() {
			for (;;) {
				while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\n' || s[pos] == '\t' || s[pos] == '\r')) ++pos;
				if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '/') {
					while (pos < s.size() && s[pos] != '\n') ++pos;
				} else if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '*') {
					const size_t end = s.find("*/", pos + 2);
					pos = end == std::string::npos ? s.size() : end + 2;
				} else {
					return;
				}
			}
		}
		char peek() const { return pos < s.size() ? s[pos] : '\0'; }
		void expect(char ch) { if (peek() != ch) fail("unexpected character"); ++pos; }

		Json parse_value() {
			skip_ws();
			const char ch = peek();
			if (ch == '{') return parse_object();
			if (ch == '[') return parse_array();
			if (ch == '"') return Json(parse_string());
			if (s.compare(pos, 4, "true") == 0) { pos += 4; return Json(true); }
			if (s.compare(pos, 5, "false") == 0) { pos += 5; return Json(false); }
			if (s.compare(pos, 4, "null") == 0) { pos += 4; return Json{}; }
			return parse_number();
		}

		Json parse_number() {
			const size_t start = pos;
			bool is_int = true;
			if (peek() == '-' || peek() == '+') ++pos;
			while (pos < s.size()) {
				const char ch = s[pos];
				if (ch >= '0' && ch <= '9') { ++pos; }
				else if (ch == '.' || ch == 'e' || ch == 'E' || ch == '-' || ch == '+') { is_int = false; ++pos; }
				else break;
			}
			if (pos == start) fail("invalid value");
			char* end = nullptr;
			const std::string tok = s.substr(start, pos - start);
			const double v = std::strtod(tok.c_str(), &end);
			if (end == tok.c_str()) fail("invalid number");
			Json j(v);
			j.m_is_int = is_int;
			return j;
		}

		std::string parse_string() {
			expect('"');
			std::string out;
			while (true) {
				if (pos >= s.size()) fail("unterminated string");
				char ch = s[pos++];
				if (ch == '"') break;
				if (ch == '\\') {
					if (pos >= s.size()) fail("bad escape");
					const char e = s[pos++];
					switch (e) {
						case '"': out += '"'; break;
						case '\\': out += '\\'; break;
						case '/': out += '/'; break;
						case 'b': out += '\b'; break;
						case 'f': out += '\f'; break;
						case 'n': out += '\n'; break;
						case 'r': out += '\r'; break;
						case 't': out += '\t'; break;
						case 'u': {
							if (pos + 4 > s.size()) fail("bad \\u escape");
							const unsigned cp = (unsigned)std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16);
							pos += 4;
							if (cp < 0x80) out += (char)cp;
							else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
							else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
							break;
						}
						default: fail("bad escape");
					}
				} else {
					out += ch;
				}
			}
			return out;
		}

		Json parse_array() {
			expect('[');
			Json j = Json::array();
			skip_ws();
			if (peek() == ']') { ++pos; return j; }
			while (true) {
				j.m_arr.push_back(parse_value());
				skip_ws();
				if (peek() == ',') { ++pos; continue; }
				expect(']');
				break;
			}
			return j;
		}

		Json parse_object() {
			expect('{');
			Json j = Json::object();
			skip_ws();
			if (peek() == '}') { ++pos; return j; }
			while (true) {
				skip_ws();
				std::string key = parse_string();
				skip_ws();
				expect(':');
				Json v = parse_value();
				j[key] = std::move(v);
				skip_ws();
				if (peek() == ',') { ++pos; continue; }
				expect('}');
				break;
			}
			return j;
		}
	};

	Kind m_kind = Kind::Null;
	bool m_bool = false;
	double m_num = 0.0;
	bool m_is_int = false;
	bool m_is_float32 = false;
	std::string m_str;
	std::vector<Json> m_arr;
	std::vector<std::pair<std::string, Json>> m_obj;
	std::vector<uint8_t> m_bin;

public:
	// ---- MessagePack, the byte format callers store snapshots in (instant-ngp: json::to_msgpack(trainer->serialize())).
	// Encoding choices follow nlohmann::json so that the bytes match: object keys in sorted order (its objects are std::map),
	// integers in the smallest format, a float as float32 when that is exact and float64 otherwise, binary as bin8/16/32.
	static std::vector<uint8_t> to_msgpack(const Json& j) {
		std::vector<uint8_t> out;
		j.pack(out);
		return out;
	}
	static Json from_msgpack(const uint8_t* data, size_t size) {
		Unpacker u{data, size};
		Json j = u.value();
		if (u.pos != size) throw std::runtime_error{"msgpack: trailing bytes"};
		return j;
	}
	static Json from_msgpack(const std::vector<uint8_t>& bytes) { return from_msgpack(bytes.data(), bytes.size()); }

private:
	static void put_be(std::vector<uint8_t>& out, uint64_t v, int n_bytes) {
		for (int i = n_bytes - 1; i >= 0; --i) out.push_back((uint8_t)(v >> (8 * i)));
	}
	static void pack_length(std::vector<uint8_t>& out, size_t n, uint8_t fix_base, size_t fix_max, uint8_t code8, uint8_t code16, uint8_t code32) {
		if (fix_base && n <= fix_max) out.push_back((uint8_t)(fix_base | n));
		else if (code8 && n <= 0xff) { out.push_back(code8); out.push_back((uint8_t)n); }
		else if (n <= 0xffff) { out.push_back(code16); put_be(out, n, 2); }
		else { out.push_back(code32); put_be(out, n, 4); }
	}
	void pack(std::vector<uint8_t>& out) const {
		switch (m_kind) {
			case Kind::Null: out.push_back(0xc0); break;
			case Kind::Bool: out.push_back(m_bool ? 0xc3 : 0xc2); break;
			case Kind::Number: {
				if (m_is_int) {
					if (m_num >= 0) {
						const uint64_t v = (uint64_t)m_num;
						if (v < 128) out.push_back((uint8_t)v);
						else if (v <= 0xff) { out.push_back(0xcc); put_be(out, v, 1); }
						else if (v <= 0xffff) { out.push_back(0xcd); put_be(out, v, 2); }
						else if (v <= 0xffffffffull) { out.push_back(0xce); put_be(out, v, 4); }
						else { out.push_back(0xcf); put_be(out, v, 8); }
					} else {
						const int64_t v = (int64_t)m_num;
						if (v >= -32) out.push_back((uint8_t)v);
						else if (v >= -128) { out.push_back(0xd0); put_be(out, (uint64_t)v, 1); }
						else if (v >= -32768) { out.push_back(0xd1); put_be(out, (uint64_t)v, 2); }
						else if (v >= -2147483648ll) { out.push_back(0xd2); put_be(out, (uint64_t)v, 4); }
						else { out.push_back(0xd3); put_be(out, (uint64_t)v, 8); }
					}
				} else if ((double)(float)m_num == m_num) {
					const float f = (float)m_num;
					uint32_t bits;
					memcpy(&bits, &f, 4);
					out.push_back(0xca);
					put_be(out, bits, 4);
				} else {
					uint64_t bits;
					memcpy(&bits, &m_num, 8);
					out.push_back(0xcb);
					put_be(out, bits, 8);
				}
				break;
			}
			case Kind::String:
				pack_length(out, m_str.size(), 0xa0, 31, 0xd9, 0xda, 0xdb);
				out.insert(out.end(), m_str.begin(), m_str.end());
				break;
			case Kind::Binary:
				pack_length(out, m_bin.size(), 0, 0, 0xc4, 0xc5, 0xc6);
				out.insert(out.end(), m_bin.begin(), m_bin.end());
				break;
			case Kind::Array:
				pack_length(out, m_arr.size(), 0x90, 15, 0, 0xdc, 0xdd);
				for (const Json& e : m_arr) e.pack(out);
				break;
			case Kind::Object: {
				pack_length(out, m_obj.size(), 0x80, 15, 0, 0xde, 0xdf);
				std::vector<const std::pair<std::string, Json>*> sorted;
				for (const auto& kv : m_obj) sorted.push_back(&kv);
				for (size_t i = 1; i < sorted.size(); ++i) // insertion sort: snapshots have a handful of keys
					for (size_t k = i; k > 0 && sorted[k]->first < sorted[k - 1]->first; --k) std::swap(sorted[k], sorted[k - 1]);
				for (const auto* kv : sorted) {
					Json(kv->first).pack(out);
					kv->second.pack(out);
				}
				break;
			}
		}
	}

	struct Unpacker {
		const uint8_t* data;
		size_t size, pos = 0;
		Unpacker(const uint8_t* d, size_t n) : data{d}, size{n} {}
		void need(size_t n) const { if (size - pos < n) throw std::runtime_error{"msgpack: truncated input"}; }
		uint64_t be(int n_bytes) {
			need((size_t)n_bytes);
			uint64_t v = 0;
			for (int i = 0; i < n_bytes; ++i) v = v << 8 | data[pos++];
			return v;
		}
		Json string(size_t n) { need(n); Json j(std::string((const char*)data + pos, n)); pos += n; return j; }
		Json bin(size_t n) { need(n); Json j = Json::binary(std::vector<uint8_t>(data + pos, data + pos + n)); pos += n; return j; }
		Json array(size_t n) { Json j = Json::array(); for (size_t i = 0; i < n; ++i) j.push_back(value()); return j; }
		Json map(size_t n) {
			Json j = Json::object();
			for (size_t i = 0; i < n; ++i) {
				const Json key = value();
				j[key.as_string()] = value();
			}
			return j;
		}
		Json value() {
			need(1);
			const uint8_t c = data[pos++];
			if (c < 0x80) return Json((uint64_t)c);
			if (c >= 0xe0) return Json((int64_t)(int8_t)c);
			if ((c & 0xf0) == 0x80) return map(c & 0x0f);
			if ((c & 0xf0) == 0x90) return array(c & 0x0f);
			if ((c & 0xe0) == 0xa0) return string(c & 0x1f);
			switch (c) {
				case 0xc0: return Json{};
				case 0xc2: return Json(false);
				case 0xc3: return Json(true);
				case 0xc4: return bin((size_t)be(1));
				case 0xc5: return bin((size_t)be(2));
				case 0xc6: return bin((size_t)be(4));
				case 0xc7: { const size_t n = (size_t)be(1); be(1); return bin(n); } // ext: nlohmann writes binaries with a subtype this way
				case 0xc8: { const size_t n = (size_t)be(2); be(1); return bin(n); }
				case 0xc9: { const size_t n = (size_t)be(4); be(1); return bin(n); }
				case 0xca: { const uint32_t b = (uint32_t)be(4); float f; memcpy(&f, &b, 4); return Json(f); }
				case 0xcb: { const uint64_t b = be(8); double d; memcpy(&d, &b, 8); return Json(d); }
				case 0xcc: return Json((uint64_t)be(1));
				case 0xcd: return Json((uint64_t)be(2));
				case 0xce: return Json((uint64_t)be(4));
				case 0xcf: return Json((uint64_t)be(8));
				case 0xd0: return Json((int64_t)(int8_t)be(1));
				case 0xd1: return Json((int64_t)(int16_t)be(2));
				case 0xd2: return Json((int64_t)(int32_t)be(4));
				case 0xd3: return Json((int64_t)be(8));
				case 0xd4: be(1); return bin(1);
				case 0xd5: be(1); return bin(2);
				case 0xd6: be(1); return bin(4);
				case 0xd7: be(1); return bin(8);
				case 0xd8: be(1); return bin(16);
				case 0xd9: return string((size_t)be(1));
				case 0xda: return string((size_t)be(2));
				case 0xdb: return string((size_t)be(4));
				case 0xdc: return array((size_t)be(2));
				case 0xdd: return array((size_t)be(4));
				case 0xde: return map((size_t)be(2));
				case 0xdf: return map((size_t)be(4));
				default: throw std::runtime_error{"msgpack: unsupported type byte"};
			}
		}
	};
};

} // namespace tcnn_amd
