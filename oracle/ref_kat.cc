/*
 * ref_kat.cc -- known-answer generator that INCLUDES the reference's own headers from /root/reference
 * (nothing copied).  Built by `make -C oracle kat` into oracle/_ref/ref_kat; prints tests/golden/kat.json.
 * Build container only.
 */
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>
#include <random>
#include <stdexcept>
#include "arith/coder.h"
#include "arith/stat_adaptive.h"
#include "arith/model.h"
#include "formats/hry/prediction.h"
#include <iostream>
#include "structs/types.h"
#include "structs/attr.h"
#include "structs/quant.h"

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static std::string hex(const std::string &s) { static const char *d = "0123456789abcdef"; std::string o; for (unsigned char c : s) { o += d[c >> 4]; o += d[c & 15]; } return o; }

int main()
{
	std::mt19937_64 rng(12345);
	printf("{\n");
	// float residual folding
	printf(" \"delta_f32\": [\n");
	std::vector<std::pair<float, float>> fp = { {1.25f, 1.0f}, {-0.5f, 0.25f}, {0.f, 0.f}, {1.f, 0.f}, {-1.f, 0.f}, {0.f, 1.f}, {0.f, -1.f}, {3.4e38f, -3.4e38f}, {1e-40f, 1e-39f}, {-2.5f, -2.5f}, {-2.5f, -2.4999998f} };
	for (int i = 0; i < 40; ++i) { float a = u2f((uint32_t)rng()), b = u2f((uint32_t)rng()); if (a == a && b == b) fp.push_back({a, b}); }
	for (int i = 0; i < 40; ++i) { float a = (float)((int64_t)(rng() % 2001) - 1000) / 37.f; float b = a + (float)((int64_t)(rng() % 201) - 100) / 4096.f; fp.push_back({a, b}); }
	for (size_t i = 0; i < fp.size(); ++i) {
		float d = hry::pred::encodeDelta<float>(fp[i].first, fp[i].second, 0);
		float back = hry::pred::decodeDelta<float>(d, fp[i].second, 0);
		printf("  [%u, %u, %u, %u]%s\n", f2u(fp[i].first), f2u(fp[i].second), f2u(d), f2u(back), i + 1 < fp.size() ? "," : "");
	}
	printf(" ],\n \"delta_u\": [\n");
	// integer residual folding / predictor: [bytes, q, raw, pred, enc, dec(enc)]
	bool first = true;
	auto emit_u = [&](int bytes, int q, uint32_t raw, uint32_t pred) {
		uint32_t e, d;
		if (bytes == 1) { e = hry::pred::encodeDelta<uint8_t>(raw, pred, q); d = hry::pred::decodeDelta<uint8_t>(e, pred, q); }
		else if (bytes == 2) { e = hry::pred::encodeDelta<uint16_t>(raw, pred, q); d = hry::pred::decodeDelta<uint16_t>(e, pred, q); }
		else { e = hry::pred::encodeDelta<uint32_t>(raw, pred, q); d = hry::pred::decodeDelta<uint32_t>(e, pred, q); }
		printf("%s  [%d, %d, %u, %u, %u, %u]", first ? "" : ",\n", bytes, q, raw, pred, e, d);
		first = false;
	};
	emit_u(2, 14, 1000, 1003);
	int cfgs[][2] = { {1, 8}, {1, 5}, {1, 0}, {2, 14}, {2, 16}, {2, 9}, {2, 0}, {4, 20}, {4, 32}, {4, 0}, {4, 17} };
	for (auto &c : cfgs) {
		int bits = c[1] == 0 ? c[0] * 8 : c[1];
		uint64_t mx = bits == 32 ? 0xffffffffull : ((1ull << bits) - 1);
		uint32_t edge[] = { 0, 1, 2, (uint32_t)(mx / 2), (uint32_t)(mx - 1), (uint32_t)mx };
		for (uint32_t a : edge) for (uint32_t b : edge) emit_u(c[0], c[1], a, b);
		for (int i = 0; i < 12; ++i) emit_u(c[0], c[1], (uint32_t)(rng() % (mx + 1)), (uint32_t)(rng() % (mx + 1)));
	}
	printf("\n ],\n \"predict_u\": [\n");
	first = true;
	auto emit_p = [&](int bytes, int q, uint32_t a, uint32_t b, uint32_t c) {
		uint32_t r;
		if (bytes == 1) r = hry::pred::predict<uint8_t>(a, b, c, q); else if (bytes == 2) r = hry::pred::predict<uint16_t>(a, b, c, q); else r = hry::pred::predict<uint32_t>(a, b, c, q);
		printf("%s  [%d, %d, %u, %u, %u, %u]", first ? "" : ",\n", bytes, q, a, b, c, r);
		first = false;
	};
	emit_p(2, 14, 10, 16380, 3); emit_p(2, 14, 5, 3, 100);
	for (auto &c : cfgs) {
		int bits = c[1] == 0 ? c[0] * 8 : c[1];
		uint64_t mx = bits == 32 ? 0xffffffffull : ((1ull << bits) - 1);
		for (int i = 0; i < 16; ++i) {
			uint32_t a = rng() % (mx + 1), b = rng() % (mx + 1), d = rng() % (mx + 1);
			if (i < 4) { a = (i & 1) ? (uint32_t)mx : 0; }
			emit_p(c[0], c[1], a, b, d);
		}
	}
	printf("\n ],\n \"predict_f32\": [\n");
	for (int i = 0; i < 24; ++i) {
		float a = (float)((int64_t)(rng() % 20001) - 10000) / 317.f, b = (float)((int64_t)(rng() % 20001) - 10000) / 311.f, c = (float)((int64_t)(rng() % 20001) - 10000) / 313.f;
		printf("  [%u, %u, %u, %u]%s\n", f2u(a), f2u(b), f2u(c), f2u(hry::pred::predict<float>(a, b, c, 0)), i < 23 ? "," : "");
	}
	printf(" ],\n \"requant_f32\": [\n");
	// scalar float quantisation exactly as quant.h:134-136 evaluates it
	for (int i = 0; i < 48; ++i) {
		float mn = (float)((int64_t)(rng() % 2001) - 1000) / 129.f, sc = (float)(rng() % 1000 + 1) / 77.f;
		float v = mn + sc * (float)(rng() % 100001) / 100000.f;
		if (i % 8 == 0) v = mn; if (i % 8 == 1) v = mn + sc;
		int q = (int)(rng() % 30) + 1;
		uint64_t r = quant::rescale<float>(v - mn, sc, (1 << (uint32_t)q) - 1) + 0.5f;
		printf("  [%u, %u, %u, %d, %llu]%s\n", f2u(v), f2u(mn), f2u(sc), q, (unsigned long long)r, i < 47 ? "," : "");
	}
	printf(" ],\n \"range_bytes\": [\n");
	// adaptive 256-ary model + 64-bit coder over byte strings (arith/README usage)
	std::vector<std::string> strs = { "Hello, my name is Max!", "", "a", std::string(300, 'z'), std::string("\x00\xff\x00\xff\x80", 5) };
	{ std::string s; for (int i = 0; i < 2000; ++i) s += (char)(rng() % 7 == 0 ? rng() % 256 : rng() % 4); strs.push_back(s); }
	{ std::string s; for (int i = 0; i < 500; ++i) s += (char)(rng() % 256); strs.push_back(s); }
	for (size_t k = 0; k < strs.size(); ++k) {
		std::ostringstream os;
		{
			arith::Encoder<> coder(os);
			arith::ModelMult<uint8_t, arith::AdaptiveStatisticsModule<>> model;
			for (unsigned char c : strs[k]) model.encode<uint8_t>(coder, c);
			coder.flush();
		}
		printf("  [\"%s\", \"%s\"]%s\n", hex(strs[k]).c_str(), hex(os.str()).c_str(), k + 1 < strs.size() ? "," : "");
	}
	printf(" ],\n \"range_lht\": [\n");
	// raw coder on explicit triples, including the h == t branch and tiny/huge totals
	for (int k = 0; k < 6; ++k) {
		std::ostringstream os;
		std::vector<uint64_t> tr;
		{
			arith::Encoder<> coder(os);
			int n = 50 + 200 * k;
			for (int i = 0; i < n; ++i) {
				uint64_t t = k == 0 ? 2 : (k == 1 ? (1ull << 32) - 5 : 1 + rng() % (1ull << (4 + 5 * k)));
				uint64_t l = rng() % t, h = l + 1 + rng() % (t - l);
				if (i % 5 == 0) h = t;
				coder(l, h, t);
				tr.push_back(l); tr.push_back(h); tr.push_back(t);
			}
			coder.flush();
		}
		printf("  [[");
		for (size_t i = 0; i < tr.size(); ++i) printf("%llu%s", (unsigned long long)tr[i], i + 1 < tr.size() ? "," : "");
		printf("], \"%s\"]%s\n", hex(os.str()).c_str(), k < 5 ? "," : "");
	}
	printf(" ],\n \"range_bytes32\": [\n");
	// the same coder and model templates instantiated with 32-bit registers (arith::Encoder<uint32_t>): what every stream of
	// the chunked container uses
	{ std::string s; for (int i = 0; i < 40000; ++i) s += (char)(rng() % 5 == 0 ? rng() % 256 : rng() % 3); strs.push_back(s); }
	for (size_t k = 0; k < strs.size(); ++k) {
		std::ostringstream os;
		{
			arith::Encoder<uint32_t> coder(os);
			arith::ModelMult<uint8_t, arith::AdaptiveStatisticsModule<uint32_t>, uint32_t> model;
			for (unsigned char c : strs[k]) model.encode<uint8_t>(coder, c);
			coder.flush();
		}
		// the decoder of the same instantiation must invert it (checked here so that the fixture is known to be decodable)
		{
			std::istringstream is(os.str());
			arith::Decoder<uint32_t> dec(is);
			arith::ModelMult<uint8_t, arith::AdaptiveStatisticsModule<uint32_t>, uint32_t> model;
			for (unsigned char c : strs[k]) if (model.decode<uint8_t>(dec) != c) { fprintf(stderr, "32-bit coder does not round-trip\n"); return 1; }
		}
		printf("  [\"%s\", \"%s\"]%s\n", hex(strs[k]).c_str(), hex(os.str()).c_str(), k + 1 < strs.size() ? "," : "");
	}
	printf(" ],\n \"range_lht32\": [\n");
	for (int k = 0; k < 6; ++k) {
		std::ostringstream os;
		std::vector<uint64_t> tr;
		{
			arith::Encoder<uint32_t> coder(os);
			int n = 50 + 200 * k;
			for (int i = 0; i < n; ++i) {
				uint32_t t = k == 0 ? 2 : (k == 1 ? (1u << 30) - 5 : 1 + (uint32_t)(rng() % (1ull << (4 + 4 * k))));
				uint32_t l = (uint32_t)(rng() % t), h = l + 1 + (uint32_t)(rng() % (t - l));
				if (i % 5 == 0) h = t;
				coder(l, h, t);
				tr.push_back(l); tr.push_back(h); tr.push_back(t);
			}
			coder.flush();
		}
		printf("  [[");
		for (size_t i = 0; i < tr.size(); ++i) printf("%llu%s", (unsigned long long)tr[i], i + 1 < tr.size() ? "," : "");
		printf("], \"%s\"]%s\n", hex(os.str()).c_str(), k < 5 ? "," : "");
	}
	printf(" ]\n}\n");
	return 0;
}
