// Exactness of the reciprocal the serial range recurrence divides by (codec_math.hpp: make_magic / div_by_magic) against the
// hardware division, over every total below 70 000 at the boundaries of the range register and random totals below 2^32.
// Built and run by tests/test_host_cpu.py.
#include <cstdint>
#include <cstdio>
#include <random>
#include "codec_math.hpp"
int main() {
	std::mt19937_64 g(1);
	uint64_t bad = 0, n_checked = 0;
	auto check = [&](uint32_t t, uint64_t n) {
		uint64_t m; uint32_t sh; hry::cm::make_magic(t, m, sh);
		if (hry::cm::div_by_magic(n, m, sh) != n / t) { if (bad < 5) printf("BAD t=%u n=%llu\n", t, (unsigned long long)n); ++bad; }
		++n_checked;
	};
	for (uint32_t t = 2; t < 70000; ++t) {
		check(t, 1ull << 63); check(t, (1ull << 63) - 1); check(t, (1ull << 62) + 1);
		uint64_t k = (1ull << 63) / t; check(t, k * t); check(t, k * t - 1); if (k * t + t - 1 <= (1ull << 63)) check(t, k * t + t - 1);
		check(t, 0); check(t, t - 1); check(t, t);
	}
	for (int i = 0; i < 3000000; ++i) {
		uint32_t t = (uint32_t)(g() >> (32 + (g() % 31))); if (t < 2) t = 2;
		uint64_t n = (g() >> 1) ; if (n > (1ull << 63)) n = 1ull << 63;
		check(t, n);
		uint64_t k = n / t; check(t, k * t); if (k) check(t, k * t - 1);
		check(0xffffffffu - (uint32_t)(i & 1023), n);
	}
	printf("%llu checked, %llu bad\n", (unsigned long long)n_checked, (unsigned long long)bad);
	return bad != 0;
}
