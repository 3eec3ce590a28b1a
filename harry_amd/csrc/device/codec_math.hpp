// Scalar arithmetic of the attribute path, usable from HIP kernels (and host code of the product).
// Behavioural contract: formats/hry/transform.h:19-48 (ordered-int map of floats), formats/hry/prediction.h:21-147
// (balanced residual folding, saturating parallelogram predictor), structs/quant.h:98-136 (float -> uintN).
// Must be compiled with -ffp-contract=off: every fp operation is individually rounded.
#pragma once
#include <cstdint>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define HRY_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define HRY_HD inline
#endif

namespace hry {
namespace cm {

template <int N> struct word;
template <> struct word<1> { typedef uint8_t u; typedef int8_t s; };
template <> struct word<2> { typedef uint16_t u; typedef int16_t s; };
template <> struct word<4> { typedef uint32_t u; typedef int32_t s; };
template <> struct word<8> { typedef uint64_t u; typedef int64_t s; };

template <typename T> struct is_fp { static constexpr bool value = false; };
template <> struct is_fp<float> { static constexpr bool value = true; };
template <> struct is_fp<double> { static constexpr bool value = true; };

template <typename To, typename From> HRY_HD To bits(From v)
{
	static_assert(sizeof(To) == sizeof(From), "size");
	To r;
	__builtin_memcpy(&r, &v, sizeof(To));
	return r;
}

// order-preserving map float bits -> signed int: negative values get their magnitude bits inverted (transform.h:19-23)
HRY_HD uint32_t ordered_from_f32(float f)
{
	uint32_t b = bits<uint32_t>(f);
	return b ^ ((uint32_t)(0u - (b >> 31)) >> 1);
}
HRY_HD float f32_from_ordered(uint32_t o) { return bits<float>(o ^ ((uint32_t)(0u - (o >> 31)) >> 1)); }

template <typename T> HRY_HD int width_bits(int q) { return q == 0 ? (int)sizeof(T) * 8 : q; }
template <typename T> HRY_HD T ones(int nbits) { return nbits == (int)(sizeof(T) * 8) ? T(-1) : T((1 << nbits) - 1); }   // prediction.h:27-31

// balanced residual code of an integer against its prediction (prediction.h:81-99); expressions keep the
// reference's operand types so that promotion and truncation are identical for every width
template <typename T> HRY_HD T fold_int(const T raw, const T pred, int nbits)
{
	const T room = ones<T>(nbits) - pred;
	if (pred == T(0)) return raw;
	const T bal = T(pred) < room ? T(pred) : room;
	if (raw < pred) {
		const T d = pred - raw;
		if (d > bal) return d + bal;
		return T(d << 1) - 1;
	}
	const T d = raw - pred;
	if (d > bal) return d + bal;
	return T(d << 1);
}
// inverse (prediction.h:46-64)
template <typename T> HRY_HD T unfold_int(const T code, const T pred, int nbits)
{
	const T room = ones<T>(nbits) - pred;
	if (pred == T(0)) return code;
	const T pm1 = T(pred - T(1));
	const T bal = pm1 < room ? pm1 : room;
	if ((code >> 1) > bal) {
		if (room >= pred) return pred + code - bal - T(1);
		return pred - code + bal;
	}
	const T flip = (code & 1) ? T(~T(0)) : T(0);
	return pred + (T(code >> 1) ^ flip);
}

// residual of a stored value of type T against its prediction, returned as the stored bit pattern.
// float: both operands go through the ordered-int map; the reference's sign-flip table is indexed off by one so
// that NO sign flip is applied to 4-byte values (prediction.h:33-44).
template <typename T> HRY_HD typename word<sizeof(T)>::u residual_bits(T raw, T pred, int q)
{
	if constexpr (is_fp<T>::value) {
		static_assert(sizeof(T) == 4, "8-byte floating residuals are unspecified in the reference");
		return fold_int<uint32_t>(ordered_from_f32(raw), ordered_from_f32(pred), width_bits<T>(q));
	} else {
		return bits<typename word<sizeof(T)>::u>(fold_int<T>(raw, pred, width_bits<T>(q)));
	}
}
template <typename T> HRY_HD T value_from_residual(typename word<sizeof(T)>::u code, T pred, int q)
{
	if constexpr (is_fp<T>::value) {
		static_assert(sizeof(T) == 4, "8-byte floating residuals are unspecified in the reference");
		return f32_from_ordered(unfold_int<uint32_t>(code, ordered_from_f32(pred), width_bits<T>(q)));
	} else {
		return unfold_int<T>(bits<T>(code), pred, width_bits<T>(q));
	}
}

// parallelogram rule v0 + v1 - v2 (prediction.h:121-147): clamped to [0, 2^bits-1] for integers
template <typename T> HRY_HD T parallelogram(const T v0, const T v1, const T v2, int q)
{
	if constexpr (is_fp<T>::value) {
		return v0 + (v1 - v2);
	} else {
		const T top = ones<T>(width_bits<T>(q));
		if (v1 < v2) {
			const T d = v2 - v1;
			if (d > v0) return T(0);
			return v0 - d;
		}
		const T d = v1 - v2;
		const T v = v0 + d;
		if ((v > top) || (v < v0)) return top;
		return v;
	}
}

// accumulator type for candidate means (mixing.h:110-127)
template <typename T> struct wide { typedef int64_t type; };
template <> struct wide<float> { typedef double type; };
template <> struct wide<double> { typedef double type; };
template <> struct wide<uint64_t> { typedef uint64_t type; };
HRY_HD double mean_of(double sum, double n) { return sum / n; }                       // transform.h:90
HRY_HD int64_t mean_of(int64_t sum, int64_t n) { return (sum + (n >> 1)) / n; }       // transform.h:91
HRY_HD uint64_t mean_of(uint64_t sum, uint64_t n) { return (sum + (n >> 1)) / n; }

// float -> q-bit unsigned (quant.h:98-102,134-136): ((v - min) / scale) * float(2^q - 1) + 0.5f, truncated
HRY_HD uint64_t quantise_f32(float v, float mn, float scale, int q)
{
	float levels = (float)((1 << (uint32_t)q) - 1);
	float x = (v - mn) / scale * levels + 0.5f;
	return (uint64_t)x;
}

// ---- exact division by a context total through a precomputed reciprocal ---------------------------------
// r = floor(n / t) for every n <= 2^63 (the coder's range register never exceeds HALF, coder.h:47,92-101) and 2 <= t < 2^32:
//   t not a power of two: s = floor(log2 t), magic = floor(2^(64+s) / t) + 1 = 2^(64+s)/t + e with 0 < e <= 1; mulhi(magic, n)
//   >> s overshoots n / t by n e / 2^(64+s) <= 2^63 / 2^(64+s) = 1 / 2^(s+1) < 1 / t, so the floor is unchanged;
//   t = 2^s: magic = 2^63 is the exact reciprocal at shift s - 1.
// (One multiplication and one shift in the serial chain; the 65-bit reciprocal that is exact for every 64-bit n costs three
// more dependent operations per symbol.)
HRY_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#ifdef __HIP_DEVICE_COMPILE__
	return __umul64hi(a, b);
#else
	return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
HRY_HD uint64_t div_by_magic(uint64_t n, uint64_t magic, uint32_t shift)
{
	return mulhi64(magic, n) >> shift;
}
// reciprocal of t (2 <= t < 2^32)
HRY_HD void make_magic(uint32_t t, uint64_t &magic, uint32_t &shift)
{
	uint32_t k = 31u - (uint32_t)__builtin_clz(t);
	if ((t & (t - 1)) == 0) { magic = 1ull << 63; shift = k - 1; return; }
	// floor(2^(64+k) / t) in two 64-bit steps (2^(32+k) / t < 2^32)
	uint64_t n1 = (uint64_t)(1u << k) << 32;
	uint64_t q1 = n1 / t, r1 = n1 % t;
	uint64_t n0 = r1 << 32;
	uint64_t q0 = n0 / t;
	magic = ((q1 << 32) | q0) + 1;
	shift = k;
}

// ---- the same for 32-bit coder registers (arith::Encoder<uint32_t>, the streams of the chunked container) ----
// r = floor(R / t) == mulhi32(R, m) >> sh for every R <= 2^31 and 2 <= t < 2^31:
//   t not a power of two: s = floor(log2 t), m = floor(2^(32+s) / t) + 1 = 2^(32+s)/t + e with 0 < e <= 1; the product
//   overshoots R/t by R e / 2^(32+s) < 1/t because R <= 2^31 < 2^(32+s)/t, so the floor is unchanged;
//   t = 2^s: m = 2^31 is the exact reciprocal at shift s - 1.
// t == 1 gets m = 0 (r = 0): the only codable symbol then has l = 0 and h = t, for which r is never used.
HRY_HD void make_magic32(uint32_t t, uint32_t &m, uint32_t &sh)
{
	if (t < 2) { m = 0; sh = 0; return; }
	uint32_t s = 31u - (uint32_t)__builtin_clz(t);
	if ((t & (t - 1)) == 0) { m = 1u << 31; sh = s - 1; return; }
	m = (uint32_t)((1ull << (32 + s)) / t) + 1u;
	sh = s;
}
HRY_HD uint32_t div_by_magic32(uint32_t n, uint32_t m, uint32_t sh)
{
#ifdef __HIP_DEVICE_COMPILE__
	return __umulhi(n, m) >> sh;
#else
	return (uint32_t)(((uint64_t)n * m) >> 32) >> sh;
#endif
}

}   // namespace cm
}   // namespace hry
