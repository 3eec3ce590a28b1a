// Decoder-side attribute reconstruction (formats/hry/attrcode.h:443-550, prediction.h:46-78).
//
// The decoder numbers vertices in decode order, so vertex id == traversal rank == attribute index
// (cbm/decoder.h:48-75,145; attrcode.h:454).  A vertex is predicted from already reconstructed neighbours, and in a
// cut-border traversal nearly every vertex depends on its immediate predecessor: reconstruction is one dependency
// chain per mesh (SURVEY.md finding 0-2).  What can be parallel is done in parallel:
//   k_candidates       : thread per vertex, fan walk on the complete connectivity -> candidate triples (ids < v)
//   k_residuals_to_rec : byte planes -> residual codes in the attribute records
//   k_faces_unfold     : faces have no candidates (App. B-16) -> independent
//   k_unpredict        : the chain.  One wavefront, lane c owns component c (components are independent chains
//                        sharing the candidate list); reconstructed values of the last N vertices live in LDS.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <stdexcept>
#include <type_traits>

#include "codec_math.hpp"
#include "dev_types.hpp"
#include "kernels.hpp"

namespace hry {
namespace dev {

template <typename T> __device__ __forceinline__ T ldq(const uint8_t *p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }
template <typename T> __device__ __forceinline__ void stq(uint8_t *p, T v) { __builtin_memcpy(p, &v, sizeof(T)); }

struct TopoD {
	ConnView c;
	__device__ __forceinline__ uint32_t next(uint32_t e) const
	{
		if (c.eface) { uint32_t f = c.eface[e]; return e + 1 == c.foff[f + 1] ? c.foff[f] : e + 1; }
		uint32_t k = e % c.udeg;
		return k + 1 == c.udeg ? e - k : e + 1;
	}
	__device__ __forceinline__ uint32_t prev(uint32_t e) const
	{
		if (c.eface) { uint32_t f = c.eface[e]; return e == c.foff[f] ? c.foff[f + 1] - 1 : e - 1; }
		uint32_t k = e % c.udeg;
		return k == 0 ? e + c.udeg - 1 : e - 1;
	}
	__device__ __forceinline__ uint32_t degree(uint32_t e) const
	{
		if (c.eface) { uint32_t f = c.eface[e]; return c.foff[f + 1] - c.foff[f]; }
		return c.udeg;
	}
};

// fan order of attrcode.h:83-106,155-171; a candidate needs all three vertices decoded earlier (id < v)
template <typename F> __device__ __forceinline__ void fan_ids(const TopoD &tp, uint32_t ein, uint32_t v, F &&f)
{
	auto offer = [&](uint32_t a, uint32_t b, uint32_t o) { if (a < v && b < v && o < v) f(a, b, o); };
	// A twin at or beyond c.ne is no neighbour: the pipelined decode has uploaded the connectivity up to there only (c.ne = the
	// half-edges on the device).  Such a link was made after the vertex became final and leads to a face with a younger vertex,
	// which the rank filter would reject -- but what lies behind it on the device is not that face yet, it is whatever the buffer
	// held before.  (Elsewhere c.ne is the table's size and this reads a damaged entry as a border.)
	auto twin_of = [&](uint32_t e) { const uint32_t t = tp.c.twin[e]; return t < tp.c.ne ? t : e; };
	auto visit = [&](uint32_t e) {
		uint32_t d = tp.degree(e);
		if (d == 3) {
			uint32_t e1 = tp.next(e), t = twin_of(e1);
			if (t == e1) return;
			uint32_t tn = tp.next(t);
			offer(tp.c.org[t], tp.c.org[tn], tp.c.org[tp.next(tn)]);
			return;
		}
		uint32_t e0 = tp.next(e), e1 = tp.prev(e);
		uint32_t a = tp.c.org[e0], b = tp.c.org[e1];
		offer(a, b, tp.c.org[tp.next(e0)]);
		if (d > 4) offer(a, b, b);
	};
	const int kMaxSteps = 1 << 16;
	uint32_t e = ein, t;
	int steps = 0;
	bool border = false;
	for (;;) {
		visit(e);
		t = twin_of(e);
		if (t == e) { border = true; break; }
		e = tp.next(t);
		if (e == ein || ++steps > kMaxSteps) break;
	}
	if (!border) return;
	e = tp.prev(ein);
	t = twin_of(e);
	if (e == t) return;
	e = t;
	do {
		visit(e);
		e = tp.prev(e);
		t = twin_of(e);
		if (e == t) break;
		e = t;
	} while (e != ein && ++steps <= kMaxSteps);
}

constexpr int kCandMax = 8;
// Candidate table of the decoder (k_candidates_ids): per vertex the ids of its FIRST TWO candidate triples (24 bytes -- a regular
// mesh uses no more) and the candidate count; a vertex with three to eight candidates has its whole row (eight triples, 96
// bytes) in an overflow area, its compact row holds the row's number there.  Round 2 wrote eight triples for every vertex: 96
// bytes written and read back per vertex, ten times the algorithmic bytes of a decode.  Layout of the one allocation:
// nvtx x 6 words | 16 words (word 0: rows handed out) | overflow rows of 24 words.
constexpr int kCand2 = 6;
__host__ __device__ __forceinline__ size_t cand_over_at(uint32_t nvtx_total) { return (size_t)nvtx_total * kCand2; }   // in words
__device__ __forceinline__ const uint32_t *cand_full_row(const uint32_t *cand, uint32_t nvtx_total, uint32_t v)
{
	return cand + cand_over_at(nvtx_total) + 16 + (size_t)cand[(size_t)v * kCand2] * (kCandMax * 3);
}
// the row that holds candidate triples 0 .. n - 1 of vertex v
__device__ __forceinline__ const uint32_t *cand_row(const uint32_t *cand, uint32_t nvtx_total, uint32_t v, uint32_t n)
{
	return n > 2 ? cand_full_row(cand, nvtx_total, v) : cand + (size_t)v * kCand2;
}

__global__ __launch_bounds__(256) void k_residuals_to_rec(const uint8_t *planes, uint32_t n, ListDesc ld, uint8_t *rec)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint8_t *r = rec + (size_t)i * ld.stride;
	for (int c = 0; c < ld.ncomp; ++c) {
		int nb = ld.stype[c] == 0 || ld.stype[c] == 4 || ld.stype[c] == 5 ? 4 : ld.stype[c] == 6 || ld.stype[c] == 7 ? 2 : ld.stype[c] >= 8 ? 1 : 8;
		for (int b = 0; b < nb; ++b) r[ld.off[c] + b] = planes[(size_t)(ld.plane[c] + b) * n + i];
	}
}

template <typename F> __device__ __forceinline__ void with_st(int st, F &&f)
{
	switch (st) {
	case 0: f(float()); break;
	case 2: f(uint64_t()); break;
	case 3: f(int64_t()); break;
	case 4: f(uint32_t()); break;
	case 5: f(int32_t()); break;
	case 6: f(uint16_t()); break;
	case 7: f(int16_t()); break;
	case 8: f(uint8_t()); break;
	case 9: f(int8_t()); break;
	default: break;
	}
}

__global__ __launch_bounds__(256) void k_faces_unfold(uint32_t n, ListDesc ld, uint8_t *rec)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint8_t *r = rec + (size_t)i * ld.stride;
	for (int c = 0; c < ld.ncomp; ++c)
		with_st(ld.stype[c], [&](auto tag) {
			typedef decltype(tag) T;
			typedef typename cm::word<sizeof(T)>::u U;
			stq<T>(r + ld.off[c], cm::value_from_residual<T>(ldq<U>(r + ld.off[c]), T(0), ld.quant[c]));
		});
}

// small exact division for candidate means: sums of <= 8 values of <= 16 bits stay below 2^20
__device__ __forceinline__ uint32_t div_small(uint32_t x, uint32_t n)
{
	// floor(x / n) == (x * ceil(2^32 / n)) >> 32 whenever x * n < 2^32
	// ceil(2^32 / n) for n = 2 .. 8 spelled out: the 64-bit division that would compute it costs more than a vertex of the chain
	const uint32_t inv = n == 2 ? 0x80000000u : n == 3 ? 0x55555556u : n == 4 ? 0x40000000u : n == 5 ? 0x33333334u : n == 6 ? 0x2aaaaaabu : n == 7 ? 0x24924925u : n == 8 ? 0x20000000u
	                     : n <= 1 ? 0u : (uint32_t)(0x100000000ull / n) + 1u;
	return n <= 1 ? x : __umulhi(x, inv);
}

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane_idx) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane_idx); }

// ---------------------------------------------------------------------------------------------------------
// k_unpredict2: the reconstruction chain.  Components are independent chains that share only the candidate lists,
// so every component gets its own wavefront (one block per component, own LDS ring).  A single wavefront issues
// about one instruction every 5-8 cycles whatever the instruction is, so the chain is organised to need as few
// instructions per vertex as possible: a batch of 64 vertices is handled "systolically".
//   prepare (lane j <-> vertex base+j, all lanes in parallel): gather every prediction source that lies BEFORE the
//           batch (LDS ring, or HBM when older than the ring) into registers; a source inside the batch is recorded
//           as the lane that will produce it.  The residual code is read straight from the byte planes.
//   chain   (step i = 0..nb-1, fully unrolled): EVERY lane evaluates its own vertex from the sources it has (~20
//           vector instructions for 16-bit components); lane i holds all its sources by then, so its value is final.
//           The value is broadcast (v_readlane) and every lane that waits for lane i picks it up (compare + select per
//           source).  No memory access, no branch and no scalar<->vector hand-over sits on the dependency chain.
// Vertices with more than two candidates (0.3 % of a regular triangle mesh) end a batch and are evaluated on their own,
// candidates across lanes; all their sources are older than the batch by construction.
// Exactly the arithmetic of attrcode.h:182-208 / prediction.h:46-78,121-147 per vertex, in coding order.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t med3_i32(int32_t x, int32_t lo, int32_t hi)
{
	int32_t r;
	asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
	return (uint32_t)r;
}

template <typename T>
__device__ __forceinline__ T chain_predict(uint32_t ncu, const T *pv)
{
	typedef typename cm::wide<T>::type W;
	if (ncu == 0) return T(0);
	W acc = 0;
#pragma unroll
	for (int k = 0; k < kCandMax; ++k) if ((uint32_t)k < ncu) acc = acc + (W)pv[k];
	T avg;
	if constexpr (!cm::is_fp<T>::value && !(T(-1) < T(0))) {
		// unsigned sums of <= 8 values: (sum + n/2) / n without the 64-bit divide
		uint64_t sum = (uint64_t)acc + (ncu >> 1);
		if (ncu == 1) avg = (T)sum;
		else if (ncu == 2) avg = (T)(sum >> 1);
		else if (ncu == 4) avg = (T)(sum >> 2);
		else if (ncu == 8) avg = (T)(sum >> 3);
		else if (sizeof(T) <= 2) avg = (T)div_small((uint32_t)sum, ncu);
		else avg = (T)(sum / ncu);
	} else avg = (T)cm::mean_of(acc, (W)ncu);
	if constexpr (!cm::is_fp<T>::value) return avg;
	else {
		T best = 3.402823466e+38f;
#pragma unroll
		for (int k = 0; k < kCandMax; ++k) {
			if ((uint32_t)k < ncu) {
				T db = avg > best ? avg - best : best - avg;
				T dp = avg > pv[k] ? avg - pv[k] : pv[k] - avg;
				best = db < dp ? best : pv[k];
			}
		}
		return best;
	}
}

// the same for a number of candidates known at compile time; pk: lane k holds candidate k
template <typename T, int N>
__device__ __forceinline__ T chain_predict_n(uint32_t pk)
{
	typedef typename cm::word<sizeof(T)>::u U;
	T pv[kCandMax];
#pragma unroll
	for (int k = 0; k < kCandMax; ++k) pv[k] = k < N ? cm::bits<T>((U)rl(pk, k)) : T(0);
	if constexpr (!cm::is_fp<T>::value) return chain_predict<T>((uint32_t)N, pv);
	else {
		double acc = 0;
#pragma unroll
		for (int k = 0; k < N; ++k) acc = acc + (double)pv[k];
		// acc / N, correctly rounded, without the division (20 issue slots): with y = RN(1 / N), q0 = RN(acc y), r = acc - N q0 (exact in
		// an fma), RN(q0 + r y) is the correctly rounded quotient (Markstein); zeros, infinities and NaNs are q0 already
		double mean;
		if constexpr ((N & (N - 1)) == 0) mean = acc / (double)N;
		else {
			const double y = 1.0 / (double)N, q0 = acc * y;
			const double r = __builtin_fma(-(double)N, q0, acc), q1 = __builtin_fma(r, y, q0);
			mean = (__builtin_fabs(q0) > 0.0 && __builtin_fabs(q0) < __builtin_inf()) ? q1 : q0;
		}
		const T avg = (T)mean;
		T best = 3.402823466e+38f;
#pragma unroll
		for (int k = 0; k < N; ++k) {
			const T db = avg > best ? avg - best : best - avg;
			const T dp = avg > pv[k] ? avg - pv[k] : pv[k] - avg;
			best = db < dp ? best : pv[k];
		}
		return best;
	}
}

// ---- per-lane evaluation of one vertex with at most two candidates: six source bit patterns -> value bit pattern.
// Three tiers: unsigned components of at most 16 bits (quantised attributes, the headline case) in trimmed 32-bit
// arithmetic, float / 32-bit unsigned branch-free, every other type through the generic functions of codec_math.hpp.
template <typename T> struct LaneEval {
	typedef typename cm::word<sizeof(T)>::u U;
	int q;
	uint32_t nc, code;
	__device__ __forceinline__ void setup(uint32_t nc_, uint32_t code_, int q_) { nc = nc_; code = code_; q = q_; }
	__device__ __forceinline__ uint32_t eval(const uint32_t (&s)[6]) const
	{
		T p0 = cm::parallelogram<T>(cm::bits<T>((U)s[0]), cm::bits<T>((U)s[1]), cm::bits<T>((U)s[2]), q);
		T p1 = cm::parallelogram<T>(cm::bits<T>((U)s[3]), cm::bits<T>((U)s[4]), cm::bits<T>((U)s[5]), q);
		typedef typename cm::wide<T>::type W;
		T two = (T)cm::mean_of((W)p0 + (W)p1, (W)2);
		T pred = nc == 2 ? two : nc == 1 ? p0 : T(0);
		return (uint32_t)cm::bits<U>(cm::value_from_residual<T>((U)code, pred, q));
	}
};

// prediction.h:46-64 for unsigned words, far branches reduced (room >= pred: bal = pred - 1, pred + code - bal - 1 == code;
// room < pred: bal = room, pred - code + bal == top - code); everything that does not depend on the prediction is
// computed once per vertex
struct UnfoldPre {
	uint32_t code, half, top_minus_code, delta;
	__device__ __forceinline__ void setup(uint32_t c, uint32_t top, uint32_t wrap_mask)
	{
		code = c;
		half = c >> 1;
		top_minus_code = (top - c) & wrap_mask;
		delta = (c & 1u) ? 0u - half - 1u : half;
	}
	__device__ __forceinline__ uint32_t apply(uint32_t pred, uint32_t top) const
	{
		uint32_t room = top - pred;
		uint32_t pm1 = pred - 1u;
		uint32_t bal = pm1 < room ? pm1 : room;
		uint32_t far = room >= pred ? code : top_minus_code;
		uint32_t r = half > bal ? far : pred + delta;
		return pred == 0 ? code : r;
	}
};

template <typename T> struct LaneEvalSmall {   // uint8_t / uint16_t
	uint32_t top;
	UnfoldPre uf;
	__device__ __forceinline__ void setup(uint32_t nc_, uint32_t code_, int q_)
	{
		top = (uint32_t)cm::ones<T>(q_ == 0 ? (int)sizeof(T) * 8 : q_);
		uf.setup(code_, top, (uint32_t)(T)(~T(0)));
	}
	// The caller duplicates the single candidate of a vertex with one candidate ((2p + 1) >> 1 == p) and zeroes the
	// sources of a vertex without candidates, so the mean of two is the prediction in every case.
	__device__ __forceinline__ uint32_t eval(const uint32_t (&s)[6]) const
	{
		// prediction.h:121-138: v1 < v2 -> max(0, v0 - d), else min(top, v0 + d) (type wrap of v0 + d included); with all
		// three values in [0, top] both are the clamp of v0 + v1 - v2 to [0, top]
		uint32_t p0 = med3_i32((int32_t)(s[0] + s[1] - s[2]), 0, (int32_t)top);
		uint32_t p1 = med3_i32((int32_t)(s[3] + s[4] - s[5]), 0, (int32_t)top);
		return uf.apply((p0 + p1 + 1u) >> 1, top);
	}
};
template <> struct LaneEval<uint8_t> : LaneEvalSmall<uint8_t> {};
template <> struct LaneEval<uint16_t> : LaneEvalSmall<uint16_t> {};

template <> struct LaneEval<uint32_t> {
	uint32_t top, nc;
	UnfoldPre uf;
	__device__ __forceinline__ void setup(uint32_t nc_, uint32_t code_, int q_)
	{
		nc = nc_;
		top = cm::ones<uint32_t>(q_ == 0 ? 32 : q_);
		uf.setup(code_, top, 0xffffffffu);
	}
	static __device__ __forceinline__ uint32_t paral(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t top)
	{
		uint32_t dn = v2 - v1, dp = v1 - v2;
		uint32_t lo = dn > v0 ? 0u : v0 - dn;
		uint32_t v = v0 + dp;
		uint32_t hi = ((v > top) | (v < v0)) ? top : v;
		return v1 < v2 ? lo : hi;
	}
	__device__ __forceinline__ uint32_t eval(const uint32_t (&s)[6]) const
	{
		uint32_t p0 = paral(s[0], s[1], s[2], top), p1 = paral(s[3], s[4], s[5], top);
		uint32_t two = (uint32_t)(((uint64_t)p0 + p1 + 1u) >> 1);
		uint32_t pred = nc == 2 ? two : nc == 1 ? p0 : 0u;
		return uf.apply(pred, top);
	}
};

template <> struct LaneEval<float> {
	uint32_t nc;
	UnfoldPre uf;
	__device__ __forceinline__ void setup(uint32_t nc_, uint32_t code_, int)
	{
		nc = nc_;
		uf.setup(code_, 0xffffffffu, 0xffffffffu);
	}
	__device__ __forceinline__ uint32_t eval(const uint32_t (&s)[6]) const
	{
		float p0 = cm::bits<float>(s[0]) + (cm::bits<float>(s[1]) - cm::bits<float>(s[2]));
		float p1 = cm::bits<float>(s[3]) + (cm::bits<float>(s[4]) - cm::bits<float>(s[5]));
		// n = 2: mean in double, nearest candidate with strict <, the first one wins (attrcode.h:182-208)
		float avg = (float)(((double)p0 + (double)p1) / 2.0);
		float best = 3.402823466e+38f;
		float db = avg > best ? avg - best : best - avg, dp = avg > p0 ? avg - p0 : p0 - avg;
		best = db < dp ? best : p0;
		db = avg > best ? avg - best : best - avg; dp = avg > p1 ? avg - p1 : p1 - avg;
		float two = db < dp ? best : p1;
		float pred = nc == 2 ? two : nc == 1 ? p0 : 0.0f;
		// transform.h:19-23 ordered-int map on both sides, prediction.h:33-44: no sign flip for 4-byte values
		return cm::bits<uint32_t>(cm::f32_from_ordered(uf.apply(cm::ordered_from_f32(pred), 0xffffffffu)));
	}
};

template <typename T> __device__ __forceinline__ uint32_t ev_top(int q) { return (uint32_t)cm::ones<T>(q == 0 ? (int)sizeof(T) * 8 : q); }
constexpr uint32_t kNoLane = 64;

#ifdef HRY_CHAIN_CLOCKS   // development: where the ticks of a chain go (wavefront 0 of every chain prints its sums)
#define HRY_CLK(...) __VA_ARGS__
#else
#define HRY_CLK(...)
#endif
#ifdef HRY_CHAIN_LOG   // development: when every tile of component 0's chain was handed on, and what kind of tile it was (scripts/chain_log.py)
__device__ unsigned long long g_chain_log[1u << 18];
__device__ unsigned long long g_chain_marks[1u << 18];   // (-DHRY_CHAIN_MARKS: ticks from the value's arrival to the end of the first run, of the first head, of the tile)
extern "C" int hry_debug_chain_log(unsigned long long *dst, unsigned n)
{
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_chain_log), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int hry_debug_chain_marks(unsigned long long *dst, unsigned n)
{
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_chain_marks), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#ifdef HRY_CHAIN_MARKS
#define HRY_MARK(...) __VA_ARGS__
#else
#define HRY_MARK(...)
#endif
#define HRY_LOG(...) __VA_ARGS__
#else
#define HRY_LOG(...)
#define HRY_MARK(...)
#endif
// Chains of different connected components run in ONE launch.  A component that names vertices of an earlier one (shared
// non-manifold vertices, cbm/encoder.h:79-113,187) reads their reconstructed values from the records; it waits until the chain
// of that component -- same attribute component -- has PROGRESSED past that vertex: done[] holds, per chain, the first vertex that
// is not final yet, published when the component's chain ends (publishing every eighth tile as well was measured in round 3: the fence
// it needs also waits for the next tile's prefetch, +4.5 % on a lone chain, and the configs[3] share gained nothing from it).
// Workgroups start in the order of their indices and a chain only ever waits for a component before it in coding order (= a lower workgroup index), which is therefore running
// or finished: the waits cannot deadlock.  Bounded like every other wait of the chains (g_chain_timeout).
// cache: eight 64-bit words of the chain's LDS, or nullptr -- owners it waited for, as (first vertex, progress seen) in ONE word each
// (whichever lane writes one, every lane reads a pair that belongs together): a vertex inside a remembered owner's finished part
// needs neither the owner search (fourteen dependent loads in a table of 19 000 components) nor a look at the progress word.
struct CrossSync { const uint32_t *seg_start; uint32_t nseg; uint32_t *done; unsigned long long *cache; uint32_t *gave_up; };   // done[attribute component * nseg + component of the mesh]; gave_up: the launch's own word behind that table
__device__ uint32_t g_chain_timeout;
constexpr uint32_t kSpinLimit = 1u << 22;
// A wait for another component's chain is bounded in WALL-CLOCK time (the 100 MHz counter of s_memrealtime, looked at every 256
// slow polls = every millisecond or so): two seconds -- the longest chain of the largest mesh this was built for takes 20 ms; a
// context that shares a busy device (eight contexts on one GPU is a tested mode) is slowed down, not stopped.  Once a chain of
// THIS decode has given up (the word behind its flag table: contexts that share a device do not see each other's) every other
// wait ends at its next look and k_unpredict2 starts no further component: the grid drains in the time of the components under
// way, and the host turns the word (and g_chain_timeout) into an error instead of returning a wrong mesh.
constexpr unsigned long long kOwnerWaitTicks = 200000000ull;
__device__ __attribute__((noinline)) void wait_owner(const CrossSync &xs, int c, uint32_t id)
{
	if (!xs.done) return;
	if (xs.cache) { const unsigned long long e = xs.cache[(id >> 2) & 7u]; if (id >= (uint32_t)e && id < (uint32_t)(e >> 32)) return; }
	uint32_t lo = 0, hi = xs.nseg, first = 0;   // owner: the last component that starts at or before the vertex
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1, at = xs.seg_start[mid]; if (at <= id) { lo = mid; first = at; } else hi = mid; }
	const uint32_t *flag = xs.done + (size_t)c * xs.nseg + lo;
	uint32_t spins = 0, seen;
	unsigned long long t_begin = 0;
#pragma nounroll
	while ((seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= id) {
		// (a few quick looks, then one every 3 us: thousands of slivers wait for the END of the component they hang on, each
		// look is a load past the caches, and the chains that do the work share that path)
		if (spins < 16) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(127);
		if ((++spins & 255u) == 0u) {
			if (xs.gave_up && __hip_atomic_load(xs.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // given up elsewhere in this decode: drain
			const unsigned long long now = wall_clock64();
			if (!t_begin) t_begin = now;
			else if (now - t_begin > kOwnerWaitTicks || spins > kSpinLimit) {
				atomicOr(&g_chain_timeout, 4u);
				if (xs.gave_up) __hip_atomic_store(xs.gave_up, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
		}
	}
	if (xs.cache && lo) xs.cache[(id >> 2) & 7u] = ((unsigned long long)seen << 32) | first;
}
// value of a vertex reconstructed by another chain or long ago by this one: past this compute unit's L1 (the line may have been
// cached while a neighbouring attribute component of the same record was still unwritten)
template <typename U> __device__ __forceinline__ U far_load(const uint8_t *addr) { return __hip_atomic_load((const U*)addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// upto: every vertex of the chain below it is final in the records
__device__ __forceinline__ void raise_flag(const CrossSync &xs, int c, uint32_t seg_idx, uint32_t upto)
{
	__syncthreads();
	if (xs.done) {
		__threadfence();   // every lane's stores into the records are visible device-wide before the progress word
		if (threadIdx.x == 0) __hip_atomic_store(xs.done + (size_t)c * xs.nseg + seg_idx, upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// ---- one ordinary step of the dense form of the float chain (unpredict2_component), hand-scheduled ----------------------------------
// Every lane: the two parallelogram predictions of its vertex, the one nearer to their mean (the later one on equal distances:
// sign of |mean - p0| - |mean - p1|), the residual code on its bits (+- delta by its sign; lanes with a fixed value take that);
// then lane `step` broadcasts its value and every lane takes it into the sources that wait for that vertex.  One block, so that
// no wait state is spent (a VALU-written SGPR needs two other instructions before the VALU reads it): 29 issue slots; the
// compiler's version of the loop body had 83 with its branches, the ring write and the hazards' s_nop.
// Temporaries: v236-v240, v246, s86-s96, vcc.
#define HRY_DS_HIT(k, sp)  "v_cmp_eq_u32_e64 " sp ", %[step], %[t" #k "]\n\t"
#define HRY_DS_PICK(k, sp) "v_cndmask_b32_e64 %[d" #k "], %[d" #k "], v246, " sp "\n\t"
__device__ __forceinline__ uint32_t dense_step(uint32_t (&d)[6], const uint32_t (&tag)[6], uint32_t delta, uint32_t fixmask, uint32_t fixedval, uint32_t step)
{
	uint32_t out;
	asm volatile(
		"v_sub_f32 v236, %[d1], %[d2]\n\t"
		"v_sub_f32 v237, %[d4], %[d5]\n\t"
		"v_add_f32 v236, %[d0], v236\n\t"
		"v_add_f32 v237, %[d3], v237\n\t"
		"v_mul_f32 v238, 0.5, v237\n\t"
		"v_fma_f32 v238, v236, 0.5, v238\n\t"
		"v_sub_f32 v239, v238, v236\n\t"
		"v_sub_f32 v240, v238, v237\n\t"
		"v_sub_f32_e64 v239, |v239|, |v240|\n\t"
		"v_ashrrev_i32 v239, 31, v239\n\t"
		"v_bfi_b32 v236, v239, v236, v237\n\t"
		"v_ashrrev_i32 v239, 31, v236\n\t"
		"v_xad_u32 v240, %[delta], v239, v236\n\t"
		"v_sub_u32 v240, v240, v239\n\t"
		"v_bfi_b32 %[out], %[fixmask], %[fixedval], v240\n\t"
		HRY_DS_HIT(0, "s[86:87]")
		"v_readlane_b32 s96, %[out], %[step]\n\t"
		HRY_DS_HIT(1, "s[88:89]") HRY_DS_HIT(2, "s[90:91]") HRY_DS_HIT(3, "s[92:93]") HRY_DS_HIT(4, "s[94:95]") HRY_DS_HIT(5, "vcc")
		"v_mov_b32 v246, s96\n\t"
		HRY_DS_PICK(0, "s[86:87]") HRY_DS_PICK(1, "s[88:89]") HRY_DS_PICK(2, "s[90:91]") HRY_DS_PICK(3, "s[92:93]") HRY_DS_PICK(4, "s[94:95]") HRY_DS_PICK(5, "vcc")
		: [out] "=&v"(out), [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5])
		: [t0] "v"(tag[0]), [t1] "v"(tag[1]), [t2] "v"(tag[2]), [t3] "v"(tag[3]), [t4] "v"(tag[4]), [t5] "v"(tag[5]),
		  [delta] "v"(delta), [fixmask] "v"(fixmask), [fixedval] "v"(fixedval), [step] "s"(step)
		: "v236", "v237", "v238", "v239", "v240", "v246", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "vcc");
	return out;
}

template <typename T>
__device__ void unpredict2_component(const TopoD &tp, const uint32_t *order_v, uint32_t nvtx_total, uint32_t seg_begin, uint32_t nvtx, const uint32_t *cand, const uint8_t *ncand,
                                     const uint8_t *planes, uint8_t *rec, int stride, int off, int q, int plane0,
                                     typename cm::word<sizeof(T)>::u *ring, uint32_t ring_n, uint32_t *queue, const CrossSync &xs, int comp, uint32_t seg_idx)
{
	// this call reconstructs the vertices [seg_begin, nvtx) of one component; the ring holds only vertices >= seg_begin.
	// The workgroup is ONE wavefront: its LDS accesses execute in program order, no barrier is needed anywhere.
	typedef typename cm::word<sizeof(T)>::u U;
	typedef typename cm::wide<T>::type W;
	static_assert(sizeof(T) <= 4, "8-byte components use the generic kernel");
	constexpr bool kSmallUnsigned = !cm::is_fp<T>::value && sizeof(T) <= 2 && !(T(-1) < T(0));
	const int lane = threadIdx.x;
	const uint32_t mask = ring_n - 1;
	// Per-vertex inputs (the ids of the first two candidates, candidate count, residual code) live in REGISTERS, one 64-vertex tile
	// at a time (lane j <-> vertex tile_base + j), the next tile's loads in flight while this one is reconstructed.  A batch starts
	// wherever the previous one ended (batches are cut at vertices that need special handling): the tile's registers are shifted
	// down by the start lane through the LDS crossbar (ds_bpermute: no memory behind it), so the batch code always sees lane j <->
	// vertex base + j; a batch ends at the tile's end at the latest.  Round 2 staged four tiles through a column-major LDS queue
	// to let batches run across tiles: 52 LDS accesses per batch, 1 500 of the 8 400 cycles a batch cost once the float chain
	// itself was down to 3 600.  The rows of vertices with more than two candidates (0.3 %) go to LDS when their tile starts.
	uint32_t nx_id[6], nx_nc = 0, nx_byte[sizeof(T)];   // raw loads only: anything computed from them here would wait for the memory
	const uint32_t n_tiles = (nvtx - seg_begin + 63) / 64;
	auto tile_request = [&](uint32_t t) {     // loads of tile t into registers
		const uint32_t v = seg_begin + 64 * t + lane;
		nx_nc = 0;
#pragma unroll
		for (int b8 = 0; b8 < (int)sizeof(T); ++b8) nx_byte[b8] = 0;
#pragma unroll
		for (int k = 0; k < 6; ++k) nx_id[k] = 0;
		if (v < nvtx) {
			const uint2 *src = (const uint2*)(cand + (size_t)v * kCand2);   // 24-byte rows: 8-byte aligned
			const uint2 a = src[0], b2 = src[1], c2 = src[2];
			nx_id[0] = a.x; nx_id[1] = a.y; nx_id[2] = b2.x; nx_id[3] = b2.y; nx_id[4] = c2.x; nx_id[5] = c2.y;
			nx_nc = ncand[v];
#pragma unroll
			for (int b8 = 0; b8 < (int)sizeof(T); ++b8) nx_byte[b8] = planes[(size_t)(plane0 + b8) * nvtx_total + v];
		}
	};
	uint32_t *bigrow = queue;                 // 24 ids per lane of the tile, written by the lanes with more than two candidates
	// value of an already reconstructed vertex that lies before the current batch.  ring_floor: the dense form below writes a
	// batch's values into the ring WHILE the batch runs and may run it again -- the ring slot of vertex base + j is the slot of vertex
	// base + j - ring_n, so sources that far back are read from the records there
	uint32_t ring_floor = 0;
	auto old_value = [&](uint32_t id, uint32_t base) -> U {
		if (base - id <= ring_n && id >= seg_begin && id >= ring_floor) return ring[id & mask];
		if (id < seg_begin) wait_owner(xs, comp, id);
		return far_load<U>(rec + (size_t)id * stride + off);
	};
	// value of a vertex with more than two candidates, evaluated on its own from the ring (every source is older than it): up to
	// kCandMax candidates sit in its row (lane k evaluates candidate k, the mean and the selection run over them in table order);
	// beyond that the fan is walked right here.  Uniform: every lane returns the same value.  row_lane: its lane in the tile.
	HRY_CLK(unsigned long long ck_m1 = 0, ck_m2 = 0, ck_m3 = 0, ck_m4 = 0, ck_mn = 0;)
	auto many_candidates_value = [&](uint32_t v, uint32_t row_lane, uint32_t n0, uint32_t c0) -> T {
		HRY_CLK(const unsigned long long m_t0 = __builtin_amdgcn_s_memtime(); unsigned long long m_t1 = m_t0, m_t2 = m_t0, m_t3 = m_t0;)
		T pred = T(0);
		if (n0 != 0xff) {
			// lane k: candidate k.  Straight-line: the row and the ring are read unconditionally (addresses always inside) and the
			// results selected -- a branch per source cost this lone wavefront an LDS round trip each; only a source older than the
			// ring takes the branch to memory
			const bool mine = (uint32_t)lane < n0;
			const uint32_t *row = bigrow + row_lane * 24 + 3 * (mine ? lane : 0);
			const uint32_t c0i = row[0], c1i = row[1], c2i = row[2];
			U s0 = ring[c0i & mask], s1 = ring[c1i & mask], s2 = ring[c2i & mask];
			auto in_ring = [&](uint32_t id) { return (v - id <= ring_n) & (id >= seg_begin) & (id >= ring_floor); };
			if (__ballot(mine && !(in_ring(c0i) & in_ring(c1i) & in_ring(c2i)))) {
				if (mine) { s0 = old_value(c0i, v); s1 = old_value(c1i, v); s2 = old_value(c2i, v); }
			}
			const uint32_t pk = mine ? (uint32_t)cm::bits<U>(cm::parallelogram<T>(cm::bits<T>(s0), cm::bits<T>(s1), cm::bits<T>(s2), q)) : 0u;
			HRY_CLK({ asm volatile("" :: "v"(pk)); m_t1 = __builtin_amdgcn_s_memtime(); })
			// (compiled per number of candidates: the generic loop tests it eight times twice over)
			if (n0 == 3) pred = chain_predict_n<T, 3>(pk);
			else if (n0 == 4) pred = chain_predict_n<T, 4>(pk);
			else if (n0 == 5) pred = chain_predict_n<T, 5>(pk);
			else if (n0 == 6) pred = chain_predict_n<T, 6>(pk);
			else {
				T pv[kCandMax];
#pragma unroll
				for (int k = 0; k < kCandMax; ++k) pv[k] = cm::bits<T>((U)rl(pk, k));
				pred = chain_predict<T>(n0, pv);
			}
			HRY_CLK({ asm volatile("" :: "v"((uint32_t)cm::bits<U>(pred))); m_t2 = __builtin_amdgcn_s_memtime(); })
		} else {
			W acc = 0;
			uint32_t n = 0;
			fan_ids(tp, order_v[v], v, [&](uint32_t a, uint32_t b, uint32_t o) {
				acc = acc + (W)cm::parallelogram<T>(cm::bits<T>(old_value(a, v)), cm::bits<T>(old_value(b, v)), cm::bits<T>(old_value(o, v)), q);
				++n;
			});
			if (n) {
				T avg = (T)cm::mean_of(acc, (W)n);
				if constexpr (!cm::is_fp<T>::value) pred = avg;
				else {
					T best = 3.402823466e+38f;
					fan_ids(tp, order_v[v], v, [&](uint32_t a, uint32_t b, uint32_t o) {
						T p = cm::parallelogram<T>(cm::bits<T>(old_value(a, v)), cm::bits<T>(old_value(b, v)), cm::bits<T>(old_value(o, v)), q);
						T db = avg > best ? avg - best : best - avg;
						T dp = avg > p ? avg - p : p - avg;
						best = db < dp ? best : p;
					});
					pred = best;
				}
			}
		}
		const T m_out = cm::value_from_residual<T>((U)c0, pred, q);
		HRY_CLK({ asm volatile("" :: "v"((uint32_t)cm::bits<U>(m_out))); m_t3 = __builtin_amdgcn_s_memtime(); ck_m1 += m_t1 - m_t0; ck_m2 += m_t2 - m_t1; ck_m3 += m_t3 - m_t2; ++ck_mn; })
		return m_out;
	};
	tile_request(0);
	uint32_t dense_skip = 0;                     // float chain, dense form: batches that take the exact step at once after the short one kept failing
	uint32_t fast_skip = 0, fast_backoff = 0;   // float chain: batches for which the speculative form is not tried after it failed
	HRY_CLK(unsigned long long ck_p1 = 0, ck_p2 = 0, ck_p3 = 0, ck_prep = 0, ck_chain = 0, ck_verify = 0, ck_pub = 0, ck_exact = 0, ck_batches = 0, ck_nb = 0, ck_retry = 0, ck_exact_n = 0, ck_bigs = 0, ck_general = 0, ck_t = 0, ck_bigt = 0, ck_gen_n = 0, ck_gen_nb = 0;
	        const unsigned long long ck_begin = __builtin_amdgcn_s_memtime();)
	for (uint32_t tile = 0; tile < n_tiles; ++tile) {
	HRY_CLK(ck_t = __builtin_amdgcn_s_memtime();)
	const uint32_t tile_base = seg_begin + 64 * tile, tile_n = min(64u, nvtx - tile_base);
	uint32_t t_id[6], t_code = 0;
#pragma unroll
	for (int j = 0; j < 6; ++j) t_id[j] = nx_id[j];
	const uint32_t t_nc = nx_nc;
#pragma unroll
	for (int b8 = 0; b8 < (int)sizeof(T); ++b8) t_code |= nx_byte[b8] << (8 * b8);
	if (tile + 1 < n_tiles) tile_request(tile + 1);   // in flight for the whole of this tile
	if (__ballot(t_nc > 2 && t_nc != 0xff)) {
		if (t_nc > 2 && t_nc != 0xff) {
			const uint4 *src = (const uint4*)cand_full_row(cand, nvtx_total, tile_base + lane);   // (96-byte rows behind a 64-byte header: 16-byte aligned)
#pragma unroll
			for (int k = 0; k < 6; ++k) { const uint4 r = src[k]; bigrow[lane * 24 + 4 * k] = r.x; bigrow[lane * 24 + 4 * k + 1] = r.y; bigrow[lane * 24 + 4 * k + 2] = r.z; bigrow[lane * 24 + 4 * k + 3] = r.w; }
		}
	}
	HRY_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_p1 += n - ck_t; })
	uint32_t pos = 0;
	while (pos < tile_n) {
		HRY_CLK(ck_t = __builtin_amdgcn_s_memtime();)
		const uint32_t base = tile_base + pos;
		const bool in_range = (uint32_t)lane < tile_n - pos;
		uint32_t ids[6], nc, code;
		if (pos) {
			const int from = (int)(((uint32_t)lane + pos) & 63u) * 4;
#pragma unroll
			for (int j = 0; j < 6; ++j) ids[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)t_id[j]);
			nc = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)t_nc);
			code = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)t_code);
		} else {
#pragma unroll
			for (int j = 0; j < 6; ++j) ids[j] = t_id[j];
			nc = t_nc; code = t_code;
		}
		nc = in_range ? nc : 0u;
		uint32_t nb = tile_n - pos;
		const uint64_t big = __ballot(nc > 2);
		if constexpr (std::is_same<T, float>::value) {
			// ---- mixed-polygon meshes (BASELINE configs[3] / [4]): a quad brings two new vertices and the second one's parallelograms
			// read both, one vertex in twelve has more than two candidates -- "chained" runs are eight long there and every cut
			// costs a batch start.  Where such vertices come thick the rest of the tile is ONE batch in the general form: at step
			// i lane i's value is final and is broadcast (every lane that waits for vertex i picks it up; it also goes into the
			// ring), a vertex with many candidates is evaluated at its own step from the ring.  The step is the SHORT form of the
			// arithmetic (as in the run form below), verified exactly after the batch: every lane holds the final values of its
			// sources then and evaluates the reference arithmetic on them; the many-candidate values are exact functions of what was
			// broadcast before them, so all are right iff every lane agrees.
			const uint32_t nbf = tile_n - pos;
			const uint64_t in_mask = nbf >= 64 ? ~0ull : (1ull << nbf) - 1ull;
			const uint64_t bigm = big & in_mask;
			if ((uint32_t)__builtin_popcountll(bigm) * 16u > nbf) {
				ring_floor = base + 64u > ring_n ? base + 64u - ring_n : 0u;
				const bool isbig = nc > 2;
				const uint32_t nc2 = isbig ? 0u : nc;
				uint32_t dsrc[6], dtag[6], far_any = 0;
#pragma unroll
				for (int s6 = 0; s6 < 6; ++s6) {
					const uint32_t id = ids[s6];
					const uint32_t ringv = (uint32_t)ring[id & mask];
					const bool valid = (uint32_t)(s6 / 3) < nc2;
					const bool inb = valid & (id >= base), old = valid & (id < base);
					far_any |= (old & ((base - id > ring_n) | (id < seg_begin))) ? 1u : 0u;
					dsrc[s6] = old ? ringv : 0u;
					dtag[s6] = inb ? id - base : kNoLane;
				}
				if (__ballot(far_any != 0)) {
#pragma unroll
					for (int s6 = 0; s6 < 6; ++s6) {
						const uint32_t id = ids[s6];
						if ((uint32_t)(s6 / 3) < nc2 && id < base && ((base - id > ring_n) | (id < seg_begin))) {
							if (id < seg_begin) wait_owner(xs, comp, id);
							dsrc[s6] = (uint32_t)far_load<U>(rec + (size_t)id * stride + off);
						}
					}
					__builtin_amdgcn_s_waitcnt(0);
				}
				LaneEval<T> dev;
				dev.setup(nc2, code, q);
				if (nc2 == 1) {   // a lone candidate is both candidates of the short form (the mean of p and p is p)
#pragma unroll
					for (int j = 0; j < 3; ++j) { dsrc[3 + j] = dsrc[j]; dtag[3 + j] = dtag[j]; }
				}
				const uint32_t half = code >> 1, delta = (code & 1u) ? 0u - half - 1u : half;
				// fixed lanes (all-ones mask): no candidate -- the residual code is the value (exact) --, many candidates (its own step),
				// a lane that was put right
				uint32_t fixmask = nc2 == 0 ? ~0u : 0u;
				uint32_t fixedval = dev.eval(dsrc), myval = 0, out = 0;
				uint32_t from = 0;
				bool exact_steps = dense_skip != 0;      // the stretch keeps failing (a coordinate that is 0 everywhere): the exact step at once
				const uint32_t my_slot = (base + (uint32_t)lane) & mask;
				for (uint32_t tries = 0;; ++tries) {
					// Stretches of ordinary vertices between the many-candidate ones.  A lane's value is final from its own step on (its
					// sources are earlier lanes) and every later step computes it again, so nothing is kept per step; the ring -- which
					// only the many-candidate evaluations read inside the batch -- gets the values of a stretch right before the next
					// such vertex, all lanes at once (round 3 wrote it from lane 0 at every step).
					uint32_t at = from, flushed = from;
					while (at < nbf) {
						const uint64_t ahead = bigm >> at;
						const uint32_t stop = ahead ? at + (uint32_t)__builtin_ctzll(ahead) : nbf;
						if (exact_steps) {
							for (uint32_t i = at; i < stop; ++i) {
								const uint32_t sv = rl(dev.eval(dsrc), i);
								myval = (uint32_t)lane == i ? sv : myval;
#pragma unroll
								for (int j = 0; j < 6; ++j) dsrc[j] = dtag[j] == i ? sv : dsrc[j];
							}
						} else {
							for (uint32_t i = at; i < stop; ++i) out = dense_step(dsrc, dtag, delta, fixmask, fixedval, i);
						}
						{   // the values of [flushed, stop) into the ring
							const uint32_t mine = (isbig || exact_steps) ? myval : out;
							if ((uint32_t)lane >= flushed && (uint32_t)lane < stop) ring[my_slot] = (U)mine;
							flushed = stop;
						}
						if (stop < nbf) {
							const uint32_t sv = cm::bits<uint32_t>(many_candidates_value(base + stop, pos + stop, rl(nc, stop), rl(code, stop)));
							myval = (uint32_t)lane == stop ? sv : myval;
#pragma unroll
							for (int j = 0; j < 6; ++j) dsrc[j] = dtag[j] == stop ? sv : dsrc[j];
						}
						at = stop + 1;
					}
					if (flushed < nbf) {   // (the batch ended with a many-candidate vertex)
						if ((uint32_t)lane >= flushed && (uint32_t)lane < nbf) ring[my_slot] = (U)myval;
					}
					if (!isbig && !exact_steps) myval = out;
					if (exact_steps) break;                                // (every value is the reference arithmetic on final sources)
					const uint32_t ref = dev.eval(dsrc);
					const uint64_t bad = __ballot((uint32_t)lane < nbf && !isbig && ref != myval);
					if (!bad) break;
					const uint32_t f = (uint32_t)__builtin_ctzll(bad);     // every lane before f is right, so ref of lane f is its exact value
					if ((uint32_t)lane == f) { fixmask = ~0u; fixedval = ref; }
					from = f;
					if (tries >= 2) { exact_steps = true; dense_skip = 8; }   // the rest of the batch, and the next batches, in the exact step
				}
				if (dense_skip) --dense_skip;
				ring_floor = 0;
				if ((uint32_t)lane < nbf) stq<T>(rec + (size_t)(base + lane) * stride + off, cm::bits<T>((U)myval));
				pos += nbf;
				HRY_CLK(++ck_batches; ck_nb += nbf; ck_general += __builtin_amdgcn_s_memtime() - ck_t; ++ck_gen_n; ck_gen_nb += nbf;)
				continue;
			}
		}
		if (big & 1ull) {
			// More than two candidates: this vertex is evaluated on its own; every source is older than it.
			const uint32_t v = base;
			const T val = many_candidates_value(v, pos, (uint32_t)__builtin_amdgcn_readfirstlane((int)nc), (uint32_t)__builtin_amdgcn_readfirstlane((int)code));
			if (lane == 0) {
				stq<T>(rec + (size_t)v * stride + off, val);
				ring[v & mask] = cm::bits<U>(val);
			}
			pos += 1;
			HRY_CLK(++ck_bigs; ck_bigt += __builtin_amdgcn_s_memtime() - ck_t;)
			continue;
		}
		if (big) nb = min(nb, (uint32_t)__builtin_ctzll(big));
		// ---- shape of the batch and sources, lane j <-> vertex base + j.  Straight-line code on purpose: every branch a lone
		// wavefront takes costs it an instruction-fetch bubble, so the ring is read unconditionally (the address is always
		// inside the ring) and the results are selected.  A vertex is "chained" when at most one of its sources lies inside
		// the batch, that source is the vertex right before it and enters its parallelogram with a plus sign - the rule in a
		// cut-border traversal (99.5 % of the vertices of a regular triangle mesh).  A run of chained vertices needs no
		// broadcast at all: the previous lane's value arrives through a DPP wave shift (chain variant "run" below).
		const uint32_t ncl = lane < (int)nb ? nc : 0u;
		uint32_t src[6], tag[6];
		uint32_t npend = 0, ngood = 0, pend_slot = 6, far_any = 0;
#pragma unroll
		for (int s6 = 0; s6 < 6; ++s6) {
			const uint32_t id = ids[s6];
			const uint32_t ringv = (uint32_t)ring[id & mask];
			const bool valid = (uint32_t)(s6 / 3) < ncl;
			const bool inb = valid & (id >= base);
			const bool old = valid & (id < base);
			npend += inb ? 1u : 0u;
			ngood += (inb & (s6 % 3 != 2) & (id + 1u == base + (uint32_t)lane)) ? 1u : 0u;
			pend_slot = inb ? (uint32_t)s6 : pend_slot;
			far_any |= (old & ((base - id > ring_n) | (id < seg_begin))) ? 1u : 0u;
			src[s6] = old ? ringv : 0u;
			tag[s6] = inb ? id - base : kNoLane;   // produced inside this batch, by an earlier lane
		}
		bool run_mode = false;
		{
			const bool chained = (npend == 0) | ((npend == 1) & (ngood == 1));
			const uint64_t unchained = __ballot(!chained);   // never lane 0: all its sources are older; lanes >= nb count as chained
			const uint32_t run = unchained ? min(nb, (uint32_t)__builtin_ctzll(unchained)) : nb;
			// A run costs ~60 cycles per vertex, a step of the general form ~370, a batch start ~2 000.  An unchained vertex is
			// chained again as the FIRST lane of a batch (every source is older then), so a run is cut in front of it; only where
			// unchained vertices come thick (fewer than six chained ones in front) does the general form take them -- up to the
			// first stretch of eight chained vertices, not to the end of the tile (round 2: 5 % of the batches of a regular mesh
			// went through 64 general steps for one such vertex each, 16 % of the chain's time).
			// ... unless they come thick all over the batch (quad and mixed-polygon meshes: a face brings two new vertices and the
			// second one's parallelograms read both): cutting would leave batches of five with 4 000 cycles of overhead each
			const uint32_t n_unchained = (uint32_t)__builtin_popcountll(unchained);
			if (run == nb || (run >= 6 && n_unchained * 8u <= nb)) { run_mode = true; nb = run; }
			else if (n_unchained * 8u > nb) { /* the general form for the whole batch */ }
			else {
				uint32_t g = run + 1;
				for (;;) {
					const uint32_t w = g < 64 ? (uint32_t)(unchained >> g) & 0xffu : 0u;
					if (!w) break;
					g += 32u - (uint32_t)__builtin_clz(w);
				}
				nb = min(nb, g);
			}
		}
		// Only a source older than the ring costs a global round trip, and only then is the vector-memory counter waited
		// for: neither the stores of the previous batch nor the tile in flight are ever waited for here.
		if (__ballot(far_any != 0)) {
#pragma unroll
			for (int s6 = 0; s6 < 6; ++s6) {
				const uint32_t id = ids[s6];
				if ((uint32_t)(s6 / 3) < ncl && id < base && ((base - id > ring_n) | (id < seg_begin))) {
					if (id < seg_begin) wait_owner(xs, comp, id);
					src[s6] = (uint32_t)far_load<U>(rec + (size_t)id * stride + off);
				}
			}
			__builtin_amdgcn_s_waitcnt(0);   // here, inside the rare branch: nothing after it may wait for vector memory
		}
		if constexpr (kSmallUnsigned) {
			if (!run_mode) {   // LaneEvalSmall: a lone candidate counts twice
				const bool lone1 = nc == 1;
#pragma unroll
				for (int j = 0; j < 3; ++j) { src[3 + j] = lone1 ? src[j] : src[3 + j]; tag[3 + j] = lone1 ? tag[j] : tag[3 + j]; }
			}
		}
		LaneEval<T> ev;
		ev.setup(nc, code, q);
		HRY_CLK({ asm volatile("" :: "v"(src[0]), "v"(src[5]), "v"(tag[0])); const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_p2 += n - ck_t; })
		// ---- chain
		uint32_t val = 0;
		bool done_run = false;
		if constexpr (kSmallUnsigned) {
			if (run_mode) {
				// Chain variant "run": lane l waits for lane l-1 only.  What bounds a lone wavefront here is the latency of
				// DEPENDENT instructions (~10 cycles each), so the step is arranged for depth, not only for count:
				//   x0   = value[l-1] + bo0        the add reads lane l-1 through a DPP wave shift; bo0 = the other two sources
				//   p0   = clamp(x0, 0, top)
				//   sum  = p0 * m + c              two candidates: m = 1, c = p1 + 1; a lone candidate: m = 2, c = 1 ((2p+1)>>1 == p);
				//                                  a vertex without a source inside the batch: m = 0, c = its finished sum
				//   pred = sum >> 1, and the inverse residual code is decided on sum directly (prediction.h:46-64):
				//     near  <=>  (sum - 2(half+1)) <u 2(top - 2 half);  far value = sum <= 2(top>>1)+1 ? code : top - code
				// 10 instructions + one s_nop (DPP read-after-write distance), 6 of them on the dependency path.
				const uint32_t top = ev_top<T>(q);
				const uint32_t kp = pend_slot < 3 ? 0u : pend_slot < 6 ? 1u : 0u;          // candidate holding the chained source
				const uint32_t sa = kp ? src[3] : src[0], sb = kp ? src[4] : src[1], so = kp ? src[5] : src[2];
				const uint32_t oa = kp ? src[0] : src[3], ob = kp ? src[1] : src[4], oo = kp ? src[2] : src[5];   // the other candidate
				const bool pend_is_b = pend_slot == 1 || pend_slot == 4;
				const uint32_t bo0 = (pend_is_b ? sa : sb) - so;
				const uint32_t p1c = med3_i32((int32_t)(oa + ob - oo), 0, (int32_t)top);
				const uint32_t p0c = med3_i32((int32_t)(sa + sb - so), 0, (int32_t)top);   // meaningful when every source is older
				const bool two = nc == 2, lone = nc == 1, keepl = pend_slot == 6;
				const uint32_t p1 = two ? p1c : 0u;
				// every source older than the batch: the sum is a constant (m = 0)
				const uint32_t m = keepl ? 0u : two ? 1u : 2u;
				const uint32_t c = keepl ? (two ? p0c + p1 + 1u : lone ? 2u * p0c + 1u : 1u) : (two ? p1 + 1u : 1u);
				UnfoldPre uf;
				uf.setup(code, top, (uint32_t)(T)(~T(0)));
				const uint32_t htop2 = 2u * (top >> 1) + 1u, c_hp2 = 2u * (uf.half + 1u), c_lim2 = top >= 2u * uf.half ? 2u * (top - 2u * uf.half) : 0u;
#define HRY_RSTEP                                                                                                       \
				    "v_add_u32_dpp v100, %[val], %[bo0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"            \
				    "v_med3_i32 v100, v100, 0, %[top]\n\t"                                                                 \
				    "v_mad_u32_u24 v102, v100, %[m], %[c]\n\t"                                                             \
				    "v_lshrrev_b32 v103, 1, v102\n\t"                                                                      \
				    "v_sub_u32 v104, v102, %[hp2]\n\t"                                                                     \
				    "v_cmp_ge_u32 s[80:81], %[htop2], v102\n\t"                                                            \
				    "v_cmp_lt_u32 s[82:83], v104, %[lim2]\n\t"                                                             \
				    "v_add_u32 v106, v103, %[delta]\n\t"                                                                   \
				    "v_cndmask_b32 v107, %[tmc], %[code], s[80:81]\n\t"                                                    \
				    "v_cndmask_b32 %[val], v107, v106, s[82:83]\n\t"                                                       \
				    "s_nop 1\n\t"
#define HRY_RSTEP8                                                                                                      \
				asm(HRY_RSTEP HRY_RSTEP HRY_RSTEP HRY_RSTEP HRY_RSTEP HRY_RSTEP HRY_RSTEP HRY_RSTEP                        \
				    : [val] "+v"(val)                                                                                     \
				    : [bo0] "v"(bo0), [m] "v"(m), [c] "v"(c), [top] "s"(top), [htop2] "s"(htop2), [code] "v"(uf.code),          \
				      [tmc] "v"(uf.top_minus_code), [delta] "v"(uf.delta), [hp2] "v"(c_hp2), [lim2] "v"(c_lim2)               \
				    : "v100", "v102", "v103", "v104", "v106", "v107", "s80", "s81", "s82", "s83");
				asm volatile("s_nop 1");   // the first DPP read of val
				for (uint32_t i = 0; i < nb; i += 8) { HRY_RSTEP8 }
#undef HRY_RSTEP8
#undef HRY_RSTEP
				done_run = true;
			}
		}
		if constexpr (kSmallUnsigned) {
			if (!done_run) {
			// One step, hand-scheduled: 27 instructions, every hazard distance of gfx950 (VALU-written SGPR read by a VALU: 2
			// wait states, VALU-written VGPR read by v_readlane: 1) is covered by independent instructions instead of s_nop,
			// each of which would cost this lone wavefront a full issue slot.  Same arithmetic as LaneEvalSmall::eval with
			// the inverse residual code reduced further (prediction.h:46-64):
			//   near case   half <= min(pred - 1, top - pred)  <=>  (pred - (half + 1)) <u top - 2 half   (0 when 2 half > top)
			//   far value   room >= pred  <=>  pred <= top >> 1 ? code : top - code      (pred == 0 lands here and yields code)
			// The third source of every candidate is kept negated, so that a parallelogram is one three-operand add.
			const uint32_t top = ev.top, half_top = ev.top >> 1, c_code = ev.uf.code, c_tmc = ev.uf.top_minus_code, c_delta = ev.uf.delta;
			const uint32_t c_hp1 = ev.uf.half + 1u, c_lim = top >= 2u * ev.uf.half ? top - 2u * ev.uf.half : 0u;
			src[2] = 0u - src[2]; src[5] = 0u - src[5];
#define HRY_STEP(I)                                                                                                     \
			asm("v_add3_u32 v100, %[a0], %[b0], %[o0]\n\t"                                                              \
			    "v_add3_u32 v101, %[a1], %[b1], %[o1]\n\t"                                                              \
			    "v_med3_i32 v100, v100, 0, %[top]\n\t"                                                                  \
			    "v_med3_i32 v101, v101, 0, %[top]\n\t"                                                                  \
			    "v_add3_u32 v102, v100, v101, 1\n\t"                                                                    \
			    "v_lshrrev_b32 v103, 1, v102\n\t"                                                                       \
			    "v_sub_u32 v104, v103, %[hp1]\n\t"                                                                      \
			    "v_cmp_ge_u32 s[80:81], %[htop], v103\n\t"                                                              \
			    "v_add_u32 v106, v103, %[delta]\n\t"                                                                    \
			    "v_cmp_lt_u32 s[82:83], v104, %[lim]\n\t"                                                               \
			    "v_cmp_eq_u32 s[84:85], " #I ", %[t0]\n\t"                                                              \
			    "v_cmp_eq_u32 s[86:87], " #I ", %[t1]\n\t"                                                              \
			    "v_cndmask_b32 v107, %[tmc], %[code], s[80:81]\n\t"                                                     \
			    "v_cmp_eq_u32 s[88:89], " #I ", %[t2]\n\t"                                                              \
			    "v_cndmask_b32 %[val], v107, v106, s[82:83]\n\t"                                                        \
			    "v_cmp_eq_u32 s[90:91], " #I ", %[t3]\n\t"                                                              \
			    "v_readlane_b32 s96, %[val], " #I "\n\t"                                                                \
			    "v_cmp_eq_u32 s[92:93], " #I ", %[t4]\n\t"                                                              \
			    "v_cmp_eq_u32 s[94:95], " #I ", %[t5]\n\t"                                                              \
			    "v_mov_b32 v108, s96\n\t"                                                                               \
			    "v_sub_u32 v109, 0, v108\n\t"                                                                           \
			    "v_cndmask_b32 %[a0], %[a0], v108, s[84:85]\n\t"                                                        \
			    "v_cndmask_b32 %[b0], %[b0], v108, s[86:87]\n\t"                                                        \
			    "v_cndmask_b32 %[a1], %[a1], v108, s[90:91]\n\t"                                                        \
			    "v_cndmask_b32 %[b1], %[b1], v108, s[92:93]\n\t"                                                        \
			    "v_cndmask_b32 %[o0], %[o0], v109, s[88:89]\n\t"                                                        \
			    "v_cndmask_b32 %[o1], %[o1], v109, s[94:95]"                                                             \
			    : [a0] "+v"(src[0]), [b0] "+v"(src[1]), [o0] "+v"(src[2]), [a1] "+v"(src[3]), [b1] "+v"(src[4]), [o1] "+v"(src[5]),  \
			      [val] "=&v"(val)                                                                                     \
			    : [t0] "v"(tag[0]), [t1] "v"(tag[1]), [t2] "v"(tag[2]), [t3] "v"(tag[3]), [t4] "v"(tag[4]), [t5] "v"(tag[5]),   \
			      [top] "s"(top), [htop] "s"(half_top), [code] "v"(c_code), [tmc] "v"(c_tmc), [delta] "v"(c_delta),       \
			      [hp1] "v"(c_hp1), [lim] "v"(c_lim)                                                                    \
			    : "v100", "v101", "v102", "v103", "v104", "v106", "v107", "v108", "v109", "s80", "s81", "s82", "s83", "s84", "s85",  \
			      "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96");
#define HRY_STEP8(A, B, C, D, E, F, G, H) HRY_STEP(A) HRY_STEP(B) HRY_STEP(C) HRY_STEP(D) HRY_STEP(E) HRY_STEP(F) HRY_STEP(G) HRY_STEP(H)
			do {   // steps past the end of a short batch only produce unused values: leave at multiples of eight
				HRY_STEP8(0, 1, 2, 3, 4, 5, 6, 7)
				if (nb <= 8) break;
				HRY_STEP8(8, 9, 10, 11, 12, 13, 14, 15)
				if (nb <= 16) break;
				HRY_STEP8(16, 17, 18, 19, 20, 21, 22, 23)
				if (nb <= 24) break;
				HRY_STEP8(24, 25, 26, 27, 28, 29, 30, 31)
				if (nb <= 32) break;
				HRY_STEP8(32, 33, 34, 35, 36, 37, 38, 39)
				if (nb <= 40) break;
				HRY_STEP8(40, 41, 42, 43, 44, 45, 46, 47)
				if (nb <= 48) break;
				HRY_STEP8(48, 49, 50, 51, 52, 53, 54, 55)
				if (nb <= 56) break;
				HRY_STEP8(56, 57, 58, 59, 60, 61, 62, 63)
			} while (0);
#undef HRY_STEP8
#undef HRY_STEP
			}
		} else if (run_mode && cm::is_fp<T>::value) {
			// Chain variant "run" for float components (lossless meshes).  Per lane:
			//   x  = value of the previous vertex (DPP wave shift)
			//   pp = the parallelogram that waits for it: x + T (x is its first source, T = v1 - v2) or A + (x - O) (second source)
			//   (p0, p1) = (pp, q) or (q, pp): q is the finished other parallelogram, the order of the candidates is kept because
			//   the selection is not symmetric (attrcode.h:182-208: mean in double, candidate nearest to it, strict <, starting
			//   from FLT_MAX); a lone candidate is the prediction itself
			//   value = inverse residual code on the ordered-int images (transform.h:19-23, prediction.h:46-64, no sign flip)
			// A lone wavefront pays ~4 cycles per instruction and ~6 per DEPENDENT instruction, so the chain is evaluated SPECULATIVELY
			// in 11 - 12 instructions per vertex (8 on the dependency) and then VERIFIED exactly, all lanes at once (the scheme of k_unpredict3): every lane
			// evaluates the reference arithmetic (LaneEval<float>::eval) on its predecessor's final value; lane 0 of a batch is exact
			// by construction, so all values are right iff every lane agrees (induction).  What the short form assumes:
			//   mean   fma(pp, 0.5, q/2) in float = the float of the double mean unless the sum leaves the float range or the two
			//          exponents are > 29 bits apart;  the FLT_MAX start of the sweep never wins;
			//   code   the "near" case of the residual code (prediction.h:58-63): value bits = prediction bits +- delta, the sign
			//          taken from the prediction (no crossing of zero, prediction not 0).
			// The first lane that disagrees is given its exact value (it becomes a lane without a source inside the batch) and the
			// chain is run again from there, twice at most; after that -- and for the following batches of a chain that keeps
			// failing (a coordinate plane z = 0: every prediction is 0) -- the exact chain below (33 instructions) takes over.
			const uint32_t kp = pend_slot < 3 ? 0u : pend_slot < 6 ? 1u : 0u;
			const float fa = cm::bits<float>(kp ? src[3] : src[0]), fb = cm::bits<float>(kp ? src[4] : src[1]), fo = cm::bits<float>(kp ? src[5] : src[2]);
			const float ga = cm::bits<float>(kp ? src[0] : src[3]), gb = cm::bits<float>(kp ? src[1] : src[4]), go = cm::bits<float>(kp ? src[2] : src[5]);
			const uint32_t cT = cm::bits<uint32_t>(fb - fo), cA = cm::bits<uint32_t>(fa), cO = cm::bits<uint32_t>(fo);
			const uint32_t cq = cm::bits<uint32_t>(ga + (gb - go));
			// the finished value of a vertex without a source inside the batch -- the exact chain needs it exactly; the speculative
			// one takes the short form of the arithmetic for it too (it is verified like every other lane)
			uint32_t valc = 0;
			const bool isBl = pend_slot == 1 || pend_slot == 4;
			const uint64_t isB = __ballot(isBl), isK1 = __ballot(kp == 1);
			const uint64_t lone = __ballot(nc == 1);
			const bool keepl0 = pend_slot == 6;
			const uint64_t keep = __ballot(keepl0);
			UnfoldPre uf;
			uf.setup(code, 0xffffffffu, 0xffffffffu);
			bool exact_chain = fast_skip != 0;
			if (fast_skip) --fast_skip;
			if (!exact_chain) {
				// pp = P + (x + M): (M, P) = (T, -0.0) when x is the first source (-0.0 + y == y for every y), (-O, A) when it is the second.
				// A lane without a source inside the batch holds a finished value: M = NaN makes its pp, mean and distances NaN, the
				// difference of the |distances| a positive NaN -- never "pp is nearer" -- so the select yields q = that value, and
				// delta = 0 leaves it alone: no instruction of the step is spent on such lanes.
				const bool lonel = nc == 1;
				{
					const float p0 = cm::bits<float>(src[0]) + (cm::bits<float>(src[1]) - cm::bits<float>(src[2]));
					const float p1 = cm::bits<float>(src[3]) + (cm::bits<float>(src[4]) - cm::bits<float>(src[5]));
					const float avg = __builtin_fmaf(p0, 0.5f, 0.5f * p1);
					const float e = __builtin_fabsf(avg - p0) - __builtin_fabsf(avg - p1);
					const float two = e < 0.0f ? p0 : p1;                        // ties go to the later candidate
					const uint32_t pb = cm::bits<uint32_t>(nc == 2 ? two : nc == 1 ? p0 : 0.0f);
					valc = nc == 0 ? uf.code : (int32_t)pb < 0 ? pb - uf.delta : pb + uf.delta;
				}
				uint32_t cM = keepl0 ? 0x7fc00000u : isBl ? cm::bits<uint32_t>(-fo) : cT;
				const uint32_t cP = isBl ? cA : 0x80000000u;
				uint32_t q2 = keepl0 ? valc : lonel ? 0x7f7fffffu : cq;               // a lone candidate: the other one is out of reach
				const uint32_t qh = lonel ? 0u : cm::bits<uint32_t>(0.5f * cm::bits<float>(cq));
				const uint32_t kk = kp == 1 ? 1u : 0u;                                 // ties go to the later candidate (strict <)
				uint32_t dlt = keepl0 ? 0u : uf.delta;
				// the value either candidate would give: its bits +- delta by its own sign; the one for q is finished here, the one for
				// pp runs beside the selection (the dependency is add, add, fma, sub, sub, sub, shift, select: eight deep)
				uint32_t vq = (int32_t)q2 < 0 ? q2 - dlt : q2 + dlt;
				const bool all_first = (isB & ((nb >= 64 ? ~0ull : (1ull << nb) - 1ull))) == 0;
#define HRY_QSTEP_HEAD_A                                                                                                 \
			    "v_add_f32_dpp v100, %[val], %[M] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define HRY_QSTEP_HEAD_B                                                                                                 \
			    "v_add_f32_dpp v100, %[val], %[M] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                   \
			    "v_add_f32 v100, %[P], v100\n\t"
#define HRY_QSTEP_TAIL                                                                                                   \
			    "v_fma_f32 v101, v100, 0.5, %[qh]\n\t"                                                                     \
			    "v_ashrrev_i32 v105, 31, v100\n\t"                                                                         \
			    "v_sub_f32 v102, v101, v100\n\t"                                                                           \
			    "v_sub_f32 v103, v101, %[q]\n\t"                                                                           \
			    "v_xad_u32 v106, %[dlt], v105, v100\n\t"                                                                   \
			    "v_sub_f32_e64 v102, |v102|, |v103|\n\t"                                                                   \
			    "v_sub_u32 v106, v106, v105\n\t"                                                                           \
			    "v_sub_u32 v102, v102, %[kk]\n\t"                                                                          \
			    "v_ashrrev_i32 v102, 31, v102\n\t"                                                                         \
			    "v_bfi_b32 %[val], v102, v106, %[vq]\n\t"                                                                  \
			    "s_nop 1\n\t"
#define HRY_QRUN(HEAD)                                                                                                   \
				for (uint32_t i = s0; i < nb; i += 8) {                                                                    \
					asm(HEAD HRY_QSTEP_TAIL HEAD HRY_QSTEP_TAIL HEAD HRY_QSTEP_TAIL HEAD HRY_QSTEP_TAIL                         \
					    HEAD HRY_QSTEP_TAIL HEAD HRY_QSTEP_TAIL HEAD HRY_QSTEP_TAIL HEAD HRY_QSTEP_TAIL                         \
					    : [val] "+v"(val)                                                                                   \
					    : [M] "v"(cM), [P] "v"(cP), [q] "v"(q2), [qh] "v"(qh), [kk] "v"(kk), [dlt] "v"(dlt), [vq] "v"(vq)             \
					    : "v100", "v101", "v102", "v103", "v104", "v105", "v106");                                          \
				}
				uint32_t s0 = 0;
				HRY_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_prep += n - ck_t; ck_t = n; })
				for (uint32_t tries = 0;; ++tries) {
					asm volatile("s_nop 1");   // the first DPP read of val, and the ballots above
					if (all_first) { HRY_QRUN(HRY_QSTEP_HEAD_A) } else { HRY_QRUN(HRY_QSTEP_HEAD_B) }
					HRY_CLK({ asm volatile("" :: "v"(val)); const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_chain += n - ck_t; ck_t = n; })
					// exact check, all lanes at once: the reference arithmetic on the predecessor's final value
					const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)val, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
					uint32_t sv[6];
#pragma unroll
					for (int j = 0; j < 6; ++j) sv[j] = pend_slot == (uint32_t)j ? prev : src[j];
					const uint32_t ref = ev.eval(sv);
					const uint64_t bad = __ballot(lane < (int)nb && ref != val);
					HRY_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_verify += n - ck_t; ck_t = n; if (bad) ++ck_retry; })
					if (!bad) break;
					if (tries >= 2) { exact_chain = true; break; }
					const uint32_t f = (uint32_t)__builtin_ctzll(bad);     // every lane before f is right, so ref of lane f is its exact value
					const bool fix = (uint32_t)lane == f;                  // ... and the lane now holds a finished value
					cM = fix ? 0x7fc00000u : cM; q2 = fix ? ref : q2; vq = fix ? ref : vq; dlt = fix ? 0u : dlt;
					s0 = f & ~7u;
				}
#undef HRY_QRUN
#undef HRY_QSTEP_TAIL
#undef HRY_QSTEP_HEAD_B
#undef HRY_QSTEP_HEAD_A
				HRY_CLK(if (exact_chain) ++ck_exact_n;)
				if (exact_chain) { fast_backoff = fast_backoff ? min(64u, 2u * fast_backoff) : 2u; fast_skip = fast_backoff; val = 0; }
				else fast_backoff = 0;
			}
			if (exact_chain) {
			valc = ev.eval(src);
			// the exact chain, hand-scheduled: 33 instructions + 3 s_nop per vertex; LaneEval<float>::eval is the readable form
			const uint32_t c_hp1 = uf.half + 1u, c_lim = 0xffffffffu - 2u * uf.half, fltmax = 0x7f7fffffu, htop = 0x7fffffffu;
#define HRY_FSTEP                                                                                                       \
			    "v_add_f32_dpp v100, %[val], %[T] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                   \
			    "v_sub_f32_dpp v101, %[val], %[O] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                   \
			    "v_add_f32 v101, %[A], v101\n\t"                                                                           \
			    "v_cndmask_b32 v100, v100, v101, %[isB]\n\t"                                                               \
			    "v_cndmask_b32 v102, v100, %[q], %[isK1]\n\t"                                                              \
			    "v_cndmask_b32 v103, %[q], v100, %[isK1]\n\t"                                                              \
			    "v_cvt_f64_f32 v[104:105], v102\n\t"                                                                       \
			    "v_cvt_f64_f32 v[106:107], v103\n\t"                                                                       \
			    "v_add_f64 v[104:105], v[104:105], v[106:107]\n\t"                                                         \
			    "v_mul_f64 v[104:105], v[104:105], 0.5\n\t"                                                                \
			    "v_cvt_f32_f64 v108, v[104:105]\n\t"                                                                       \
			    "v_sub_f32 v109, v108, v102\n\t"                                                                           \
			    "v_sub_f32 v110, v108, %[fmax]\n\t"                                                                        \
			    "v_cmp_lt_f32 s[80:81], |v110|, |v109|\n\t"                                                                \
			    "v_sub_f32 v111, v108, v103\n\t"                                                                           \
			    "s_nop 0\n\t"                                                                                              \
			    "v_cndmask_b32 v112, v102, %[fmax], s[80:81]\n\t"                                                          \
			    "v_sub_f32 v114, v108, v112\n\t"                                                                           \
			    "v_cmp_lt_f32 s[82:83], |v114|, |v111|\n\t"                                                                \
			    "s_nop 1\n\t"                                                                                              \
			    "v_cndmask_b32 v115, v103, v112, s[82:83]\n\t"                                                             \
			    "v_cndmask_b32 v115, v115, v102, %[lone]\n\t"                                                              \
			    "v_ashrrev_i32 v116, 31, v115\n\t"                                                                         \
			    "v_lshrrev_b32 v116, 1, v116\n\t"                                                                          \
			    "v_xor_b32 v116, v116, v115\n\t"                                                                           \
			    "v_sub_u32 v117, v116, %[hp1]\n\t"                                                                         \
			    "v_cmp_ge_u32 s[80:81], %[htop], v116\n\t"                                                                 \
			    "v_cmp_lt_u32 s[82:83], v117, %[lim]\n\t"                                                                  \
			    "v_add_u32 v118, v116, %[delta]\n\t"                                                                       \
			    "v_cndmask_b32 v119, %[tmc], %[code], s[80:81]\n\t"                                                        \
			    "v_cndmask_b32 v119, v119, v118, s[82:83]\n\t"                                                             \
			    "v_ashrrev_i32 v120, 31, v119\n\t"                                                                         \
			    "v_lshrrev_b32 v120, 1, v120\n\t"                                                                          \
			    "v_xor_b32 v119, v120, v119\n\t"                                                                           \
			    "v_cndmask_b32 %[val], v119, %[valc], %[keep]\n\t"                                                         \
			    "s_nop 1\n\t"
			asm volatile("s_nop 1");   // the first DPP read of val, and the ballots above
			for (uint32_t i = 0; i < nb; i += 4) {
				asm(HRY_FSTEP HRY_FSTEP HRY_FSTEP HRY_FSTEP
				    : [val] "+v"(val)
				    : [T] "v"(cT), [A] "v"(cA), [O] "v"(cO), [q] "v"(cq), [fmax] "v"(fltmax), [valc] "v"(valc), [code] "v"(uf.code),
				      [tmc] "v"(uf.top_minus_code), [delta] "v"(uf.delta), [hp1] "v"(c_hp1), [lim] "v"(c_lim), [htop] "s"(htop),
				      [isB] "s"(isB), [isK1] "s"(isK1), [lone] "s"(lone), [keep] "s"(keep)
				    : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v114", "v115",
				      "v116", "v117", "v118", "v119", "v120", "s80", "s81", "s82", "s83");
			}
#undef HRY_FSTEP
			HRY_CLK({ asm volatile("" :: "v"(val)); const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_exact += n - ck_t; ck_t = n; })
			}
		} else if (run_mode) {
			// Chain variant "run" for the other component types (32-bit and signed integers): the same wave shift, the
			// arithmetic is LaneEval<T>::eval itself with the previous vertex substituted for the one source that waits for it.
			const bool w0 = pend_slot == 0, w1 = pend_slot == 1, w3 = pend_slot == 3, w4 = pend_slot == 4;
			for (uint32_t i = 0; i < nb; ++i) {
				const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)val, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
				const uint32_t s[6] = { w0 ? prev : src[0], w1 ? prev : src[1], src[2], w3 ? prev : src[3], w4 ? prev : src[4], src[5] };
				val = ev.eval(s);
			}
		} else {
			bool settled = false;
			if constexpr (std::is_same<T, float>::value) {
				// the general form of a float batch (any lane may read any earlier lane): the short step of the dense form above,
				// verified exactly after the batch; the exact loop below takes the batch when two repeats did not settle it
				if (nc == 1) {   // a lone candidate is both candidates of the short form
#pragma unroll
					for (int j = 0; j < 3; ++j) { src[3 + j] = src[j]; tag[3 + j] = tag[j]; }
				}
				const uint32_t half = code >> 1, delta = (code & 1u) ? 0u - half - 1u : half;
				uint32_t fixmask = (nc == 0 || lane >= (int)nb) ? ~0u : 0u, fixedval = ev.eval(src), from = 0;
				for (uint32_t tries = 0; tries < 3 && !settled; ++tries) {
					for (uint32_t i = from; i < nb; ++i) val = dense_step(src, tag, delta, fixmask, fixedval, i);
					const uint32_t ref = ev.eval(src);
					const uint64_t bad = __ballot(lane < (int)nb && ref != val);
					if (!bad) { settled = true; break; }
					const uint32_t f = (uint32_t)__builtin_ctzll(bad);     // every lane before f is right, so ref of lane f is its exact value
					if ((uint32_t)lane == f) { fixmask = ~0u; fixedval = ref; }
					from = f;
				}
			}
			if (!settled) {
#pragma unroll
				for (uint32_t i = 0; i < 64; ++i) {
					if ((i & 7u) == 0 && i >= nb) break;   // steps past the end of a short batch only produce unused values
					val = ev.eval(src);
					const uint32_t s = rl(val, i);
#pragma unroll
					for (int j = 0; j < 6; ++j) src[j] = tag[j] == i ? s : src[j];
				}
			}
		}
		// ---- publish the batch
		HRY_CLK({ asm volatile("" :: "v"(val)); const unsigned long long n = __builtin_amdgcn_s_memtime(); if (!run_mode) { ck_general += n - ck_t; ++ck_gen_n; ck_gen_nb += nb; } ck_t = n; })
		if (lane < (int)nb) {
			const uint32_t v = base + lane;
			ring[v & mask] = (U)val;
			stq<T>(rec + (size_t)v * stride + off, cm::bits<T>((U)val));
		}
		pos += nb;
		HRY_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_pub += n - ck_t; ++ck_batches; ck_nb += nb; })
	}
	}
	HRY_CLK(if (lane == 0 && ck_batches > 100) printf("chain2 comp %d: %llu vertices, %llu batches (mean %llu), bigs %llu, retries %llu, exact batches %llu | per batch: prep %llu (commit %llu, to sources %llu) chain %llu verify %llu exact %llu general %llu publish %llu | bigs at %llu, %llu general batches (mean %llu) | many-candidate: sources %llu predict %llu code %llu | total %llu per vertex %llu\n",
	        comp, (unsigned long long)(nvtx - seg_begin), ck_batches, ck_nb / ck_batches, ck_bigs, ck_retry, ck_exact_n, ck_prep / ck_batches, ck_p1 / ck_batches, ck_p2 / ck_batches, ck_chain / ck_batches, ck_verify / ck_batches, ck_exact / ck_batches,
	        ck_general / ck_batches, ck_pub / ck_batches, ck_bigt / (ck_bigs ? ck_bigs : 1), ck_gen_n, ck_gen_nb / (ck_gen_n ? ck_gen_n : 1), ck_m1 / (ck_mn ? ck_mn : 1), ck_m2 / (ck_mn ? ck_mn : 1), ck_m3 / (ck_mn ? ck_mn : 1), __builtin_amdgcn_s_memtime() - ck_begin, (__builtin_amdgcn_s_memtime() - ck_begin) / (nvtx - seg_begin));)
}

struct CompSel { int32_t n; int32_t comp[kMaxComp]; };
// work lists: block y reconstructs the segments segs[list_off[y] .. list_off[y+1]) one after the other

// one kernel per component type (keeps each instantiation's register allocation to itself: no scratch in the chain)
template <typename T>
__global__ __launch_bounds__(64) void k_unpredict2(ConnView cv, const uint32_t *order_v, uint32_t nvtx, const uint32_t *cand, const uint8_t *ncand,
                                                   const uint8_t *planes, ListDesc ld, uint8_t *rec, uint32_t ring_bytes, CompSel sel,
                                                   const uint32_t *segs, const uint32_t *list_off, CrossSync xs, uint32_t n_lists)
{
	extern __shared__ unsigned long long ring_raw2[];
	// The chains of a mesh component's attribute components (x, y, z) write 4 bytes each into the same 12-byte records: they run on
	// ONE XCD -- workgroups go round-robin over the eight, so a group of 8 sel.n consecutive workgroups holds eight lists, workgroup
	// c * 8 + j of the group = attribute component c of list j -- and their stores meet in one L2 instead of three (the
	// configs[3] share wrote 334 MB for 76 MB of records with (component, list) -> (blockIdx.x, blockIdx.y)).  A chain still
	// only waits for the same attribute component of an EARLIER list, i.e. for a lower workgroup index.
	const uint32_t per = 8u * (uint32_t)sel.n, grp = blockIdx.x / per, r = blockIdx.x % per;
	const uint32_t list = grp * 8u + (r & 7u);
	if (list >= n_lists) return;
	const int c = sel.comp[r >> 3];
	TopoD tp{ cv };
	CrossSync xl = xs;   // ... with the chain's remembered owners behind the ring and the rows (wait_owner)
	xl.cache = (unsigned long long*)((uint8_t*)ring_raw2 + ring_bytes + 64 * kCandMax * 3 * 4);
	if (threadIdx.x < 8) xl.cache[threadIdx.x] = 0xffffffffull;   // (first vertex 2^32 - 1, nothing seen: no vertex is inside)
	// segs: triples (first decode rank, end, component of the mesh)
	for (uint32_t k = list_off[list]; k < list_off[list + 1]; ++k) {
		const uint32_t b = segs[3 * k], e = segs[3 * k + 1];
		// a chain of this device gave up a wait: no further component is started (the result is an error on the host either way)
		if (xs.gave_up && __builtin_amdgcn_readfirstlane(__hip_atomic_load(xs.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
		if (b < e)
			unpredict2_component<T>(tp, order_v, nvtx, b, e, cand, ncand, planes, rec, ld.stride, ld.off[c], ld.quant[c], ld.plane[c],
			                        (typename cm::word<sizeof(T)>::u*)ring_raw2, ring_bytes / (uint32_t)sizeof(T),
			                        (uint32_t*)((uint8_t*)ring_raw2 + ring_bytes), xl, c, segs[3 * k + 2]);
		raise_flag(xs, c, segs[3 * k + 2], e);
	}
}

// ---------------------------------------------------------------------------------------------------------
// k_unpredict3: the chain for unsigned components of at most 16 bits (every quantised attribute), evaluated as a
// PREFIX SCAN instead of vertex by vertex.
//
// Everything that depends only on the connectivity is computed beforehand, in parallel, by k_chain_records: per vertex a
// 16-byte record with the LDS ring slots of its (at most two) candidates' sources, which source is "the vertex right
// before it" (the chained source), and the earliest batch start the vertex tolerates (every other source must be final
// before its batch starts).  The chain kernel walks 64-aligned tiles (record + residual code of a tile prefetched one tile
// ahead, one coalesced load each) and cuts a tile into runs at those marks; inside a run every vertex depends on the run
// only through its predecessor.
//
// A run is a composition of per-vertex maps x -> value (x = the predecessor's value).  With the parallelogram unclamped
// and the residual code in its "near" form (prediction.h:58-63) the map is
//     lone candidate : x -> x + (b - o) + delta
//     two candidates : x -> floor((x + (b - o) + p1 + 1) / 2) + delta          (mean of two, transform.h:91)
//     no source inside the run : x -> constant                                    (evaluated exactly, not speculated)
// i.e. x -> floor((x + A) / 2^k) + D with 0 <= A < 2^k.  These maps are closed under composition
//     g(f(x)) = floor((x + A1 + 2^k1 ((D1 + A2) mod 2^k2)) / 2^(k1+k2)) + D2 + floor((D1 + A2) / 2^k2),
// and on the value range 0 <= x < 2^16 a map with k > 16 is a step function D + [x >= theta], which is again of the form
// floor((x + A') / 2^16) + D: the representation never grows.  A wavefront inclusive scan (4 DPP row shifts + 2 row
// broadcasts) composes the maps of a run; lane l then holds the value of its vertex -- provided no lane left the speculated
// form (clamped parallelogram, "far" or raw residual code: rare).  That is verified exactly: every lane evaluates its true
// arithmetic (attrcode.h:182-208, prediction.h:46-64,121-138) on its predecessor's scanned value; by induction all values
// are right iff every lane agrees.  The first disagreeing lane is replaced by its true value (a constant map) and the
// scan is repeated once; a second disagreement finishes the run vertex by vertex.  Results are bit-identical to the
// sequential evaluation in every case; only the speed depends on the data.
// Reconstructed values live in an LDS ring of kRing3 entries and go to the records tile by tile, by the tile's owner and behind
// its hand-over, so the chain never waits for a store.
// ---------------------------------------------------------------------------------------------------------
struct alignas(16) ChainRec { uint16_t slot[6]; uint16_t flags; uint16_t pad; };
constexpr uint32_t kRing3 = 16384;          // ring entries (32 KB of LDS for 16-bit values)
constexpr uint32_t kRing3Near = kRing3 - 64;   // a source this close to its vertex is still in the ring when the run is prepared
#ifndef HRY_CHAIN_POLL_PAUSE_ASM
#define HRY_CHAIN_POLL_PAUSE_ASM
#endif
#ifndef HRY_CHAIN_MAX_HEADS
#define HRY_CHAIN_MAX_HEADS 8
#endif
constexpr uint32_t kMaxHeads = HRY_CHAIN_MAX_HEADS;   // heads per prepared tile
#ifndef HRY_CHAIN_MAX_HEADS_LATE
#define HRY_CHAIN_MAX_HEADS_LATE 12
#endif
constexpr uint32_t kMaxHeadsLate = HRY_CHAIN_MAX_HEADS_LATE;   // heads of a tile that is prepared when its turn has come
constexpr uint32_t kHand0 = 2, kHand = 64;     // hand-over words of the chain's wavefront team behind its two control words
enum { CR_NC = 3, CR_BIG = 3, CR_POS_SHIFT = 2, CR_POS_NONE = 7, CR_FAR = 1 << 5, CR_NEED_SHIFT = 6 };

// ring_floor: ids below it are not in the LDS ring when the vertex is reconstructed (they belong to an earlier component, or
// to an earlier slice beyond the part of the ring that is reloaded): such sources are read by vertex id ("far")
__device__ __forceinline__ ChainRec make_chain_rec_ids(uint32_t nc, const uint32_t (&row)[6], uint32_t v, uint32_t seg_begin, uint32_t ring_floor)
{
	ChainRec r;
#pragma unroll
	for (int j = 0; j < 6; ++j) r.slot[j] = 0;
	r.pad = 0;
	if (nc > 2) { r.flags = (uint16_t)(CR_BIG | (CR_POS_NONE << CR_POS_SHIFT)); return r; }
	uint32_t pos = CR_POS_NONE, need = 0, far = 0;
#pragma unroll
	for (uint32_t j = 0; j < 6; ++j) {
		if (j >= 3 * nc) continue;
		const uint32_t id = row[j];
		if (id + 1u == v && j % 3 != 2 && pos == CR_POS_NONE && v > seg_begin) pos = j;   // the chained source: predecessor, plus sign, once
		else { need = max(need, id + 1u); }
		far |= (v - id > kRing3Near || id < ring_floor) ? 1u : 0u;
		r.slot[j] = (uint16_t)(id & (kRing3 - 1));
	}
	if (nc == 1) { r.slot[3] = r.slot[0]; r.slot[4] = r.slot[1]; r.slot[5] = r.slot[2]; }   // (2 p + 1) >> 1 == p
	const uint32_t tile = v & ~63u;
	const uint32_t need_rel = need > tile ? need - tile : 0u;   // <= v - tile <= 63
	r.flags = (uint16_t)(nc | (pos << CR_POS_SHIFT) | (far ? CR_FAR : 0) | (need_rel << CR_NEED_SHIFT));
	r.pad = (uint16_t)min(65535u, need ? v + 1u - need : 65535u);   // distance to the most recent source other than the chained one
	return r;
}
__device__ __forceinline__ ChainRec make_chain_rec(const uint32_t *cand, const uint8_t *ncand, uint32_t v, uint32_t seg_begin, uint32_t ring_floor)
{
	const uint32_t nc = ncand[v];
	uint32_t row[6] = { 0, 0, 0, 0, 0, 0 };
	if (nc <= 2) { const uint32_t *p = cand + (size_t)v * kCand2; for (uint32_t j = 0; j < 3 * nc; ++j) row[j] = p[j]; }
	return make_chain_rec_ids(nc, row, v, seg_begin, ring_floor);
}
// cand / ncand as written by k_candidates_ids; seg_start: first decode rank of every component + end sentinel
__global__ __launch_bounds__(256) void k_chain_records(const uint32_t *cand, const uint8_t *ncand, uint32_t n, const uint32_t *seg_start, uint32_t nseg, ChainRec *out)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= n) return;
	uint32_t lo = 0, hi = nseg;   // component of v: the last one that starts at or before v
	while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_start[mid] <= v) lo = mid; else hi = mid; }
	const uint32_t seg_begin = seg_start[lo];
	out[v] = make_chain_rec(cand, ncand, v, seg_begin, seg_begin);
}

// g_chain_timeout (above) is set when a wavefront gave up waiting for another one (a hand-over that takes longer than ~a second
// is a bug, not load): the grid still drains, and the host turns the flag into an error instead of returning a wrong mesh.

// What the chain's rare paths need, kept in LDS behind the team's words (written once per segment): the out-of-line functions
// below take a vertex id and the team's words, nothing else.  With their operands as arguments -- the records' base, stride and
// offset, ring, bounds, the other components' progress table: eleven scalars that have to be live at every call site -- the chain's
// loop had 36 more of its own values in spill lanes, every one of them a v_readlane on the serial path (- 4 % of the kernel).
struct ChainCold {
	const uint8_t *rec;      // records, already at the component's offset
	const void *ring;
	CrossSync xs;
	int32_t stride, q, comp;
	uint32_t ring_floor, seg_begin;
	ConnView cv;             // (for the fan walk of a vertex with more candidates than a table row holds)
	const uint32_t *order_v;
};
constexpr uint32_t kCold0 = kHand0 + kHand;                          // (8-byte aligned: 66 words)
constexpr uint32_t kSync3Words = kCold0 + (sizeof(ChainCold) + 3) / 4;
__device__ __forceinline__ const ChainCold &chain_cold(const uint32_t *sync) { return *(const ChainCold*)(sync + kCold0); }

// rare: a source older than the LDS ring (or of an earlier launch), read from the records by vertex id.  Out of line on purpose:
// the chain's hot loop has to stay small enough for the instruction cache.
template <typename T>
__device__ __attribute__((noinline)) uint32_t chain_far_value(uint32_t id, uint32_t *sync)
{
	const ChainCold &c = chain_cold(sync);
	if (id < c.seg_begin) wait_owner(c.xs, c.comp, id);   // another component's chain (or an earlier slice: no flags, already final)
	else {   // this chain's own output: wait for its flush
		uint32_t spins = 0;
#pragma nounroll
		while (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= id) {
			__builtin_amdgcn_s_sleep(2);
			if ((++spins & 1023u) == 0u && __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;   // the chain was given up
			if (spins > kSpinLimit) { atomicOr(&g_chain_timeout, 2u); __hip_atomic_store(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
		}
	}
	return (uint32_t)__hip_atomic_load((const T*)(c.rec + (size_t)id * c.stride), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rare: a vertex with more candidates than a table row holds, predicted by walking its fan
template <typename T>
__device__ __attribute__((noinline)) uint32_t chain_fan_pred(uint32_t vb, uint32_t *sync)
{
	const ChainCold &c = chain_cold(sync);
	const TopoD tp{ c.cv };
	const T *ring = (const T*)c.ring;
	auto old_value = [&](uint32_t id) -> uint32_t {   // (every source of a vertex that is evaluated on its own is final)
		if (id >= vb) return 0u;
		if (id >= c.ring_floor && vb - id <= kRing3 - 64u) return (uint32_t)ring[id & (kRing3 - 1)];
		return chain_far_value<T>(id, sync);
	};
	int64_t acc = 0;
	uint32_t n = 0;
	fan_ids(tp, c.order_v[vb], vb, [&](uint32_t a, uint32_t b, uint32_t o) {
		acc += (int64_t)cm::parallelogram<T>((T)old_value(a), (T)old_value(b), (T)old_value(o), c.q);
		++n;
	});
	return n ? (uint32_t)(T)cm::mean_of(acc, (int64_t)n) : 0u;
}

struct Map3 { int32_t k, A, D; };   // x -> floor((x + A) / 2^k) + D, 0 <= A < 2^k, k <= 16
// g after f, exact for 0 <= x < 2^16
__device__ __forceinline__ Map3 compose3(const Map3 &f, const Map3 &g)
{
	const int32_t T = f.D + g.A;
	const int32_t rT = T & ((1 << g.k) - 1), q = T >> g.k;
	const int32_t K = f.k + g.k;
	const int32_t A_small = f.A + (rT << f.k);
	// K > 16: step function with threshold theta = (2^k2 - rT) 2^k1 - A1 (>= 1); at or above 2^16 it never fires
	const uint32_t u = (uint32_t)((1 << g.k) - rT);
	const bool sat = (u >> (17 - f.k)) != 0u;            // u 2^k1 >= 2^17  (f.k == 0: u <= 2^16, never)
	const uint32_t theta = (u << f.k) - (uint32_t)f.A;   // meaningful when !sat: < 2^17
	const int32_t A_big = (sat || theta >= 65536u) ? 0 : (int32_t)(65536u - theta);
	Map3 h;
	h.k = K <= 16 ? K : 16;
	h.A = K <= 16 ? A_small : A_big;
	h.D = g.D + q;
	return h;
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ Map3 dpp3(const Map3 &m)
{
	Map3 r;   // lanes without a source (row start, masked rows) receive the identity map (0, 0, 0)
	r.k = __builtin_amdgcn_update_dpp(0, m.k, CTRL, ROWMASK, 0xf, false);
	r.A = __builtin_amdgcn_update_dpp(0, m.A, CTRL, ROWMASK, 0xf, false);
	r.D = __builtin_amdgcn_update_dpp(0, m.D, CTRL, ROWMASK, 0xf, false);
	return r;
}
// g after f where f.k + g.k <= 16 is known: no step function to fold (8 instructions instead of 20)
__device__ __forceinline__ Map3 compose3_small(const Map3 &f, const Map3 &g)
{
	const int32_t T = f.D + g.A;
	Map3 h;
	h.k = f.k + g.k;
	h.A = f.A + ((T & ((1 << g.k) - 1)) << f.k);
	h.D = g.D + (T >> g.k);
	return h;
}
// inclusive scan over the wavefront: lane l gets m_l o m_(l-1) o ... o m_0.  SMALL: every lane's k is 0 or 1 (the caller has
// looked), so the spans of the first four steps -- 2, 4, 8, 16 lanes -- stay at k <= 16 and take the short composition: a tile
// of a long regular stretch is prepared in ~ 400 instructions, half of them this scan, and eight wavefronts that prepare
// one tile each are what bounds such a stretch (round 6: every eighth tile of a long row of tiles without heads waited
// 2 000 ticks for its preparation on the 28 M-triangle torus)
template <bool SMALL> __device__ __forceinline__ Map3 scan3_t(Map3 m)
{
	if (SMALL) {
		m = compose3_small(dpp3<0x111, 0xf>(m), m);
		m = compose3_small(dpp3<0x112, 0xf>(m), m);
		m = compose3_small(dpp3<0x114, 0xf>(m), m);
		m = compose3_small(dpp3<0x118, 0xf>(m), m);
	} else {
		m = compose3(dpp3<0x111, 0xf>(m), m);   // row_shr:1
		m = compose3(dpp3<0x112, 0xf>(m), m);   // row_shr:2
		m = compose3(dpp3<0x114, 0xf>(m), m);   // row_shr:4
		m = compose3(dpp3<0x118, 0xf>(m), m);   // row_shr:8
	}
	m = compose3(dpp3<0x142, 0xa>(m), m);   // row_bcast:15 into rows 1 and 3
	m = compose3(dpp3<0x143, 0xc>(m), m);   // row_bcast:31 into rows 2 and 3
	return m;
}
__device__ __forceinline__ Map3 scan3(Map3 m) { return scan3_t<false>(m); }

// (max of the lower bounds, min of the upper bounds of the lanes a scan step combines; lanes without a partner keep theirs)
template <int CTRL, int ROWMASK> __device__ __forceinline__ void fold_bounds(int32_t &lower, int32_t &upper)
{
	lower = max(lower, __builtin_amdgcn_update_dpp(0, lower, CTRL, ROWMASK, 0xf, false));              // (every lower bound is >= 0)
	upper = min(upper, __builtin_amdgcn_update_dpp(0x7fffffff, upper, CTRL, ROWMASK, 0xf, false));
}

// segmented: bit 8 of k marks a lane that starts a run; lane l gets m_l o ... o m_s, s = the last start at or below l
constexpr int32_t kRunStart = 256;
template <bool SMALL> __device__ __forceinline__ Map3 scan3_runs_t(Map3 m)
{
	auto step = [&](const Map3 &p, bool small) {
		const bool start = (m.k & kRunStart) != 0;
		Map3 pm = p, mm = m;
		pm.k &= kRunStart - 1; mm.k &= kRunStart - 1;
		Map3 c = small ? compose3_small(pm, mm) : compose3(pm, mm);
		c.k |= p.k & kRunStart;   // a start anywhere in the span
		m.k = start ? m.k : c.k; m.A = start ? m.A : c.A; m.D = start ? m.D : c.D;
	};
	step(dpp3<0x111, 0xf>(m), SMALL);
	step(dpp3<0x112, 0xf>(m), SMALL);
	step(dpp3<0x114, 0xf>(m), SMALL);
	step(dpp3<0x118, 0xf>(m), SMALL);
	step(dpp3<0x142, 0xa>(m), false);
	step(dpp3<0x143, 0xc>(m), false);
	m.k &= kRunStart - 1;
	return m;
}
__device__ __forceinline__ Map3 scan3_runs(Map3 m) { return scan3_runs_t<false>(m); }

// Several wavefronts share one chain (blockDim.x / 64 of them): wavefront w owns the tiles t = w (mod W).  A tile whose
// first run depends on the tiles before it only through its predecessor's value is PREPARED (gathers, maps, scan) while
// the tiles before it are still being finished by the other wavefronts; only "take the predecessor's value, apply the
// composed maps, verify, publish" is serial.  Hand-over through LDS, one word per tile: sync[kHand0 + (t & 63)] = (t + 1) << 16 |
// value of the tile's last vertex, written after the tile's values -- the word is flag and operand at once, so the next
// tile's owner has its input with the load that ends its wait (no second LDS round trip, no counter, no s_waitcnt on the
// serial path).  sync[0] = the chain was given up (a wait ran into its bound), sync[1] = vertices below it have reached global
// memory.  Every wait is for an earlier tile, whose owner never waits for a later one, so the chain always advances.
// What is behind the wait, by the tile's kind (ticks per tile on the headline mesh, scripts/chain_log.py): no heads -- a range
// test and the values (~ 500, the LDS round trip of the hand-over included); one head from its record's slots -- the same per
// run with the head between them (~ 900); more heads, or a value outside its interval -- runs verified against their true
// arithmetic, heads one by one (2 000 - 3 500); tiles whose sources are too recent to be prepared ahead -- prepared beside
// the d tiles that separate them from their latest source (1 700 - 3 300), or run by run when the vertices name each other.
template <typename T>
__device__ void unpredict3_segment(const TopoD &tp, const uint32_t *order_v, uint32_t nvtx_total, uint32_t seg_begin, uint32_t seg_end,
                                   const uint32_t *cand, const uint8_t *ncand, const ChainRec *crec, const uint8_t *planes, uint8_t *rec,
                                   int stride, int off, int q, int plane0, T *ring, uint32_t ring_floor, uint32_t *sync, const CrossSync &xs, int comp)
{
	static_assert(sizeof(T) <= 2 && !(T(-1) < T(0)), "unsigned components of at most 16 bits");
	const int lane = threadIdx.x & 63;
	const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), W = blockDim.x >> 6;   // uniform, and known to be: scalar loop control
	constexpr uint32_t mask = kRing3 - 1;
	const uint32_t top = ev_top<T>(q), wrap = (uint32_t)(T)(~T(0));
	const uint32_t t_first = seg_begin & ~63u;
	// LDS executes a wavefront's accesses in order, so "values, then counter" on one side and "counter, then values" on the
	// other need no fence -- and must not get one: acquire / release would also wait for vector memory, i.e. for the prefetch
	// of the next tile, on every hand-over.  Relaxed atomics keep the compiler from caching or reordering the counters.
	// value of a vertex that is final before the current run, wherever it lives
	auto old_value = [&](uint32_t id, uint32_t cur) -> uint32_t {
		if (id >= cur) return 0u;   // the chained source of a vertex inside the run: not final yet, and never used from here
		if (id >= ring_floor && cur - id <= kRing3Near) return (uint32_t)ring[id & mask];
		return chain_far_value<T>(id, sync);
	};
	uint4 nx_rec = make_uint4(0, 0, 0, 0);
	uint32_t nx_b0 = 0, nx_b1 = 0;
	auto request = [&](uint32_t tb) {
		const uint32_t v = tb + lane;
		nx_rec = make_uint4(0, 0, 0, 0); nx_b0 = nx_b1 = 0;
		if (v >= seg_begin && v < seg_end) {
			nx_rec = *(const uint4*)(crec + v);
			nx_b0 = planes[(size_t)plane0 * nvtx_total + v];
			if (sizeof(T) == 2) nx_b1 = planes[(size_t)(plane0 + 1) * nvtx_total + v];
		}
	};
	// a slice that continues a chain finds the end of the previous slice in the ring again
	for (uint32_t b = ring_floor + 64 * wv; b < seg_begin; b += 64 * W) { const uint32_t v = b + lane; if (v < seg_begin) ring[v & mask] = ldq<T>(rec + (size_t)v * stride + off); }
	if (threadIdx.x == 0) {
		sync[0] = 0; sync[1] = seg_begin;
		ChainCold c;
		c.cv = tp.c; c.order_v = order_v; c.rec = rec + off; c.ring = ring; c.xs = xs; c.stride = stride; c.q = q; c.comp = comp; c.ring_floor = ring_floor; c.seg_begin = seg_begin;
		*(ChainCold*)(sync + kCold0) = c;
	}
	if (threadIdx.x < kHand) sync[kHand0 + threadIdx.x] = 0;
	__syncthreads();
	HRY_CLK(unsigned long long ck_wait = 0, ck_serial = 0, ck_prep = 0, ck_t0 = 0, ck_t1 = 0, ck_tiles = 0, ck_early = 0, ck_retry = 0, ck_runs = 0, ck_bigs = 0, ck_clean = 0, ck_clean_n = 0, ck_r2 = 0, ck_r2len = 0, ck_r2_8 = 0, ck_r2_16 = 0, ck_r2_32 = 0, ck_r2_afterbig = 0, ck_rowt = 0, ck_rown = 0, ck_fast = 0, ck_slott = 0, ck_slotn = 0, ck_runt = 0, ck_dense = 0, ck_begin = __builtin_amdgcn_s_memtime(); bool ck_is_clean = true;)
	request(t_first + 64 * wv);
	for (uint32_t tb = t_first + 64 * wv; tb < seg_end; tb += 64 * W) {
		if (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;   // another wavefront's wait ran into its bound
		HRY_CLK(ck_t0 = __builtin_amdgcn_s_memtime(); ++ck_tiles; ck_is_clean = true; const unsigned long long ck_runs0 = ck_runs;)
		const uint4 cr = nx_rec;
		const uint32_t code = nx_b0 | (nx_b1 << 8);
		if (tb + 64 * W < seg_end) request(tb + 64 * W);
		const uint32_t tile_idx = (tb - t_first) >> 6;
		bool waited = false;
		HRY_LOG(uint32_t log_kind = 0;)
		HRY_MARK(unsigned long long mk0 = 0, mk1 = 0, mk2 = 0;)
		uint32_t x_prev = 0;    // the value of vertex tb - 1, as handed over by its tile
		uint32_t x_out = 0;     // the value of this tile's last finished vertex (uniform)
		auto wait_prev = [&]() {
			if (waited) return;
			waited = true;
			HRY_CLK(const unsigned long long w0 = __builtin_amdgcn_s_memtime(); ck_prep += w0 - ck_t0;)
			HRY_CLK(struct Stamp { unsigned long long &a, &b, w0; __device__ ~Stamp() { b = __builtin_amdgcn_s_memtime(); a += b - w0; } } stamp{ ck_wait, ck_t1, w0 };)
			if (tile_idx == 0u) return;
			// hand-written: the compiler turns this loop into an exec-mask loop with a dozen mask operations per round and on
			// the way out, all of them on the serial path.  The word is the same for every lane: one SGPR, scalar branches.
			const uint32_t slot_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&sync[kHand0 + ((tile_idx - 1u) & (kHand - 1u))];   // LDS byte address
			const uint32_t want = tile_idx & 0xffffu;
			uint32_t w, left = kSpinLimit, tmp_v, tmp_s;
			asm volatile("1:\n\t"
			             "ds_read_b32 %[v], %[addr]\n\t"
			             "s_waitcnt lgkmcnt(0)\n\t"
			             "v_readfirstlane_b32 %[w], %[v]\n\t"
			             "s_lshr_b32 %[t], %[w], 16\n\t"
			             "s_cmp_eq_u32 %[t], %[want]\n\t"
			             "s_cbranch_scc1 2f\n\t"
			             HRY_CHAIN_POLL_PAUSE_ASM
			             "s_sub_u32 %[left], %[left], 1\n\t"
			             "s_cmp_lg_u32 %[left], 0\n\t"
			             "s_cbranch_scc1 1b\n\t"
			             "2:\n\t"
			             "s_and_b32 %[t], %[w], 0xffff"
			             : [w] "=&s"(w), [t] "=&s"(tmp_s), [v] "=&v"(tmp_v), [left] "+s"(left)
			             : [addr] "v"(slot_addr), [want] "s"(want)
			             : "memory", "scc");
			if (left == 0u) { atomicOr(&g_chain_timeout, 1u); __hip_atomic_store(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // (the loop ends at its next turn: sync[0])
			asm volatile("" ::: "memory");   // ring reads stay behind the wait
			HRY_MARK(mk0 = __builtin_amdgcn_s_memtime();)
			x_prev = tmp_s;
		};
		const uint32_t v = tb + lane;
		const uint32_t lo = seg_begin > tb ? seg_begin - tb : 0u, hi = min(64u, seg_end - tb);
		const bool valid = (uint32_t)lane >= lo && (uint32_t)lane < hi;
		const uint32_t slot0 = cr.x & 0xffffu, slot1 = cr.x >> 16, slot2 = cr.y & 0xffffu, slot3 = cr.y >> 16, slot4 = cr.z & 0xffffu, slot5 = cr.z >> 16;
		const uint32_t flags = cr.w & 0xffffu, gap = cr.w >> 16;
		const uint32_t nc = flags & CR_NC, pos = (flags >> CR_POS_SHIFT) & 7u, need_rel = flags >> CR_NEED_SHIFT;
		const bool far = (flags & CR_FAR) != 0;
		// every source other than the predecessor lies in a tile this wavefront has already seen finished (<= t - W)
		const bool settled = gap > (uint32_t)lane + 64u * (W - 1u);
		// HEADS: vertices that are evaluated on their own in the serial part, from the ring -- the ones with more than two
		// candidates (no map of the scan's family) and the ones with a source, other than the predecessor, that is not final yet
		// when the tile is prepared.  Everything between two heads is a run whose maps are composed NOW: behind a head only
		// "apply the composed maps to the head's value, verify" is left.
		const bool head = valid && (nc == CR_BIG || (nc != 0u && !settled));
		const uint64_t headmask = __ballot(head);
		const uint32_t nheads = (uint32_t)__builtin_popcountll(headmask);
		// a head with at most two candidates, all of them in the ring, is evaluated on its own lane from its record's slots;
		// the others (more candidates, or a source older than the ring) from their candidate rows, which are fetched ahead
		const uint64_t bigmask = __ballot(valid && nc == CR_BIG);
		const uint64_t rowheads = __ballot(head && (nc == CR_BIG || far));
		// otherwise: a tile with many recent sources (small or irregular meshes), cut into runs as they come
		const bool prepared = nheads <= kMaxHeads && (uint32_t)__builtin_popcountll(rowheads) <= 8u;
		const uint64_t rowmask = prepared ? rowheads : bigmask;
		// The candidate rows of the tile's first eight heads are fetched now, ahead of the chain (0.2 - 0.7 heads per tile): lane l
		// holds candidate l & 7 of the (l >> 3)-th of them.  Sources older than the ring are final in the records by now and are
		// fetched here as well: on the serial path each of them is a trip to memory (1 - 2 us, as much as two whole tiles).
		uint32_t pf_pos = 64u, pf_n = 0, pf_a = 0, pf_b = 0, pf_o = 0, pf_va = 0, pf_vb = 0, pf_vo = 0, pf_far = 0;
		if (rowmask) {
			uint64_t bm = rowmask;
#pragma nounroll
			for (uint32_t j = 0; j < 8u && bm; ++j) { const uint32_t p = (uint32_t)__builtin_ctzll(bm); bm &= bm - 1ull; if ((uint32_t)lane >> 3 == j) pf_pos = p; }
			if (pf_pos < 64u) {
				const uint32_t vb = tb + pf_pos;
				pf_n = ncand[vb];
				if (pf_n != 0xffu && (uint32_t)(lane & 7) < pf_n) {
					const uint32_t *row = cand_row(cand, nvtx_total, vb, pf_n) + 3 * (lane & 7);
					pf_a = row[0]; pf_b = row[1]; pf_o = row[2];
					auto is_far = [&](uint32_t id) { return id < vb && !(id >= ring_floor && vb - id <= kRing3Near); };
					if (is_far(pf_a)) { pf_va = chain_far_value<T>(pf_a, sync); pf_far |= 1u; }
					if (is_far(pf_b)) { pf_vb = chain_far_value<T>(pf_b, sync); pf_far |= 2u; }
					if (is_far(pf_o)) { pf_vo = chain_far_value<T>(pf_o, sync); pf_far |= 4u; }
				}
			}
		}
		UnfoldPre uf;
		uf.setup(code, top, wrap);
		// ---- the map of a vertex, from the ring as it is when this is called: x (the predecessor's value) -> value
		bool keepl = false;
		uint32_t v0 = 0, p1c = 0;
		int32_t bo0 = 0;
		const bool two = nc == 2;
		Map3 g;
		auto build_maps = [&](bool sel, uint32_t cur, uint32_t const_lane) {   // cur: sources from here on are not final (only the chained one is)
			uint32_t sv0 = ring[slot0], sv1 = ring[slot1], sv2 = ring[slot2], sv3 = ring[slot3], sv4 = ring[slot4], sv5 = ring[slot5];
			if (__ballot(sel && far)) {
				if (sel && far) {   // some source is older than the ring or belongs to an earlier component: by vertex id
					const uint32_t *row = cand + (size_t)v * kCand2;   // (at most two candidates here)
					const uint32_t a0 = old_value(row[0], cur), a1 = old_value(row[1], cur), a2 = old_value(row[2], cur);
					uint32_t a3 = a0, a4 = a1, a5 = a2;
					if (nc == 2) { a3 = old_value(row[3], cur); a4 = old_value(row[4], cur); a5 = old_value(row[5], cur); }
					// the chained source is not final yet unless this vertex starts the run; its slot value is ignored below
					sv0 = a0; sv1 = a1; sv2 = a2; sv3 = a3; sv4 = a4; sv5 = a5;
				}
			}
			if (nc == 0) { sv0 = sv1 = sv2 = sv3 = sv4 = sv5 = 0; }
			// a vertex without a source inside the run is a constant; const_lane reads its predecessor like any older source
			keepl = pos == CR_POS_NONE || (uint32_t)lane == const_lane;
			const uint32_t p0e = med3_i32((int32_t)(sv0 + sv1 - sv2), 0, (int32_t)top), p1e = med3_i32((int32_t)(sv3 + sv4 - sv5), 0, (int32_t)top);
			v0 = uf.apply((p0e + p1e + 1u) >> 1, top) & wrap;
			// chained form: candidate kp holds the predecessor as source a or b
			const bool kp = pos >= 3u, isb = pos == 1u || pos == 4u;
			const uint32_t sa = kp ? sv3 : sv0, sb = kp ? sv4 : sv1, so = kp ? sv5 : sv2;
			bo0 = (int32_t)((isb ? sa : sb) - so);
			p1c = kp ? p0e : p1e;
			const int32_t tsum = bo0 + (int32_t)p1c + 1;
			g.k = keepl ? 16 : two ? 1 : 0;
			g.A = keepl ? 0 : two ? (tsum & 1) : 0;
			g.D = keepl ? (int32_t)v0 : two ? (tsum >> 1) + (int32_t)uf.delta : bo0 + (int32_t)uf.delta;
		};
		auto scan_run = [&](uint32_t s, uint32_t e) -> Map3 {   // lane l of [s, e): g_l o ... o g_s
			Map3 m = g;
			if ((uint32_t)lane < s || (uint32_t)lane >= e) { m.k = 0; m.A = 0; m.D = 0; }
			return scan3(m);
		};
		// a vertex evaluated on its own from the ring: candidate k on lane l0 + k (table order), or by walking the fan
		auto eval_alone = [&](uint32_t s, bool fetched, uint32_t bj) -> uint32_t {
			HRY_CLK(++ck_bigs; ck_is_clean = false;)
			const uint32_t vb = tb + s;
			const uint32_t l0 = fetched ? 8u * bj : 0u;   // lanes l0 .. l0 + 7 hold its candidates
			const uint32_t n0 = fetched ? rl(pf_n, l0) : (uint32_t)ncand[vb];
			HRY_LOG(if (comp == 0 && lane == 0 && !HRY_MARK(1 +) 0) { unsigned long long &mk = g_chain_marks[(tb >> 6) & ((1u << 18) - 1u)]; mk = (mk << 8) | (n0 & 0xffu); })
			T pred = T(0);
			if (n0 != 0xff) {
				uint32_t pk = 0;
				if (fetched) {
					// branch-free: every lane reads the ring at its own (valid or zero) ids; prediction.h:121-138 as in LaneEvalSmall
					uint32_t ia = pf_a, ib = pf_b, io = pf_o, fr = pf_far;
					asm volatile("" : "+v"(ia), "+v"(ib), "+v"(io), "+v"(fr));   // addresses and tests computed here, for a head, not hoisted into every tile's serial part
					const uint32_t ra = ring[ia & mask], rb = ring[ib & mask], ro = ring[io & mask];
					const uint32_t a = (fr & 1u) ? pf_va : ra, b = (fr & 2u) ? pf_vb : rb, o = (fr & 4u) ? pf_vo : ro;
					const bool mine = (uint32_t)lane >> 3 == bj && (uint32_t)(lane & 7) < n0;
					pk = mine ? med3_i32((int32_t)(a + b - o), 0, (int32_t)top) : 0u;
				} else if ((uint32_t)lane < n0) {
					const uint32_t *row = cand_row(cand, nvtx_total, vb, n0) + 3 * lane;
					pk = (uint32_t)cm::parallelogram<T>((T)old_value(row[0], vb), (T)old_value(row[1], vb), (T)old_value(row[2], vb), q);
				}
				// the candidates' sum: the lanes outside l0 .. l0 + 7 hold 0, three row shifts put the group's total on its last lane
				// (transform.h:91 for unsigned values: (sum + n / 2) / n, as chain_predict)
				uint32_t sum = pk & wrap;
				sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x111, 0xf, 0xf, true);   // row_shr:1
				sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x112, 0xf, 0xf, true);   // row_shr:2
				sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x114, 0xf, 0xf, true);   // row_shr:4
				const uint32_t tot = rl(sum, l0 + 7u) + (n0 >> 1);
				pred = (T)(n0 == 1u ? tot : n0 == 2u ? tot >> 1 : n0 == 4u ? tot >> 2 : n0 == 8u ? tot >> 3 : div_small(tot, n0));
			} else {
				pred = (T)chain_fan_pred<T>(vb, sync);
			}
			// prediction.h:46-64 through the vertex's own lane (uf holds its residual code)
			const uint32_t val = uf.apply((uint32_t)pred, top) & wrap;
			if ((uint32_t)lane == s) ring[vb & mask] = (T)val;
			return rl(val, s);
		};
		// the run [s, e) from the value x before it: apply the composed maps, verify every vertex against its true arithmetic
		auto finish_run = [&](uint32_t s, uint32_t e, Map3 F, uint32_t x) -> uint32_t {
			const bool active = (uint32_t)lane >= s && (uint32_t)lane < e;
			uint32_t xh = 0;
			for (int attempt = 0;; ++attempt) {
				if (attempt) F = scan_run(s, e);
				xh = (uint32_t)((((int32_t)x + F.A) >> F.k) + F.D);
				uint32_t xp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xh, 0x138, 0xf, 0xf, false);   // wave_shr:1
				xp = (uint32_t)lane == s ? x : xp;
				const uint32_t p0 = med3_i32((int32_t)(xp + (uint32_t)bo0), 0, (int32_t)top);
				const uint32_t pred = two ? (p0 + p1c + 1u) >> 1 : p0;
				uint32_t ap = uf.apply(pred, top) & wrap;
				asm volatile("" : "+v"(ap));   // straight-line: a branch around ten instructions costs a lone wavefront more than they do
				const uint32_t tv = keepl ? v0 : ap;
				const uint64_t bad = __ballot(active && tv != xh);
				if (!bad) break;
				HRY_CLK(++ck_retry; ck_is_clean = false;)
				const uint32_t j = (uint32_t)__builtin_ctzll(bad);
				if ((uint32_t)lane == j) { keepl = true; v0 = tv; g.k = 16; g.A = 0; g.D = (int32_t)tv; xh = tv; }
				if (attempt == 1) {
					// the data keeps leaving the speculated form: finish the run vertex by vertex
					for (uint32_t i = j + 1; i < e; ++i) {
						const uint32_t xq = rl(xh, i - 1);
						const uint32_t q0 = med3_i32((int32_t)(xq + (uint32_t)bo0), 0, (int32_t)top);
						const uint32_t pr = two ? (q0 + p1c + 1u) >> 1 : q0;
						const uint32_t t2 = keepl ? v0 : (uf.apply(pr, top) & wrap);
						if ((uint32_t)lane == i) xh = t2;
					}
					break;
				}
			}
			if (active) ring[v & mask] = (T)xh;
			return rl(xh, e - 1u);
		};
		uint32_t x = 0;   // value of the vertex before the current position (uniform)
		// the maps of the vertices that are not heads and ONE segmented scan for all the runs between the heads (a run starts at
		// the tile's first vertex and behind every head)
		auto compose_runs = [&](uint64_t hm) -> Map3 {
			const bool hd = (hm >> lane) & 1ull;
			build_maps(valid && !hd, tb + lo, 64u);
			Map3 F = g;
			if (!valid || hd) { F.k = 0; F.A = 0; F.D = 0; }
#ifndef HRY_CHAIN_FULL_SCANS
			// (the usual tile: every map is of a lone candidate or of the mean of two -- k = 0 or 1 -- and mostly there is no head)
			if (!__ballot(F.k > 1)) {
				if (hm == 0ull) return scan3_t<true>(F);   // one run (the lanes before it hold the identity): no runs' flags either
				if ((uint32_t)lane == lo || hd || ((hm << 1) >> lane) & 1ull) F.k |= kRunStart;
				return scan3_runs_t<true>(F);
			}
#endif
			if ((uint32_t)lane == lo || hd || ((hm << 1) >> lane) & 1ull) F.k |= kRunStart;
			return scan3_runs(F);
		};
		// heads one by one (alone: from their candidate rows, fetched ahead for the first eight of pfm; else on their own lane from
		// the record's slots, the predecessor included: LaneEvalSmall::eval), the runs between them from their composed maps
		auto serial_part = [&](uint64_t hm, uint64_t alone, uint64_t pfm, const Map3 &Fm) {
			for (uint32_t s = lo; s < hi;) {
				if ((hm >> s) & 1ull) {
					HRY_CLK(const unsigned long long h0t = __builtin_amdgcn_s_memtime();)
					if ((alone >> s) & 1ull) {
						const uint32_t bj = (uint32_t)__builtin_popcountll(pfm & ((1ull << s) - 1ull));
						x = eval_alone(s, ((pfm >> s) & 1ull) != 0 && bj < 8u, bj);
						HRY_CLK(asm volatile("" :: "s"(x)); ck_rowt += __builtin_amdgcn_s_memtime() - h0t; ++ck_rown;)
					} else {
						HRY_CLK(++ck_bigs; ck_is_clean = false;)
						const uint32_t h0 = ring[slot0], h1 = ring[slot1], h2 = ring[slot2], h3 = ring[slot3], h4 = ring[slot4], h5 = ring[slot5];
						const uint32_t q0 = med3_i32((int32_t)(h0 + h1 - h2), 0, (int32_t)top), q1 = med3_i32((int32_t)(h3 + h4 - h5), 0, (int32_t)top);
						const uint32_t val = uf.apply((q0 + q1 + 1u) >> 1, top) & wrap;
						if ((uint32_t)lane == s) ring[v & mask] = (T)val;
						x = rl(val, s);
						HRY_CLK(asm volatile("" :: "s"(x)); ck_slott += __builtin_amdgcn_s_memtime() - h0t; ++ck_slotn;)
					}
					HRY_MARK(if (!mk2) { asm volatile("" :: "s"(x)); mk2 = __builtin_amdgcn_s_memtime(); })
					++s;
					continue;
				}
				const uint64_t above = hm & ~((1ull << s) - 1ull);   // the run that starts at s ends before the next head
				const uint32_t e = above ? (uint32_t)__builtin_ctzll(above) : hi;
				HRY_CLK(++ck_runs; if (s != lo) { ++ck_r2; ck_r2len += e - s; })
				HRY_CLK(const unsigned long long r0t = __builtin_amdgcn_s_memtime();)
				x = finish_run(s, e, Fm, x);
				HRY_MARK(if (!mk1) { asm volatile("" :: "s"(x)); mk1 = __builtin_amdgcn_s_memtime(); })
				HRY_CLK(asm volatile("" :: "s"(x)); ck_runt += __builtin_amdgcn_s_memtime() - r0t;)
				s = e;
			}
		};
		// the hand-over word of an earlier tile: that tile and every one before it are finished
		auto wait_tile = [&](uint32_t u) {
			uint32_t spins = 0;
#pragma nounroll
			while ((__hip_atomic_load(&sync[kHand0 + (u & (kHand - 1u))], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 16) != ((u + 1u) & 0xffffu)) {
				if (++spins > kSpinLimit) { atomicOr(&g_chain_timeout, 1u); __hip_atomic_store(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
				if ((spins & 1023u) == 0u && __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
			}
			asm volatile("" ::: "memory");   // ring reads stay behind the wait
		};
		uint64_t hm = headmask, alone = rowheads, pfm = rowheads;
		bool composed = prepared;
		if (!prepared) {
			// ---- a tile with many recent sources (small or irregular meshes, the first and the last rings of a closing border).  Most
			// of them are recent but not of this tile: once the tiles that hold them are finished they are final, and the tile is
			// prepared THEN, with heads only where a source lies inside the tile itself.  The latest source of the tile's other
			// vertices lies d tiles back (d = 0: in the tile before this one): the preparation waits for tile t - 1 - d and runs
			// beside the serial parts of the d tiles between
			const bool inner = valid && (nc == CR_BIG || (nc != 0u && gap <= (uint32_t)lane - lo && gap <= (uint32_t)lane));   // the source is a vertex of this tile
			const uint64_t innermask = __ballot(inner);
			if ((uint32_t)__builtin_popcountll(innermask) <= kMaxHeadsLate) {
				composed = true;
				hm = innermask; alone = __ballot(inner && (nc == CR_BIG || far)); pfm = bigmask;
				const uint32_t back = valid && nc != 0u && !inner ? (gap - (uint32_t)lane - 1u) >> 6 : 0xffffu;   // whole tiles between this one and the vertex's latest source
				uint32_t d = 0;
#pragma nounroll
				for (uint32_t dd = W - 1u; dd >= 1u; --dd) if (!__ballot(back < dd)) { d = dd; break; }
#ifdef HRY_CHAIN_NO_LATE_OVERLAP
				d = 0;
#endif
				if (d == 0u) wait_prev();
				else if (tile_idx > d) wait_tile(tile_idx - 1u - d);
				HRY_LOG(log_kind = 3u | ((uint32_t)__builtin_popcountll(innermask) << 4) | (d << 10);)
			}
		}
		if (composed) {
			// ---- prepared tile: everything but the heads before the tile before this one is finished
			Map3 Fm = compose_runs(hm);
			asm volatile("" : "+v"(Fm.k), "+v"(Fm.A), "+v"(Fm.D));   // the scan is computed before the wait below, not sunk behind it
			HRY_CLK(++ck_early;)
#ifndef HRY_CHAIN_NO_FAST_TILES
			// A tile without heads is ONE run, and whether its speculation holds is a question about the predecessor's value alone:
			// lane l's form (parallelogram inside [0, top], residual code near: prediction.h:58-63) holds iff the value before it
			// lies in an interval [a, b]; that value is the composed map of the lanes before it -- monotone in x, the value handed
			// over -- so the condition pulls back to an interval of x, and the lanes' intervals intersect to ONE: [x_lo, x_lo +
			// x_width].  All of that is computed HERE, before the wait; behind it are a scalar range test, three instructions for
			// the tile's values, three scalar ones for the value handed on.  (x outside: the verified path below, as before.)
			// (round 6) A tile with ONE head that is evaluated from its record's slots -- a corner of the walk's spiral, one tile in four
			// of a regular mesh -- is two runs with the head between them: the same test per run (run 2's is about the head's value)
			// leaves "values of run 1, the head from the ring, values of run 2" behind the wait, without the verified path's scans.
			bool fast = false, one = false;
			int32_t x_lo = 0x7fffffff, x1_lo = 0, x2_lo = 0;   // (x_lo: no interval yet -- no 16-bit value passes the test behind the wait)
			uint32_t x_width = 0, x1_width = 0, x2_width = 0, hs = 0;
			int32_t end_k = 0, end_A = 0, end_D = 0;
#ifndef HRY_CHAIN_NO_ONE_HEAD
			const bool one_cand = (hm & (hm - 1ull)) == 0ull && (hm & alone) == 0ull;
#else
			const bool one_cand = false;
#endif
			if (hm == 0ull || one_cand) {
				const int32_t hf = (int32_t)uf.half, tp_ = (int32_t)top;
				int32_t a = two ? max(0, 2 * hf + 1 - (int32_t)p1c) - bo0 : hf + 1 - bo0;
				int32_t b = two ? min(tp_, 2 * (tp_ - hf) - (int32_t)p1c) - bo0 : tp_ - hf - bo0;
				if (keepl || !valid || ((hm >> lane) & 1ull)) { a = -(1 << 29); b = 1 << 29; }
				// the map in front of the lane (the first vertex of the tile, and of the run behind a head: the value itself -- a head's
				// own map in the scan is the identity)
				Map3 P;
				P.k = __builtin_amdgcn_update_dpp(0, Fm.k, 0x138, 0xf, 0xf, false);   // wave_shr:1
				P.A = __builtin_amdgcn_update_dpp(0, Fm.A, 0x138, 0xf, 0xf, false);
				P.D = __builtin_amdgcn_update_dpp(0, Fm.D, 0x138, 0xf, 0xf, false);
				if ((uint32_t)lane <= lo) { P.k = 0; P.A = 0; P.D = 0; }
				// floor((x + A) / 2^k) + D >= a  <=>  x >= (a - D) 2^k - A;   <= b  <=>  x <= (b - D + 1) 2^k - A - 1   (x + A < 2^17)
				const int32_t room = 0x20000 >> P.k;
				const int32_t t = a - P.D, u = b - P.D + 1;
				int32_t lower = t <= 0 ? 0 : t > room ? 0x7fffffff : (t << P.k) - P.A;
				int32_t upper = u <= 0 ? -1 : u > room ? 0x7fffffff : (u << P.k) - P.A - 1;
				if (hm == 0ull) {
					fold_bounds<0x111, 0xf>(lower, upper);   // (the scan's six steps: lane 63 ends up with every lane's bounds)
					fold_bounds<0x112, 0xf>(lower, upper);
					fold_bounds<0x114, 0xf>(lower, upper);
					fold_bounds<0x118, 0xf>(lower, upper);
					fold_bounds<0x142, 0xa>(lower, upper);
					fold_bounds<0x143, 0xc>(lower, upper);
					const int32_t L = (int32_t)rl((uint32_t)lower, 63u), U = (int32_t)rl((uint32_t)upper, 63u);
					fast = U >= L;
					if (fast) { x_lo = L; x_width = (uint32_t)(U - L); }
					end_k = (int32_t)rl((uint32_t)Fm.k, hi - 1u); end_A = (int32_t)rl((uint32_t)Fm.A, hi - 1u); end_D = (int32_t)rl((uint32_t)Fm.D, hi - 1u);
				} else if (one_cand) {
					hs = (uint32_t)__builtin_ctzll(hm);
					int32_t l1 = (uint32_t)lane < hs ? lower : 0, u1 = (uint32_t)lane < hs ? upper : 0x7fffffff;   // run 1: the lanes below the head
					int32_t l2 = (uint32_t)lane > hs ? lower : 0, u2 = (uint32_t)lane > hs ? upper : 0x7fffffff;   // run 2: the lanes above it
					fold_bounds<0x111, 0xf>(l1, u1); fold_bounds<0x111, 0xf>(l2, u2);
					fold_bounds<0x112, 0xf>(l1, u1); fold_bounds<0x112, 0xf>(l2, u2);
					fold_bounds<0x114, 0xf>(l1, u1); fold_bounds<0x114, 0xf>(l2, u2);
					fold_bounds<0x118, 0xf>(l1, u1); fold_bounds<0x118, 0xf>(l2, u2);
					fold_bounds<0x142, 0xa>(l1, u1); fold_bounds<0x142, 0xa>(l2, u2);
					fold_bounds<0x143, 0xc>(l1, u1); fold_bounds<0x143, 0xc>(l2, u2);
					const int32_t L1 = (int32_t)rl((uint32_t)l1, 63u), U1 = (int32_t)rl((uint32_t)u1, 63u), L2 = (int32_t)rl((uint32_t)l2, 63u), U2 = (int32_t)rl((uint32_t)u2, 63u);
					one = U1 >= L1 && U2 >= L2;
					x1_lo = L1; x1_width = (uint32_t)(U1 - L1);
					x2_lo = L2; x2_width = (uint32_t)(U2 - L2);
					end_k = (int32_t)rl((uint32_t)Fm.k, hi - 1u); end_A = (int32_t)rl((uint32_t)Fm.A, hi - 1u); end_D = (int32_t)rl((uint32_t)Fm.D, hi - 1u);   // (a head in the last lane: the identity)
				}
			}
			wait_prev();
			x = x_prev;   // the first vertex of a slice is never chained
			if ((uint32_t)((int32_t)x - x_lo) <= x_width) {
				const uint32_t xh = (uint32_t)((((int32_t)x + Fm.A) >> Fm.k) + Fm.D);
				if (valid) ring[v & mask] = (T)xh;
				x = (uint32_t)((((int32_t)x + end_A) >> end_k) + end_D);
				HRY_CLK(++ck_runs; ++ck_fast;)
				HRY_LOG(log_kind = prepared ? 1u : 5u | (log_kind & ~0xfu);)
			} else if (one && (uint32_t)((int32_t)x - x1_lo) <= x1_width) {
				const uint32_t xh1 = (uint32_t)((((int32_t)x + Fm.A) >> Fm.k) + Fm.D);
				if (valid && (uint32_t)lane < hs) ring[v & mask] = (T)xh1;
				// the head, on its own lane from its record's slots (the vertex before it is in the ring now)
				const uint32_t h0 = ring[slot0], h1 = ring[slot1], h2 = ring[slot2], h3 = ring[slot3], h4 = ring[slot4], h5 = ring[slot5];
				const uint32_t q0 = med3_i32((int32_t)(h0 + h1 - h2), 0, (int32_t)top), q1 = med3_i32((int32_t)(h3 + h4 - h5), 0, (int32_t)top);
				const uint32_t val = uf.apply((q0 + q1 + 1u) >> 1, top) & wrap;
				const uint32_t xm = rl(val, hs);
				if ((uint32_t)((int32_t)xm - x2_lo) <= x2_width) {
					const uint32_t xh2 = (uint32_t)((((int32_t)xm + Fm.A) >> Fm.k) + Fm.D);
					if (valid && (uint32_t)lane >= hs) ring[v & mask] = (T)((uint32_t)lane == hs ? val : xh2);
					x = (uint32_t)((((int32_t)xm + end_A) >> end_k) + end_D);
				} else {
					if ((uint32_t)lane == hs) ring[v & mask] = (T)val;
					x = hs + 1u < hi ? finish_run(hs + 1u, hi, Fm, xm) : xm;
				}
				HRY_LOG(log_kind = 6u | (log_kind & ~0xfu);)
			} else { HRY_LOG(if (prepared) log_kind = 2u | (nheads << 4) | ((uint32_t)__builtin_popcountll(rowheads) << 10);) serial_part(hm, alone, pfm, Fm); }
#else
			wait_prev();
			x = x_prev;   // the first vertex of a slice is never chained
			serial_part(hm, alone, pfm, Fm);
#endif
		} else {
			// ---- a tile whose vertices need each other: runs are cut where a vertex needs a source inside the run, and prepared
			// when the vertices before them are final
			wait_prev();
			HRY_CLK(++ck_dense;)
			{
				HRY_LOG(log_kind = 4u | ((uint32_t)__builtin_popcountll(bigmask) << 10);)
				for (uint32_t s = lo; s < hi;) {
					if ((bigmask >> s) & 1ull) {
						const uint32_t bj = (uint32_t)__builtin_popcountll(bigmask & ((1ull << s) - 1ull));
						x = eval_alone(s, bj < 8u, bj);
						++s;
						continue;
					}
					const uint64_t above = s >= 63u ? 0ull : ~((2ull << s) - 1ull);
					const uint64_t cut = (__ballot(valid && need_rel > s) | bigmask) & above;
					const uint32_t e = cut ? (uint32_t)__builtin_ctzll(cut) : hi;
					HRY_CLK(++ck_runs; ck_is_clean = false;)
					build_maps((uint32_t)lane >= s && (uint32_t)lane < e, tb + s, s);
					x = finish_run(s, e, scan_run(s, e), 0u);
					s = e;
				}
			}
		}
		x_out = x;
		// the tile is finished: its values are in the ring before the counter moves (release)
		wait_prev();
		asm volatile("" ::: "memory");   // the ring writes of this tile are issued before the word that announces them (LDS runs a wavefront's accesses in order)
		__hip_atomic_store(&sync[kHand0 + (tile_idx & (kHand - 1u))], ((tile_idx + 1u) << 16) | (x_out & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // every lane, the same word: no exec juggling
#ifndef HRY_CHAIN_NO_SCHED_BARRIER
		__builtin_amdgcn_sched_barrier(0);   // (the loop's bookkeeping and the flush's test stay behind the store that everybody waits for: the scheduler had put nine of their instructions in front of it)
#endif
		HRY_MARK(if (comp == 0 && lane == 0) { const unsigned long long mk3 = __builtin_amdgcn_s_memtime(); auto d = [&](unsigned long long a, unsigned long long b) { return a && b && b > a ? (b - a > 0xffffull ? 0xffffull : b - a) : 0ull; };
			g_chain_marks[(tb >> 6) & ((1u << 18) - 1u)] = d(mk0, mk1) | (d(mk0, mk2) << 16) | (d(mk0, mk3) << 32); })
		HRY_LOG(if (comp == 0 && lane == 0) g_chain_log[(tb >> 6) & ((1u << 18) - 1u)] = ((unsigned long long)__builtin_amdgcn_s_memtime() << 16) | log_kind;)
		HRY_CLK(const unsigned long long ck_d = __builtin_amdgcn_s_memtime() - ck_t1; ck_serial += ck_d; if (ck_is_clean && ck_runs - ck_runs0 == 1) { ck_clean += ck_d; ++ck_clean_n; })
		// The tile's values go to the records, from the ring, by its owner (round 6: until then the owner of every 64th tile sent 4 096
		// values and fenced -- 8 000 ticks in which it did not prepare its next tile: one tile in 64 arrived 4 000 - 10 000 ticks late, 8 %
		// of the kernel).  No fence here: the store is complete when this wavefront takes the record it requests at its NEXT turn
		// (vector memory returns in order: that record's wait covers everything issued before the request), i.e. before it can finish
		// the tile 2 W behind this one; so when a tile is handed over, every tile 3 W or more before it is in memory, and sync[1] says
		// so every 64 tiles -- readers of old values are 255 tiles behind.  The last tiles: the kernel's end, or raise_flag()'s fence.
#ifndef HRY_CHAIN_FLUSH_BLOCKS
		if (valid) stq<T>(rec + (size_t)v * stride + off, ring[v & mask]);
		if (((tile_idx + 1u) & 63u) == 0u && tile_idx + 1u > 3u * W && lane == 0)
			__hip_atomic_fetch_max(&sync[1], t_first + 64u * (tile_idx + 1u - 3u * W), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
		const bool last_tile = tb + 64 >= seg_end;
		if (((tile_idx + 1u) & 63u) == 0u || last_tile) {
			const uint32_t upto = min(tb + 64u, seg_end);
			const uint32_t from = max(seg_begin, t_first + ((tile_idx + 1u - 1u) & ~63u) * 64u);
			for (uint32_t b = from; b < upto; b += 64) { const uint32_t u = b + lane; if (u < upto) stq<T>(rec + (size_t)u * stride + off, ring[u & mask]); }
			__threadfence();   // the stores have reached the device's coherence point (far readers load past their L1)
			if (lane == 0) __hip_atomic_fetch_max(&sync[1], upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // flushes of different wavefronts may finish out of order
		}
#endif
	}
	HRY_CLK(if (threadIdx.x == 0 && ck_tiles > 100) printf("chain comp %d: tiles %llu (dense %llu, fast %llu) runs %llu at %llu (later runs %llu, mean length %llu), heads from rows %llu at %llu, from slots %llu at %llu, retries %llu clean %llu at %llu | per tile: prep %llu wait %llu serial %llu | total %llu per tile of the team %llu\n", comp, ck_tiles, ck_dense, ck_fast, ck_runs, ck_runt / (ck_runs ? ck_runs : 1), ck_r2, ck_r2len / (ck_r2 ? ck_r2 : 1), ck_rown, ck_rowt / (ck_rown ? ck_rown : 1), ck_slotn, ck_slott / (ck_slotn ? ck_slotn : 1), ck_retry, ck_clean_n, ck_clean / (ck_clean_n ? ck_clean_n : 1),
	                                    ck_prep / ck_tiles, ck_wait / ck_tiles, ck_serial / ck_tiles, (unsigned long long)__builtin_amdgcn_s_memtime() - ck_begin, ((unsigned long long)__builtin_amdgcn_s_memtime() - ck_begin) / (ck_tiles * W));)
}

template <typename T>
__global__ __launch_bounds__(512) void k_unpredict3(ConnView cv, const uint32_t *order_v, uint32_t nvtx, const uint32_t *cand, const uint8_t *ncand, const ChainRec *crec,
                                                   const uint8_t *planes, ListDesc ld, uint8_t *rec, CompSel sel, const uint32_t *segs, const uint32_t *list_off, CrossSync xs,
                                                   uint32_t n_lists)
{
	__shared__ T ring3[kRing3];
	__shared__ __attribute__((aligned(16))) uint32_t sync3[kSync3Words];
	// (the attribute components of a list on ONE XCD: see k_unpredict2)
	const uint32_t per = 8u * (uint32_t)sel.n, grp = blockIdx.x / per, r = blockIdx.x % per;
	const uint32_t list = grp * 8u + (r & 7u);
	if (list >= n_lists) return;
	const int c = sel.comp[r >> 3];
	TopoD tp{ cv };
	for (uint32_t k = list_off[list]; k < list_off[list + 1]; ++k) {
		const uint32_t b = segs[3 * k], e = segs[3 * k + 1];
		if (b < e) unpredict3_segment<T>(tp, order_v, nvtx, b, e, cand, ncand, crec, planes, rec, ld.stride, ld.off[c], ld.quant[c], ld.plane[c], ring3, b, sync3, xs, c);
		raise_flag(xs, c, segs[3 * k + 2], e);
	}
}
template <typename T>
__global__ __launch_bounds__(1024) void k_unpredict3_range(ConnView cv, const uint32_t *order_v, uint32_t nvtx, const uint32_t *cand, const uint8_t *ncand, const ChainRec *crec,
                                                         const uint8_t *planes, ListDesc ld, uint8_t *rec, CompSel sel, uint32_t v_begin, uint32_t v_end, uint32_t ring_floor)
{
	__shared__ T ring3[kRing3];
	__shared__ __attribute__((aligned(16))) uint32_t sync3[kSync3Words];
	// The chains of the attribute components read the same chain records and write into the same vertex records.  Workgroups
	// go round-robin over the 8 XCDs (each with its own L2): only every eighth workgroup of the launch carries a chain, so
	// that all of them share ONE L2 -- the records are fetched from memory once, and the components' 2-byte stores into a
	// 12-byte record meet in one cache instead of three.
	if (blockIdx.x & 7u) return;
	// (the chain is a handful of wavefronts on a serial dependency; beside it run hundreds of entropy-decoding wavefronts of the same
	// decode, and where one of them shares a SIMD with a chain wavefront the arbiter should know which of the two everybody waits for)
	__builtin_amdgcn_s_setprio(3);
	const int c = sel.comp[blockIdx.x >> 3];
	TopoD tp{ cv };
	const CrossSync none{ nullptr, 0, nullptr, nullptr };   // one component: no other chain to wait for
	unpredict3_segment<T>(tp, order_v, nvtx, v_begin, v_end, cand, ncand, crec, planes, rec, ld.stride, ld.off[c], ld.quant[c], ld.plane[c], ring3, ring_floor, sync3, none, c);
}

// ---------------------------------------------------------------------------------------------------------
void launch_residuals_to_rec(hipStream_t st, const uint8_t *planes, uint32_t n, const ListDesc &ld, uint8_t *rec)
{
	if (n && ld.ncomp) hipLaunchKernelGGL(k_residuals_to_rec, dim3((n + 255) / 256), dim3(256), 0, st, planes, n, ld, rec);
}
void launch_faces_unfold(hipStream_t st, uint32_t n, const ListDesc &ld, uint8_t *rec)
{
	if (n && ld.ncomp) hipLaunchKernelGGL(k_faces_unfold, dim3((n + 255) / 256), dim3(256), 0, st, n, ld, rec);
}
bool unpredict2_applicable(const ListDesc &ld)
{
	for (int c = 0; c < ld.ncomp; ++c) if (ld.stype[c] == 1 || ld.stype[c] == 2 || ld.stype[c] == 3) return false;
	return ld.ncomp > 0;
}
// candidate lists with plain vertex ids (k_unpredict2 resolves ring slots itself)
// (virtual block (b % 8) * per + b / 8: one contiguous range of vertices per XCD and L2, see k_predict_vtx)
// crec != nullptr: the chain record of the vertex (k_unpredict3) from the same registers, for the slice [v0, n) of a chain that
// continues an earlier slice -- one pass over the candidates instead of a second kernel that reads them back
__global__ __launch_bounds__(256) void k_candidates_ids(ConnView cv, const uint32_t *order_v, uint32_t v0, uint32_t n, uint32_t *cand, uint8_t *ncand, uint32_t blocks_per_xcd,
                                                        uint32_t nvtx_total, ChainRec *crec, uint32_t ring_floor)
{
	uint32_t v = v0 + ((blockIdx.x & 7u) * blocks_per_xcd + (blockIdx.x >> 3)) * blockDim.x + threadIdx.x;
	if (v >= n) return;
	TopoD tp{ cv };
	// The first two triples stay in registers, the third to eighth go to LDS (one column per thread: no bank conflicts, nothing
	// behind it).  A private array indexed by the running candidate count lives in scratch memory: round 2's version of this kernel
	// wrote 122 bytes per vertex to HBM, 96 of them its own scratch.
	__shared__ uint32_t s_more[(kCandMax - 2) * 3 * 256];
	uint32_t k = 0, a0 = 0, b0 = 0, o0 = 0, a1 = 0, b1 = 0, o1 = 0;
	fan_ids(tp, order_v[v], v, [&](uint32_t a, uint32_t b, uint32_t o) {
		if (k == 0) { a0 = a; b0 = b; o0 = o; }
		else if (k == 1) { a1 = a; b1 = b; o1 = o; }
		else if (k < (uint32_t)kCandMax) { uint32_t *p = s_more + (size_t)(k - 2) * 3 * 256 + threadIdx.x; p[0] = a; p[256] = b; p[512] = o; }
		++k;
	});
	const uint32_t m = k > (uint32_t)kCandMax ? 0 : k;
	if (m < 2) { a1 = b1 = o1 = 0; }
	if (m < 1) { a0 = b0 = o0 = 0; }
	if (crec) { const uint32_t row[6] = { a0, b0, o0, a1, b1, o1 }; crec[v] = make_chain_rec_ids(k > 2 ? 3u : k, row, v, v0, ring_floor); }
	const bool wide = m > 2;
	const uint64_t need = __ballot(wide);
	if (wide) {   // the whole row goes to the overflow area; the lanes of a wavefront that need one take their rows with ONE atomic
		uint32_t *over = cand + cand_over_at(nvtx_total);
		const uint32_t lane = threadIdx.x & 63u;
		uint32_t base = 0;
		if (lane == (uint32_t)__builtin_ctzll(need)) base = atomicAdd(over, (uint32_t)__builtin_popcountll(need));
		base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(need));
		const uint32_t slot = base + (uint32_t)__builtin_popcountll(need & ((1ull << lane) - 1ull));
		uint32_t *row = over + 16 + (size_t)slot * (kCandMax * 3);
		row[0] = a0; row[1] = b0; row[2] = o0; row[3] = a1; row[4] = b1; row[5] = o1;
		for (uint32_t j = 2; j < (uint32_t)kCandMax; ++j) {
			const uint32_t *p = s_more + (size_t)(j - 2) * 3 * 256 + threadIdx.x;
			const bool have = j < m;
			row[3 * j] = have ? p[0] : 0u; row[3 * j + 1] = have ? p[256] : 0u; row[3 * j + 2] = have ? p[512] : 0u;
		}
		a0 = slot;
	}
	uint2 *out = (uint2*)(cand + (size_t)v * kCand2);
	out[0] = make_uint2(a0, b0); out[1] = make_uint2(o0, a1); out[2] = make_uint2(b1, o1);
	ncand[v] = k > (uint32_t)kCandMax ? 0xff : (uint8_t)k;
}
// words of the candidate table of nvtx vertices (compact rows, header, worst-case overflow)
size_t cand_table_words(uint32_t nvtx) { return cand_over_at(nvtx) + 16 + (size_t)nvtx * (kCandMax * 3); }
void cand_table_reset(hipStream_t st, uint32_t *cand, uint32_t nvtx) { (void)hipMemsetAsync(cand + cand_over_at(nvtx), 0, 64, st); }
void launch_candidates_ids(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t *cand, uint8_t *ncand)
{
	if (!nvtx) return;
	const uint32_t per = ((nvtx + 255) / 256 + 7) / 8;
	cand_table_reset(st, cand, nvtx);
	hipLaunchKernelGGL(k_candidates_ids, dim3(per * 8), dim3(256), 0, st, cv, order_v, 0u, nvtx, cand, ncand, per, nvtx, (ChainRec*)nullptr, 0u);
}
// wavefronts per reconstruction chain (k_unpredict3): HRY_CHAIN_WAVES = 1..8; by default 6 for a large mesh (long rings: more
// look-ahead for the preparation costs no heads; 5 until a tile without heads got its short serial part, round 5: the preparation
// grew by the tile's interval, the chain's turn shrank -- 1 M-triangle torus, one launch: 5.20 ms before, 4.86 with five, 4.69 with
// six, 4.89 with eight; round 6, with the late tiles' preparation beside the tiles before them: 4.85 with six, 4.65 with seven,
// 4.75 with eight; with every tile sent to the records by its owner instead of 64 at a time by one: 4.00 with six, 3.64 with
// seven, 3.47 with eight, 3.44 with ten or twelve, 3.64 with sixteen.  Long regular stretches are bound by the preparation -- a
// tile takes ~ 5 000 ticks to prepare where two wavefronts share a SIMD, and every W-th tile of a row of tiles without heads
// waits for it --: the 28 M-triangle torus 98.7 ms with eight, 81.4 with twelve, 80.6 with sixteen: twelve), 4 otherwise
static uint32_t chain_waves(uint32_t nvtx)
{
	static const uint32_t forced = [] { const char *e = getenv("HRY_CHAIN_WAVES"); int v = e ? atoi(e) : 0; return (uint32_t)(v < 0 ? 0 : v > 16 ? 16 : v); }();
	return forced ? forced : nvtx >= (1u << 18) ? 12u : nvtx >= (1u << 16) ? 8u : 4u;   // (90 000 vertices: 1.08 ms with four, 1.00 with eight or twelve; 202 000: 2.16 / 1.77 / 1.88; 32 000 and below: the same with four and eight)
}
// ---- pipelined decode: one slice [v_begin, v_end) of the vertex chain
// gave_up: the decode's own give-up word (behind its flag table), or nullptr -- a context that shares its device with others may find
// g_chain_timeout read and reset by one of them, its own word not
uint32_t chain_timeout_flags(hipStream_t st, const uint32_t *gave_up)
{
	uint32_t f = 0, own = 0, zero = 0;
	if (hipMemcpyFromSymbolAsync(&f, HIP_SYMBOL(g_chain_timeout), 4, 0, hipMemcpyDeviceToHost, st) != hipSuccess) return 0;
	if (gave_up && hipMemcpyAsync(&own, gave_up, 4, hipMemcpyDeviceToHost, st) != hipSuccess) own = 0;
	(void)hipStreamSynchronize(st);
	if (f) { (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_chain_timeout), &zero, 4, 0, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
	return f | own;
}
uint32_t chain_ring_floor(uint32_t v_begin) { return v_begin > kRing3Near ? v_begin - kRing3Near : 0u; }
bool unpredict3_covers(const ListDesc &ld)
{
	for (int c = 0; c < ld.ncomp; ++c) if (ld.stype[c] != 6 && ld.stype[c] != 8) return false;
	return ld.ncomp > 0;
}
void launch_slice_prepare(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t v_begin, uint32_t v_end, uint32_t *cand, uint8_t *ncand, void *crec)
{
	if (v_end <= v_begin) return;
	const uint32_t n = v_end - v_begin;
	const uint32_t per = ((n + 255) / 256 + 7) / 8;
	hipLaunchKernelGGL(k_candidates_ids, dim3(per * 8), dim3(256), 0, st, cv, order_v, v_begin, v_end, cand, ncand, per, nvtx, (ChainRec*)crec, chain_ring_floor(v_begin));
}
void launch_slice_chain(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t v_begin, uint32_t v_end, const uint32_t *cand, const uint8_t *ncand,
                        const void *crec, const uint8_t *planes, const ListDesc &ld, uint8_t *rec)
{
	if (v_end <= v_begin) return;
	auto go3 = [&](auto kern, int stype) {
		CompSel sel{};
		for (int c = 0; c < ld.ncomp; ++c) if (ld.stype[c] == stype) sel.comp[sel.n++] = c;
		if (!sel.n) return;
		hipLaunchKernelGGL(kern, dim3((sel.n - 1) * 8 + 1), dim3(64 * chain_waves(nvtx)), 0, st, cv, order_v, nvtx, cand, ncand, (const ChainRec*)crec, planes, ld, rec, sel, v_begin, v_end, chain_ring_floor(v_begin));
	};
	go3(k_unpredict3_range<uint16_t>, 6); go3(k_unpredict3_range<uint8_t>, 8);
}
// segs: pairs (begin, end) of decode ranks; list_off: n_lists + 1 offsets into segs.  Lists run in parallel blocks, the
// segments of one list one after the other.  Two launches: independent components first, then the dependent ones.
void launch_chain_records(hipStream_t st, const uint32_t *cand, const uint8_t *ncand, uint32_t nvtx, const uint32_t *seg_start, uint32_t nseg, void *crec)
{
	if (nvtx && nseg) hipLaunchKernelGGL(k_chain_records, dim3((nvtx + 255) / 256), dim3(256), 0, st, cand, ncand, nvtx, seg_start, nseg, (ChainRec*)crec);
}
bool unpredict3_wanted(const ListDesc &ld)
{
	for (int c = 0; c < ld.ncomp; ++c) if (ld.stype[c] == 6 || ld.stype[c] == 8) return true;
	return false;
}
// segs: triples (first decode rank, end, component of the mesh); list_off: n_lists + 1 offsets into segs.  Lists run in parallel
// workgroups, the segments of one list one after the other.  seg_start (every component's first decode rank + end) and done
// (ld.ncomp x nseg flags, zeroed) let a chain wait for the component that owns an older vertex it reads.
void launch_unpredict2(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t *cand, uint8_t *ncand, const void *crec,
                       const uint8_t *planes, const ListDesc &ld, uint8_t *rec, const uint32_t *segs, const uint32_t *list_off, uint32_t n_lists,
                       const uint32_t *seg_start, uint32_t nseg, uint32_t *done)
{
	if (!nvtx || !ld.ncomp || !n_lists) return;
	// (the callers' flag tables hold ld.ncomp x nseg progress words and one more behind them, zeroed with them: the give-up word)
	const CrossSync xs{ seg_start, nseg, done, nullptr, done ? done + (size_t)ld.ncomp * nseg : nullptr };
	auto go3 = [&](auto kern, int stype) {
		CompSel sel{};
		for (int c = 0; c < ld.ncomp; ++c) if (ld.stype[c] == stype) sel.comp[sel.n++] = c;
		if (!sel.n) return;
		hipLaunchKernelGGL(kern, dim3(((n_lists + 7) / 8) * 8 * (uint32_t)sel.n), dim3(64 * std::min(8u, chain_waves(nvtx))), 0, st,   /* (the kernel of many chains keeps its 512 threads: with 1 024 it would not fit its registers) */ cv, order_v, nvtx, (const uint32_t*)cand, (const uint8_t*)ncand, (const ChainRec*)crec, planes, ld, rec, sel,
		                   segs, list_off, xs, n_lists);
	};
	// the ring + the rows of a tile's many-candidate vertices (64 x 24 words): 38 KB, four chains per compute unit (round 2's input
	// queue made it 58 KB and two)
	// ... when there are more chains than that gives places for (256 compute units x 4), a ring of half the size lets seven share a
	// compute unit: a chain is a lone wavefront that issues an instruction every five or six cycles, two of them on a SIMD hardly
	// slow each other, and a source older than the ring is simply read from the records (HRY_CHAIN_RING_KB: 8, 16 or 32)
	static const uint32_t ring_kb_env = [] { const char *e = getenv("HRY_CHAIN_RING_KB"); const int v = e ? atoi(e) : 0; return v == 8 || v == 16 || v == 32 ? (uint32_t)v : 0u; }();
	const uint32_t ring_bytes = (ring_kb_env ? ring_kb_env : (uint64_t)n_lists * (uint32_t)ld.ncomp > 1024u ? 16u : 32u) * 1024u, lds_bytes = ring_bytes + 64 * kCandMax * 3 * 4 + 64;   // ring, rows, remembered owners
	auto go = [&](auto kern, int stype) {
		CompSel sel{};
		for (int c = 0; c < ld.ncomp; ++c) if (ld.stype[c] == stype) sel.comp[sel.n++] = c;
		if (!sel.n) return;
		(void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		hipLaunchKernelGGL(kern, dim3(((n_lists + 7) / 8) * 8 * (uint32_t)sel.n), dim3(64), lds_bytes, st, cv, order_v, nvtx, (const uint32_t*)cand, (const uint8_t*)ncand, planes, ld, rec, ring_bytes, sel,
		                   segs, list_off, xs, n_lists);
	};
	// components of different types are independent chains too: their kernels may overlap on the device
	go(k_unpredict2<float>, 0); go(k_unpredict2<uint32_t>, 4); go(k_unpredict2<int32_t>, 5);
	go(k_unpredict2<int16_t>, 7); go(k_unpredict2<int8_t>, 9);
	if (crec) { go3(k_unpredict3<uint16_t>, 6); go3(k_unpredict3<uint8_t>, 8); }
	else { go(k_unpredict2<uint16_t>, 6); go(k_unpredict2<uint8_t>, 8); }
}
}   // namespace dev
}   // namespace hry
