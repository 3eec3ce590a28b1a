// HIP kernels of the .hry hot path for gfx950 (wave64).  All integer / byte work bounded by HBM traffic or by
// serial dependencies; no MFMA.  See DESIGN.md for the data layout and the roofline of each kernel.
//
// Reference behaviour restated by each kernel is cited at its definition.
#include <hip/hip_runtime.h>

#include "codec_math.hpp"
#include "dev_types.hpp"
#include "fan.hpp"
#include "kernels.hpp"

namespace hry {
namespace dev {

// ---------------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------------
__constant__ int c_type_size[11] = { 4, 8, 8, 8, 4, 4, 2, 2, 1, 1, 0 };

// prediction of one component (attrcode.h:182-208): mean of candidate predictions in fan order (double / int64),
// integers take the mean, floats the candidate nearest to the mean (strict <, first wins).
template <typename T>
__device__ __forceinline__ T predict_component(const Topo &tp, const uint32_t *rank, const uint8_t *rec, int stride, int off, int q,
                                               uint32_t e, uint32_t my_rank, uint32_t lo, const uint32_t *attr_of)
{
	typedef typename cm::wide<T>::type W;
	W acc = 0;
	uint32_t n = 0;
	auto value = [&](uint32_t v) { return ldg<T>(rec + (size_t)(attr_of ? attr_of[v] : v) * stride + off); };
	fan_candidates(tp, rank, e, my_rank, lo, [&](uint32_t v0, uint32_t v1, uint32_t vo) {
		acc = acc + (W)cm::parallelogram<T>(value(v0), value(v1), value(vo), q);
		++n;
	});
	if (n == 0) return T(0);
	T avg = (T)cm::mean_of(acc, (W)n);
	if constexpr (!cm::is_fp<T>::value) return avg;
	else {
		T best = 3.402823466e+38f;   // numeric_limits<float>::max()
		fan_candidates(tp, rank, e, my_rank, lo, [&](uint32_t v0, uint32_t v1, uint32_t vo) {
			T p = cm::parallelogram<T>(value(v0), value(v1), value(vo), q);
			T db = avg > best ? avg - best : best - avg;
			T dp = avg > p ? avg - p : p - avg;
			best = db < dp ? best : p;
		});
		return best;
	}
}

// ---------------------------------------------------------------------------------------------------------
// k_bounds: per-component min / max with the reference's initial values and first-wins ties
// (structs/quant.h:30-38: min starts at numeric_limits::max(), max at numeric_limits::min() == FLT_MIN for floats)
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct Lim;
template <> struct Lim<float> { static __device__ float hi() { return 3.402823466e+38f; } static __device__ float lo() { return 1.175494351e-38f; } };
template <> struct Lim<double> { static __device__ double hi() { return 1.7976931348623157e+308; } static __device__ double lo() { return 2.2250738585072014e-308; } };
template <> struct Lim<uint64_t> { static __device__ uint64_t hi() { return ~0ull; } static __device__ uint64_t lo() { return 0; } };
template <> struct Lim<int64_t> { static __device__ int64_t hi() { return 0x7fffffffffffffffll; } static __device__ int64_t lo() { return -0x7fffffffffffffffll - 1; } };
template <> struct Lim<uint32_t> { static __device__ uint32_t hi() { return ~0u; } static __device__ uint32_t lo() { return 0; } };
template <> struct Lim<int32_t> { static __device__ int32_t hi() { return 0x7fffffff; } static __device__ int32_t lo() { return -0x7fffffff - 1; } };
template <> struct Lim<uint16_t> { static __device__ uint16_t hi() { return 0xffff; } static __device__ uint16_t lo() { return 0; } };
template <> struct Lim<int16_t> { static __device__ int16_t hi() { return 0x7fff; } static __device__ int16_t lo() { return -0x8000; } };
template <> struct Lim<uint8_t> { static __device__ uint8_t hi() { return 0xff; } static __device__ uint8_t lo() { return 0; } };
template <> struct Lim<int8_t> { static __device__ int8_t hi() { return 0x7f; } static __device__ int8_t lo() { return -0x80; } };

// (value, first index) pairs make the parallel reduction reproduce the sequential scan bit for bit (+-0.0 ties)
template <typename T> struct Ext { T v; uint32_t i; };
template <typename T> __device__ __forceinline__ Ext<T> pick_min(Ext<T> a, Ext<T> b)
{
	if (b.v < a.v) return b;
	if (a.v < b.v) return a;
	return a.i <= b.i ? a : b;
}
template <typename T> __device__ __forceinline__ Ext<T> pick_max(Ext<T> a, Ext<T> b)
{
	if (b.v > a.v) return b;
	if (a.v > b.v) return a;
	return a.i <= b.i ? a : b;
}

// 64-bit pattern of a value and back (wave shuffles move 32-bit words)
template <typename T> __device__ __forceinline__ uint64_t to_bits(T v) { uint64_t b = 0; __builtin_memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> __device__ __forceinline__ T from_bits(uint64_t b) { T v; __builtin_memcpy(&v, &b, sizeof(T)); return v; }
template <typename T> __device__ __forceinline__ Ext<T> shfl_xor_ext(Ext<T> a, int mask)
{
	uint64_t b = to_bits(a.v);
	uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, mask), hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), mask);
	return Ext<T>{ from_bits<T>(((uint64_t)hi << 32) | lo), (uint32_t)__shfl_xor((int)a.i, mask) };
}
template <typename T> __device__ __forceinline__ void wave_reduce_ext(Ext<T> &mn, Ext<T> &mx)
{
	for (int m = 32; m > 0; m >>= 1) { mn = pick_min(mn, shfl_xor_ext(mn, m)); mx = pick_max(mx, shfl_xor_ext(mx, m)); }
}

// grid (nparts, ncomp): block (p, c) scans its grid-stride share of component c; wavefronts reduce by shuffles, the four
// wavefronts of a block through LDS
template <typename T>
__device__ void bounds_component(const uint8_t *rec, uint32_t count, int stride, int off, uint8_t *part_min, uint8_t *part_max, uint32_t *part_idx)
{
	// index 0 is reserved for the initial value so that it wins ties against every element
	Ext<T> mn{ Lim<T>::hi(), 0 }, mx{ Lim<T>::lo(), 0 };
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
		T e = ldg<T>(rec + (size_t)i * stride + off);
		if (e == e) {   // NaN never replaces the running value in the reference's comparisons
			mn = pick_min(mn, Ext<T>{ e, i + 1 });
			mx = pick_max(mx, Ext<T>{ e, i + 1 });
		}
	}
	wave_reduce_ext(mn, mx);
	__shared__ uint64_t sv[2][4];
	__shared__ uint32_t si[2][4];
	const int wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) { sv[0][wave] = to_bits(mn.v); si[0][wave] = mn.i; sv[1][wave] = to_bits(mx.v); si[1][wave] = mx.i; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 4; ++w) {
			mn = pick_min(mn, Ext<T>{ from_bits<T>(sv[0][w]), si[0][w] });
			mx = pick_max(mx, Ext<T>{ from_bits<T>(sv[1][w]), si[1][w] });
		}
		const size_t p = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
		stg<uint64_t>(part_min + p * 8, to_bits(mn.v));
		stg<uint64_t>(part_max + p * 8, to_bits(mx.v));
		part_idx[2 * p] = mn.i;
		part_idx[2 * p + 1] = mx.i;
	}
}
// every component type, doubles included (they can be quantised, quant.h:137-139; only their lossless residuals are undefined)
template <typename F> __device__ __forceinline__ void with_any_type(int t, F &&f)
{
	if (t == 1) f(double()); else with_stype(t, f);
}
__global__ __launch_bounds__(256) void k_bounds_partial(const uint8_t *rec, uint32_t count, BoundsPlan plan,
                                                        uint8_t *part_min, uint8_t *part_max, uint32_t *part_idx)
{
	const int c = blockIdx.y;
	with_any_type(plan.type[c], [&](auto tag) { bounds_component<decltype(tag)>(rec, count, plan.stride, plan.off[c], part_min, part_max, part_idx); });
}
// one wavefront per component folds its partials: lanes stride over them, then a shuffle reduction.
// out: per component { u64 min bits, u64 max bits, u32 first index of the min + 1, u32 of the max + 1 } (24 bytes)
template <typename T>
__device__ void bounds_final(const uint8_t *part_min, const uint8_t *part_max, const uint32_t *part_idx, int nparts, uint8_t *out)
{
	Ext<T> mn{ Lim<T>::hi(), 0 }, mx{ Lim<T>::lo(), 0 };
	const size_t base = (size_t)blockIdx.x * nparts;
	for (int p = threadIdx.x; p < nparts; p += 64) {
		mn = pick_min(mn, Ext<T>{ from_bits<T>(ldg<uint64_t>(part_min + (base + p) * 8)), part_idx[2 * (base + p)] });
		mx = pick_max(mx, Ext<T>{ from_bits<T>(ldg<uint64_t>(part_max + (base + p) * 8)), part_idx[2 * (base + p) + 1] });
	}
	wave_reduce_ext(mn, mx);
	if (threadIdx.x == 0) {
		uint8_t *o = out + (size_t)blockIdx.x * 24;
		stg<uint64_t>(o, to_bits(mn.v)); stg<uint64_t>(o + 8, to_bits(mx.v));
		stg<uint32_t>(o + 16, mn.i); stg<uint32_t>(o + 20, mx.i);
	}
}
__global__ __launch_bounds__(64) void k_bounds_final(const uint8_t *part_min, const uint8_t *part_max, const uint32_t *part_idx, int nparts, BoundsPlan plan, uint8_t *out)
{
	with_any_type(plan.type[blockIdx.x], [&](auto tag) { bounds_final<decltype(tag)>(part_min, part_max, part_idx, nparts, out); });
}

// ---------------------------------------------------------------------------------------------------------
// k_requant: in-place requantisation of selected components (structs/quant.h:114-178).  Sources: unquantised
// float / 1-2-4 byte integers, or an already quantised value (q -> q').  src and dst alias the same slot
// (attr.h:95-98); the destination is written in its storage type into the low bytes of the slot.
// ---------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T rescale_int(T val, T from, T to) { return val / from * to + val % from * to / from; }   // quant.h:103-107

template <typename T> __device__ __forceinline__ T rescale_fp(T val, T from, T to) { return val / from * to; }   // quant.h:98-102 (every operation rounded on its own)

__global__ __launch_bounds__(256) void k_requant(uint8_t *rec, uint32_t count, int stride, RequantPlan plan)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
		uint8_t *r = rec + (size_t)i * stride;
		for (int k = 0; k < plan.n; ++k) {
			const RequantComp &c = plan.c[k];
			uint8_t *slot = r + c.off;
			uint64_t q = 0;
			const int lv = (1 << (uint32_t)c.dst_bits) - 1;   // levels of the destination; evaluated in int like the reference (quant.h:135)
			if (c.src_bits) {   // already quantised: read the storage type (quant.h:121-129)
				switch (c.src_type) {
				case 8: q = ldg<uint8_t>(slot); break;
				case 6: q = ldg<uint16_t>(slot); break;
				case 4: q = ldg<uint32_t>(slot); break;
				default: q = ldg<uint64_t>(slot); break;
				}
				if (c.dst_bits) q = rescale_int<uint64_t>(q, (uint64_t)((1 << (uint32_t)c.src_bits) - 1), (uint64_t)lv);   // q -> q' (quant.h:169-171)
			} else {
				switch (c.src_type) {   // quant.h:131-166
				case 0: q = cm::quantise_f32(ldg<float>(slot), cm::bits<float>((uint32_t)c.mn), cm::bits<float>((uint32_t)c.scale), c.dst_bits); break;
				case 1: q = (uint64_t)(rescale_fp<double>(ldg<double>(slot) - cm::bits<double>(c.mn), cm::bits<double>(c.scale), (double)lv) + 0.5); break;
				case 2: q = rescale_int<uint64_t>(ldg<uint64_t>(slot) - c.mn, c.scale, (uint64_t)lv); break;
				case 3: q = (uint64_t)rescale_int<int64_t>(ldg<int64_t>(slot) - (int64_t)c.mn, (int64_t)c.scale, (int64_t)lv); break;
				case 4: q = rescale_int<uint32_t>(ldg<uint32_t>(slot) - (uint32_t)c.mn, (uint32_t)c.scale, (uint32_t)lv); break;
				case 5: q = (uint64_t)rescale_int<int32_t>(ldg<int32_t>(slot) - (int32_t)c.mn, (int32_t)c.scale, (int32_t)lv); break;
				case 6: q = rescale_int<uint16_t>((uint16_t)(ldg<uint16_t>(slot) - (uint16_t)c.mn), (uint16_t)c.scale, (uint16_t)lv); break;
				case 7: q = (uint64_t)rescale_int<int16_t>((int16_t)(ldg<int16_t>(slot) - (int16_t)c.mn), (int16_t)c.scale, (int16_t)lv); break;
				case 8: q = rescale_int<uint8_t>((uint8_t)(ldg<uint8_t>(slot) - (uint8_t)c.mn), (uint8_t)c.scale, (uint8_t)lv); break;
				case 9: q = (uint64_t)rescale_int<int8_t>((int8_t)(ldg<int8_t>(slot) - (int8_t)c.mn), (int8_t)c.scale, (int8_t)lv); break;
				default: break;
				}
			}
			if (c.dst_bits) {
				if (c.dst_bits <= 8) stg<uint8_t>(slot, (uint8_t)q);
				else if (c.dst_bits <= 16) stg<uint16_t>(slot, (uint16_t)q);
				else stg<uint32_t>(slot, (uint32_t)q);
				continue;
			}
			// dequantisation into the original type (quant.h:180-212): rescale(q, 2^bits - 1, extent) + min, in that type
			const int sl = (1 << (uint32_t)c.src_bits) - 1;
			switch (c.dst_type) {
			case 0: stg<float>(slot, rescale_fp<float>((float)q, (float)sl, cm::bits<float>((uint32_t)c.scale)) + cm::bits<float>((uint32_t)c.mn)); break;
			case 1: stg<double>(slot, rescale_fp<double>((double)q, (double)sl, cm::bits<double>(c.scale)) + cm::bits<double>(c.mn)); break;
			case 2: stg<uint64_t>(slot, rescale_int<uint64_t>(q, (uint64_t)sl, c.scale) + c.mn); break;
			case 3: stg<int64_t>(slot, rescale_int<int64_t>((int64_t)q, (int64_t)sl, (int64_t)c.scale) + (int64_t)c.mn); break;
			case 4: stg<uint32_t>(slot, rescale_int<uint32_t>((uint32_t)q, (uint32_t)sl, (uint32_t)c.scale) + (uint32_t)c.mn); break;
			case 5: stg<int32_t>(slot, rescale_int<int32_t>((int32_t)q, (int32_t)sl, (int32_t)c.scale) + (int32_t)c.mn); break;
			case 6: stg<uint16_t>(slot, (uint16_t)(rescale_int<uint16_t>((uint16_t)q, (uint16_t)sl, (uint16_t)c.scale) + (uint16_t)c.mn)); break;
			case 7: stg<int16_t>(slot, (int16_t)(rescale_int<int16_t>((int16_t)q, (int16_t)sl, (int16_t)c.scale) + (int16_t)c.mn)); break;
			case 8: stg<uint8_t>(slot, (uint8_t)(rescale_int<uint8_t>((uint8_t)q, (uint8_t)sl, (uint8_t)c.scale) + (uint8_t)c.mn)); break;
			case 9: stg<int8_t>(slot, (int8_t)(rescale_int<int8_t>((int8_t)q, (int8_t)sl, (int8_t)c.scale) + (int8_t)c.mn)); break;
			default: break;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// traversal rank: rank[v] = position of v in the coding order, kNoRank for vertices never coded
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank(const uint32_t *order_v, uint32_t n, const uint32_t *org, uint32_t *rank)
{
	uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < n) rank[org[order_v[k]]] = k;
}

// ---- the same kernels over RUNS of the coding order (chunked.cpp: EncodePipeline).  The groups of a multi-component mesh finish
// their walks one by one, in no particular order; what a finished group has coded is a handful of runs [first, first + n) of the
// coded vertices / faces, final from then on.  A batch of such runs is one launch: position p of the batch lies in run r =
// the last one with start[r] <= p (start: exclusive scan of the runs' lengths, start[nruns] = the batch's size).
struct RunTable { const uint32_t *start, *first; uint32_t nruns, total; };
__device__ __forceinline__ uint32_t run_lookup(const RunTable &rt, uint32_t p)
{
	uint32_t lo = 0, hi = rt.nruns;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rt.start[mid] <= p) lo = mid; else hi = mid; }
	return rt.first[lo] + (p - rt.start[lo]);
}
// packed: the runs' entries of order_v back to back, as they came up (position p of the batch); they take their places in order_v here
// (packed == nullptr: the runs' entries are in their places already, copied there run by run)
__global__ __launch_bounds__(256) void k_rank_runs(RunTable rt, const uint32_t *packed, uint32_t *order_v, const uint32_t *org, uint32_t *rank)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= rt.total) return;
	const uint32_t k = run_lookup(rt, p);
	uint32_t e;
	if (packed) { e = packed[p]; order_v[k] = e; } else e = order_v[k];
	rank[org[e]] = k;
}

// ---------------------------------------------------------------------------------------------------------
// k_predict_vtx: prediction + residual folding + byte symbolisation of vertex attributes, one thread per coded
// vertex (attrcode.h:209-225 vtx, :321-344 vtx_post; io.h:90-94; models.h:168-173).  Output: SoA byte planes,
// plane p holds byte p of every vertex' residual record in coding order (coalesced for the model kernels).
// ---------------------------------------------------------------------------------------------------------
// The fan is walked ONCE per vertex: the (at most kEncCand) candidate triples that pass the rank filter are kept as vertex
// ids in LDS and every component is evaluated from them -- the mean, and for floats the nearest-to-the-mean selection, which
// the reference evaluates in a second sweep over the same candidates (attrcode.h:182-208).  A fan with more candidates
// (high-valence vertices) takes predict_component, which walks again per component.
// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2: virtual block (b % 8) * per + b / 8 gives every
// XCD one contiguous range of the coding order, so the connectivity / rank / record lines its wavefronts gather are shared
// inside one L2 instead of being fetched by all eight.
constexpr int kEncCand = 8;
// coded vertex k of n (n = the stride of the byte planes)
__device__ __forceinline__ void predict_vertex(const ConnView &cv, const uint32_t *order_v, uint32_t k, uint32_t n, const uint32_t *rank,
                                               const uint8_t *rec, const ListDesc &ld, uint8_t *planes, uint32_t (*s_cand)[256])
{
	Topo tp{ cv };
	const uint32_t e = order_v[k];
	const uint32_t v = cv.org[e];
	int nc = 0;
	fan_candidates(tp, rank, e, k, 0, [&](uint32_t v0, uint32_t v1, uint32_t vo) {
		if (nc < kEncCand) { s_cand[3 * nc][threadIdx.x] = v0; s_cand[3 * nc + 1][threadIdx.x] = v1; s_cand[3 * nc + 2][threadIdx.x] = vo; }
		++nc;
	});
	for (int c = 0; c < ld.ncomp; ++c) {
		with_stype(ld.stype[c], [&](auto tag) {
			typedef decltype(tag) T;
			typedef typename cm::wide<T>::type W;
			const int off = ld.off[c], q = ld.quant[c];
			auto value = [&](uint32_t x) { return ldg<T>(rec + (size_t)x * ld.stride + off); };
			T pred;
			if (nc > kEncCand) pred = predict_component<T>(tp, rank, rec, ld.stride, off, q, e, k, 0, nullptr);
			else if (nc == 0) pred = T(0);
			else {
				T p[kEncCand];
				W acc = 0;
#pragma unroll
				for (int i = 0; i < kEncCand; ++i)
					if (i < nc) {
						p[i] = cm::parallelogram<T>(value(s_cand[3 * i][threadIdx.x]), value(s_cand[3 * i + 1][threadIdx.x]), value(s_cand[3 * i + 2][threadIdx.x]), q);
						acc = acc + (W)p[i];   // fan order (the float mean is a double sum in candidate order)
					}
				const T avg = (T)cm::mean_of(acc, (W)nc);
				if constexpr (!cm::is_fp<T>::value) pred = avg;
				else {
					T best = 3.402823466e+38f;   // numeric_limits<float>::max()
#pragma unroll
					for (int i = 0; i < kEncCand; ++i)
						if (i < nc) {
							T db = avg > best ? avg - best : best - avg;
							T dp = avg > p[i] ? avg - p[i] : p[i] - avg;
							best = db < dp ? best : p[i];
						}
					pred = best;
				}
			}
			T raw = value(v);
			auto code = cm::residual_bits<T>(raw, pred, q);
			for (int b = 0; b < (int)sizeof(T); ++b) planes[(size_t)(ld.plane[c] + b) * n + k] = (uint8_t)(code >> (8 * b));
		});
	}
}
__global__ __launch_bounds__(256) void k_predict_vtx(ConnView cv, const uint32_t *order_v, uint32_t n, const uint32_t *rank,
                                                     const uint8_t *rec, ListDesc ld, uint8_t *planes, uint32_t blocks_per_xcd)
{
	const uint32_t vb = (blockIdx.x & 7u) * blocks_per_xcd + (blockIdx.x >> 3);
	const uint32_t k = vb * blockDim.x + threadIdx.x;
	__shared__ uint32_t s_cand[kEncCand * 3][256];
	if (k >= n) return;
	predict_vertex(cv, order_v, k, n, rank, rec, ld, planes, s_cand);
}
// a batch of runs of the coding order (RunTable above); n: all coded vertices of the mesh = the planes' stride
__global__ __launch_bounds__(256) void k_predict_vtx_runs(ConnView cv, RunTable rt, const uint32_t *order_v, uint32_t n, const uint32_t *rank,
                                                          const uint8_t *rec, ListDesc ld, uint8_t *planes, uint32_t blocks_per_xcd)
{
	const uint32_t vb = (blockIdx.x & 7u) * blocks_per_xcd + (blockIdx.x >> 3);
	const uint32_t p = vb * blockDim.x + threadIdx.x;
	__shared__ uint32_t s_cand[kEncCand * 3][256];
	if (p >= rt.total) return;
	predict_vertex(cv, order_v, run_lookup(rt, p), n, rank, rec, ld, planes, s_cand);
}

// faces: prediction is always "no candidate" (attrcode.h:227-254) => residual against 0
__device__ __forceinline__ void face_planes_of(const ConnView &cv, const uint32_t *order_f, uint32_t j, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes);
__global__ __launch_bounds__(256) void k_face_planes(ConnView cv, const uint32_t *order_f, uint32_t n, const uint8_t *rec, ListDesc ld, uint8_t *planes)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	face_planes_of(cv, order_f, j, n, rec, ld, planes);
}
// packed: the runs' entries of order_f back to back (position p of the batch)
__global__ __launch_bounds__(256) void k_face_planes_runs(ConnView cv, RunTable rt, const uint32_t *packed, uint32_t *order_f, uint32_t n, const uint8_t *rec, ListDesc ld, uint8_t *planes)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= rt.total) return;
	const uint32_t j = run_lookup(rt, p);
	if (packed) order_f[j] = packed[p];
	face_planes_of(cv, order_f, j, n, rec, ld, planes);
}
// one 32-bit value per symbol -> byte planes, over runs (the polygons' triangle counts, one per coded face; n = all of them);
// packed: the runs' values back to back
// (packed == nullptr: the values are in their places in val)
__global__ __launch_bounds__(256) void k_split_bytes_runs(RunTable rt, const uint32_t *packed, const uint32_t *val, uint32_t n, int nbytes, uint8_t *planes)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= rt.total) return;
	const uint32_t j = run_lookup(rt, p), v = packed ? packed[p] : val[j];
	for (int b = 0; b < nbytes; ++b) planes[(size_t)b * n + j] = (uint8_t)(v >> (8 * b));
}
__device__ __forceinline__ void face_planes_of(const ConnView &cv, const uint32_t *order_f, uint32_t j, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes)
{
	Topo tp{ cv };
	uint32_t f = tp.face(order_f[j]);
	for (int c = 0; c < ld.ncomp; ++c) {
		with_stype(ld.stype[c], [&](auto tag) {
			typedef decltype(tag) T;
			T raw = ldg<T>(rec + (size_t)f * ld.stride + ld.off[c]);
			auto code = cm::residual_bits<T>(raw, T(0), ld.quant[c]);
			for (int b = 0; b < (int)sizeof(T); ++b) planes[(size_t)(ld.plane[c] + b) * n + j] = (uint8_t)(code >> (8 * b));
		});
	}
}

// ---------------------------------------------------------------------------------------------------------
// reciprocal table: magic[t] for every context total that can occur (t = 2 .. n-1)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_magic_table(MagicEnt *tab, uint32_t from, uint32_t to)
{
	uint32_t t = from + blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= to) return;
	MagicEnt m{ 0, 0, 0 };
	if (t >= 2) cm::make_magic(t, m.magic, m.shift);
	uint32_t sh32 = 0;
	cm::make_magic32(t, m.m32, sh32);
	m.shift |= sh32 << kMagicSh32Shift;
	tab[t] = m;
}

__device__ __forceinline__ void emit_symbol(SymRec *rec, uint32_t *sym_l, const MagicEnt *magic, uint32_t g, uint32_t l, uint32_t c, uint32_t t)
{
	bool sub = l + c == t;
	bool noop = sub && l == 0;
	MagicEnt m = magic[t];
	SymRec r;
	r.magic = m.magic;
	r.x = sub ? l : c;
	r.meta = (m.shift & 63u) | (sub ? kMetaSub : 0u) | (noop ? kMetaNoop : 0u);
	rec[g] = r;
	sym_l[g] = l;
}

// connectivity groups arrive as one 32-bit value per symbol; split them into byte planes
__global__ __launch_bounds__(256) void k_split_bytes(const uint32_t *val, uint32_t n, int nbytes, uint8_t *planes)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	uint32_t v = val[j];
	for (int b = 0; b < nbytes; ++b) planes[(size_t)b * n + j] = (uint8_t)(v >> (8 * b));
}

// cut-border operations: the order-conditioned model was evaluated by the walk (models.h:91-119)
__global__ __launch_bounds__(256) void k_op_records(const uint32_t *l, const uint32_t *h, const uint32_t *t, const uint32_t *pos, uint32_t n,
                                                    const MagicEnt *magic, SymRec *rec, uint32_t *sym_l)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) emit_symbol(rec, sym_l, magic, pos[j], l[j], h[j] - l[j], t[j]);
}

// The same model evaluated HERE, by counting (models.h:49-120): the state before operation i is a set of prefix counts over the
// operation stream -- plain[s] = 1 + #(symbol s before i) for the five rare symbols, c_new[k] / c_fwd[k] = 1 + #(NEWVTX / CONNFWD of
// order class k before i), c_all = 2 + #(NEWVTX or CONNFWD before i) -- and the mixed frequency of NEWVTX is c_new[k] c_all /
// (c_new[k] + c_fwd[k]) in 64-bit integers.  Counters: 0..4 plain, 5..12 new, 13..20 fwd.  One wavefront per kOpChunk operations;
// inside a batch of 64 the counts before a lane come from ballots (per symbol, and three for "same order class").
constexpr uint32_t kOpChunk = 4096, kOpCounters = 21;
__device__ __forceinline__ uint32_t op_counter(uint32_t s, uint32_t k) { return s < 5u ? s : s == 5u ? 5u + k : 13u + k; }
__global__ __launch_bounds__(256) void k_opmodel_hist(const uint8_t *op, uint32_t n, uint32_t *hist)
{
	__shared__ uint32_t c[kOpCounters];
	if (threadIdx.x < kOpCounters) c[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t b = blockIdx.x * kOpChunk, e = min(n, b + kOpChunk);
	for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) { const uint32_t x = op[i]; atomicAdd(&c[op_counter(x & 7u, x >> 3)], 1u); }
	__syncthreads();
	if (threadIdx.x < kOpCounters) hist[(size_t)blockIdx.x * kOpCounters + threadIdx.x] = c[threadIdx.x];
}
// exclusive scan over the chunks, in place; one workgroup: thread t owns a contiguous range of chunks
__global__ __launch_bounds__(256) void k_opmodel_scan(uint32_t *hist, uint32_t nchunks)
{
	__shared__ uint32_t part[256][kOpCounters];
	const uint32_t per = (nchunks + 255u) / 256u, b = min(nchunks, threadIdx.x * per), e = min(nchunks, b + per);
	uint32_t sum[kOpCounters];
	for (uint32_t c = 0; c < kOpCounters; ++c) sum[c] = 0;
	for (uint32_t k = b; k < e; ++k) for (uint32_t c = 0; c < kOpCounters; ++c) sum[c] += hist[(size_t)k * kOpCounters + c];
	for (uint32_t c = 0; c < kOpCounters; ++c) part[threadIdx.x][c] = sum[c];
	__syncthreads();
	if (threadIdx.x < kOpCounters) { uint32_t run = 0; for (uint32_t t = 0; t < 256u; ++t) { const uint32_t v = part[t][threadIdx.x]; part[t][threadIdx.x] = run; run += v; } }
	__syncthreads();
	for (uint32_t c = 0; c < kOpCounters; ++c) sum[c] = part[threadIdx.x][c];
	for (uint32_t k = b; k < e; ++k)
		for (uint32_t c = 0; c < kOpCounters; ++c) { uint32_t *h = hist + (size_t)k * kOpCounters + c; const uint32_t v = *h; *h = sum[c]; sum[c] += v; }
}
// thr / cum: op_position_table (host.hpp) -- the connectivity groups between the operations
__global__ __launch_bounds__(64) void k_opmodel_records(const uint8_t *op, uint32_t n, const uint32_t *hist, const uint32_t *thr, const uint32_t *cum, uint32_t ngroups,
                                                         const MagicEnt *magic, SymRec *rec, uint32_t *sym_l)
{
	__shared__ uint32_t cnt[kOpCounters];
	const uint32_t lane = threadIdx.x;
	if (lane < kOpCounters) cnt[lane] = hist[(size_t)blockIdx.x * kOpCounters + lane];
	__syncthreads();
	uint32_t all = 0;   // NEWVTX + CONNFWD before the batch
	for (uint32_t c = 5; c < kOpCounters; ++c) all += cnt[c];
	const uint64_t earlier = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	const uint32_t b = blockIdx.x * kOpChunk, e = min(n, b + kOpChunk);
	for (uint32_t base = b; base < e; base += 64) {
		const uint32_t i = base + lane;
		const bool valid = i < e;
		const uint32_t x = valid ? op[i] : 0xffu, s = x & 7u, k = (x >> 3) & 7u;
		uint64_t ms[7];
#pragma unroll
		for (uint32_t y = 0; y < 7; ++y) ms[y] = __ballot(valid && s == y);
		uint64_t same = ~0ull;   // lanes of my order class
#pragma unroll
		for (int bit = 0; bit < 3; ++bit) { const bool mine = (k >> bit) & 1u; const uint64_t m = __ballot(mine); same &= mine ? m : ~m; }
		if (valid) {
			uint64_t f[7];
#pragma unroll
			for (uint32_t y = 0; y < 5; ++y) f[y] = 1u + cnt[y] + (uint32_t)__popcll(ms[y] & earlier);
			const uint64_t c_new = 1u + cnt[5u + k] + (uint32_t)__popcll(ms[5] & same & earlier), c_fwd = 1u + cnt[13u + k] + (uint32_t)__popcll(ms[6] & same & earlier);
			const uint64_t c_all = 2u + (uint64_t)all + (uint32_t)__popcll((ms[5] | ms[6]) & earlier);
			const uint64_t nv = c_new * c_all / (c_new + c_fwd);
			f[5] = nv; f[6] = c_all - nv;
			uint64_t l = 0;
#pragma unroll
			for (uint32_t y = 0; y < 7; ++y) l += y < s ? f[y] : 0ull;
			uint64_t fs = 0;
#pragma unroll
			for (uint32_t y = 0; y < 7; ++y) fs = y == s ? f[y] : fs;
			const uint64_t t = f[0] + f[1] + f[2] + f[3] + f[4] + c_all;
			// position in the symbol sequence: behind the groups that come before this operation
			const uint32_t oi = i;
			uint32_t lo = 0, hi = ngroups;   // first group with thr > oi
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (thr[mid] <= oi) lo = mid + 1; else hi = mid; }
			const uint32_t pos = oi + (lo ? cum[lo - 1] : 0u);
			emit_symbol(rec, sym_l, magic, pos, (uint32_t)l, (uint32_t)fs, (uint32_t)t);
		}
		__syncthreads();   // (one wavefront) the batch has read the counters
		if (valid) atomicAdd(&cnt[op_counter(s, k)], 1u);
		all += (uint32_t)__popcll(ms[5] | ms[6]);
		__syncthreads();
	}
}

// attr_type symbols of a list whose elements all carry private data: the j-th symbol is DATA with counts
// {DATA: 1 + j, HIST: 1} (models.h:201-203) => l = 0, h = 1 + j, t = 2 + j
__global__ __launch_bounds__(256) void k_type_records(uint32_t n, uint32_t pos_base, uint32_t pos_stride, const MagicEnt *magic, SymRec *rec, uint32_t *sym_l)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) emit_symbol(rec, sym_l, magic, pos_base + j * pos_stride, 0, 1 + j, 2 + j);
}

// ---------------------------------------------------------------------------------------------------------
// exact adaptive-model evaluation by counting (arith/stat_adaptive.h:46-54,77-82):
//   k_model_hist  : 256-bin histogram of every chunk
//   k_model_scan  : exclusive scan over the chunks of one plane (+ initial counts) -> table at each chunk start
//   k_model_lht   : one wavefront owns a chunk, keeps its count / cumulative tables in LDS and produces (l, h-l, t)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_model_hist(const PlaneJob *jobs, const ChunkRef *chunks, uint32_t *hist)
{
	ChunkRef cr = chunks[blockIdx.x];
	const PlaneJob &jb = jobs[cr.job];
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	uint32_t end = min(cr.first + (uint32_t)kChunk, jb.n);
	for (uint32_t j = cr.first + threadIdx.x; j < end; j += 256) atomicAdd(&h[jb.sym[j]], 1u);
	__syncthreads();
	hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_model_scan(const PlaneJob *jobs, uint32_t *hist)
{
	const PlaneJob &jb = jobs[blockIdx.x];
	uint32_t nch = (jb.n + kChunk - 1) / kChunk;
	uint32_t run = jb.init[threadIdx.x];
	uint32_t *h = hist + (size_t)jb.chunk0 * 256 + threadIdx.x;
	for (uint32_t c = 0; c < nch; ++c) {
		uint32_t v = h[(size_t)c * 256];
		h[(size_t)c * 256] = run;
		run += v;
	}
}

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t o = __shfl_up(inc, d, 64);
		if ((int)(threadIdx.x & 63) >= d) inc += o;
	}
	total = __shfl(inc, 63, 64);
	return inc - v;
}

__global__ __launch_bounds__(64) void k_model_lht(const PlaneJob *jobs, const ChunkRef *chunks, const uint32_t *hist, const MagicEnt *magic,
                                                  SymRec *rec, uint32_t *sym_l)
{
	ChunkRef cr = chunks[blockIdx.x];
	const PlaneJob jb = jobs[cr.job];
	const int lane = threadIdx.x;
	__shared__ uint32_t cnt[256], cum[256], bh[256];
	{   // table at chunk start; cum[s] = sum of counts of symbols < s
		const uint32_t *st = hist + (size_t)blockIdx.x * 256 + 4 * lane;
		uint32_t a = st[0], b = st[1], c = st[2], d = st[3], tot;
		uint32_t ex = wave_excl_scan(a + b + c + d, tot);
		cnt[4 * lane] = a; cnt[4 * lane + 1] = b; cnt[4 * lane + 2] = c; cnt[4 * lane + 3] = d;
		cum[4 * lane] = ex; cum[4 * lane + 1] = ex + a; cum[4 * lane + 2] = ex + a + b; cum[4 * lane + 3] = ex + a + b + c;
	}
	__syncthreads();
	const uint32_t end = min(cr.first + (uint32_t)kChunk, jb.n);
	for (uint32_t base = cr.first; base < end; base += 64) {
		uint32_t j = base + lane;
		bool valid = j < end;
		uint32_t s = valid ? jb.sym[j] : 0x100u;
		uint32_t l = valid ? cum[s] : 0, c = valid ? cnt[s] : 0;
		bh[4 * lane] = 0; bh[4 * lane + 1] = 0; bh[4 * lane + 2] = 0; bh[4 * lane + 3] = 0;
		// symbols of this batch that precede lane: smaller ones raise l, equal ones raise the count
		uint32_t nb = min(64u, end - base);
		for (uint32_t i = 0; i < nb; ++i) {
			uint32_t si = (uint32_t)__builtin_amdgcn_readlane(s, i);
			if ((int)i < lane) { l += si < s ? 1u : 0u; c += si == s ? 1u : 0u; }
		}
		if (valid) {
			uint32_t g = jb.pos_tab ? jb.pos_tab[j] + jb.pos_add : jb.pos_base + j * jb.pos_stride;
			emit_symbol(rec, sym_l, magic, g, l, c, jb.t0 + j);
		}
		__syncthreads();
		if (valid) atomicAdd(&bh[s], 1u);
		__syncthreads();
		uint32_t a = bh[4 * lane], b = bh[4 * lane + 1], cc = bh[4 * lane + 2], d = bh[4 * lane + 3], tot;
		uint32_t ex = wave_excl_scan(a + b + cc + d, tot);
		cnt[4 * lane] += a; cnt[4 * lane + 1] += b; cnt[4 * lane + 2] += cc; cnt[4 * lane + 3] += d;
		cum[4 * lane] += ex; cum[4 * lane + 1] += ex + a; cum[4 * lane + 2] += ex + a + b; cum[4 * lane + 3] += ex + a + b + cc;
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------------------
// k_rchain: the serial range recurrence of the reference's single stream (arith/coder.h:69-91):
//     r = R / t;  R' = (h < t) ? r * (h - l) : R - r * l;  while (R' <= 2^62) R' <<= 1
// R does not depend on the low register L, so this chain is split off: it produces r_k and the bit position S_k
// (number of shifts before symbol k); the low register is assembled in parallel by k_low_accumulate.
// One wavefront; the state lives in scalar registers, symbols are fetched 64 at a time (one dwordx4 per lane).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rchain(const SymRec *rec, uint32_t n, uint64_t *r_out, uint32_t *s_out, uint64_t *state)
{
	const int lane = threadIdx.x;
	uint64_t R = state[0];
	uint64_t S = state[1];
	const SymRec idle{ 0, 0, kMetaSub | kMetaNoop };
	SymRec cur = lane < (int)n ? rec[lane] : idle;
	for (uint32_t base = 0; base < n; base += 64) {
		uint32_t nidx = base + 64 + lane;
		SymRec nxt = nidx < n ? rec[nidx] : idle;
		uint32_t mlo = (uint32_t)cur.magic, mhi = (uint32_t)(cur.magic >> 32), mx = cur.x, mm = cur.meta;
		uint64_t my_r = 0;
		uint32_t my_s = 0;
		uint32_t cntb = min(64u, n - base);
		for (uint32_t i = 0; i < cntb; ++i) {
			// an exact no-op (l = 0, h = t) is the form R' = R - r * 0: same code path, no branch in the chain
			uint32_t meta = (uint32_t)__builtin_amdgcn_readlane(mm, i);
			uint32_t s_before = (uint32_t)S;
			uint64_t magic = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mhi, i) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(mlo, i);
			uint32_t x = (uint32_t)__builtin_amdgcn_readlane(mx, i);
			uint64_t r = cm::div_by_magic(R, magic, meta & 63u);
			uint64_t prod = r * x;
			uint64_t Rn = (meta & kMetaSub) ? R - prod : prod;
			uint64_t y = Rn - 1;
			uint32_t sh = (y ? (uint32_t)__builtin_clzll(y) : 64u) - 1u;
			R = Rn << sh;
			S += sh;
			if (lane == (int)i) { my_r = r; my_s = s_before; }
		}
		if (base + lane < n) { r_out[base + lane] = my_r; s_out[base + lane] = my_s; }
		cur = nxt;
	}
	if (lane == 0) { state[0] = R; state[1] = S; }
}

// ---------------------------------------------------------------------------------------------------------
// low register as one big number: L = sum_k (r_k * l_k) << (stream position), added into 64-bit accumulators
// per 32-bit output word (word 0 = most significant).  coder.h:71 (L += r * l) and :73-90 (carry handling by
// bit-plus-follow) are exactly big-number addition with carry propagation; flush (:58-67) appends the 64 bits of L.
// ---------------------------------------------------------------------------------------------------------
constexpr int kAccWin = 640;   // LDS window in 32-bit words per block of 256 symbols (256 x 63 shifts = 504 words + 3)

__global__ __launch_bounds__(256) void k_low_accumulate(const uint64_t *r, const uint32_t *s, const uint32_t *sym_l, uint32_t n, unsigned long long *acc)
{
	// symbols of a block touch a narrow window of output words (positions are monotone): sum them in LDS first, then one
	// global atomic per touched word instead of three contended ones per symbol
	__shared__ unsigned long long win[kAccWin];
	__shared__ uint32_t s_lo, s_hi;
	const uint32_t g0 = blockIdx.x * blockDim.x, g = g0 + threadIdx.x;
	if (threadIdx.x == 0) { s_lo = s[g0]; s_hi = s[min(g0 + 255u, n - 1)]; }
	for (int k = threadIdx.x; k < kAccWin; k += 256) win[k] = 0;
	__syncthreads();
	const uint32_t w0 = s_lo >> 5, span = (s_hi >> 5) - w0 + 3;
	uint64_t a = 0;
	uint32_t w = 0, sh = 0;
	if (g < n) {
		uint32_t l = sym_l[g];
		if (l) { a = r[g] * l; uint32_t pos = s[g]; w = pos >> 5; sh = pos & 31; }
	}
	uint64_t hi = a >> (32 + sh), low = a << (32 - sh);
	uint32_t mid = (uint32_t)(low >> 32), lo = (uint32_t)low;
	if (span <= (uint32_t)kAccWin) {
		if (a) {
			if (hi) atomicAdd(&win[w - w0], (unsigned long long)hi);
			if (mid) atomicAdd(&win[w - w0 + 1], (unsigned long long)mid);
			if (lo) atomicAdd(&win[w - w0 + 2], (unsigned long long)lo);
		}
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < span; k += 256) {
			unsigned long long v = win[k];
			if (v) atomicAdd(&acc[w0 + k], v);
		}
	} else if (a) {
		if (hi) atomicAdd(&acc[w], (unsigned long long)hi);
		if (mid) atomicAdd(&acc[w + 1], (unsigned long long)mid);
		if (lo) atomicAdd(&acc[w + 2], (unsigned long long)lo);
	}
}

// first normalisation: v[k] = low32(acc[k]) + high32(acc[k+1]) < 2^33, after which carries are single bits (formed where it is
// needed: until round 5 a kernel of its own wrote the v[] -- 8 bytes per output word -- for the two below to read)
__device__ __forceinline__ unsigned long long carry_folded(const unsigned long long *acc, uint32_t nw, uint32_t k)
{
	const unsigned long long up = k + 1 < nw ? acc[k + 1] >> 32 : 0ull;
	return (acc[k] & 0xffffffffull) + up;
}

// carry-lookahead over words, processed from the least significant word (index nw-1) upwards.
// pair (g, p): g = the segment generates a carry, p = it propagates an incoming carry.
__device__ __forceinline__ uint32_t gp_of(unsigned long long v) { return (v >> 32 ? 1u : 0u) | ((v == 0xffffffffull) ? 2u : 0u); }
// (generate, propagate) of a lower part followed by a higher part
__device__ __forceinline__ uint32_t gp_then(uint32_t lo, uint32_t hi)
{
	const uint32_t g = (hi & 1u) | (((hi >> 1) & 1u) & (lo & 1u));
	const uint32_t p = ((hi >> 1) & 1u) & ((lo >> 1) & 1u);
	return g | (p << 1);
}

constexpr int kCarryRows = 16;                        // rows of 64 words per wavefront
constexpr int kCarryBlock = 4 * 64 * kCarryRows;      // words per block (four wavefronts)
// A block covers reversed indices [b * kCarryBlock, ...): reversed index i <-> word nw-1-i; wavefront w of the block its kCarryRows
// rows of 64 from b * kCarryBlock + w * 64 * kCarryRows, lane l of a row the word at l: one coalesced load per row.  A row's 64
// look-ahead pairs are two ballots, G (generates) and P (propagates; never both), and carry look-ahead over them is ONE 64-bit
// addition on the scalar unit: (G | P) + G + c has, XORed with its operands, the carry INTO every position, and overflows iff the
// row hands one on.  (Until round 5: four words per thread in a row, thread 0 folding the block's 256 pairs one by one and every
// thread of k_carry_apply the pairs of all threads below it -- 6 ms of carry kernels for the 292 MB of streams of the
// 100 M-triangle mesh, at 3 % of what HBM delivers; the folded words v[] came from a kernel of their own.)
struct CarryRow { unsigned long long G, P; };
__device__ __forceinline__ unsigned long long carry_word(const unsigned long long *acc, uint32_t nw, uint32_t i)
{
	return i < nw ? carry_folded(acc, nw, nw - 1 - i) : 0xffffffffull;   // (behind the end: propagates, never stored)
}
__device__ __forceinline__ void carry_row_masks(unsigned long long v, CarryRow &row)
{
	// (the callers give lane l the row's position 63 - l, so that the lanes' addresses ascend -- descending ones were fetched and
	// written sector by sector: 1.8 GB each way for 292 MB of words, profiles/r5 -- and the ballots are turned round to match)
	row.G = __builtin_bitreverse64(__ballot((v >> 32) != 0ull));
	row.P = __builtin_bitreverse64(__ballot(v == 0xffffffffull));
}
// the row as one pair: bit 0 = hands a carry on by itself, bit 1 = hands an incoming one on
__device__ __forceinline__ uint32_t carry_row_pair(const CarryRow &row)
{
	const unsigned long long X = row.G | row.P, sum = X + row.G;
	return (sum < X ? 1u : 0u) | (row.P == ~0ull ? 2u : 0u);
}
// (round 6) The chunked container's streams are numbers of their own, laid out at their CAPACITY: 443 M words of accumulators
// for 73 M used at 100 M triangles, and the three kernels swept all of it (3.6 GB each way, 2 ms).  A stream's words beyond its
// bit count are zero -- a wavefront's 1 024 of them generate nothing, propagate nothing and their bytes are never packed -- so the
// wavefronts' ranges that hold a used word are marked beforehand and the others load and store nothing.
__global__ __launch_bounds__(256) void k_carry_mark_used(const StreamJob *jobs, const uint32_t *stream_bits, uint32_t ns, uint32_t nw, uint8_t *used)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ns) return;
	const uint32_t w0 = jobs[i].word_base, n = (stream_bits[i] + 31u) >> 5;   // (the low register's last word included: stream_bits = shifts + 32)
	if (!n || w0 >= nw) return;
	const uint32_t w1 = min(nw, w0 + n);
	constexpr uint32_t per = 64u * kCarryRows;   // words of one wavefront
	for (uint32_t g = (nw - w1) / per; g <= (nw - 1u - w0) / per; ++g) used[g] = 1;   // (reversed index nw - 1 - word)
}
__global__ __launch_bounds__(256) void k_carry_block_summary(const unsigned long long *acc, uint32_t nw, uint32_t *summary, const uint8_t *used)
{
	__shared__ uint32_t sm[4];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t base = blockIdx.x * kCarryBlock + wave * 64u * kCarryRows;
	uint32_t a = 2u;   // identity: generates nothing, propagates
	if (used && !used[blockIdx.x * 4u + wave]) a = 0u;   // zeros: no carry out of them, none through them
	else {
		unsigned long long v[kCarryRows];
#pragma unroll
		for (int r = 0; r < kCarryRows; ++r) v[r] = carry_word(acc, nw, base + (uint32_t)r * 64u + (63u - lane));   // (every load in flight before the first ballot waits for one)
#pragma unroll
		for (int r = 0; r < kCarryRows; ++r) {
			CarryRow row;
			carry_row_masks(v[r], row);
			a = gp_then(a, carry_row_pair(row));
		}
	}
	if (lane == 0) sm[wave] = a;
	__syncthreads();
	if (threadIdx.x == 0) summary[blockIdx.x] = gp_then(gp_then(gp_then(sm[0], sm[1]), sm[2]), sm[3]);
}
// summary[b] becomes the carry INTO block b.  One workgroup: every thread folds a contiguous run of blocks, the 1024 run
// aggregates are scanned in LDS (carry look-ahead is associative), then every thread pushes its carry through its run.
// (The first version walked the blocks with one thread: 11 ms on a 28 M-triangle mesh.)
__global__ __launch_bounds__(1024) void k_carry_scan_blocks(uint32_t *summary, uint32_t nblocks)
{
	__shared__ uint32_t agg[1024];
	const uint32_t t = threadIdx.x;
	const uint32_t per = (nblocks + 1023u) / 1024u;
	const uint32_t b0 = min(nblocks, t * per), b1 = min(nblocks, b0 + per);
	uint32_t a = 2u;   // identity: generates nothing, propagates
	for (uint32_t b = b0; b < b1; ++b) a = gp_then(a, summary[b]);
	agg[t] = a;
	__syncthreads();
	// inclusive scan of the aggregates (Hillis-Steele)
	for (uint32_t d = 1; d < 1024u; d <<= 1) {
		const uint32_t mine = agg[t], lower = t >= d ? agg[t - d] : 2u;
		__syncthreads();
		agg[t] = gp_then(lower, mine);
		__syncthreads();
	}
	uint32_t carry = t ? (agg[t - 1] & 1u) : 0u;   // nothing enters block 0
	for (uint32_t b = b0; b < b1; ++b) {
		const uint32_t gp = summary[b];
		summary[b] = carry;
		carry = (gp & 1u) | (((gp >> 1) & 1u) & carry);
	}
}
__global__ __launch_bounds__(256) void k_carry_apply(const unsigned long long *acc, uint32_t nw, const uint32_t *block_carry, uint8_t *bytes, const uint8_t *used)
{
	__shared__ uint32_t sm[4];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t base = blockIdx.x * kCarryBlock + wave * 64u * kCarryRows;
	const bool mine = !used || used[blockIdx.x * 4u + wave];   // (nobody reads the bytes of words no stream uses; zeros hand nothing on)
	unsigned long long v[kCarryRows];
	CarryRow row[kCarryRows];
	uint32_t a = mine ? 2u : 0u;
	if (mine) {
#pragma unroll
		for (int r = 0; r < kCarryRows; ++r) v[r] = carry_word(acc, nw, base + (uint32_t)r * 64u + (63u - lane));   // (every load in flight before the first ballot waits for one)
#pragma unroll
		for (int r = 0; r < kCarryRows; ++r) {
			carry_row_masks(v[r], row[r]);
			a = gp_then(a, carry_row_pair(row[r]));
		}
	}
	if (lane == 0) sm[wave] = a;
	__syncthreads();
	if (!mine) return;
	uint32_t before = 2u;
	for (uint32_t w = 0; w < wave; ++w) before = gp_then(before, sm[w]);
	// the carry into this wavefront's first row: the block's, pushed through the wavefronts below
	unsigned long long c = (before & 1u) | (((before >> 1) & 1u) & block_carry[blockIdx.x]);
	uint32_t *words = (uint32_t*)bytes;   // (big-endian bytes of word k at 4 k: one swapped store)
#pragma unroll
	for (int r = 0; r < kCarryRows; ++r) {
		const unsigned long long X = row[r].G | row[r].P, s1 = X + row[r].G, s2 = s1 + c;
		const unsigned long long into = s2 ^ X ^ row[r].G;   // bit j: the carry into the word at position j of the row
		const uint32_t i = base + (uint32_t)r * 64u + (63u - lane);
		if (i < nw) words[nw - 1 - i] = __builtin_bswap32((uint32_t)(v[r] + ((into >> (63u - lane)) & 1ull)));
		c = (s1 < X || s2 < s1) ? 1ull : 0ull;
	}
}

// ---------------------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

void launch_bounds(hipStream_t st, const uint8_t *rec, uint32_t count, const BoundsPlan &plan,
                   uint8_t *part_min, uint8_t *part_max, uint32_t *part_idx, int nparts, uint8_t *out)
{
	if (plan.n <= 0) return;
	hipLaunchKernelGGL(k_bounds_partial, dim3(nparts, plan.n), dim3(256), 0, st, rec, count, plan, part_min, part_max, part_idx);
	hipLaunchKernelGGL(k_bounds_final, dim3(plan.n), dim3(64), 0, st, part_min, part_max, part_idx, nparts, plan, out);
}
void launch_requant(hipStream_t st, uint8_t *rec, uint32_t count, int stride, const RequantPlan &plan)
{
	if (!count || !plan.n) return;
	unsigned nb = std::min(blocks_for(count, 256), 4096u);
	hipLaunchKernelGGL(k_requant, dim3(nb), dim3(256), 0, st, rec, count, stride, plan);
}
void launch_rank(hipStream_t st, const uint32_t *order_v, uint32_t n, const uint32_t *org, uint32_t *rank)
{
	if (n) hipLaunchKernelGGL(k_rank, dim3(blocks_for(n, 256)), dim3(256), 0, st, order_v, n, org, rank);
}
void launch_predict_vtx(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t n, const uint32_t *rank, const uint8_t *rec,
                        const ListDesc &ld, uint8_t *planes)
{
	if (!n) return;
	const unsigned per = (blocks_for(n, 256) + 7) / 8;   // blocks per XCD; the grid is padded to 8 * per, surplus blocks find k >= n
	hipLaunchKernelGGL(k_predict_vtx, dim3(per * 8), dim3(256), 0, st, cv, order_v, n, rank, rec, ld, planes, per);
}
void launch_face_planes(hipStream_t st, const ConnView &cv, const uint32_t *order_f, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes)
{
	if (n && ld.ncomp) hipLaunchKernelGGL(k_face_planes, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, order_f, n, rec, ld, planes);
}
// ---- over a batch of runs of the coding order: start / first = device arrays of nruns + 1 / nruns entries, total = start[nruns]
void launch_rank_runs(hipStream_t st, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *packed, uint32_t *order_v, const uint32_t *org, uint32_t *rank)
{
	if (total) hipLaunchKernelGGL(k_rank_runs, dim3(blocks_for(total, 256)), dim3(256), 0, st, RunTable{ start, first, nruns, total }, packed, order_v, org, rank);
}
void launch_predict_vtx_runs(hipStream_t st, const ConnView &cv, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *order_v, uint32_t n,
                             const uint32_t *rank, const uint8_t *rec, const ListDesc &ld, uint8_t *planes)
{
	if (!total) return;
	const unsigned per = (blocks_for(total, 256) + 7) / 8;
	hipLaunchKernelGGL(k_predict_vtx_runs, dim3(per * 8), dim3(256), 0, st, cv, RunTable{ start, first, nruns, total }, order_v, n, rank, rec, ld, planes, per);
}
void launch_face_planes_runs(hipStream_t st, const ConnView &cv, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *packed, uint32_t *order_f, uint32_t n,
                             const uint8_t *rec, const ListDesc &ld, uint8_t *planes)
{
	if (total && ld.ncomp) hipLaunchKernelGGL(k_face_planes_runs, dim3(blocks_for(total, 256)), dim3(256), 0, st, cv, RunTable{ start, first, nruns, total }, packed, order_f, n, rec, ld, planes);
}
void launch_split_bytes_runs(hipStream_t st, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *packed, const uint32_t *val, uint32_t n, int nbytes, uint8_t *planes)
{
	if (total) hipLaunchKernelGGL(k_split_bytes_runs, dim3(blocks_for(total, 256)), dim3(256), 0, st, RunTable{ start, first, nruns, total }, packed, val, n, nbytes, planes);
}
// face of every half-edge of a mixed-degree mesh, from the face offsets (the table the topology helpers read)
__global__ __launch_bounds__(256) void k_edge_faces(const uint32_t *foff, uint32_t first, uint32_t nf, uint32_t *eface)
{
	uint32_t f = first + blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= nf) return;
	for (uint32_t e = foff[f]; e < foff[f + 1]; ++e) eface[e] = f;
}
// faces [first, nf)
void launch_edge_faces(hipStream_t st, const uint32_t *foff, uint32_t nf, uint32_t *eface, uint32_t first)
{
	if (nf > first) hipLaunchKernelGGL(k_edge_faces, dim3(blocks_for(nf - first, 256)), dim3(256), 0, st, foff, first, nf, eface);
}
void launch_magic_table(hipStream_t st, MagicEnt *tab, uint32_t from, uint32_t to)
{
	if (to > from) hipLaunchKernelGGL(k_magic_table, dim3(blocks_for(to - from, 256)), dim3(256), 0, st, tab, from, to);
}
void launch_split_bytes(hipStream_t st, const uint32_t *val, uint32_t n, int nbytes, uint8_t *planes)
{
	if (n) hipLaunchKernelGGL(k_split_bytes, dim3(blocks_for(n, 256)), dim3(256), 0, st, val, n, nbytes, planes);
}
void launch_op_records(hipStream_t st, const uint32_t *l, const uint32_t *h, const uint32_t *t, const uint32_t *pos, uint32_t n,
                       const MagicEnt *magic, SymRec *rec, uint32_t *sym_l)
{
	if (n) hipLaunchKernelGGL(k_op_records, dim3(blocks_for(n, 256)), dim3(256), 0, st, l, h, t, pos, n, magic, rec, sym_l);
}
size_t op_model_scratch_bytes(uint32_t n) { return (size_t)((n + kOpChunk - 1) / kOpChunk + 1) * kOpCounters * 4; }
void launch_op_model(hipStream_t st, const uint8_t *op, uint32_t n, const uint32_t *thr, const uint32_t *cum, uint32_t ngroups, void *scratch,
                     const MagicEnt *magic, SymRec *rec, uint32_t *sym_l)
{
	if (!n) return;
	const uint32_t nchunks = (n + kOpChunk - 1) / kOpChunk;
	hipLaunchKernelGGL(k_opmodel_hist, dim3(nchunks), dim3(256), 0, st, op, n, (uint32_t*)scratch);
	hipLaunchKernelGGL(k_opmodel_scan, dim3(1), dim3(256), 0, st, (uint32_t*)scratch, nchunks);
	hipLaunchKernelGGL(k_opmodel_records, dim3(nchunks), dim3(64), 0, st, op, n, (const uint32_t*)scratch, thr, cum, ngroups, magic, rec, sym_l);
}
void launch_type_records(hipStream_t st, uint32_t n, uint32_t pos_base, uint32_t pos_stride, const MagicEnt *magic, SymRec *rec, uint32_t *sym_l)
{
	if (n) hipLaunchKernelGGL(k_type_records, dim3(blocks_for(n, 256)), dim3(256), 0, st, n, pos_base, pos_stride, magic, rec, sym_l);
}
void launch_model(hipStream_t st, const PlaneJob *jobs, uint32_t njobs, const ChunkRef *chunks, uint32_t nchunks, uint32_t *hist,
                  const MagicEnt *magic, SymRec *rec, uint32_t *sym_l)
{
	if (!nchunks) return;
	hipLaunchKernelGGL(k_model_hist, dim3(nchunks), dim3(256), 0, st, jobs, chunks, hist);
	hipLaunchKernelGGL(k_model_scan, dim3(njobs), dim3(256), 0, st, jobs, hist);
	hipLaunchKernelGGL(k_model_lht, dim3(nchunks), dim3(64), 0, st, jobs, chunks, hist, magic, rec, sym_l);
}
void launch_rchain(hipStream_t st, const SymRec *rec, uint32_t n, uint64_t *r_out, uint32_t *s_out, uint64_t *state)
{
	hipLaunchKernelGGL(k_rchain, dim3(1), dim3(64), 0, st, rec, n, r_out, s_out, state);
}
void launch_low_accumulate(hipStream_t st, const uint64_t *r, const uint32_t *s, const uint32_t *sym_l, uint32_t n, uint64_t *acc)
{
	if (n) hipLaunchKernelGGL(k_low_accumulate, dim3(blocks_for(n, 256)), dim3(256), 0, st, r, s, sym_l, n, (unsigned long long*)acc);
}
// jobs / stream_bits (ns streams, device): the streams' places and lengths, for a container of many streams (nullptr: ONE number
// of nw words).  summary: nw / 1024 + 2 words (the blocks' pairs, behind them the blocks' marks)
// Up to four ranges of 32-bit words from PINNED HOST memory into device arrays, by ONE kernel (round 6: the decoder's helper threads
// leave their stretches of the connectivity in pinned mirrors; as four hipMemcpyAsync per stretch those 24 MB of the headline
// mesh went up as ~ 45 serialized transfers of 20 - 45 us each, 2 ms in front of the chains of everything behind the first
// stretch; a kernel reads the host memory over the link with thousands of requests in flight)
__global__ __launch_bounds__(256) void k_pull_ranges(PullRanges r)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t n = r.words[k];
		const uint32_t *src = r.src[k];
		uint32_t *dst = r.dst[k];
		for (uint32_t i = t; i < n; i += step) dst[i] = __builtin_nontemporal_load(src + i);
	}
}
void launch_pull_ranges(hipStream_t st, const PullRanges &r)
{
	const uint64_t total = (uint64_t)r.words[0] + r.words[1] + r.words[2] + r.words[3];
	if (!total) return;
	const unsigned nb = (unsigned)std::min<uint64_t>(1024, (total + 1023) / 1024);   // four words a thread at least
	hipLaunchKernelGGL(k_pull_ranges, dim3(nb), dim3(256), 0, st, r);
}
void launch_carry(hipStream_t st, const uint64_t *acc, uint32_t nw, uint64_t *v, uint32_t *summary, uint8_t *bytes, const StreamJob *jobs, const uint32_t *stream_bits, uint32_t ns)
{
	unsigned nb = blocks_for(nw, kCarryBlock);
	(void)v;   // (the folded words are formed inside the kernels)
	uint8_t *used = nullptr;
	static const bool sweep_all = [] { const char *e = getenv("HRY_CARRY_SWEEP_ALL"); return e && *e && *e != '0'; }();
	if (jobs && stream_bits && ns > 1 && !sweep_all) {
		used = (uint8_t*)(summary + nb + 1);   // (a mark per wavefront: 4 nb bytes behind the nb words of the pairs; summary has nw / 1024 + 2 words)
		(void)hipMemsetAsync(used, 0, (size_t)nb * 4, st);
		hipLaunchKernelGGL(k_carry_mark_used, dim3(blocks_for(ns, 256)), dim3(256), 0, st, jobs, stream_bits, ns, nw, used);
	}
	hipLaunchKernelGGL(k_carry_block_summary, dim3(nb), dim3(256), 0, st, (const unsigned long long*)acc, nw, summary, (const uint8_t*)used);
	hipLaunchKernelGGL(k_carry_scan_blocks, dim3(1), dim3(1024), 0, st, summary, nb);
	hipLaunchKernelGGL(k_carry_apply, dim3(nb), dim3(256), 0, st, (const unsigned long long*)acc, nw, summary, bytes, (const uint8_t*)used);
}

}   // namespace dev
}   // namespace hry
