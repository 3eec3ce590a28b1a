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
	auto visit = [&](uint32_t e) {
		uint32_t d = tp.degree(e);
		if (d == 3) {
			uint32_t e1 = tp.next(e), t = tp.c.twin[e1];
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
		t = tp.c.twin[e];
		if (t == e) { border = true; break; }
		e = tp.next(t);
		if (e == ein || ++steps > kMaxSteps) break;
	}
	if (!border) return;
	e = tp.prev(ein);
	t = tp.c.twin[e];
	if (e == t) return;
	e = t;
	do {
		visit(e);
		e = tp.prev(e);
		t = tp.c.twin[e];
		if (e == t) break;
		e = t;
	} while (e != ein && ++steps <= kMaxSteps);
}

constexpr int kCandMax = 8;

__global__ __launch_bounds__(256) void k_candidates(ConnView cv, const uint32_t *order_v, uint32_t n, uint32_t *cand, uint8_t *ncand)
{
	uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= n) return;
	TopoD tp{ cv };
	uint32_t k = 0;
	uint32_t *out = cand + (size_t)v * (kCandMax * 3);
	fan_ids(tp, order_v[v], v, [&](uint32_t a, uint32_t b, uint32_t o) {
		if (k < (uint32_t)kCandMax) { out[3 * k] = a; out[3 * k + 1] = b; out[3 * k + 2] = o; }
		++k;
	});
	ncand[v] = k > (uint32_t)kCandMax ? 0xff : (uint8_t)k;
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

// one lane = one component.  ring: reconstructed values (as 64-bit bit patterns) of the last `ring_n` vertices.
template <typename T>
__device__ __forceinline__ void unpredict_lane(const TopoD &tp, const uint32_t *order_v, uint32_t nvtx, const uint32_t *cand, const uint8_t *ncand,
                                               uint8_t *rec, int stride, int off, int q, unsigned long long *ring, uint32_t ring_n, int ring_stride, int lane)
{
	typedef typename cm::wide<T>::type W;
	typedef typename cm::word<sizeof(T)>::u U;
	auto fetch = [&](uint32_t id, uint32_t v) -> T {
		if (v - id <= ring_n) return cm::bits<T>((U)ring[(size_t)(id & (ring_n - 1)) * ring_stride + lane]);
		return ldq<T>(rec + (size_t)id * stride + off);
	};
	for (uint32_t v = 0; v < nvtx; ++v) {
		uint32_t nc = ncand[v];
		T pv[kCandMax];
		T pred = T(0);
		if (nc != 0xff) {
			const uint32_t *cd = cand + (size_t)v * (kCandMax * 3);
			W acc = 0;
#pragma unroll
			for (int k = 0; k < kCandMax; ++k) {
				if ((uint32_t)k < nc) {
					pv[k] = cm::parallelogram<T>(fetch(cd[3 * k], v), fetch(cd[3 * k + 1], v), fetch(cd[3 * k + 2], v), q);
					acc = acc + (W)pv[k];
				}
			}
			if (nc) {
				T avg = (T)cm::mean_of(acc, (W)nc);
				if constexpr (!cm::is_fp<T>::value) pred = avg;
				else {
					T best = 3.402823466e+38f;
#pragma unroll
					for (int k = 0; k < kCandMax; ++k) {
						if ((uint32_t)k < nc) {
							T db = avg > best ? avg - best : best - avg;
							T dp = avg > pv[k] ? avg - pv[k] : pv[k] - avg;
							best = db < dp ? best : pv[k];
						}
					}
					pred = best;
				}
			}
		} else {
			// more candidates than the table holds (high-valence vertex): walk the fan here
			W acc = 0;
			uint32_t n = 0;
			fan_ids(tp, order_v[v], v, [&](uint32_t a, uint32_t b, uint32_t o) {
				acc = acc + (W)cm::parallelogram<T>(fetch(a, v), fetch(b, v), fetch(o, v), q);
				++n;
			});
			if (n) {
				T avg = (T)cm::mean_of(acc, (W)n);
				if constexpr (!cm::is_fp<T>::value) pred = avg;
				else {
					T best = 3.402823466e+38f;
					fan_ids(tp, order_v[v], v, [&](uint32_t a, uint32_t b, uint32_t o) {
						T p = cm::parallelogram<T>(fetch(a, v), fetch(b, v), fetch(o, v), q);
						T db = avg > best ? avg - best : best - avg;
						T dp = avg > p ? avg - p : p - avg;
						best = db < dp ? best : p;
					});
					pred = best;
				}
			}
		}
		uint8_t *slot = rec + (size_t)v * stride + off;
		T val = cm::value_from_residual<T>(ldq<U>(slot), pred, q);
		stq<T>(slot, val);
		ring[(size_t)(v & (ring_n - 1)) * ring_stride + lane] = (unsigned long long)cm::bits<U>(val);
	}
}

__global__ __launch_bounds__(64) void k_unpredict(ConnView cv, const uint32_t *order_v, uint32_t nvtx, const uint32_t *cand, const uint8_t *ncand,
                                                  ListDesc ld, uint8_t *rec, uint32_t ring_n)
{
	extern __shared__ unsigned long long ring[];
	const int lane = threadIdx.x;
	if (lane >= ld.ncomp) return;
	TopoD tp{ cv };
	with_st(ld.stype[lane], [&](auto tag) {
		unpredict_lane<decltype(tag)>(tp, order_v, nvtx, cand, ncand, rec, ld.stride, ld.off[lane], ld.quant[lane], ring, ring_n, ld.ncomp, lane);
	});
}

// ---------------------------------------------------------------------------------------------------------
void launch_candidates(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t n, uint32_t *cand, uint8_t *ncand)
{
	if (n) hipLaunchKernelGGL(k_candidates, dim3((n + 255) / 256), dim3(256), 0, st, cv, order_v, n, cand, ncand);
}
void launch_residuals_to_rec(hipStream_t st, const uint8_t *planes, uint32_t n, const ListDesc &ld, uint8_t *rec)
{
	if (n && ld.ncomp) hipLaunchKernelGGL(k_residuals_to_rec, dim3((n + 255) / 256), dim3(256), 0, st, planes, n, ld, rec);
}
void launch_faces_unfold(hipStream_t st, uint32_t n, const ListDesc &ld, uint8_t *rec)
{
	if (n && ld.ncomp) hipLaunchKernelGGL(k_faces_unfold, dim3((n + 255) / 256), dim3(256), 0, st, n, ld, rec);
}
void launch_unpredict(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, const uint32_t *cand, const uint8_t *ncand,
                      const ListDesc &ld, uint8_t *rec)
{
	if (!nvtx || !ld.ncomp) return;
	uint32_t ring_n = 4096;
	while ((size_t)ring_n * ld.ncomp * 8 > 96 * 1024 && ring_n > 64) ring_n >>= 1;
	size_t lds = (size_t)ring_n * ld.ncomp * 8;
	hipFuncSetAttribute((const void*)k_unpredict, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(k_unpredict, dim3(1), dim3(64), lds, st, cv, order_v, nvtx, cand, ncand, ld, rec, ring_n);
}

}   // namespace dev
}   // namespace hry
