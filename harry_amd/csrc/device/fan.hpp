// Device-side helpers shared by the kernel files: unaligned loads / stores of record slots, dispatch on a storage type, the
// half-edge navigation of the flat connectivity and the reference's walk around a vertex (attrcode.h:83-106).
#pragma once
#include <hip/hip_runtime.h>

#include "codec_math.hpp"
#include "dev_types.hpp"

namespace hry {
namespace dev {

template <typename T> __device__ __forceinline__ T ldg(const uint8_t *p)
{
	T v;
	__builtin_memcpy(&v, p, sizeof(T));
	return v;
}
template <typename T> __device__ __forceinline__ void stg(uint8_t *p, T v) { __builtin_memcpy(p, &v, sizeof(T)); }

template <typename F> __device__ __forceinline__ void with_stype(int st, F &&f)
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
	default: break;   // DOUBLE is rejected on the host (prediction.h:33-44 reads out of bounds for 8-byte floats)
	}
}

struct Topo {
	ConnView c;
	__device__ __forceinline__ uint32_t face(uint32_t e) const { return c.eface ? c.eface[e] : e / c.udeg; }
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

// Visit the prediction candidates of the vertex at half-edge `ein` in the reference's fan order
// (attrcode.h:83-106 TFAN_IT, :155-171 paral, :117-121 use_paral).  A candidate (v0, v1, vo) is kept iff all three
// vertices were coded before the current one and not before `lo`: rank in [lo, my_rank).
// The walk is bounded so that a corrupt twin table cannot hang the wave.
template <typename F>
__device__ __forceinline__ void fan_candidates(const Topo &tp, const uint32_t *rank, uint32_t ein, uint32_t my_rank, uint32_t lo, F &&f)
{
	auto offer = [&](uint32_t v0, uint32_t v1, uint32_t vo) {
		uint32_t r0 = rank[v0], r1 = rank[v1], r2 = rank[vo];
		if (r0 < my_rank && r1 < my_rank && r2 < my_rank && r0 >= lo && r1 >= lo && r2 >= lo) f(v0, v1, vo);
	};
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

// Every half-edge that starts at the vertex of `ein`, in the reference's order (attrcode.h:83-106 TFAN_IT): forward through
// the twins until the walk returns or meets a border, then backward from the start.  Bounded like fan_candidates.
template <typename F>
__device__ __forceinline__ void fan_each(const Topo &tp, uint32_t ein, F &&f)
{
	const int kMaxSteps = 1 << 16;
	uint32_t e = ein, t;
	int steps = 0;
	bool border = false;
	for (;;) {
		f(e);
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
		f(e);
		e = tp.prev(e);
		t = tp.c.twin[e];
		if (e == t) break;
		e = t;
	} while (e != ein && ++steps <= kMaxSteps);
}

}   // namespace dev
}   // namespace hry
