// Half-edge twin matching on the device (reference: structs/conn.h:172-234, the builder every
// reader feeds).  The reference matches while it reads: a directed edge (a, b) takes the pending (b, a) if there is one and
// otherwise becomes pending itself -- unless an (a, b) is pending already, in which case it stays unmatched for good
// (conn.h:201-214).  Only half-edges over the same undirected edge interact, in the order of their indices, so:
//   k_twin_count    one counter per vertex: half-edges whose smaller endpoint it is          (atomics, 4 B per half-edge)
//   k_scan_*        exclusive scan of the counters
//   k_twin_scatter  (larger endpoint, half-edge) pairs into the segment of the smaller endpoint (any order)
//   k_twin_match    one thread per vertex: sorts its segment (a handful of entries) by (larger endpoint, half-edge) and
//                   replays the reference's rule over every run of equal larger endpoints
// HBM-bound integer work: 4 B read + 8 B written + 8 B read + 4 B written per half-edge.
#include <hip/hip_runtime.h>

#include "dev_types.hpp"
#include "fan.hpp"
#include "kernels.hpp"

namespace hry {
namespace dev {

__global__ __launch_bounds__(256) void k_twin_count(ConnView cv, uint32_t *twin, uint32_t *count)
{
	const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h >= cv.ne) return;
	Topo tp{ cv };
	const uint32_t a = cv.org[h], c = cv.org[tp.next(h)];
	twin[h] = h;
	atomicAdd(&count[min(a, c)], 1u);
}

// ---- exclusive scan over n counters: block sums, scan of the sums by one block, apply --------------------------------------
constexpr int kScanBlock = 1024;
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &block_total)   // 1024 threads
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
	if (lane == 63) s_wave[wave] = inc;
	__syncthreads();
	if (wave == 0) {
		uint32_t w = lane < kScanBlock / 64 ? s_wave[lane] : 0u, wi = w;
#pragma unroll
		for (int d = 1; d < 16; d <<= 1) { uint32_t o = __shfl_up(wi, d, 64); if (lane >= d) wi += o; }
		if (lane < kScanBlock / 64) s_wave[lane] = wi - w;
		if (lane == kScanBlock / 64 - 1) s_wave[16] = wi;
	}
	__syncthreads();
	block_total = s_wave[16];
	return s_wave[wave] + inc - v;
}
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(const uint32_t *in, uint32_t n, uint32_t *sums)
{
	__shared__ uint32_t s_wave[17];
	const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
	uint32_t total;
	block_excl_scan(i < n ? in[i] : 0u, s_wave, total);
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanBlock) void k_scan_top(uint32_t *sums, uint32_t nb)   // one block: exclusive scan in place, any nb
{
	__shared__ uint32_t s_wave[17];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nb; base += kScanBlock) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < nb ? sums[i] : 0u;
		uint32_t total;
		const uint32_t ex = block_excl_scan(v, s_wave, total);
		if (i < nb) sums[i] = carry + ex;
		carry += total;
		__syncthreads();
	}
}
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const uint32_t *in, uint32_t n, const uint32_t *sums, uint32_t *out)   // out[n] = total
{
	__shared__ uint32_t s_wave[17];
	const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
	const uint32_t v = i < n ? in[i] : 0u;
	uint32_t total;
	const uint32_t ex = block_excl_scan(v, s_wave, total) + sums[blockIdx.x];
	if (i < n) out[i] = ex;
	if (i == n - 1) out[n] = ex + v;
}

__global__ __launch_bounds__(256) void k_twin_scatter(ConnView cv, const uint32_t *start, uint32_t *fill, unsigned long long *ent)
{
	const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h >= cv.ne) return;
	Topo tp{ cv };
	const uint32_t a = cv.org[h], c = cv.org[tp.next(h)];
	const uint32_t lo = min(a, c), hi = max(a, c);
	const uint32_t pos = start[lo] + atomicAdd(&fill[lo], 1u);
	ent[pos] = ((unsigned long long)hi << 32) | h;
}

// vertices with more than kTwinSegMax half-edges in their segment (the hub of a fan with thousands of spokes) are not for one
// thread's insertion sort: they are listed in `over` (over[0] = how many, then the vertex ids, at most kTwinOverMax of them) and
// matched by the host, which has the twins anyway (Context::upload_mesh)
constexpr uint32_t kTwinSegMax = 48, kTwinOverMax = 4096;
__global__ __launch_bounds__(256) void k_twin_match(ConnView cv, uint32_t nv, const uint32_t *start, unsigned long long *ent, uint32_t *twin, uint32_t *over)
{
	const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
	if (lo >= nv) return;
	const uint32_t b = start[lo], e = start[lo + 1];
	if (e - b < 2) return;
	if (e - b > kTwinSegMax) {
		const uint32_t k = atomicAdd(&over[0], 1u);
		if (k < kTwinOverMax) over[1 + k] = lo;
		return;
	}
	for (uint32_t i = b + 1; i < e; ++i) {   // insertion sort by (larger endpoint, half-edge): segments are a handful of entries
		const unsigned long long x = ent[i];
		uint32_t j = i;
		while (j > b && ent[j - 1] > x) { ent[j] = ent[j - 1]; --j; }
		ent[j] = x;
	}
	constexpr uint32_t NONE = 0xffffffffu;
	uint32_t cur_hi = NONE, pend_out = NONE, pend_in = NONE;   // pending lo -> hi, pending hi -> lo
	for (uint32_t i = b; i < e; ++i) {
		const unsigned long long x = ent[i];
		const uint32_t hi = (uint32_t)(x >> 32), h = (uint32_t)x;
		if (hi != cur_hi) { cur_hi = hi; pend_out = pend_in = NONE; }
		if (hi == lo) {   // a == b: the edge is its own reverse (one pending slot)
			if (pend_out != NONE) { twin[h] = pend_out; twin[pend_out] = h; pend_out = NONE; }
			else pend_out = h;
			continue;
		}
		const bool out = cv.org[h] == lo;
		uint32_t &opposite = out ? pend_in : pend_out, &same = out ? pend_out : pend_in;
		if (opposite != NONE) { twin[h] = opposite; twin[opposite] = h; opposite = NONE; }
		else if (same == NONE) same = h;   // a second pending edge of the same direction is dropped (conn.h:210-213)
	}
}

static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// ws: (2 * nv + 2 + blocks) * 4 bytes of counters + ne * 8 bytes of entries (8-byte aligned first) + the overflow list
size_t twin_workspace_bytes(uint32_t nv, uint32_t ne) { return (size_t)ne * 8 + ((size_t)2 * nv + 2 + blocks_for(nv, kScanBlock) + 2 + kTwinOverMax + 1) * 4; }
uint32_t twin_overflow_capacity() { return kTwinOverMax; }
// over_out: device pointer of the overflow list (count, then vertex ids) inside ws
void launch_twins(hipStream_t st, const ConnView &cv, uint32_t nv, uint32_t *twin, void *ws, const uint32_t **over_out)
{
	if (!cv.ne) return;
	unsigned long long *ent = (unsigned long long*)ws;
	uint32_t *count = (uint32_t*)(ent + cv.ne), *start = count + nv, *sums = start + nv + 1;
	uint32_t *over = sums + blocks_for(nv, kScanBlock) + 2;
	(void)hipMemsetAsync(over, 0, 4, st);
	*over_out = over;
	const unsigned nb = blocks_for(nv, kScanBlock);
	(void)hipMemsetAsync(count, 0, (size_t)nv * 4, st);
	hipLaunchKernelGGL(k_twin_count, dim3(blocks_for(cv.ne, 256)), dim3(256), 0, st, cv, twin, count);
	hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(kScanBlock), 0, st, count, nv, sums);
	hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanBlock), 0, st, sums, nb);
	hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kScanBlock), 0, st, count, nv, sums, start);
	(void)hipMemsetAsync(count, 0, (size_t)nv * 4, st);   // reused as the fill cursors
	hipLaunchKernelGGL(k_twin_scatter, dim3(blocks_for(cv.ne, 256)), dim3(256), 0, st, cv, start, count, ent);
	hipLaunchKernelGGL(k_twin_match, dim3(blocks_for(nv, 256)), dim3(256), 0, st, cv, nv, start, ent, twin, over);
}


// ---------------------------------------------------------------------------------------------------------------------------
// The connected components of the faces and what the walk on several host threads needs to know of them before it starts
// (host/cbm_walk.cpp: analyse_impl -- which faces form a component, the coding order, how many faces / half-edges / new vertices
// each component brings, which components share a vertex), on the device where the connectivity is resident anyway: on the
// host these passes are 1.9 CPU-seconds for the 100 M-triangle configs[3] mesh -- 117 ms on the 16 CPUs a box grants, as much
// as the walks themselves.  Data-parallel integer work over 300 M half-edges and 50 M vertices with random access into tables
// of a few hundred megabytes: atomics on label / vertex words, per-component sums aggregated per wavefront first (faces and
// vertices of one component are neighbours in memory: a wavefront holds one or two components).
//   k_cc_init / k_cc_hook / k_cc_flatten   lock-free union-find, the root of a set is its smallest face (what the host's is)
//   k_cc_roots + k_scan_*                  dense component numbers in face order of the roots
//   k_cc_face_stats                        faces, half-edges, face interval, smallest key of the start-face sequence per component
//   k_cc_vertex_first / _ties / _tie_pairs / _stats   the first component (coding order) at every vertex, components tied by a
//                                          shared vertex (noted as pairs, then a second union-find over components), new
//                                          vertices and vertex interval
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uf_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t uf_find(uint32_t *par, uint32_t x)
{
	// parents only ever decrease and a face that has got a parent never becomes a root again: a stale read is a valid (higher)
	// ancestor, and the halving store -- a plain one -- can only replace an ancestor by another ancestor
	uint32_t p = uf_load(par + x);
	while (p != x) {
		const uint32_t g = uf_load(par + p);
		if (g != p) __hip_atomic_store(par + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		x = p; p = g;
	}
	return x;
}
__device__ __forceinline__ void uf_unite(uint32_t *par, uint32_t a, uint32_t b)
{
	for (;;) {
		a = uf_find(par, a); b = uf_find(par, b);
		if (a == b) return;
		if (a > b) { const uint32_t t = a; a = b; b = t; }
		const uint32_t old = atomicCAS(par + b, b, a);   // b is a root still: it hangs under the smaller root
		if (old == b) return;
		b = old;
	}
}
__global__ __launch_bounds__(256) void k_cc_init(uint32_t *label, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) label[i] = i;
}
// An edge with a twin on both sides is taken from its larger face, a one-sided twin from the side that has it.
//
// Round 5, in two passes.  Faces that lie near each other in the arrays mostly ARE neighbours (a file is written patch by patch), so a
// workgroup first unites its own kHookFaces faces in LDS -- the same lock-free union-find, no traffic --
// and writes every face's local root as its label; the second pass takes only the edges that leave a workgroup's faces to the
// union-find in HBM, where a set now arrives as one root per workgroup instead of face by face.  (One pass over HBM: 19.0 ms for
// the 78.5 M faces of the configs[3] mesh, its searches and compare-and-swaps all in the L2.)
constexpr uint32_t kHookFaces = 16384, kHookThreads = 1024;   // 64 KB of parents in LDS: two workgroups a compute unit
__device__ __forceinline__ uint32_t lds_find(uint32_t *par, uint32_t x)
{
	uint32_t p = __hip_atomic_load(par + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	while (p != x) {
		const uint32_t g = __hip_atomic_load(par + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (g != p) __hip_atomic_store(par + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		x = p; p = g;
	}
	return x;
}
__global__ __launch_bounds__(kHookThreads) void k_cc_hook_local(ConnView cv, uint32_t *label)
{
	__shared__ uint32_t par[kHookFaces];
	const uint32_t base = blockIdx.x * kHookFaces;
	for (uint32_t t = threadIdx.x; t < kHookFaces; t += kHookThreads) par[t] = t;
	__syncthreads();
	Topo tp{ cv };
	for (uint32_t t = threadIdx.x; t < kHookFaces; t += kHookThreads) {
		const uint32_t f = base + t;
		if (f >= cv.nf) break;
		const uint32_t h0 = cv.eface ? cv.foff[f] : f * cv.udeg, h1 = cv.eface ? cv.foff[f + 1] : h0 + cv.udeg;
		for (uint32_t h = h0; h < h1; ++h) {
			const uint32_t o = cv.twin[h];
			if (o == h || o >= cv.ne) continue;
			const uint32_t b = tp.face(o);
			if (b == f || (b > f && cv.twin[o] == h)) continue;
			if (b - base >= kHookFaces) continue;   // leaves the workgroup's faces: the second pass
			uint32_t x = t, y = b - base;
			for (;;) {
				x = lds_find(par, x); y = lds_find(par, y);
				if (x == y) break;
				if (x > y) { const uint32_t s = x; x = y; y = s; }
				const uint32_t old = atomicCAS(par + y, y, x);
				if (old == y) break;
				y = old;
			}
		}
	}
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < kHookFaces; t += kHookThreads) {   // (the unions are over: a root read is final)
		const uint32_t f = base + t;
		if (f >= cv.nf) break;
		uint32_t x = t, q = par[x];
		while (q != x) { x = q; q = par[x]; }
		label[f] = base + x;
	}
}
__global__ __launch_bounds__(256) void k_cc_hook(ConnView cv, uint32_t *label)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= cv.nf) return;
	Topo tp{ cv };
	const uint32_t base = f - f % kHookFaces;
	const uint32_t h0 = cv.eface ? cv.foff[f] : f * cv.udeg, h1 = cv.eface ? cv.foff[f + 1] : h0 + cv.udeg;
	for (uint32_t h = h0; h < h1; ++h) {
		const uint32_t o = cv.twin[h];
		if (o == h || o >= cv.ne) continue;
		const uint32_t b = tp.face(o);
		if (b == f || (b > f && cv.twin[o] == h)) continue;
		if (b - base < kHookFaces) continue;   // united in LDS by the first pass
		uf_unite(label, f, b);
	}
}
// every label becomes its root.  No halving here: a halving store into an element another thread has already set to its root would
// leave an ancestor there; an element's own store is the only one it gets, and a root read is final (the unions are over)
__global__ __launch_bounds__(256) void k_cc_flatten(uint32_t *label, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t x = i, p = uf_load(label + x);
	while (p != x) { x = p; p = uf_load(label + x); }
	if (x != i) __hip_atomic_store(label + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_cc_roots(const uint32_t *label, uint32_t n, uint32_t *flag)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[i] = label[i] == i ? 1u : 0u;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64); return v; }
__device__ __forceinline__ unsigned long long wave_min64(unsigned long long v)
{
	for (int d = 32; d; d >>= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
	return v;
}
// spans: the start-face sequence as (lowest face, highest face, position of the span's first face, ascending?) sorted by the
// lowest face (host/cbm_walk.cpp: StartFaces::index_blocks); label[] holds a face's root on entry, its component number on exit
__global__ __launch_bounds__(256) void k_cc_face_stats(ConnView cv, uint32_t *label, const uint32_t *num, const uint32_t *spans, uint32_t nspans,
                                                       uint32_t *nfaces, uint32_t *nhe, uint32_t *flo, uint32_t *fhi, unsigned long long *first_key)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = f < cv.nf;
	uint32_t c = 0xffffffffu, deg = 0;
	unsigned long long key = ~0ull;
	if (valid) {
		c = num[label[f]];
		label[f] = c;
		deg = cv.eface ? cv.foff[f + 1] - cv.foff[f] : cv.udeg;
		uint32_t a = 0, b = nspans;
		while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (spans[4 * mid] <= f) a = mid; else b = mid; }
		const uint32_t lo = spans[4 * a], hi = spans[4 * a + 1], pos0 = spans[4 * a + 2], asc = spans[4 * a + 3];
		const uint32_t pos = pos0 + (asc ? f - lo : hi - f);
		key = f == 0 ? 0ull : (((unsigned long long)pos + 1ull) << 32) | f;   // the reference takes face 0 first whatever the set's order (writer.cc:40-46)
	}
	// per wavefront the sums of every component it holds (one or two, where the components are more than slivers), then -- round 5 --
	// per WORKGROUP through a list in LDS: its four wavefronts mostly hold the same component, and five atomics per wavefront and
	// component met on that component's counters from every wavefront in flight (4.1 ms at 100 M triangles)
	__shared__ uint32_t l_c[256], l_n[256], l_deg[256], l_first[256], l_last[256], l_count;
	__shared__ unsigned long long l_key[256];
	if (threadIdx.x == 0) l_count = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	unsigned long long todo = __ballot(valid);
	while (todo) {
		const int leader = __ffsll((long long)todo) - 1;
		const uint32_t c0 = (uint32_t)__shfl((int)c, leader, 64);
		const bool mine = valid && c == c0;
		const unsigned long long mask = __ballot(mine);
		const uint32_t n = (uint32_t)__popcll(mask), sdeg = wave_sum(mine ? deg : 0u);
		const unsigned long long kmin = wave_min64(mine ? key : ~0ull);
		const int last = 63 - __clzll((long long)mask);
		const uint32_t f_first = (uint32_t)__shfl((int)f, leader, 64), f_last = (uint32_t)__shfl((int)f, last, 64);
		if (lane == leader) {
			const uint32_t at = atomicAdd(&l_count, 1u);   // (at most one entry per face: 256)
			l_c[at] = c0; l_n[at] = n; l_deg[at] = sdeg; l_first[at] = f_first; l_last[at] = f_last; l_key[at] = kmin;
		}
		todo &= ~mask;
	}
	__syncthreads();
	const uint32_t entries = l_count;
	if (threadIdx.x < entries) {
		// the first entry of a component gathers the later ones of the same component and speaks for them
		const uint32_t i = threadIdx.x, c0 = l_c[i];
		bool first = true;
		for (uint32_t j = 0; j < i && first; ++j) first = l_c[j] != c0;
		if (first) {
			uint32_t n = l_n[i], sdeg = l_deg[i], lo = l_first[i], hi = l_last[i];
			unsigned long long kmin = l_key[i];
			for (uint32_t j = i + 1; j < entries; ++j) if (l_c[j] == c0) {
				n += l_n[j]; sdeg += l_deg[j]; lo = min(lo, l_first[j]); hi = max(hi, l_last[j]); kmin = min(kmin, l_key[j]);
			}
			atomicAdd(nfaces + c0, n); atomicAdd(nhe + c0, sdeg);
			atomicMin(flo + c0, lo); atomicMax(fhi + c0, hi + 1u);
			atomicMin(first_key + c0, kmin);
		}
	}
}
// comp[] = component number of every face (k_cc_face_stats), rank_of[] = its place in the coding order; one thread per face
__global__ __launch_bounds__(256) void k_cc_vertex_first(ConnView cv, const uint32_t *comp, const uint32_t *rank_of, uint32_t *vfirst)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= cv.nf) return;
	const uint32_t k = rank_of[comp[f]];
	const uint32_t h0 = cv.eface ? cv.foff[f] : f * cv.udeg, h1 = cv.eface ? cv.foff[f + 1] : h0 + cv.udeg;
	for (uint32_t h = h0; h < h1; ++h) atomicMin(vfirst + cv.org[h], k);
}
// A corner whose vertex another component reached first ties the two.  Such corners are rare (three in a thousand on the configs[3]
// mesh) but every wavefront of 64 faces holds one, and a union inline -- searches, a compare-and-swap, retries -- kept all 64 lanes
// waiting for it: 13 ms at 100 M triangles against 2.7 ms for the pass before it, which touches the same words.  The pairs are
// appended to a list instead (one atomic per wavefront and corner round) and united by a kernel of their own; what does not fit
// in the list is united on the spot.
constexpr uint32_t kTieLists = 64;   // (a power of two that divides kTiePairs)
__global__ __launch_bounds__(256) void k_cc_vertex_ties(ConnView cv, const uint32_t *comp, const uint32_t *rank_of, const uint32_t *vfirst, uint32_t *tie,
                                                        uint2 *pairs, uint32_t cap, uint32_t *count)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = f < cv.nf;
	const uint32_t k = live ? rank_of[comp[f]] : 0u;
	const uint32_t h0 = !live ? 0u : cv.eface ? cv.foff[f] : f * cv.udeg, h1 = !live ? 0u : cv.eface ? cv.foff[f + 1] : h0 + cv.udeg;
	const int lane = threadIdx.x & 63;
	uint32_t rounds = h1 - h0;
	for (int d = 32; d; d >>= 1) rounds = max(rounds, (uint32_t)__shfl_xor((int)rounds, d, 64));
	for (uint32_t i = 0; i < rounds; ++i) {
		const uint32_t h = h0 + i;
		const uint32_t first = h < h1 ? vfirst[cv.org[h]] : k;
		const bool hit = first != k;
		const unsigned long long mask = __ballot(hit);
		if (!mask) continue;
		// (kTieLists lists with a counter each, a workgroup appends to the one of its index: every wavefront of the configs[3] mesh
		// notes a pair or two, and 2.5 M additions to ONE word were most of this kernel's 8.1 ms)
		uint32_t base = 0;
		const int leader = __ffsll((long long)mask) - 1;
		const uint32_t sub = blockIdx.x & (kTieLists - 1u), sub_cap = cap / kTieLists;
		if (lane == leader) base = atomicAdd(count + sub, (uint32_t)__popcll(mask));
		base = (uint32_t)__shfl((int)base, leader, 64);
		if (hit) {
			const uint32_t at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
			if (at < sub_cap) pairs[(size_t)sub * sub_cap + at] = make_uint2(first, k);
			else uf_unite(tie, first, k);
		}
	}
}
__global__ __launch_bounds__(256) void k_cc_tie_pairs(const uint2 *pairs, uint32_t cap, const uint32_t *count, uint32_t *tie)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t sub_cap = cap / kTieLists, sub = i / sub_cap, at = i - sub * sub_cap;
	if (sub < kTieLists && at < min(count[sub], sub_cap)) uf_unite(tie, pairs[i].x, pairs[i].y);
}
__global__ __launch_bounds__(256) void k_cc_vertex_stats(const uint32_t *vfirst, uint32_t nv, uint32_t *fresh, uint32_t *vlo, uint32_t *vhi)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t k = v < nv ? vfirst[v] : 0xffffffffu;
	const bool valid = k != 0xffffffffu;
	// (per wavefront, then per workgroup through LDS, as k_cc_face_stats)
	__shared__ uint32_t l_k[256], l_n[256], l_first[256], l_last[256], l_count;
	if (threadIdx.x == 0) l_count = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	unsigned long long todo = __ballot(valid);
	while (todo) {
		const int leader = __ffsll((long long)todo) - 1;
		const uint32_t k0 = (uint32_t)__shfl((int)k, leader, 64);
		const unsigned long long mask = __ballot(valid && k == k0);
		const int last = 63 - __clzll((long long)mask);
		const uint32_t v_first = (uint32_t)__shfl((int)v, leader, 64), v_last = (uint32_t)__shfl((int)v, last, 64);
		if (lane == leader) {
			const uint32_t at = atomicAdd(&l_count, 1u);
			l_k[at] = k0; l_n[at] = (uint32_t)__popcll(mask); l_first[at] = v_first; l_last[at] = v_last;
		}
		todo &= ~mask;
	}
	__syncthreads();
	const uint32_t entries = l_count;
	if (threadIdx.x < entries) {
		const uint32_t i = threadIdx.x, k0 = l_k[i];
		bool first = true;
		for (uint32_t j = 0; j < i && first; ++j) first = l_k[j] != k0;
		if (first) {
			uint32_t n = l_n[i], lo = l_first[i], hi = l_last[i];
			for (uint32_t j = i + 1; j < entries; ++j) if (l_k[j] == k0) { n += l_n[j]; lo = min(lo, l_first[j]); hi = max(hi, l_last[j]); }
			atomicAdd(fresh + k0, n); atomicMin(vlo + k0, lo); atomicMax(vhi + k0, hi + 1u);
		}
	}
}

constexpr uint32_t kTiePairs = 1u << 22;   // capacity of the tie list (8 bytes each; the configs[3] mesh at 100 M triangles notes 0.9 M)
size_t components_workspace_bytes(uint32_t nv, uint32_t nf) { return ((size_t)3 * nf + nv + blocks_for(nf, kScanBlock) + 16) * 4 + (size_t)kTiePairs * 8 + kTieLists * 4 + 64; }
// where everything lies in that workspace -- the ONE place that knows (the driver, analysis.cpp, asks here)
ComponentsWorkspace components_workspace(void *ws, uint32_t nv, uint32_t nf)
{
	ComponentsWorkspace w;
	w.label = (uint32_t*)ws; w.flag = w.label + nf; w.num = w.flag + nf; w.sums = w.num + nf + 1;
	w.vfirst = w.sums + blocks_for(nf, kScanBlock) + 8;
	// the tie lists behind the vertex words, 8-byte aligned; their counters (kTieLists words) in front of them
	w.tie_count = (uint32_t*)(((uintptr_t)(w.vfirst + nv) + 7) & ~(uintptr_t)7);
	w.tie_pairs = w.tie_count + kTieLists;
	return w;
}
// stage 1: label[f] = root of f's component, num = exclusive scan of the root flags (num[nf] = number of components)
void launch_components_label(hipStream_t st, const ConnView &cv, const ComponentsWorkspace &w)
{
	uint32_t *label = w.label, *flag = w.flag, *num = w.num, *sums = w.sums;
	if (!cv.nf) return;
	const unsigned nb = blocks_for(cv.nf, kScanBlock);
	if (cv.ne) {
		hipLaunchKernelGGL(k_cc_hook_local, dim3(blocks_for(cv.nf, kHookFaces)), dim3(kHookThreads), 0, st, cv, label);
		hipLaunchKernelGGL(k_cc_hook, dim3(blocks_for(cv.nf, 256)), dim3(256), 0, st, cv, label);
	} else hipLaunchKernelGGL(k_cc_init, dim3(blocks_for(cv.nf, 256)), dim3(256), 0, st, label, cv.nf);
	hipLaunchKernelGGL(k_cc_flatten, dim3(blocks_for(cv.nf, 256)), dim3(256), 0, st, label, cv.nf);
	hipLaunchKernelGGL(k_cc_roots, dim3(blocks_for(cv.nf, 256)), dim3(256), 0, st, label, cv.nf, flag);
	hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(kScanBlock), 0, st, flag, cv.nf, sums);
	hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanBlock), 0, st, sums, nb);
	hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kScanBlock), 0, st, flag, cv.nf, sums, num);
}
// stage 2: per component (numbered in face order of the roots; tables zeroed / set to their neutral values by the caller)
void launch_components_faces(hipStream_t st, const ConnView &cv, uint32_t *label, const uint32_t *num, const uint32_t *spans, uint32_t nspans,
                             uint32_t *nfaces, uint32_t *nhe, uint32_t *flo, uint32_t *fhi, uint64_t *first_key)
{
	if (cv.nf) hipLaunchKernelGGL(k_cc_face_stats, dim3(blocks_for(cv.nf, 256)), dim3(256), 0, st, cv, label, num, spans, nspans, nfaces, nhe, flo, fhi, (unsigned long long*)first_key);
}
// stage 3: per component in coding order (rank_of: component number -> rank; vfirst 0xff-filled, tie[k] = k, fresh 0, vlo 0xff, vhi 0 by the caller)
void launch_components_vertices(hipStream_t st, const ConnView &cv, uint32_t nv, uint32_t ncomp, const ComponentsWorkspace &w, const uint32_t *rank_of,
                                uint32_t *tie, uint32_t *fresh, uint32_t *vlo, uint32_t *vhi)
{
	if (!cv.ne || !nv) return;
	const uint32_t *comp = w.label;
	uint32_t *vfirst = w.vfirst, *count = w.tie_count;
	uint2 *pairs = (uint2*)w.tie_pairs;
	(void)hipMemsetAsync(count, 0, kTieLists * 4, st);
	hipLaunchKernelGGL(k_cc_init, dim3(blocks_for(ncomp, 256)), dim3(256), 0, st, tie, ncomp);
	hipLaunchKernelGGL(k_cc_vertex_first, dim3(blocks_for(cv.nf, 256)), dim3(256), 0, st, cv, comp, rank_of, vfirst);
	hipLaunchKernelGGL(k_cc_vertex_ties, dim3(blocks_for(cv.nf, 256)), dim3(256), 0, st, cv, comp, rank_of, (const uint32_t*)vfirst, tie, pairs, kTiePairs, count);
	hipLaunchKernelGGL(k_cc_tie_pairs, dim3(blocks_for(kTiePairs, 256)), dim3(256), 0, st, (const uint2*)pairs, kTiePairs, (const uint32_t*)count, tie);
	hipLaunchKernelGGL(k_cc_flatten, dim3(blocks_for(ncomp, 256)), dim3(256), 0, st, tie, ncomp);
	hipLaunchKernelGGL(k_cc_vertex_stats, dim3(blocks_for(nv, 256)), dim3(256), 0, st, (const uint32_t*)vfirst, nv, fresh, vlo, vhi);
}

}   // namespace dev
}   // namespace hry
