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

}   // namespace dev
}   // namespace hry
