// Which record every element names, along the coding order -- on the device (round 5; the host's loop: host/general_events.cpp,
// which stays for the reference stream, whose symbols have positions in ONE sequence).  A mesh with general bindings (regions,
// shared records, corner lists: what the OBJ reader creates) says for every reference of a list: a new record (DATA: its residual
// bytes follow), one created earlier by its distance in creation order (HIST, GlobalHistory attrcode.h:23-53), or -- at a corner --
// one already named at this vertex by its distance in the vertex' own list of names (LHIST, LocalHistory :54-80);
// attrcode.h:321-393,395-416 without the values.
//
// The host walks the references one after the other with a table "record -> creation rank" and a list of names per (corner
// slot, vertex).  Both are answers to "who was FIRST", and first-of is a minimum over positions in the coding order:
//   k_ev_count / scan / k_ev_expand    the references of ONE list in coding order: element (half-edge of the vertex / face / corner),
//                                      slot, record.  Position p of a reference = exclusive scan of what every coded vertex / face
//                                      contributes (its region may bind the list at no slot, or at several)
//   k_ev_names (corner lists)          per (slot, vertex) a lock-free list of the records named there, each with the SMALLEST position
//                                      that names it (insert by compare-and-swap, atomic minimum on a hit)
//   k_ev_first                         a corner reference is answered by the vertex' names unless it is that smallest position; the
//                                      others -- and every reference of a vertex or face list -- go to the creation order: atomic
//                                      minimum of the position per record
//   k_ev_kind + three scans            DATA where the reference is its record's smallest position, HIST otherwise, LHIST as above;
//                                      the scans give every kind's place in its own output array
//   k_ev_data / k_ev_hist              DATA: record, element, slot, and the record's creation rank (= its place among the DATA);
//                                      HIST: records created before the reference - 1 - that rank; LHIST: names the vertex had got
//                                      after this one and before the reference (its list is newest first on the host)
// Same arrays as collect_events(), entry for entry (the containers' bytes are compared with the oracle's by every OBJ test).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>

#include "dev_types.hpp"
#include "kernels.hpp"

namespace hry {
namespace dev {

namespace {
constexpr uint32_t NONE = 0xffffffffu;
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a link (a list's head, a node's next): what it shows was written before it was published
__device__ __forceinline__ uint32_t ld_link(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

// slots of region r that lead to list l: visit(a) for each, in slot order
template <typename F> __device__ __forceinline__ void slots_of(const EvRegions &rg, int kind, int r, uint32_t l, F &&visit)
{
	const int32_t *off = kind == 0 ? rg.off_face : kind == 1 ? rg.off_vtx : rg.off_corner;
	const uint16_t *lists = kind == 0 ? rg.face_lists : kind == 1 ? rg.vtx_lists : rg.corner_lists;
	const int b = off[r], e = off[r + 1];
	for (int a = b; a < e; ++a) if (lists[a] == l) visit((uint32_t)(a - b));
}
__device__ __forceinline__ uint32_t face_of(const ConnView &cv, uint32_t h) { return cv.eface ? cv.eface[h] : h / cv.udeg; }

// ---- exclusive scan (block sums, scan of the sums by one block, apply), as twins.hip has it
constexpr int kScanBlock = 1024;
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &block_total)
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
__global__ __launch_bounds__(kScanBlock) void k_ev_scan_sums(const uint32_t *in, uint32_t n, uint32_t *sums)
{
	__shared__ uint32_t s_wave[17];
	const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
	uint32_t total;
	block_excl_scan(i < n ? in[i] : 0u, s_wave, total);
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanBlock) void k_ev_scan_top(uint32_t *sums, uint32_t nb)
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
// out[i] = exclusive prefix; *total_out = the sum (n >= 1)
__global__ __launch_bounds__(kScanBlock) void k_ev_scan_apply(const uint32_t *in, uint32_t n, const uint32_t *sums, uint32_t *out, uint32_t *total_out)
{
	__shared__ uint32_t s_wave[17];
	const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
	const uint32_t v = i < n ? in[i] : 0u;
	uint32_t total;
	const uint32_t ex = block_excl_scan(v, s_wave, total) + sums[blockIdx.x];
	if (i < n) out[i] = ex;
	if (i == n - 1) *total_out = ex + v;
}
void exclusive_scan(hipStream_t st, const uint32_t *in, uint32_t n, uint32_t *sums, uint32_t *out, uint32_t *total_out)
{
	if (!n) { (void)hipMemsetAsync(total_out, 0, 4, st); return; }
	const unsigned nb = (n + kScanBlock - 1) / kScanBlock;
	hipLaunchKernelGGL(k_ev_scan_sums, dim3(nb), dim3(kScanBlock), 0, st, in, n, sums);
	hipLaunchKernelGGL(k_ev_scan_top, dim3(1), dim3(kScanBlock), 0, st, sums, nb);
	hipLaunchKernelGGL(k_ev_scan_apply, dim3(nb), dim3(kScanBlock), 0, st, in, n, (const uint32_t*)sums, out, total_out);
}

// ---- the references of one list ----------------------------------------------------------------------------------------------------
// kind: 0 face list, 1 vertex list, 2 corner list.  order: order_v (kind 1) or order_f; one thread per coded vertex / face
__global__ __launch_bounds__(256) void k_ev_count(int kind, uint32_t list, ConnView cv, GenView gv, EvRegions rg, const uint32_t *order, uint32_t n, uint32_t *cnt)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t h = order[i];
	uint32_t c = 0;
	if (kind == 1) slots_of(rg, 1, gv.vtx_reg[cv.org[h]], list, [&](uint32_t) { ++c; });
	else {
		const uint32_t f = face_of(cv, h);
		slots_of(rg, kind, gv.face_reg[f], list, [&](uint32_t) { ++c; });
		if (kind == 2) c *= cv.eface ? cv.foff[f + 1] - cv.foff[f] : cv.udeg;
	}
	cnt[i] = c;
}
// r_q (corner lists): the reference's place among the corner references of ALL lists -- the names of a vertex are kept per corner
// SLOT, whatever list a region binds there (LocalHistory, attrcode.h:54-80: two regions may bind different lists at one slot, and
// a record number named for one answers a reference to the same number of the other); corner_base: that place for a face's first
// corner references of every list that a coded face brings: corners x the corner slots of its region
__global__ __launch_bounds__(256) void k_ev_corner_count(ConnView cv, GenView gv, EvRegions rg, const uint32_t *order_f, uint32_t n, uint32_t *cnt)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t f = face_of(cv, order_f[i]);
	const int r = gv.face_reg[f];
	cnt[i] = (uint32_t)(rg.off_corner[r + 1] - rg.off_corner[r]) * (cv.eface ? cv.foff[f + 1] - cv.foff[f] : cv.udeg);
}
__global__ __launch_bounds__(256) void k_ev_expand(int kind, uint32_t list, uint32_t list_count, ConnView cv, GenView gv, EvRegions rg, const uint32_t *order, uint32_t n,
                                                    const uint32_t *base, const uint32_t *corner_base, uint32_t *r_elem, uint8_t *r_slot, uint32_t *r_idx, uint32_t *r_q, uint32_t *err)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t h = order[i];
	uint32_t p = base[i];
	bool bad = false;
	uint32_t q = 0;
	auto put = [&](uint32_t elem, uint32_t a, uint32_t idx) {   // (a record outside its list: noted for the host, and kept off the tables)
		r_elem[p] = elem; r_slot[p] = (uint8_t)a; r_idx[p] = idx < list_count ? idx : 0u; bad |= idx >= list_count;
		if (kind == 2) r_q[p] = q + a;
		++p;
	};
	if (kind == 1) {
		const uint32_t v = cv.org[h];
		slots_of(rg, 1, gv.vtx_reg[v], list, [&](uint32_t a) { put(h, a, gv.vtx_attr[(size_t)v * gv.nb_vtx + a]); });
	} else {
		const uint32_t f = face_of(cv, h);
		const int r = gv.face_reg[f];
		if (kind == 0) slots_of(rg, 0, r, list, [&](uint32_t a) { put(f, a, rg.face_attr[(size_t)f * rg.nb_face + a]); });
		else {
			const uint32_t fb = cv.eface ? cv.foff[f] : f * cv.udeg, fe = cv.eface ? cv.foff[f + 1] : fb + cv.udeg;
			const uint32_t nca = (uint32_t)(rg.off_corner[r + 1] - rg.off_corner[r]);
			q = corner_base[i];
			uint32_t c = h;   // the corners from the face's coded half-edge on, round the face (attrcode.h:395-416)
			do {
				slots_of(rg, 2, r, list, [&](uint32_t a) { put(c, a, gv.corner_attr[(size_t)c * gv.nb_corner + a]); });
				q += nca;
				c = c + 1 == fe ? fb : c + 1;
			} while (c != h);
		}
	}
	if (bad) atomicOr(err, 1u);
}

// ---- the names a vertex has got in a corner slot: head[slot * nv + v] -> nodes (record, smallest naming position, next) ------------
constexpr uint32_t kMaxNames = 1u << 14;   // names of one vertex at one slot that a reference may walk past on the device
struct Names { uint32_t *head, *n_idx, *n_pos, *n_next, *n_nodes, *err; uint32_t nv; };
__global__ __launch_bounds__(256) void k_ev_names(ConnView cv, const uint32_t *r_elem, const uint8_t *r_slot, const uint32_t *r_idx, const uint32_t *r_q, const uint32_t *n_ptr, Names nm)
{
	const uint32_t ref = blockIdx.x * blockDim.x + threadIdx.x;
	if (ref >= *n_ptr) return;
	const uint32_t p = r_q[ref];   // (the place among the corner references of every list)
	const uint32_t idx = r_idx[ref];
	uint32_t *hd = nm.head + (size_t)r_slot[ref] * nm.nv + cv.org[r_elem[ref]];
	uint32_t mine = NONE, seen_upto = NONE;   // mine: the node this thread has filled but not linked; seen_upto: the head whose list has been searched
	uint32_t steps = 0;
	for (;;) {
		const uint32_t first = ld_link(hd);
		// search the nodes in front of what has been searched already
		for (uint32_t k = first; k != seen_upto && k != NONE; k = ld_link(nm.n_next + k)) {
			if (ld_u32(nm.n_idx + k) == idx) { atomicMin(nm.n_pos + k, p); return; }   // (a node this thread filled in vain stays out of every list)
			// a vertex with tens of thousands of different records at one slot (a hub): every reference walks its whole list, here as
			// on the host -- but a host thread is not a wavefront that the others of its kernel wait for.  The host takes such a mesh
			if (++steps > kMaxNames) { atomicOr(nm.err, 4u); return; }
		}
		seen_upto = first;
		if (mine == NONE) {
			mine = atomicAdd(nm.n_nodes, 1u);
			// (atomic stores: another thread's atomicMin may meet n_pos as soon as the link below shows the node)
			__hip_atomic_store(nm.n_idx + mine, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(nm.n_pos + mine, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__hip_atomic_store(nm.n_next + mine, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__threadfence();   // the node's fields before the link that shows it (release; the readers' ld_link is the acquire)
		if (atomicCAS(hd, first, mine) == first) return;
	}
}
// is_first[p] (corner lists: not answered by the vertex' names), and the record's smallest position among those
__global__ __launch_bounds__(256) void k_ev_first(int kind, ConnView cv, const uint32_t *r_elem, const uint8_t *r_slot, const uint32_t *r_idx, const uint32_t *r_q, const uint32_t *n_ptr, Names nm,
                                                   uint8_t *is_first, uint32_t *first_pos)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= *n_ptr) return;
	const uint32_t idx = r_idx[p];
	bool first = true;
	if (kind == 2 && (ld_u32(nm.err) & 4u)) return;   // (a hub was met: the host's loop takes the mesh, nobody walks the hub's list again)
	if (kind == 2) {
		uint32_t k = nm.head[(size_t)r_slot[p] * nm.nv + cv.org[r_elem[p]]];
		while (k != NONE && nm.n_idx[k] != idx) k = nm.n_next[k];
		first = k != NONE && nm.n_pos[k] == r_q[p];
	}
	is_first[p] = first ? 1 : 0;
	if (first) atomicMin(first_pos + idx, p);
}
__global__ __launch_bounds__(256) void k_ev_kind(const uint32_t *r_idx, const uint8_t *is_first, const uint32_t *first_pos, const uint32_t *n_ptr, uint32_t n_max, uint8_t *kind_out,
                                                  uint32_t *f_data, uint32_t *f_hist, uint32_t *f_lhist)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_max) return;
	if (p >= *n_ptr) { f_data[p] = 0; f_hist[p] = 0; f_lhist[p] = 0; return; }   // (the scans run over n_max)
	const uint32_t k = !is_first[p] ? 2u : first_pos[r_idx[p]] == p ? 0u : 1u;   // RefKind: data 0, hist 1, lhist 2 (io.h:95-98)
	kind_out[p] = (uint8_t)k;
	f_data[p] = k == 0u; f_hist[p] = k == 1u; f_lhist[p] = k == 2u;
}
__global__ __launch_bounds__(256) void k_ev_data(const uint8_t *kind, const uint32_t *at_data, const uint32_t *r_elem, const uint8_t *r_slot, const uint32_t *r_idx, const uint32_t *n_ptr,
                                                  uint32_t *d_idx, uint32_t *d_he, uint8_t *d_slot, uint32_t *rank_of)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= *n_ptr || kind[p] != 0) return;
	const uint32_t j = at_data[p];
	d_idx[j] = r_idx[p]; d_he[j] = r_elem[p]; d_slot[j] = r_slot[p];
	rank_of[r_idx[p]] = j;   // GlobalHistory::tidxlist: the record's place in the creation order
}
__global__ __launch_bounds__(256) void k_ev_hist(ConnView cv, const uint8_t *kind, const uint32_t *at_data, const uint32_t *at_hist, const uint32_t *at_lhist,
                                                  const uint32_t *r_elem, const uint8_t *r_slot, const uint32_t *r_idx, const uint32_t *r_q, const uint32_t *n_ptr, const uint32_t *rank_of, Names nm,
                                                  uint32_t *gh_val, uint32_t *lh_val, uint32_t *err)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= *n_ptr || kind[p] == 0) return;
	const uint32_t idx = r_idx[p];
	if (kind[p] == 1) { gh_val[at_hist[p]] = at_data[p] - 1u - rank_of[idx]; return; }   // records created so far - 1 - the record's rank
	if (ld_u32(nm.err) & 4u) return;   // (a hub: see k_ev_first)
	// the vertex' names are kept newest first: the distance is the number of names it got after this record's and before now
	uint32_t before_now = 0, mine_pos = NONE;
	const uint32_t h0 = nm.head[(size_t)r_slot[p] * nm.nv + cv.org[r_elem[p]]], q = r_q[p];
	for (uint32_t k = h0; k != NONE; k = nm.n_next[k]) { before_now += nm.n_pos[k] < q; if (nm.n_idx[k] == idx) mine_pos = nm.n_pos[k]; }
	uint32_t before_mine = 0;
	for (uint32_t k = h0; k != NONE; k = nm.n_next[k]) before_mine += nm.n_pos[k] < mine_pos;
	const uint32_t back = before_now - 1u - before_mine;
	if (back > 0xffffu) atomicOr(err, 2u);
	lh_val[at_lhist[p]] = back;
}
// the region of every coded vertex / face, where regions are coded
__global__ __launch_bounds__(256) void k_ev_regions(int kind, ConnView cv, GenView gv, const uint32_t *order, uint32_t n, uint8_t *out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t h = order[i];
	out[i] = (uint8_t)(kind == 1 ? gv.vtx_reg[cv.org[h]] : gv.face_reg[face_of(cv, h)]);
}
inline unsigned blocks(uint32_t n) { return (n + 255u) / 256u; }
}   // namespace

namespace {
struct ListWs {   // one list's arrays between the two phases
	uint32_t *cnt, *base, *sums, *r_elem, *r_idx, *r_q, *f_data, *f_hist, *f_lhist, *at_data, *at_hist, *at_lhist, *first_pos, *rank_of;
	uint8_t *r_slot, *is_first;
};
size_t sums_words(uint32_t n_order, uint32_t max_refs) { return (std::max<size_t>(max_refs, n_order) + kScanBlock - 1) / kScanBlock + 8; }
ListWs carve(void *ws, uint32_t n_order, uint32_t max_refs, uint32_t list_count)
{
	const size_t r = max_refs;
	uint32_t *w = (uint32_t*)ws;
	ListWs L;
	L.cnt = w; w += (size_t)n_order + 1;
	L.base = w; w += (size_t)n_order + 1;
	L.sums = w; w += sums_words(n_order, max_refs);
	L.r_elem = w; w += r; L.r_idx = w; w += r; L.r_q = w; w += r;
	L.f_data = w; w += r; L.f_hist = w; w += r; L.f_lhist = w; w += r;
	L.at_data = w; w += r; L.at_hist = w; w += r; L.at_lhist = w; w += r;
	L.first_pos = w; w += (size_t)list_count + 1;
	L.rank_of = w; w += (size_t)list_count + 1;
	L.r_slot = (uint8_t*)w;
	L.is_first = L.r_slot + r;
	return L;
}
Names names_of(void *ws, uint32_t fc, uint32_t corner_refs_max, uint32_t head_words, uint32_t nv, uint32_t **corner_base, uint32_t **corner_cnt, uint32_t **sums)
{
	uint32_t *w = (uint32_t*)ws;
	Names nm{};
	*corner_cnt = w; w += (size_t)fc + 1;
	*corner_base = w; w += (size_t)fc + 1;
	*sums = w; w += sums_words(fc, 0);
	nm.head = w; w += head_words;
	nm.n_idx = w; w += corner_refs_max; nm.n_pos = w; w += corner_refs_max; nm.n_next = w; w += corner_refs_max;
	nm.n_nodes = w; w += 4;
	nm.nv = nv;
	return nm;
}
}   // namespace

size_t events_list_workspace_bytes(uint32_t n_order, uint32_t max_refs, uint32_t list_count)
{
	return ((size_t)2 * n_order + 2 + sums_words(n_order, max_refs) + 9 * (size_t)max_refs + 2 * ((size_t)list_count + 1) + 16) * 4 + 2 * (size_t)max_refs + 64;
}
// the names of the vertices at the corner slots (shared by every corner list) + the corner references' places: fc coded faces, at
// most corner_refs_max corner references of all lists together, head_words = corner slots x vertices
size_t events_names_workspace_bytes(uint32_t fc, uint32_t corner_refs_max, uint32_t head_words)
{
	return ((size_t)2 * fc + 2 + sums_words(fc, 0) + head_words + 3 * (size_t)corner_refs_max + 16) * 4;
}
void launch_corner_places(hipStream_t st, const ConnView &cv, const GenView &gv, const EvRegions &rg, const uint32_t *order_f, uint32_t fc, uint32_t corner_refs_max,
                          uint32_t head_words, uint32_t nv, void *names_ws)
{
	uint32_t *corner_base, *corner_cnt, *sums;
	Names nm = names_of(names_ws, fc, corner_refs_max, head_words, nv, &corner_base, &corner_cnt, &sums);
	(void)hipMemsetAsync(nm.head, 0xff, (size_t)head_words * 4, st);
	(void)hipMemsetAsync(nm.n_nodes, 0, 16, st);
	if (!fc) return;
	hipLaunchKernelGGL(k_ev_corner_count, dim3(blocks(fc)), dim3(256), 0, st, cv, gv, rg, order_f, fc, corner_cnt);
	exclusive_scan(st, corner_cnt, fc, sums, corner_base, nm.n_nodes + 1);   // (the total: not used)
}
// phase 1 of a list: its references in coding order (counts[0] = how many); a corner list also enters them in the vertices' names
void launch_list_refs(hipStream_t st, int kind, uint32_t list, uint32_t list_count, const ConnView &cv, const GenView &gv, const EvRegions &rg,
                      const uint32_t *order, uint32_t n_order, uint32_t max_refs, uint32_t nv, uint32_t fc, uint32_t corner_refs_max, uint32_t head_words, void *names_ws, void *list_ws,
                      uint32_t *counts, uint32_t *err)
{
	(void)hipMemsetAsync(counts, 0, 16, st);
	if (!n_order || !max_refs) return;
	const ListWs L = carve(list_ws, n_order, max_refs, list_count);
	uint32_t *corner_base, *corner_cnt, *sums;
	Names nm = names_of(names_ws, fc, corner_refs_max, head_words, nv, &corner_base, &corner_cnt, &sums);
	nm.err = err;
	hipLaunchKernelGGL(k_ev_count, dim3(blocks(n_order)), dim3(256), 0, st, kind, list, cv, gv, rg, order, n_order, L.cnt);
	exclusive_scan(st, L.cnt, n_order, L.sums, L.base, counts + 0);
	hipLaunchKernelGGL(k_ev_expand, dim3(blocks(n_order)), dim3(256), 0, st, kind, list, list_count, cv, gv, rg, order, n_order, (const uint32_t*)L.base, (const uint32_t*)corner_base,
	                   L.r_elem, L.r_slot, L.r_idx, L.r_q, err);
	(void)hipMemsetAsync(L.first_pos, 0xff, ((size_t)list_count + 1) * 4, st);
	// (the kernels over references start max_refs threads -- an upper bound the host knows -- and stop at the references there are:
	// their number stays on the device, counts[0])
	if (kind == 2)
		hipLaunchKernelGGL(k_ev_names, dim3(blocks(max_refs)), dim3(256), 0, st, cv, (const uint32_t*)L.r_elem, (const uint8_t*)L.r_slot, (const uint32_t*)L.r_idx, (const uint32_t*)L.r_q,
		                   (const uint32_t*)counts, nm);
}
// phase 2 (every corner list has been through phase 1): the kinds (type_sym, one byte a reference: the list's first plane as it
// is), creation-order distances, per-vertex distances and the records coded as data, all in HBM; counts[1..3] = HIST, LHIST, DATA
void launch_list_kinds(hipStream_t st, int kind, uint32_t list_count, const ConnView &cv, uint32_t n_order, uint32_t max_refs, uint32_t nv, uint32_t fc, uint32_t corner_refs_max,
                       uint32_t head_words, void *names_ws, void *list_ws, uint8_t *type_sym, uint32_t *gh_val, uint32_t *lh_val, uint32_t *d_idx, uint32_t *d_he, uint8_t *d_slot,
                       uint32_t *counts, uint32_t *err)
{
	if (!n_order || !max_refs) return;
	const ListWs L = carve(list_ws, n_order, max_refs, list_count);
	uint32_t *corner_base, *corner_cnt, *sums;
	Names nm = names_of(names_ws, fc, corner_refs_max, head_words, nv, &corner_base, &corner_cnt, &sums);
	nm.err = err;
	const uint32_t *n_refs = counts + 0;
	const uint32_t nr = max_refs;
	hipLaunchKernelGGL(k_ev_first, dim3(blocks(nr)), dim3(256), 0, st, kind, cv, (const uint32_t*)L.r_elem, (const uint8_t*)L.r_slot, (const uint32_t*)L.r_idx, (const uint32_t*)L.r_q, n_refs, nm,
	                   L.is_first, L.first_pos);
	hipLaunchKernelGGL(k_ev_kind, dim3(blocks(nr)), dim3(256), 0, st, (const uint32_t*)L.r_idx, (const uint8_t*)L.is_first, (const uint32_t*)L.first_pos, n_refs, nr, type_sym, L.f_data, L.f_hist, L.f_lhist);
	exclusive_scan(st, L.f_hist, nr, L.sums, L.at_hist, counts + 1);
	exclusive_scan(st, L.f_lhist, nr, L.sums, L.at_lhist, counts + 2);
	exclusive_scan(st, L.f_data, nr, L.sums, L.at_data, counts + 3);
	hipLaunchKernelGGL(k_ev_data, dim3(blocks(nr)), dim3(256), 0, st, (const uint8_t*)type_sym, (const uint32_t*)L.at_data, (const uint32_t*)L.r_elem, (const uint8_t*)L.r_slot, (const uint32_t*)L.r_idx, n_refs,
	                   d_idx, d_he, d_slot, L.rank_of);
	hipLaunchKernelGGL(k_ev_hist, dim3(blocks(nr)), dim3(256), 0, st, cv, (const uint8_t*)type_sym, (const uint32_t*)L.at_data, (const uint32_t*)L.at_hist, (const uint32_t*)L.at_lhist,
	                   (const uint32_t*)L.r_elem, (const uint8_t*)L.r_slot, (const uint32_t*)L.r_idx, (const uint32_t*)L.r_q, n_refs, (const uint32_t*)L.rank_of, nm, gh_val, lh_val, err);
}
void launch_region_symbols(hipStream_t st, int kind, const ConnView &cv, const GenView &gv, const uint32_t *order, uint32_t n, uint8_t *out)
{
	if (n) hipLaunchKernelGGL(k_ev_regions, dim3(blocks(n)), dim3(256), 0, st, kind, cv, gv, order, n, out);
}

}   // namespace dev
}   // namespace hry
