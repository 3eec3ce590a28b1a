// Decode pipeline of the chunked profile (.hry v0.2):
//   H2D payload -> k_chunk_decode (one wavefront per stream) -> symbol planes; connectivity streams first
//   D2H connectivity planes -> host cut-border replay (cbm_replay.hpp), overlapping the attribute streams' decode
//   H2D connectivity -> k_candidates, k_residuals_to_rec, k_faces_unfold, k_unpredict -> attribute records -> D2H
// Reference: formats/hry/reader.cc:179-193, cbm/decoder.h:27-211, attrcode.h:533-550.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <thread>

#include "../host/cbm_replay.hpp"
#include "context.hpp"
#include "kernels.hpp"

namespace hry {

using namespace dev;
typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
// HRY_TRACE=1: wall-clock marks of the decode pipeline on stderr (development aid)
static bool trace_on() { static const bool on = getenv("HRY_TRACE") != nullptr; return on; }
static thread_local Clock::time_point g_t0;
#define HRY_MARK(t0, what) do { if (trace_on()) fprintf(stderr, "[hry] %8.3f ms  %s\n", ms_since(t0), what); } while (0)

namespace dev {
void launch_residuals_to_rec(hipStream_t st, const uint8_t *planes, uint32_t n, const ListDesc &ld, uint8_t *rec);
void launch_faces_unfold(hipStream_t st, uint32_t n, const ListDesc &ld, uint8_t *rec);
bool unpredict2_applicable(const ListDesc &ld);
void launch_chain_records(hipStream_t st, const uint32_t *cand, const uint8_t *ncand, uint32_t nvtx, const uint32_t *seg_start, uint32_t nseg, void *crec);
bool unpredict3_wanted(const ListDesc &ld);
void launch_unpredict2(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t *cand, uint8_t *ncand, const void *crec,
                       const uint8_t *planes, const ListDesc &ld, uint8_t *rec, const uint32_t *segs, const uint32_t *list_off, uint32_t n_lists,
                       const uint32_t *seg_start, uint32_t nseg, uint32_t *done);
void launch_candidates_ids(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t *cand, uint8_t *ncand);
bool unpredict3_covers(const ListDesc &ld);
void launch_slice_prepare(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t v_begin, uint32_t v_end, uint32_t *cand, uint8_t *ncand, void *crec);
size_t cand_table_words(uint32_t nvtx);
void cand_table_reset(hipStream_t st, uint32_t *cand, uint32_t nvtx);
void launch_slice_chain(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t nvtx, uint32_t v_begin, uint32_t v_end, const uint32_t *cand, const uint8_t *ncand,
                        const void *crec, const uint8_t *planes, const ListDesc &ld, uint8_t *rec);
void launch_scatter_u32(hipStream_t st, const uint32_t *pairs, uint32_t n, uint32_t *dst);
uint32_t chain_timeout_flags(hipStream_t st, const uint32_t *gave_up = nullptr);
}

static const int kConnPlanes = 21;

static int conn_init_kind(int i) { return i == 0 ? INIT_IOP : i == 11 ? INIT_NT0 : i == 12 ? INIT_NT1 : i >= 13 ? INIT_OP : INIT_ONES; }

// Work lists of the chain kernel (k_unpredict2: one wavefront per list and attribute component walks the list's components one
// after the other).  A component is a list of its own -- unless it is tiny: the slivers that non-manifold edges and vertices
// split off are components of a triangle or two, 19 000 of them in one GPU's share of configs[3] beside its 128 real ones, and
// a workgroup each made 57 000 workgroups of 384 chains' work: 13.7 ms of chains against 12.2 with consecutive tiny components
// sharing a list, up to kTinyRun of them, and 10.7 with the owners a chain has waited for remembered (unpredict.hip: wait_owner;
// 8.5 ms without any slivers -- scripts/chain_slivers.py).  triples: (first vertex, end, component) per component that owns vertices; starts: index of every list's first triple.
static void group_chain_lists(const uint32_t *triples, uint32_t n, uint32_t first_index, std::vector<uint32_t> &starts)
{
	constexpr uint32_t kTiny = 64, kTinyRun = 64;
	static const bool merge = getenv("HRY_CHAIN_NO_LIST_MERGE") == nullptr;
	uint32_t run = 0;   // tiny components in the list at hand (0: the next component starts a list)
	for (uint32_t i = 0; i < n; ++i) {
		const bool tiny = merge && triples[3 * i + 1] - triples[3 * i] < kTiny;
		if (!tiny || run == 0 || run >= kTinyRun) { starts.push_back(first_index + i); run = 0; }
		run = tiny ? run + 1 : 0;
	}
	starts.push_back(first_index + n);
}

// The float / 32-bit reconstruction chains of a mesh of many components, launched in BATCHES of whole components (round 4): a
// component's chain needs its connectivity on the device, the residual planes decoded, and the components before it launched --
// not the end of the replay.  The tables the chains read (lists, every component's first vertex, the progress words) are laid out
// for the most components the stream can hold and filled batch by batch; a component not yet known starts at 0xffffffff, which
// the owner search of a waiting chain (wait_owner) simply never picks.
struct ChainBatches {
	Context &cx;
	const ListDesc ldv;
	const uint8_t *d_vplanes;
	uint32_t nvc, ncomp_max;
	uint32_t comps_done = 0, v_done = 0, lists_done = 0, triples_done = 0;
	bool conn_adopted = false;             // the caller has set the context's view of the connectivity (no adopt_conn, which waits for the stream)
	uint32_t v_first_batch = 0;            // vertices of the batches launched beside the replay
	int n_timed = 0;                       // batches bracketed by cx.chain_ev pairs (elapsed_ms after the streams have drained)
	hipEvent_t first_done = nullptr;       // ... and the event behind them on their stream (nullptr: none)
	uint32_t *d_lists = nullptr, *d_off = nullptr, *d_segstart = nullptr, *d_flags = nullptr, *d_cand = nullptr;
	uint8_t *d_ncand = nullptr;
	std::deque<std::vector<uint32_t>> held;   // host tables on their way to the device
	ChainBatches(Context &c, const ListDesc &ld, const uint8_t *planes, uint32_t n_vertices, uint32_t max_components)
	    : cx(c), ldv(ld), d_vplanes(planes), nvc(n_vertices), ncomp_max(std::max(1u, max_components)) {}
	// everything that does not depend on the connectivity, on stream st (the first batch's stream)
	void init(hipStream_t st, Mesh &m)
	{
		for (size_t l = 0; l < m.lists.size() && l < 2; ++l) {
			cx.d_rec[l].ensure(std::max<size_t>(m.lists[l].data.size(), 16));
			if (!m.lists[l].data.empty()) HIP_OK(hipMemsetAsync(cx.d_rec[l].p, 0, m.lists[l].data.size(), st));
		}
		const size_t ncand_bytes = ((size_t)nvc + 63) & ~(size_t)63;
		const size_t cand_words = (cand_table_words(nvc) + 3) & ~(size_t)3;
		cx.d_cscratch.ensure(std::max<size_t>(cand_words * 4 + ncand_bytes + 64, 16));
		d_cand = cx.d_cscratch.as<uint32_t>();
		d_ncand = (uint8_t*)(d_cand + cand_words);
		cand_table_reset(st, d_cand, nvc);
		const size_t words = (size_t)3 * ncomp_max + ((size_t)ncomp_max + 1) + ((size_t)ncomp_max + 1) + (size_t)ldv.ncomp * ncomp_max + 1;   // (+ the give-up word behind the flags)
		cx.d_small.ensure(words * 4 + 64);
		d_lists = cx.d_small.as<uint32_t>();
		d_off = d_lists + (size_t)3 * ncomp_max;
		d_segstart = d_off + ncomp_max + 1;
		d_flags = d_segstart + ncomp_max + 1;
		HIP_OK(hipMemsetAsync(d_segstart, 0xff, ((size_t)ncomp_max + 1) * 4, st));
		HIP_OK(hipMemsetAsync(d_flags, 0, ((size_t)ldv.ncomp * ncomp_max + 1) * 4, st));
	}
	// components [comps_done, comps_done + n): first[i] = first vertex of component comps_done + i, v_end = first vertex behind them
	void launch(hipStream_t st, const uint32_t *first, uint32_t n, uint32_t v_end)
	{
		if (!n) return;
		if ((uint64_t)comps_done + n > ncomp_max || v_end > nvc || v_end < v_done) throw Error(HRY_E_FORMAT, "corrupt stream (more components than start symbols)");
		held.emplace_back();
		std::vector<uint32_t> &t = held.back();
		// lists (first vertex, end, component) of the components that own vertices, then their offsets, then the first vertices
		uint32_t nl = 0;
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t b = first[i], e = i + 1 < n ? first[i + 1] : v_end;
			if (b != e) { t.push_back(b); t.push_back(e); t.push_back(comps_done + i); ++nl; }
		}
		std::vector<uint32_t> starts;
		group_chain_lists(t.data(), nl, triples_done, starts);
		const uint32_t n_lists = (uint32_t)starts.size() - 1;
		const size_t off_at = t.size();
		t.insert(t.end(), starts.begin(), starts.end());
		const size_t seg_at = t.size();
		t.insert(t.end(), first, first + n);
		t.push_back(v_end);   // (the next batch overwrites it with its first component's start: the same number)
		if (nl) HIP_OK(hipMemcpyAsync(d_lists + (size_t)3 * triples_done, t.data(), (size_t)3 * nl * 4, hipMemcpyHostToDevice, st));
		HIP_OK(hipMemcpyAsync(d_off + lists_done, t.data() + off_at, ((size_t)n_lists + 1) * 4, hipMemcpyHostToDevice, st));
		HIP_OK(hipMemcpyAsync(d_segstart + comps_done, t.data() + seg_at, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st));
		const ConnView cv = cx.conn_view();
		launch_slice_prepare(st, cv, cx.d_order_v.as<uint32_t>(), nvc, v_done, v_end, d_cand, d_ncand, nullptr);
		const bool timed = n_timed < Context::kChainBatchEvents;
		if (timed) {
			for (int k = 0; k < 2; ++k) if (!cx.chain_ev[2 * n_timed + k]) HIP_OK(hipEventCreate(&cx.chain_ev[2 * n_timed + k]));
			HIP_OK(hipEventRecord(cx.chain_ev[2 * n_timed], st));
		}
		for (uint32_t done = 0; done < n_lists; done += 65535)   // a launch holds at most 65535 x 8 lists' worth of workgroups
			launch_unpredict2(st, cv, cx.d_order_v.as<uint32_t>(), nvc, d_cand, d_ncand, nullptr, d_vplanes, ldv, cx.d_rec[1].as<uint8_t>(),
			                  d_lists, d_off + lists_done + done, std::min(65535u, n_lists - done), d_segstart, ncomp_max, d_flags);
		if (timed) { HIP_OK(hipEventRecord(cx.chain_ev[2 * n_timed + 1], st)); ++n_timed; }
		comps_done += n; lists_done += n_lists; triples_done += nl; v_done = v_end;
	}
	// the chain kernels' time over every batch of the decode (the batches' streams have drained)
	double elapsed_ms() const
	{
		double sum = 0;
		for (int i = 0; i < n_timed; ++i) {
			float ms = 0;
			if (hipEventElapsedTime(&ms, cx.chain_ev[2 * i], cx.chain_ev[2 * i + 1]) == hipSuccess) sum += ms;
		}
		return sum;
	}
};

// Attribute reconstruction on the device, shared by both formats: connectivity + decode order + residual byte planes
// (already in HBM) -> attribute records.  Events 3/4 bracket the kernels.
// conn_from: the connectivity (face offsets, origins, twins) is read from that mesh instead -- one that another thread may be
// READING at the same time (general_planes_decode: the host's bookkeeping runs beside the vertex chain)
static void reconstruct_attributes(Context &cx, Mesh &mesh, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                                   const std::vector<uint32_t> &seg_level, const uint8_t *d_vplanes, const uint8_t *d_fplanes,
                                   const ListDesc &ldv, const ListDesc &ldf, bool conn_resident = false, const Mesh *conn_from = nullptr, struct ChainBatches *batches = nullptr);

static void reconstruct_attributes(Context &cx, Mesh &mesh, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                                   const std::vector<uint32_t> &seg_level, const uint8_t *d_vplanes, const uint8_t *d_fplanes,
                                   const ListDesc &ldv, const ListDesc &ldf, bool conn_resident, const Mesh *conn_from, struct ChainBatches *batches)
{
	Mesh *m = &mesh;
	const uint32_t nvc = (uint32_t)order_v.size();
	bool chain_timed = false;
	size_t early_bytes = 0;
	const uint32_t *d_gave_up = nullptr;   // the chains' give-up word of this decode (behind their flag table)
	HRY_MARK(g_t0, "reconstruct: begin");
	// connectivity up (unless it went up beside the replay: SpanUploader); the records are born on the device (zeroed there:
	// uploading the host's zeros was a fifth of this copy)
	if (batches && !conn_resident) throw Error(HRY_E_INTERNAL, "chains were launched beside the replay, but its connectivity did not reach the device");
	if (conn_resident && batches && batches->conn_adopted) {}   // (the uploader completed the device's view of the connectivity, without a wait)
	else if (conn_resident) cx.adopt_conn(*m);
	else if (conn_from) {
		if (conn_from->twins_pending || conn_from->partial) throw Error(HRY_E_INTERNAL, "lent connectivity must be complete");
		cx.upload_mesh(const_cast<Mesh&>(*conn_from), false);   // (a mesh whose twins are matched is only read)
	} else cx.upload_mesh(*m, false);
	for (size_t l = 0; l < m->lists.size() && l < 2 && !batches; ++l) {   // (a run in batches zeroed them before its first batch)
		cx.d_rec[l].ensure(std::max<size_t>(m->lists[l].data.size(), 16));
		if (!m->lists[l].data.empty()) HIP_OK(hipMemsetAsync(cx.d_rec[l].p, 0, m->lists[l].data.size(), cx.stream));
	}
	HRY_MARK(g_t0, "connectivity uploaded");
	ConnView cv = cx.conn_view();
	cx.d_order_v.ensure(std::max<size_t>((size_t)nvc * 4, 16));
	if (nvc && !conn_resident) HIP_OK(hipMemcpyAsync(cx.d_order_v.p, order_v.data(), (size_t)nvc * 4, hipMemcpyHostToDevice, cx.stream));
	const size_t ncand_bytes = ((size_t)nvc + 63) & ~(size_t)63;
	const size_t cand_words = (cand_table_words(nvc) + 3) & ~(size_t)3;
	if (!batches) cx.d_cscratch.ensure(std::max<size_t>(cand_words * 4 + (size_t)nvc * 16 + ncand_bytes + 64, 16));
	uint32_t *d_cand = cx.d_cscratch.as<uint32_t>();
	uint8_t *d_ncand = (uint8_t*)(d_cand + cand_words);
	void *d_crec = d_ncand + ncand_bytes;   // 16-byte chain records (k_unpredict3), 16-byte aligned
	HIP_OK(hipEventRecord(cx.ev[3], cx.stream));
	if (ldv.nplanes && batches) {
		// the components the batches beside the replay have not taken: one more batch, behind them
		const uint32_t nseg = (uint32_t)seg_start.size() - 1;
		if (batches->comps_done > nseg || nvc != batches->nvc) throw Error(HRY_E_FORMAT, "corrupt stream (components)");
		HIP_OK(hipEventRecord(cx.ev[7], cx.stream));
		batches->launch(cx.stream, seg_start.data() + batches->comps_done, nseg - batches->comps_done, nvc);
		HIP_OK(hipEventRecord(cx.ev[0], cx.stream));
		chain_timed = true;
		// the records of the batches that ran beside the replay come down while the last batch runs (their chains are done
		// when the event behind them is reached; vertex v's record is record v: vertex ids are handed out in decode order)
		if (batches->first_done && cx.stream3 && batches->v_first_batch && !m->lists[1].data.empty()) {
			const size_t bytes = (size_t)batches->v_first_batch * m->lists[1].stride();
			HIP_OK(hipStreamWaitEvent(cx.stream3, batches->first_done, 0));
			HIP_OK(hipMemcpyAsync(m->lists[1].data.data(), cx.d_rec[1].p, bytes, hipMemcpyDeviceToHost, cx.stream3));
			early_bytes = bytes;
		}
		HIP_OK(hipStreamSynchronize(cx.stream));
		d_cand = batches->d_cand; d_ncand = batches->d_ncand;
		d_gave_up = batches->d_flags + (size_t)batches->ldv.ncomp * batches->ncomp_max;
		HRY_MARK(g_t0, "vertex chain done");
	} else if (ldv.nplanes) {
		if (unpredict2_applicable(ldv)) {
			// One chain (a wavefront, or a team of them) per attribute component and per connected component of the mesh, all of
			// them in ONE launch in coding order; residual codes come straight from the decoded byte planes.  A component that
			// names vertices coded before it (shared non-manifold vertices) waits for the chain of the component that owns
			// them -- vertex by vertex, through per-component flags in HBM -- instead of for a whole level of components.
			(void)seg_level;
			const uint32_t nseg = (uint32_t)seg_start.size() - 1;
			std::vector<uint32_t> table;
			uint32_t n_lists = 0;
			for (uint32_t k = 0; k < nseg; ++k)
				if (seg_start[k] != seg_start[k + 1]) { table.push_back(seg_start[k]); table.push_back(seg_start[k + 1]); table.push_back(k); ++n_lists; }
			{
				std::vector<uint32_t> starts;
				group_chain_lists(table.data(), n_lists, 0, starts);
				n_lists = (uint32_t)starts.size() - 1;
				table.insert(table.end(), starts.begin(), starts.end());
			}
			const size_t off_at = table.size() - ((size_t)n_lists + 1);
			const size_t segstart_at = table.size();
			table.insert(table.end(), seg_start.begin(), seg_start.end());
			const size_t done_at = table.size();
			const bool scan_chain = unpredict3_wanted(ldv) && seg_start.size() >= 2;
			cx.d_small.ensure((table.size() + (size_t)ldv.ncomp * nseg + 1) * 4 + 64);   // (+ the give-up word behind the flags)
			HIP_OK(hipMemcpyAsync(cx.d_small.p, table.data(), table.size() * 4, hipMemcpyHostToDevice, cx.stream));
			uint32_t *d_tab = cx.d_small.as<uint32_t>();
			HIP_OK(hipMemsetAsync(d_tab + done_at, 0, ((size_t)ldv.ncomp * nseg + 1) * 4, cx.stream));
			d_gave_up = d_tab + done_at + (size_t)ldv.ncomp * nseg;
			launch_candidates_ids(cx.stream, cv, cx.d_order_v.as<uint32_t>(), nvc, d_cand, d_ncand);
			if (scan_chain) launch_chain_records(cx.stream, d_cand, d_ncand, nvc, d_tab + segstart_at, nseg, d_crec);
			HIP_OK(hipEventRecord(cx.ev[7], cx.stream));
			// a 2-D grid holds at most 65535 rows: very many components go in several launches, still in coding order
			for (uint32_t done = 0; done < n_lists; done += 65535)
				launch_unpredict2(cx.stream, cv, cx.d_order_v.as<uint32_t>(), nvc, d_cand, d_ncand, scan_chain ? d_crec : nullptr, d_vplanes, ldv, cx.d_rec[1].as<uint8_t>(),
				                  d_tab, d_tab + off_at + done, std::min(65535u, n_lists - done), d_tab + segstart_at, nseg, d_tab + done_at);
			HIP_OK(hipEventRecord(cx.ev[0], cx.stream));
			chain_timed = true;
			HIP_OK(hipStreamSynchronize(cx.stream));   // the table lives in host memory until the copy has been consumed
			HRY_MARK(g_t0, "vertex chain done");
		} else throw Error(HRY_E_UNSUPPORTED, "8-byte storage types (more than 32 quantisation bits, lossless 64-bit integers) are outside the supported subset");
	}
	if (ldf.nplanes) {
		launch_residuals_to_rec(cx.stream, d_fplanes, m->nf, ldf, cx.d_rec[0].as<uint8_t>());
		launch_faces_unfold(cx.stream, m->nf, ldf, cx.d_rec[0].as<uint8_t>());
	}
	HIP_OK(hipEventRecord(cx.ev[4], cx.stream));
	for (int l = 0; l < 2; ++l) {
		const size_t skip = l == 1 ? early_bytes : 0;   // (already on their way down, on the other stream)
		if (m->lists[l].data.size() > skip) HIP_OK(hipMemcpyAsync(m->lists[l].data.data() + skip, (const uint8_t*)cx.d_rec[l].p + skip, m->lists[l].data.size() - skip, hipMemcpyDeviceToHost, cx.stream));
	}
	HIP_OK(hipStreamSynchronize(cx.stream));
	if (early_bytes) HIP_OK(hipStreamSynchronize(cx.stream3));
	HRY_MARK(g_t0, "records on the host");
	if (uint32_t tf = chain_timeout_flags(cx.stream, d_gave_up)) throw Error(HRY_E_INTERNAL, "reconstruction chain: hand-over between wavefronts timed out (flags " + std::to_string(tf) + ")");
	// (in batches: the sum over every batch of this decode, the ones launched beside the replay included)
	cx.timing.k_chain_ms = batches && batches->n_timed ? batches->elapsed_ms() : chain_timed ? cx.elapsed(7, 0) : 0.0;
	if (cx.keep_stages) {
		cx.stage_put_host("order_v", order_v.data(), order_v.size() * 4);
		cx.stage_put("ncand", d_ncand, nvc);
		cx.stage_put("cand", d_cand, (size_t)nvc * 6 * 4);
	}
}

// ---------------------------------------------------------------------------------------------------------
// Pipelined decode (one large component, no explicitly named vertices, every vertex component quantised): the replay
// publishes its progress (ReplayLive, cbm_replay.hpp) and a consumer thread keeps the device busy behind it --
//   stream2: finished faces' origins / twins / offsets -> HBM as they appear, late twin links as patches, then for every
//            slice of vertices that can no longer change: candidates + chain records (connectivity only);
//   stream : the reconstruction chain of that slice (k_unpredict3_range), which continues the previous slice's chain.
// When the replay ends only the last slice is left.  Results are those of the sequential pipeline: the candidates of a
// complete vertex are final, and the chain is evaluated in the same order with the same arithmetic.
// ---------------------------------------------------------------------------------------------------------
struct SliceClock { hipEvent_t a, b, p0, p1; };   // chain on the main stream; candidates + chain records on the second one
// host -> device through the context's pinned staging buffer (one stream; flush() = everything has left the buffer)
struct Stager {
	Context &cx;
	hipStream_t st;
	size_t used = 0;
	Stager(Context &c, hipStream_t s) : cx(c), st(s)
	{
		const size_t want = 16u << 20;
		if (cx.h_stage_cap < want) {
			if (cx.h_stage) { (void)hipHostFree(cx.h_stage); cx.h_stage = nullptr; cx.h_stage_cap = 0; }
			HIP_OK(hipHostMalloc(&cx.h_stage, want, hipHostMallocDefault));
			cx.h_stage_cap = want;
		}
	}
	void flush() { HIP_OK(hipStreamSynchronize(st)); used = 0; }
	void put(void *dst, const void *src, size_t n)
	{
		const uint8_t *s = (const uint8_t*)src;
		uint8_t *d = (uint8_t*)dst;
		while (n) {
			if (used == cx.h_stage_cap) flush();
			const size_t k = std::min(n, cx.h_stage_cap - used);
			memcpy((uint8_t*)cx.h_stage + used, s, k);
			HIP_OK(hipMemcpyAsync(d, (uint8_t*)cx.h_stage + used, k, hipMemcpyHostToDevice, st));
			used += k; s += k; d += k; n -= k;
		}
	}
};
// Connectivity of a mesh of many components, uploaded BESIDE the parallel replay (cut_border_replay's spans): a thread of its own
// copies every finished span's face offsets, origins, twins and decode order to the device while the other spans are still being
// replayed -- after the replay nothing of the connectivity is left to copy (290 MB for the 12.6 M-triangle share of configs[3]:
// 5.6 ms that used to follow the replay).  The copies come straight from the pageable arrays: one thread, a few large ranges.
struct SpanUploader : SpanDone {
	Context &cx;
	Mesh &m;
	const OrderVec &order_v;
	ChainBatches *batches = nullptr;       // chains launched behind the copies, batch by batch (nullptr: after the replay, as one)
	hipEvent_t planes_ready = nullptr;     // ... once the residual planes are decoded
	struct Range { uint32_t index, n_spans, f0, f1, h0, h1, v0, v1; std::vector<uint32_t> comp_first; bool ends_inside = false; };
	std::vector<Range> arrived;            // by span index, for the prefix of finished spans (uploader thread only)
	std::vector<char> have;
	uint32_t prefix = 0;                   // spans [0, prefix) are on the device
	std::vector<uint32_t> pending_first;   // components of the prefix not launched yet
	uint32_t pending_v_end = 0, batches_launched = 0;
	bool planes_done = false;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<Range> todo;
	bool closing = false;
	uint64_t faces_up = 0, he_up = 0, v_up = 0;
	std::exception_ptr error;
	// Three threads take spans off the list, each with a stream of its own: a copy from pageable memory holds its thread until the
	// runtime has staged it (10 GB/s a thread), and one thread was 16 - 20 ms behind the replay of the configs[3] mesh at its end
	// (2.7 GB in 125 ms; 249 spans still queued), two 13 ms (165 spans), three none: decode 175 -> 163 -> 149 ms; a fourth thread, or
	// replay threads given up for them, gain nothing (16 CPUs of quota: `profiles/r4/decode_uploaders.txt`)
	static constexpr int kWorkers = Context::kUploadStreams + 1;
	std::thread worker[kWorkers];
	hipStream_t up_stream[kWorkers] = {};
	int n_workers = 3;
	std::mutex mu_prefix;                  // after_upload / launch_pending: one worker at a time
	Clock::time_point t_origin = g_t0;     // (the decode's clock: g_t0 is per thread)
	uint32_t eface_upto = 0;               // faces whose half-edge -> face entries are computed on the device
	SpanUploader(Context &c, Mesh &mesh, const OrderVec &ov, ChainBatches *cb = nullptr, hipEvent_t planes = nullptr) : cx(c), m(mesh), order_v(ov), batches(cb), planes_ready(planes)
	{
		cx.ensure_second_stream();
		cx.d_org.ensure(std::max<size_t>((size_t)m.declared_ne * 4, 16));
		cx.d_twin.ensure(std::max<size_t>((size_t)m.declared_ne * 4, 16));
		cx.d_foff.ensure(((size_t)m.nf + 1) * 4);
		cx.d_order_v.ensure(std::max<size_t>((size_t)m.nv * 4, 16));
		if (batches) {
			int ud = 0;
			cx.res_has_eface = !m.uniform_degree(ud);
			cx.res_udeg = (uint32_t)ud;
			if (cx.res_has_eface) cx.d_eface.ensure(std::max<size_t>((size_t)m.declared_ne * 4, 16));
			// (no wait for these: the first batch's launch follows them through an event of this stream, launch_pending -- and a wait
			// here lasted as long as the attribute streams' kernel, 40 ms of the named size's decode in front of the replay: with more
			// streams than hardware queues this stream shares a queue with one of theirs)
			HIP_OK(hipMemsetAsync(cx.d_foff.p, 0, 4, cx.stream2));
			batches->init(cx.stream2, m);
		}
		static const int wanted = [] { const char *e = getenv("HRY_SPAN_UPLOADERS"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : v > kWorkers ? kWorkers : v; }();
		n_workers = wanted;
		up_stream[0] = cx.stream2;
		for (int k = 1; k < n_workers; ++k) {
			if (!cx.up_stream[k - 1]) HIP_OK(hipStreamCreateWithFlags(&cx.up_stream[k - 1], hipStreamNonBlocking));
			if (!cx.up_ev[k - 1]) HIP_OK(hipEventCreateWithFlags(&cx.up_ev[k - 1], hipEventDisableTiming));
			up_stream[k] = cx.up_stream[k - 1];
		}
		const void *node = callers_node_cpus();
		HRY_MARK(t_origin, "span uploaders' arrays and streams ready");
		for (int k = 0; k < n_workers; ++k) worker[k] = std::thread([this, node, k] {
			try {
				stay_on_node(node);
				HIP_OK(hipSetDevice(cx.device));
				for (;;) {
					Range r;
					{
						std::unique_lock<std::mutex> lk(mu);
						cv.wait(lk, [&] { return closing || !todo.empty(); });
						if (todo.empty()) break;
						r = todo.front(); todo.pop_front();
					}
					copy_range(r, up_stream[k]);
					bool ending;
					{ std::lock_guard<std::mutex> g(mu); faces_up += r.f1 - r.f0; he_up += r.h1 - r.h0; v_up += r.v1 - r.v0; ending = closing; }
					if (batches && !ending) { std::lock_guard<std::mutex> g(mu_prefix); after_upload(std::move(r)); }   // (once the replay has ended the caller launches the rest)
				}
				HIP_OK(hipStreamSynchronize(up_stream[k]));
			} catch (...) { std::lock_guard<std::mutex> g(mu); if (!error) error = std::current_exception(); }
		});
	}
	void copy_range(const Range &r, hipStream_t st)
	{
		if (r.f1 > r.f0) HIP_OK(hipMemcpyAsync(cx.d_foff.as<uint32_t>() + r.f0 + 1, m.face_off.data() + r.f0 + 1, ((size_t)r.f1 - r.f0) * 4, hipMemcpyHostToDevice, st));
		if (r.h1 > r.h0) {
			HIP_OK(hipMemcpyAsync(cx.d_org.as<uint32_t>() + r.h0, m.org.data() + r.h0, ((size_t)r.h1 - r.h0) * 4, hipMemcpyHostToDevice, st));
			HIP_OK(hipMemcpyAsync(cx.d_twin.as<uint32_t>() + r.h0, m.twin.data() + r.h0, ((size_t)r.h1 - r.h0) * 4, hipMemcpyHostToDevice, st));
		}
		if (r.v1 > r.v0) HIP_OK(hipMemcpyAsync(cx.d_order_v.as<uint32_t>() + r.v0, order_v.data() + r.v0, ((size_t)r.v1 - r.v0) * 4, hipMemcpyHostToDevice, st));
	}
	void span(uint32_t index, uint32_t n_spans, uint32_t f0, uint32_t f1, uint32_t h0, uint32_t h1, uint32_t v0, uint32_t v1, const uint32_t *comp_first, uint32_t n_comp, bool ends_inside) override
	{
		Range r{ index, n_spans, f0, f1, h0, h1, v0, v1, {}, ends_inside };
		if (batches) r.comp_first.assign(comp_first, comp_first + n_comp);
		{ std::lock_guard<std::mutex> g(mu); todo.push_back(std::move(r)); }
		cv.notify_one();
	}
	// (an uploader thread, under mu_prefix) a span's arrays are on their way: extend the prefix of finished spans, and once the residual planes are
	// decoded launch the chains of the components the prefix has gained -- a quarter of the vertices at a time at least
	void after_upload(Range &&r)
	{
		if (have.empty()) { have.assign(r.n_spans, 0); arrived.resize(r.n_spans); }
		if (r.index >= have.size() || have[r.index]) throw Error(HRY_E_INTERNAL, "replay: span reported twice");
		const uint32_t idx = r.index;
		arrived[idx] = std::move(r);
		have[idx] = 1;
		while (prefix < have.size() && have[prefix]) {
			// (a component that goes on in the next span -- the span stops at a border snapshot -- joins the prefix with its last span)
			size_t q = prefix;
			while (q < have.size() && have[q] && arrived[q].ends_inside) ++q;
			if (q == have.size() || !have[q]) break;
			for (; prefix <= q; ++prefix) {
				Range &p = arrived[prefix];
				pending_first.insert(pending_first.end(), p.comp_first.begin(), p.comp_first.end());
				pending_v_end = p.v1;
				std::vector<uint32_t>().swap(p.comp_first);
			}
		}
		if (!planes_done) planes_done = hipEventQuery(planes_ready) == hipSuccess;
		static const uint32_t parts = [] { const char *e = getenv("HRY_CHAIN_BATCH_PARTS"); const int v = e ? atoi(e) : 4; return (uint32_t)(v < 1 ? 1 : v); }();
		if (planes_done && prefix < have.size() && !pending_first.empty() && pending_v_end - batches->v_done >= std::max(1u, m.nv / parts)) launch_pending();
	}
	void launch_pending()
	{
		// the chains run on the MAIN stream (idle while the host replays), behind an event on the copies' stream: on that stream
		// itself the copies of the later spans would queue behind tens of milliseconds of chains
		cx.res_nv = m.nv; cx.res_nf = m.nf; cx.res_ne = m.declared_ne;   // (what conn_view reports: the arrays are that large from the start)
		const uint32_t f_end = arrived[prefix - 1].f1;   // the faces of the prefix: their offsets are complete on the device
		for (int k = 1; k < n_workers; ++k) {   // (the prefix' spans went up on any of the streams: the first waits for the others)
			HIP_OK(hipEventRecord(cx.up_ev[k - 1], up_stream[k]));
			HIP_OK(hipStreamWaitEvent(cx.stream2, cx.up_ev[k - 1], 0));
		}
		if (cx.res_has_eface && f_end > eface_upto) dev::launch_edge_faces(cx.stream2, cx.d_foff.as<uint32_t>(), f_end, cx.d_eface.as<uint32_t>(), eface_upto);
		eface_upto = f_end;
		HIP_OK(hipEventRecord(cx.ev_x[2], cx.stream2));
		HIP_OK(hipStreamWaitEvent(cx.stream, cx.ev_x[2], 0));
		batches->launch(cx.stream, pending_first.data(), (uint32_t)pending_first.size(), pending_v_end);
		HIP_OK(hipEventRecord(cx.attr_ev[0], cx.stream));   // (the attribute groups' events are free by now: the planes are decoded)
		batches->first_done = cx.attr_ev[0]; batches->v_first_batch = pending_v_end;
		pending_first.clear();
		++batches_launched;
		if (getenv("HRY_TRACE")) fprintf(stderr, "[hry] %8.3f ms  chains of the components up to vertex %u launched beside the replay\n", ms_since(t_origin), pending_v_end);
	}
	// everything is on the device (true) or the caller uploads as usual (false: the replay ran as one sequence, or a copy failed)
	bool finish()
	{
		size_t left = 0;
		{ std::lock_guard<std::mutex> g(mu); closing = true; left = todo.size(); }
		cv.notify_all();
		if (trace_on()) fprintf(stderr, "[hry] %8.3f ms  replay returned: %zu finished span(s) still to be copied\n", ms_since(t_origin), left);
		// The replay is over and its thread has nothing to do: it takes spans off the list too, on the stream the attribute planes
		// were decoded on (idle by now) -- the one uploader used to be 25 - 30 ms behind at this point on the configs[3] mesh.
		// (No more batches from here on: the caller launches what is left.)
		bool failed;
		{ std::lock_guard<std::mutex> g(mu); failed = (bool)error; }   // (the uploaders set it under the same lock)
		if (!failed) {
			try {
				for (;;) {
					Range r;
					{
						std::lock_guard<std::mutex> g(mu);
						if (todo.empty()) break;
						r = std::move(todo.back()); todo.pop_back();
					}
					copy_range(r, cx.stream3 ? cx.stream3 : cx.stream2);
					std::lock_guard<std::mutex> g(mu);
					faces_up += r.f1 - r.f0; he_up += r.h1 - r.h0; v_up += r.v1 - r.v0;
				}
				if (cx.stream3) HIP_OK(hipStreamSynchronize(cx.stream3));
			} catch (...) { for (auto &w : worker) if (w.joinable()) w.join(); throw; }
		}
		for (auto &w : worker) if (w.joinable()) w.join();
		if (trace_on()) fprintf(stderr, "[hry] %8.3f ms  every span is on the device\n", ms_since(t_origin));
		if (error) return false;
		if (faces_up != m.nf || he_up != m.declared_ne || v_up != order_v.size() || order_v.size() != m.nv) return false;
		if (batches) return true;   // (the first offset went up before the first span; the main stream may still be running a batch of chains: no wait)
		const uint32_t zero = 0;
		HIP_OK(hipMemcpyAsync(cx.d_foff.p, &zero, 4, hipMemcpyHostToDevice, cx.stream));
		HIP_OK(hipStreamSynchronize(cx.stream));
		return true;
	}
	// the half-edge -> face entries of the faces behind the last batch, on stream st (the caller's last batch follows on it)
	void finish_edge_faces(hipStream_t st)
	{
		if (cx.res_has_eface && m.nf > eface_upto) dev::launch_edge_faces(st, cx.d_foff.as<uint32_t>(), m.nf, cx.d_eface.as<uint32_t>(), eface_upto);
		eface_upto = m.nf;
		cx.res_nv = m.nv; cx.res_nf = m.nf; cx.res_ne = m.declared_ne;
		m.twins_pending = false;
		cx.resident_token = 0;
	}
	~SpanUploader() { { std::lock_guard<std::mutex> g(mu); closing = true; todo.clear(); } cv.notify_all(); for (auto &w : worker) if (w.joinable()) w.join(); }
};

static bool pipelined_decode_applicable(const Mesh &m, const std::vector<RestartPoint> &restarts, const PlaneView *conn,
                                        const ListDesc &ldv, uint32_t vc)
{
	if (getenv("HRY_NO_PIPELINE")) return false;
	int ud = 0;
	uint32_t min_nv = 1u << 17;
	if (const char *e = getenv("HRY_PIPELINE_MIN_VERTICES")) min_nv = (uint32_t)strtoul(e, nullptr, 10);
	return restarts.empty() && conn[7].empty() && m.uniform_degree(ud) && unpredict3_covers(ldv) && vc == m.nv && m.nv >= min_nv && m.declared_ne != 0;
}

// attr_upto[g]: the vertex planes are decoded up to this vertex once cx.attr_ev[g] has fired (the last entry covers everything)
static void decode_pipelined(Context &cx, Mesh &mesh, const PlaneView *conn, const uint8_t *d_vplanes, const uint8_t *d_fplanes,
                             const ListDesc &ldv, const ListDesc &ldf, OrderVec &order_v, const uint32_t (&attr_upto)[Context::kAttrGroups],
                             const std::vector<SnapshotPoint> &snaps_in)
{
	Mesh *m = &mesh;
	const uint32_t nv = m->nv, nf = m->nf, ne = m->declared_ne;
	if (!cx.stream2) HIP_OK(hipStreamCreateWithFlags(&cx.stream2, hipStreamNonBlocking));
	HRY_MARK(g_t0, "pipelined decode: begin");
	// device arrays at their final size; records start as zeros (host records are zero-filled by the header reader)
	for (int l = 0; l < 2; ++l) {
		cx.d_rec[l].ensure(std::max<size_t>(m->lists[l].data.size(), 16));
		if (!m->lists[l].data.empty()) HIP_OK(hipMemsetAsync(cx.d_rec[l].p, 0, m->lists[l].data.size(), cx.stream2));
	}
	cx.d_org.ensure(std::max<size_t>((size_t)ne * 4, 16));
	cx.d_twin.ensure(std::max<size_t>((size_t)ne * 4, 16));
	cx.d_foff.ensure(((size_t)nf + 1) * 4);
	cx.d_order_v.ensure(std::max<size_t>((size_t)nv * 4, 16));
	const size_t ncand_bytes = ((size_t)nv + 63) & ~(size_t)63;
	const size_t cand_words = (cand_table_words(nv) + 3) & ~(size_t)3;
	cx.d_cscratch.ensure(std::max<size_t>(cand_words * 4 + (size_t)nv * 16 + ncand_bytes + 64, 16));
	uint32_t *d_cand = cx.d_cscratch.as<uint32_t>();
	uint8_t *d_ncand = (uint8_t*)(d_cand + cand_words);
	cand_table_reset(cx.stream2, d_cand, nv);
	void *d_crec = d_ncand + ncand_bytes;
	int ud = 0;
	m->uniform_degree(ud);
	cx.res_has_eface = false; cx.res_udeg = (uint32_t)ud; cx.res_nv = nv; cx.res_nf = nf; cx.res_ne = ne;
	const ConnView cv = cx.conn_view();

	m->face_off.resize((size_t)nf + 1); m->face_off[0] = 0;   // every entry is written before it is read (BigVec: no fill)
	m->org.resize(ne);
	order_v.assign(nv, 0);
	// Round 6: border snapshots in the directory (restart points inside the component): this thread replays the stretch up to the
	// first one and publishes as before, the stretches behind the snapshots run on helper threads meanwhile and are joined when
	// all have finished (cbm_replay.hpp SnapshotSpans) -- the replay of ONE component on several cores
	std::unique_ptr<SnapshotSpans> spans;
	if (ud == 3 && !snaps_in.empty() && host_threads() > 1 && !getenv("HRY_NO_SNAPSHOT_REPLAY") && !getenv("HRY_GENERIC_REPLAY")) spans.reset(new SnapshotSpans(*m, conn, snaps_in, order_v.data()));   // (sizes m->twin)
	else m->twin.resize(ne);
	BigVec<uint16_t> seen(nv, 0);
	ReplayLive live;
	live.on_border.assign(nv, 0);
	live.pending.reserve(1 << 16);
	HRY_MARK(g_t0, "host arrays allocated");
	if (const char *e = getenv("HRY_PIPELINE_FACES")) live.interval = std::max(1u, (uint32_t)strtoul(e, nullptr, 10));

	std::exception_ptr consumer_error;
	std::vector<SliceClock> clocks;
	uint32_t min_slice = 1u << 15;   // replay and chain take about the same time per vertex: what is left behind the replay is the last slice (16 Ki: the consumer's launches become the longer path)
	if (const char *e = getenv("HRY_PIPELINE_SLICE")) min_slice = std::max(64u, (uint32_t)strtoul(e, nullptr, 10));
	// ... but the first slices are small and double up to that size: the chain is the longer path, what counts is how early it
	// starts, and the early chunks of the vertex planes are short for exactly that (attr_chunk_len)
	uint32_t first_slice = 1u << 13;
	if (const char *e = getenv("HRY_PIPELINE_FIRST_SLICE")) first_slice = std::max(64u, (uint32_t)strtoul(e, nullptr, 10));
	first_slice = std::min(first_slice, min_slice);
	uint32_t last_piece = 1u << 16;   // pieces of what is left when the replay is done (HRY_PIPELINE_LAST_PIECE; tests: small)
	if (const char *e = getenv("HRY_PIPELINE_LAST_PIECE")) last_piece = std::max(64u, (uint32_t)strtoul(e, nullptr, 10) & ~63u);
	int attr_waited = 0;   // groups of attribute streams the chain's stream has been told to wait for
	const Clock::time_point t_begin = g_t0;
	// vertex records come back slice by slice into pinned memory and are copied into the mesh by the consumer as they land
	const size_t vrec_bytes = m->lists[1].data.size();
	if (cx.h_down_cap < vrec_bytes) {
		if (cx.h_down) { (void)hipHostFree(cx.h_down); cx.h_down = nullptr; cx.h_down_cap = 0; }
		HIP_OK(hipHostMalloc(&cx.h_down, vrec_bytes + (vrec_bytes >> 3) + 4096, hipHostMallocDefault));
		cx.h_down_cap = vrec_bytes + (vrec_bytes >> 3) + 4096;
	}
	cx.d_patch.ensure(std::max<size_t>(4u << 20, (size_t)nv));   // late twin links of a slice: (edge, twin) pairs, a few thousand per slice
	struct Landing { hipEvent_t ev; size_t off, len; };
	std::deque<Landing> landing;
	const int vstride = ldv.stride;
	const void *near = callers_neighbour_cpus();   // the consumer polls: near the replay's caches, but not on its core
	const void *node = near ? near : callers_node_cpus();
	std::thread consumer([&, node] {
		try {
			stay_on_node(node);
			HIP_OK(hipSetDevice(cx.device));
			uint32_t f_up = 0, he_up = 0, v_done = 0, v_up = 0;   // v_up: decode order on the device up to this vertex
			uint64_t seen_seq = 0, seen_pub = 0;
			std::vector<uint32_t> patches;
			std::vector<ReplayLive::Range> ranges, ranges_up;   // stretches of helper threads: announced / on the device (round 6)
			// the helpers' pinned copies go up on a stream of their own -- on the uploads' stream (stream2) the candidates of the next
			// slice of the FIRST stretch sat behind a millisecond of the other stretches' transfers -- with an event each, which
			// stream2 waits for when a publication reaches into them
			std::vector<hipEvent_t> mir_ev;        // per entry of ranges_up (nullptr: came through stream2)
			std::vector<char> mir_waited;          // per entry of mir_ev: the second stream waits for it already
			if (!cx.up_stream[0]) HIP_OK(hipStreamCreateWithFlags(&cx.up_stream[0], hipStreamNonBlocking));
			const hipStream_t mir_stream = cx.up_stream[0];
			if (!cx.up_stream[1]) HIP_OK(hipStreamCreateWithFlags(&cx.up_stream[1], hipStreamNonBlocking));
			const hipStream_t down_stream = cx.up_stream[1];   // the slices' records on their way down
			struct DropEvents { std::vector<hipEvent_t> &v; ~DropEvents() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } drop_events{ mir_ev };
			DevBuf &d_patch = cx.d_patch;   // persistent and sized before the pipeline starts: growing it here would synchronise the device (hipFree / hipMalloc) in mid-flight
			Stager up(cx, cx.stream2);
			hipEvent_t prepared;
			HIP_OK(hipEventCreateWithFlags(&prepared, hipEventDisableTiming));
			// The consumer acts on a publication only when `lag` newer ones exist: the newest part of the arrays is still hot
			// in the replay thread's cache (twins of the last ring keep changing), copying it there would slow the replay down.
			std::deque<ReplayLive::Pub> hist;
			uint32_t lag = 2;
			if (const char *e = getenv("HRY_PIPELINE_LAG")) lag = (uint32_t)strtoul(e, nullptr, 10);
			auto drain = [&](bool wait) {
				while (!landing.empty()) {
					Landing &L = landing.front();
					if (wait) HIP_OK(hipEventSynchronize(L.ev));
					else if (hipEventQuery(L.ev) != hipSuccess) break;
					memcpy(m->lists[1].data.data() + L.off, (const uint8_t*)cx.h_down + L.off, L.len);
					if (trace_on()) fprintf(stderr, "[hry] %8.3f ms  records of vertices [%zu, %zu) landed\n", ms_since(t_begin), L.off / vstride, (L.off + L.len) / vstride);
					(void)hipEventDestroy(L.ev);
					landing.pop_front();
				}
			};
			for (;;) {
				ReplayLive::Pub newest;
				drain(false);
				{
					// polled: the chain's hand-over of the next slice should not wait for a wake-up either (a decode lasts milliseconds)
					for (uint32_t spins = 0; live.announced.load(std::memory_order_acquire) == seen_seq; ++spins) {
						if (spins < 2000) __builtin_ia32_pause();
						else { drain(false); std::this_thread::sleep_for(std::chrono::microseconds(20)); }
					}
					std::unique_lock<std::mutex> lk(live.mu);
					newest = live.pub;
					patches.insert(patches.end(), live.patches.begin(), live.patches.end());
					live.patches.clear();
					ranges.insert(ranges.end(), live.ranges.begin(), live.ranges.end());
					live.ranges.clear();
					seen_seq = newest.seq;
				}
				if (newest.failed) break;
				// a helper thread's finished stretch goes up at once (nobody is writing there any more)
				for (const ReplayLive::Range &r : ranges) {
					hipEvent_t rev = nullptr;
					if (r.mirrored) {   // (the helper left a pinned copy: four transfers, no copy on this thread)
						const uint32_t *mf = cx.h_mirror.as<uint32_t>(), *mo = mf + (size_t)nf + 1, *mt = mo + ne, *mv = mt + ne;
						static const bool pull = !getenv("HRY_NO_PULL");
						if (pull) {   // one kernel reads the four ranges from the pinned mirrors (HRY_NO_PULL=1: four transfers, as until round 6)
							PullRanges pr{};
							pr.dst[0] = cx.d_foff.as<uint32_t>() + r.f0 + 1; pr.src[0] = mf + r.f0 + 1; pr.words[0] = r.f1 - r.f0;
							pr.dst[1] = cx.d_org.as<uint32_t>() + r.h0; pr.src[1] = mo + r.h0; pr.words[1] = r.h1 - r.h0;
							pr.dst[2] = cx.d_twin.as<uint32_t>() + r.h0; pr.src[2] = mt + r.h0; pr.words[2] = r.h1 - r.h0;
							pr.dst[3] = cx.d_order_v.as<uint32_t>() + r.v0; pr.src[3] = mv + r.v0; pr.words[3] = r.v1 - r.v0;
							launch_pull_ranges(mir_stream, pr);
						} else {
						if (r.f1 > r.f0) HIP_OK(hipMemcpyAsync(cx.d_foff.as<uint32_t>() + r.f0 + 1, mf + r.f0 + 1, ((size_t)r.f1 - r.f0) * 4, hipMemcpyHostToDevice, mir_stream));
						if (r.h1 > r.h0) {
							HIP_OK(hipMemcpyAsync(cx.d_org.as<uint32_t>() + r.h0, mo + r.h0, ((size_t)r.h1 - r.h0) * 4, hipMemcpyHostToDevice, mir_stream));
							HIP_OK(hipMemcpyAsync(cx.d_twin.as<uint32_t>() + r.h0, mt + r.h0, ((size_t)r.h1 - r.h0) * 4, hipMemcpyHostToDevice, mir_stream));
						}
						if (r.v1 > r.v0) HIP_OK(hipMemcpyAsync(cx.d_order_v.as<uint32_t>() + r.v0, mv + r.v0, ((size_t)r.v1 - r.v0) * 4, hipMemcpyHostToDevice, mir_stream));
						}
						HIP_OK(hipEventCreateWithFlags(&rev, hipEventDisableTiming));
						HIP_OK(hipEventRecord(rev, mir_stream));
					} else {
					if (r.f1 > r.f0) up.put(cx.d_foff.as<uint32_t>() + r.f0 + 1, m->face_off.data() + r.f0 + 1, ((size_t)r.f1 - r.f0) * 4);
					if (r.h1 > r.h0) {
						up.put(cx.d_org.as<uint32_t>() + r.h0, m->org.data() + r.h0, ((size_t)r.h1 - r.h0) * 4);
						up.put(cx.d_twin.as<uint32_t>() + r.h0, m->twin.data() + r.h0, ((size_t)r.h1 - r.h0) * 4);
					}
					if (r.v1 > r.v0) up.put(cx.d_order_v.as<uint32_t>() + r.v0, order_v.data() + r.v0, ((size_t)r.v1 - r.v0) * 4);
					}
					if (trace_on()) fprintf(stderr, "[hry] %8.3f ms  a helper's stretch (faces %u .. %u) is on its way up\n", ms_since(t_begin), r.f0, r.f1);
					ranges_up.push_back(r);
					mir_ev.push_back(rev);
				}
				ranges.clear();
				if (newest.n_pub == seen_pub) continue;   // (a helper's announcement only: no new publication of the replaying thread)
				seen_pub = newest.n_pub;
				if (trace_on() && getenv("HRY_TRACE_CONSUMER")) fprintf(stderr, "[hry] %8.3f ms    consumer: publication %llu (faces %u, vertices final up to %u)\n", ms_since(t_begin), (unsigned long long)newest.seq, newest.faces, newest.upto);
				hist.push_back(newest);
				if (!newest.done && !newest.joined && hist.size() <= lag) continue;
				const ReplayLive::Pub P = newest.done || newest.joined ? newest : hist.front();
				while (!hist.empty() && hist.front().n_pub <= P.n_pub) hist.pop_front();
				// (a publication behind the replaying thread's own stretch rests on the helpers' stretches: their transfers first)
				// (only those that lie below the publication's faces: the fans of its final vertices end there.  Waiting for every
				// transfer under way -- they arrive in the order the helpers finish, 24 MB within 0.3 ms of each other, a millisecond
				// on the link -- kept the chain idle until the LAST stretch was up: HRY_MIRROR_WAIT_ALL=1)
				if (P.joined || P.done) {
					static const bool wait_all = getenv("HRY_MIRROR_WAIT_ALL") != nullptr;
					mir_waited.resize(mir_ev.size(), 0);
					for (size_t i = 0; i < mir_ev.size(); ++i)
						if (!mir_waited[i] && mir_ev[i] && (wait_all || P.done || ranges_up[i].f0 < P.faces)) { HIP_OK(hipStreamWaitEvent(cx.stream2, mir_ev[i], 0)); mir_waited[i] = 1; }
				}
				// finished part of the connectivity -- but for what the helpers' stretches have brought up already
				if (P.faces > f_up) {
					std::vector<ReplayLive::Range> by_face(ranges_up);
					std::sort(by_face.begin(), by_face.end(), [](const ReplayLive::Range &a, const ReplayLive::Range &b) { return a.f0 < b.f0; });
					uint32_t f = f_up, h = he_up;
					auto copy_upto = [&](uint32_t f_to, uint32_t he_to) {
						if (f_to <= f) return;
						up.put(cx.d_foff.as<uint32_t>() + f, m->face_off.data() + f, ((size_t)f_to - f + 1) * 4);
						up.put(cx.d_org.as<uint32_t>() + h, m->org.data() + h, ((size_t)he_to - h) * 4);
						up.put(cx.d_twin.as<uint32_t>() + h, m->twin.data() + h, ((size_t)he_to - h) * 4);
					};
					for (const ReplayLive::Range &r : by_face) {
						if (r.f1 <= f || r.f0 >= P.faces) continue;
						copy_upto(std::min(r.f0, P.faces), std::min(r.h0, P.he));
						f = std::max(f, r.f1); h = std::max(h, r.h1);
					}
					copy_upto(P.faces, P.he);
					f_up = P.faces; he_up = P.he;
				}
				// vertices that can no longer change: whole tiles, slices of a useful size
				const uint32_t v_hi = P.done ? nv : (P.upto & ~63u);
				if (v_hi > v_done && (P.done || P.joined || v_hi - v_done >= std::min(min_slice, std::max(first_slice, v_done)))) {
					// late links of edges that were copied before (a patch of an edge that is copied later is harmless: the copy
					// carries the final value too)
					if (!patches.empty()) {
						d_patch.ensure(patches.size() * 4);
						up.put(d_patch.p, patches.data(), patches.size() * 4);
						launch_scatter_u32(cx.stream2, d_patch.as<uint32_t>(), (uint32_t)(patches.size() / 2), cx.d_twin.as<uint32_t>());
						patches.clear();
					}
					{   // the decode order of the slice's vertices (the helpers' stretches brought theirs)
						std::vector<ReplayLive::Range> by_face(ranges_up);
						std::sort(by_face.begin(), by_face.end(), [](const ReplayLive::Range &a, const ReplayLive::Range &b) { return a.f0 < b.f0; });
						uint32_t v = std::max(v_done, v_up);
						for (const ReplayLive::Range &r : by_face) {
							if (r.v1 <= v || r.v0 >= v_hi) continue;
							if (r.v0 > v) up.put(cx.d_order_v.as<uint32_t>() + v, order_v.data() + v, ((size_t)std::min(r.v0, v_hi) - v) * 4);
							v = std::max(v, r.v1);
						}
						if (v_hi > v) up.put(cx.d_order_v.as<uint32_t>() + v, order_v.data() + v, ((size_t)v_hi - v) * 4);
						v_up = std::max(v_up, v_hi);
					}
					// what the kernels of this slice may follow: the half-edges that are on the device now (a link into the part
					// that is not -- made after the publication this slice rests on -- reads as a border, which is what it was then)
					ConnView cvs = cv;
					cvs.ne = he_up;
					// (round 6: what is left when the replay is done -- since its stretches run side by side, most of the mesh -- goes in
					// pieces: a piece's records come down while the next piece's chain runs, and its candidates are found beside the chain
					// of the piece before)
					const uint32_t piece = (P.done || P.joined) && v_hi - v_done > 3 * last_piece ? last_piece : v_hi - v_done;
					for (uint32_t v_lo = v_done; v_lo < v_hi;) {
						const uint32_t v_to = v_hi - v_lo <= piece + piece / 2 ? v_hi : v_lo + piece;
						SliceClock ck;
						HIP_OK(hipEventCreate(&ck.a)); HIP_OK(hipEventCreate(&ck.b)); HIP_OK(hipEventCreate(&ck.p0)); HIP_OK(hipEventCreate(&ck.p1));
						HIP_OK(hipEventRecord(ck.p0, cx.stream2));
						launch_slice_prepare(cx.stream2, cvs, cx.d_order_v.as<uint32_t>(), nv, v_lo, v_to, d_cand, d_ncand, d_crec);
						HIP_OK(hipEventRecord(ck.p1, cx.stream2));
						HIP_OK(hipEventRecord(prepared, cx.stream2));
						HIP_OK(hipStreamWaitEvent(cx.stream, prepared, 0));
						// the residual codes of this slice: the groups of attribute streams that end inside it or before
						while (attr_waited < Context::kAttrGroups && (attr_waited == 0 || attr_upto[attr_waited - 1] < v_to)) HIP_OK(hipStreamWaitEvent(cx.stream, cx.attr_ev[attr_waited++], 0));
						HIP_OK(hipEventRecord(ck.a, cx.stream));
						launch_slice_chain(cx.stream, cvs, cx.d_order_v.as<uint32_t>(), nv, v_lo, v_to, d_cand, d_ncand, d_crec, d_vplanes, ldv, cx.d_rec[1].as<uint8_t>());
						HIP_OK(hipEventRecord(ck.b, cx.stream));
						clocks.push_back(ck);
						{
							// (the records come down on a stream of their own, behind the slice's chain: on the chain's stream the next
							// slice's chain waited for the copy -- 0.1 ms a slice, 30 ms of the 28 M-triangle mesh's 214 slices)
							Landing L;
							L.off = (size_t)v_lo * vstride; L.len = ((size_t)v_to - v_lo) * vstride;
							HIP_OK(hipStreamWaitEvent(down_stream, ck.b, 0));
							HIP_OK(hipMemcpyAsync((uint8_t*)cx.h_down + L.off, cx.d_rec[1].as<uint8_t>() + L.off, L.len, hipMemcpyDeviceToHost, down_stream));
							HIP_OK(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
							HIP_OK(hipEventRecord(L.ev, down_stream));
							landing.push_back(L);
						}
						if (trace_on()) fprintf(stderr, "[hry] %8.3f ms  slice [%u, %u) enqueued (replay at face %u)\n", ms_since(t_begin), v_lo, v_to, newest.faces);
						v_lo = v_to;
						drain(false);
					}
					v_done = v_hi;
				}
				if (P.done) break;
			}
			up.flush();
			drain(true);
			(void)hipEventDestroy(prepared);
		} catch (...) { consumer_error = std::current_exception(); }
	});

	// ---- the replay itself (this thread)
	std::exception_ptr replay_error;
	ReplayCursor cur;
	try {
		int onlydeg = ud;
		struct PlanesRd {   // bare cursors: the replay reads a byte or two per operation
			const uint8_t *cur[21], *end[21]; int fixed_numtri;
			uint32_t byte(int plane) { if (cur[plane] == end[plane]) throw Error(HRY_E_FORMAT, "corrupt stream (connectivity plane exhausted)"); return *cur[plane]++; }
			uint32_t iop() { return byte(0); }
			uint32_t u32(int first) { uint32_t v = byte(first); v |= byte(first + 1) << 8; v |= byte(first + 2) << 16; v |= byte(first + 3) << 24; return v; }
			int elem() { uint32_t z = u32(1); return (int)((z >> 1) ^ ((z & 1) ? 0xffffffffu : 0u)); }
			int part() { uint32_t v = byte(5); v |= byte(6) << 8; return (int)v; }
			uint32_t vertid() { return u32(7); }
			int numtri() { return fixed_numtri; }
			uint32_t op(int order) { int k = order - 1; if (k > 7) k = 7; if (k < 0) k = 0; return byte(13 + k); }
		} rd;
		for (int k = 0; k < 21; ++k) { rd.cur[k] = conn[k].data(); rd.end[k] = conn[k].data() + conn[k].size(); }
		rd.fixed_numtri = onlydeg - 2;
		const RestartCounters none;
		std::vector<uint32_t> comp_first;
		std::vector<std::pair<uint32_t, uint32_t>> refs;
		const bool count = getenv("HRY_PERF") != nullptr;
		PerfCounters pc;
		if (count) pc.start();
		if (spans) {
			spans->announce_to = &live;   // (the consumer copies a helper's stretch as soon as it is finished)
			{   // ... from pinned mirrors the helpers fill themselves, where the mesh is small enough for them: 1 GB (HRY_MIRROR_MAX_MB) --
				// 30 bytes a triangle, 840 MB for the 28 M triangles of configs[2], kept by the context.  Without them the consumer
				// thread stages every byte itself: 10 GB/s, 80 ms for that mesh, in front of a chain that the replay no longer holds up
				const size_t words = (size_t)nf + 1 + 2 * (size_t)ne + nv;
				static const size_t max_mb = [] { const char *e = getenv("HRY_MIRROR_MAX_MB"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1024; }();
				if (words * 4 <= (max_mb << 20) && !getenv("HRY_NO_MIRRORS")) {
					cx.h_mirror.ensure(words * 4);
					uint32_t *p0 = cx.h_mirror.as<uint32_t>();
					spans->mirror_foff = p0; spans->mirror_org = p0 + (size_t)nf + 1; spans->mirror_twin = spans->mirror_org + ne; spans->mirror_order = spans->mirror_twin + ne;
				}
			}
			spans->start(host_threads() - 1);
			BorderEnd end0;
			size_t cur_end0[21];
			const bool eom0 = replay_triangles<true>(*m, conn, seen.data(), order_v.data(), cur, comp_first, refs, &live, nullptr, spans->spans[0].cur1, spans->spans[0].stop_face, true, nullptr, &end0, cur_end0);
			if (!eom0) live.publish(cur.face, cur.he, cur.next_id, false);   // (what this stretch has finished: the consumer goes on while the others are waited for)
			HRY_MARK(g_t0, "replay: the first stretch has reached its snapshot");
			spans->finish(cur, cur_end0, std::move(end0), eom0, &live);
		}
		else if (onlydeg == 3 && !getenv("HRY_GENERIC_REPLAY")) replay_triangles<true>(*m, conn, seen.data(), order_v.data(), cur, comp_first, refs, &live);   // the lean loop
		else replay_span(*m, rd, seen.data(), order_v.data(), cur, replay_detail::NONE32, 0, none, comp_first, refs, &live);
		if (count) { pc.stop(); pc.report("cut-border replay (pipelined decode, publishing)", (double)cur.he - 2.0 * cur.face); }
		if (cur.face != nf) throw Error(HRY_E_FORMAT, "corrupt stream (face count)");
		if (cur.he != ne) throw Error(HRY_E_FORMAT, "corrupt stream (polygon edge count)");
		if (cur.next_id != nv) throw Error(HRY_E_FORMAT, "vertex plane length does not match the connectivity");
		live.publish(cur.face, cur.he, cur.next_id, true);
	} catch (...) {
		replay_error = std::current_exception();
		live.publish(cur.face, cur.he, cur.next_id, true, true);
	}
	cx.timing.host_walk_ms = ms_since(g_t0) ;
	HRY_MARK(g_t0, "replay done");
	if (trace_on()) fprintf(stderr, "[hry] %u publications, %.3f ms inside publish()\n", live.n_publish, live.t_publish_ms);
	consumer.join();
	auto drop_clocks = [&] { for (auto &c : clocks) { (void)hipEventDestroy(c.a); (void)hipEventDestroy(c.b); (void)hipEventDestroy(c.p0); (void)hipEventDestroy(c.p1); } };
	if (replay_error || consumer_error) {
		(void)hipStreamSynchronize(cx.stream); (void)hipStreamSynchronize(cx.stream2);
		for (auto &u : cx.up_stream) if (u) (void)hipStreamSynchronize(u);
		drop_clocks();
		for (auto &L : landing) (void)hipEventDestroy(L.ev);
		std::rethrow_exception(replay_error ? replay_error : consumer_error);
	}
	order_v.resize(cur.next_id);
	for (; attr_waited < Context::kAttrGroups; ++attr_waited) HIP_OK(hipStreamWaitEvent(cx.stream, cx.attr_ev[attr_waited], 0));
	if (ldf.nplanes) {
		launch_residuals_to_rec(cx.stream, d_fplanes, nf, ldf, cx.d_rec[0].as<uint8_t>());
		launch_faces_unfold(cx.stream, nf, ldf, cx.d_rec[0].as<uint8_t>());
	}
	if (!m->lists[0].data.empty()) HIP_OK(hipMemcpyAsync(m->lists[0].data.data(), cx.d_rec[0].p, m->lists[0].data.size(), hipMemcpyDeviceToHost, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	HRY_MARK(g_t0, "records on the host");
	if (uint32_t tf = chain_timeout_flags(cx.stream)) { drop_clocks(); throw Error(HRY_E_INTERNAL, "reconstruction chain: hand-over between wavefronts timed out (flags " + std::to_string(tf) + ")"); }
	double chain_ms = 0, prep_ms = 0;
	(void)hipStreamSynchronize(cx.stream2);
	for (auto &c : clocks) {
		float t = 0;
		if (hipEventElapsedTime(&t, c.a, c.b) == hipSuccess) chain_ms += t;
		if (hipEventElapsedTime(&t, c.p0, c.p1) == hipSuccess) prep_ms += t;
		static const bool slices_only = getenv("HRY_TRACE_SLICES") != nullptr;   // (these lines alone: nothing of the trace's own waits inside the decode)
		if (trace_on() || slices_only) {   // where the slices' kernels lay on the device's clock, from the decode's first event (the payload's upload)
			float a = 0, b = 0, p0 = 0, p1 = 0;
			(void)hipEventElapsedTime(&a, cx.ev[1], c.a); (void)hipEventElapsedTime(&b, cx.ev[1], c.b);
			(void)hipEventElapsedTime(&p0, cx.ev[1], c.p0); (void)hipEventElapsedTime(&p1, cx.ev[1], c.p1);
			fprintf(stderr, "[hry]   on the device: a slice's candidates %.3f .. %.3f ms, its chain %.3f .. %.3f ms after the connectivity kernel's start\n", p0, p1, a, b);
		}
	}
	drop_clocks();
	cx.timing.k_chain_ms = chain_ms;
	cx.timing.k_predict_ms = prep_ms + chain_ms;   // candidates, chain records (second stream, beside the chain of the slice before) + the chain
	if (cx.keep_stages) {
		cx.stage_put_host("order_v", order_v.data(), order_v.size() * 4);
		cx.stage_put("ncand", d_ncand, nv);
		cx.stage_put("cand", d_cand, (size_t)nv * 6 * 4);
	}
}

Mesh *decode_chunked(Context &cx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> m)
{
	auto t_all = Clock::now();
	g_t0 = t_all;
	cx.timing = hry_timing{};
	const ListDesc ldv = m->general ? ListDesc{} : make_list_desc(m->lists[1]), ldf = m->general ? ListDesc{} : make_list_desc(m->lists[0]);
	if (m->lists.size() > (size_t)kMaxLists) throw Error(HRY_E_UNSUPPORTED, "more than 16 attribute lists");
	for (const AttrList &L : m->lists)
		for (int c = 0; c < L.ncomp(); ++c) {
			if (L.stype(c) == C_DOUBLE) throw Error(HRY_E_UNSUPPORTED, "lossless double components are outside the supported subset");
			if (m->general && kTypeSize[L.stype(c)] == 8) throw Error(HRY_E_UNSUPPORTED, "8-byte storage types are outside the supported subset");
		}
	// ---- directory
	auto need = [&](size_t off, size_t k) { if (off + k > n) throw Error(HRY_E_FORMAT, "truncated chunked directory"); };
	size_t off = hdr;
	need(off, 12);
	uint32_t CH, CHC, np;
	memcpy(&CH, p + off, 4); memcpy(&CHC, p + off + 4, 4); memcpy(&np, p + off + 8, 4);
	off += 12;
	const std::vector<GenPlane> gen_layout = m->general ? general_plane_layout(*m) : std::vector<GenPlane>();
	const uint32_t expect_planes = (uint32_t)(kConnPlanes + (m->general ? (int)gen_layout.size() : ldv.nplanes + ldf.nplanes));
	if (CH == 0 || CHC == 0 || CH > (1u << 20) || CHC > CH || np != expect_planes) throw Error(HRY_E_FORMAT, "chunked directory does not match the header");
	need(off, 4ull * np);
	std::vector<uint32_t> nsym(np);
	memcpy(nsym.data(), p + off, 4ull * np);
	off += 4ull * np;
	uint64_t nstreams = 0, total_syms = 0;
	auto step_of = [&](uint32_t k, uint64_t pos) { return k < (uint32_t)kConnPlanes ? CHC : attr_chunk_len(pos, CH); };
	for (uint32_t k = 0; k < np; ++k) {
		if (k < (uint32_t)kConnPlanes) nstreams += (nsym[k] + (uint64_t)CHC - 1) / CHC;
		else for (uint64_t f = 0; f < nsym[k]; f += step_of(k, f)) ++nstreams;
		total_syms += nsym[k];
	}
	// static prior of every plane (or none: the reference's initial counts)
	std::vector<uint32_t> prior((size_t)np * 256, 0);
	std::vector<uint8_t> has_prior(np, 0);
	for (uint32_t k = 0; k < np; ++k) {
		bool use = false;
		off += read_prior(p + off, n - off, use, prior.data() + (size_t)k * 256);
		has_prior[k] = use ? 1 : 0;
	}
	need(off, 4);
	uint32_t nrs;
	memcpy(&nrs, p + off, 4);
	off += 4;
	const bool has_snapshots = (nrs & 0x80000000u) != 0;   // (round 6: a section of border snapshots follows the restart points' counters)
	nrs &= 0x7fffffffu;
	if ((uint64_t)nrs * sizeof(RestartPoint) > n) throw Error(HRY_E_FORMAT, "truncated chunked directory");
	need(off, sizeof(RestartPoint) * (size_t)nrs);
	std::vector<RestartPoint> restarts(nrs);
	if (nrs) memcpy(restarts.data(), p + off, sizeof(RestartPoint) * (size_t)nrs);
	off += sizeof(RestartPoint) * (size_t)nrs;
	std::vector<RestartCounters> rcounters(nrs);
	for (uint32_t k = 0; k < nrs; ++k) {
		need(off, 4);
		uint32_t nc;
		memcpy(&nc, p + off, 4);
		off += 4;
		if ((uint64_t)nc * 8 > n) throw Error(HRY_E_FORMAT, "truncated chunked directory");
		need(off, 8ull * nc);
		rcounters[k].resize(nc);
		for (uint32_t j = 0; j < nc; ++j) { uint32_t v[2]; memcpy(v, p + off + 8ull * j, 8); rcounters[k][j] = { v[0], v[1] }; }
		off += 8ull * nc;
	}
	std::vector<SnapshotPoint> snaps;   // restart points inside components (host.hpp BorderSnapshot)
	if (has_snapshots) { uint32_t spacing = 0; off += read_snapshot_section(p + off, n - off, m->nv, spacing, snaps); }
	need(off, 4 * nstreams);
	std::vector<uint32_t> nbytes((size_t)nstreams);
	if (nstreams) memcpy(nbytes.data(), p + off, 4 * (size_t)nstreams);
	off += 4 * (size_t)nstreams;
	std::vector<uint64_t> offs((size_t)nstreams + 1, 0);
	for (size_t i = 0; i < nstreams; ++i) offs[i + 1] = offs[i] + nbytes[i];
	if (off + offs[nstreams] > n) throw Error(HRY_E_FORMAT, "truncated chunked payload");
	const uint8_t *payload = p + off;
	const uint64_t payload_bytes = offs[nstreams];
	// plausibility of the plane sizes against the header: at most one vertex / face record per element
	const uint32_t vc = !m->general && ldv.nplanes ? nsym[kConnPlanes] : 0;
	if (!m->general) {
		for (int q = 0; q < ldv.nplanes; ++q) if (nsym[kConnPlanes + q] != vc) throw Error(HRY_E_FORMAT, "vertex planes of different length");
		for (int q = 0; q < ldf.nplanes; ++q) if (nsym[kConnPlanes + ldv.nplanes + q] != m->nf) throw Error(HRY_E_FORMAT, "face planes of wrong length");
		if (vc > m->nv) throw Error(HRY_E_FORMAT, "more coded vertices than vertices");
	} else {
		// at most one reference per vertex / face / corner slot, at most one record per reference
		const uint64_t most = (uint64_t)m->nv * m->bind.nb_vtx + (uint64_t)m->nf * m->bind.nb_face + (uint64_t)m->declared_ne * m->bind.nb_corner + 16;
		for (size_t q = 0; q < gen_layout.size(); ++q) {
			if (nsym[kConnPlanes + q] > most) throw Error(HRY_E_FORMAT, "implausible plane length");
			if (gen_layout[q].what == GP_DATA && nsym[kConnPlanes + q] > m->lists[gen_layout[q].list].count) throw Error(HRY_E_FORMAT, "more records than the header announces");
		}
	}
	if (total_syms > (1ull << 33)) throw Error(HRY_E_FORMAT, "implausible symbol count");

	// ---- model tables: one per plane (its prior, or the reference's initial counts of its kind)
	std::vector<uint32_t> kind_tabs;
	build_init_tables(*m, kind_tabs);
	std::vector<uint32_t> tabs((size_t)np * 256, 0);
	std::vector<uint32_t> totals(np, 0);
	uint32_t max_t0 = 256;
	for (uint32_t k = 0; k < np; ++k) {
		const int kind = k < (uint32_t)kConnPlanes ? conn_init_kind((int)k) : m->general ? gen_layout[k - kConnPlanes].init : INIT_ONES;
		memcpy(tabs.data() + (size_t)k * 256, has_prior[k] ? prior.data() + (size_t)k * 256 : kind_tabs.data() + (size_t)kind * 256, 1024);
		uint64_t t = 0;
		for (int i = 0; i < 256; ++i) t += tabs[(size_t)k * 256 + i];
		if (t == 0 && nsym[k]) throw Error(HRY_E_FORMAT, "corrupt chunked directory (empty model)");
		if (t > (1u << 24)) throw Error(HRY_E_FORMAT, "corrupt chunked directory (prior total)");
		totals[k] = (uint32_t)t;
		max_t0 = std::max(max_t0, totals[k]);
	}

	// ---- device: entropy decode of every stream
	auto t_h2d = Clock::now();
	cx.d_csyms.ensure(std::max<size_t>(total_syms + 64, 16));
	cx.d_cout.ensure(std::max<size_t>(payload_bytes + 16, 16));
	cx.d_cjobs.ensure(std::max<size_t>((size_t)nstreams * sizeof(StreamJob), 16));
	cx.d_coffs.ensure(((size_t)nstreams + 1) * 8);
	cx.d_csizes.ensure(std::max<size_t>((size_t)nstreams * 4, 16));
	cx.d_init.ensure(tabs.size() * 4);
	std::vector<StreamJob> jobs;
	jobs.reserve((size_t)nstreams);
	std::vector<uint64_t> plane_off(np + 1, 0);
	for (uint32_t k = 0; k < np; ++k) {
		for (uint64_t f = 0, step; f < nsym[k]; f += step) {
			step = step_of(k, f);
			jobs.push_back(StreamJob{ cx.d_csyms.as<uint8_t>() + plane_off[k] + f, (uint32_t)std::min<uint64_t>(step, nsym[k] - f), k, totals[k], 0 });
		}
		plane_off[k + 1] = plane_off[k] + nsym[k];
	}
	// The attribute streams are launched in groups by where in their plane they END (32 Ki, 128 Ki symbols, the rest): the
	// early chunks are short (attr_chunk_len) and done within a fraction of a millisecond, and the reconstruction chain -- which
	// walks the vertices in order -- waits only for the groups it has reached.  Jobs, offsets and sizes are permuted alike.
	uint32_t n_conn_streams = 0;
	for (int k = 0; k < kConnPlanes; ++k) n_conn_streams += (uint32_t)((nsym[k] + (uint64_t)CHC - 1) / CHC);
	uint64_t kGroupEnd[Context::kAttrGroups] = { 1u << 15, 1u << 17, ~0ull };
	if (const char *e = getenv("HRY_ATTR_GROUPS")) {   // 1: one launch for all attribute streams, 2: two groups (measurements)
		const int ng = atoi(e);
		if (ng == 1) kGroupEnd[0] = kGroupEnd[1] = ~0ull;
		else if (ng == 2) { kGroupEnd[0] = 1u << 17; kGroupEnd[1] = ~0ull; }
	}
	// A large payload goes up in two parts: the connectivity streams (the container's first streams) in front of their kernel, the
	// attribute streams -- nine tenths of it -- while that kernel runs (the copy is from the caller's pageable buffer and keeps this
	// thread, which has nothing else to do until the connectivity planes are back): 289 MB of the 100 M-triangle mesh were 7.4 ms
	// in front of everything.  HRY_NO_SPLIT_UPLOAD: one copy as before; HRY_SPLIT_UPLOAD_MIN: from how many bytes (32 MB).
	const uint64_t conn_bytes = offs[std::min<size_t>(n_conn_streams, nstreams)];
	static const uint64_t split_min = [] { const char *e = getenv("HRY_SPLIT_UPLOAD_MIN"); return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)32 << 20; }();   // (tests: 1)
	const bool split_upload = payload_bytes >= split_min && conn_bytes < payload_bytes && !getenv("HRY_NO_SPLIT_UPLOAD");
	uint32_t group_n[Context::kAttrGroups] = { 0, 0, 0 };
	// ... and inside a launch the streams a lane can decode (k_chunk_decode_lanes: t0 > 128) come first; lanes_n: how many
	uint32_t conn_lanes_n = 0, group_lanes_n[Context::kAttrGroups] = { 0, 0, 0 };
	{
		std::vector<uint32_t> perm(jobs.size());
		std::vector<uint8_t> grp(jobs.size(), 0);   // 2 * group + (not for a lane)
		for (size_t j = 0; j < jobs.size(); ++j) {
			perm[j] = (uint32_t)j;
			const uint8_t slow = jobs[j].t0 > 128u ? 0 : 1;
			if (j < n_conn_streams) { grp[j] = slow; conn_lanes_n += !slow; continue; }
			const uint64_t end = (uint64_t)(jobs[j].sym - (cx.d_csyms.as<uint8_t>() + plane_off[jobs[j].init])) + jobs[j].n;   // (init holds the plane index)
			int g = 0;
			while (end > kGroupEnd[g]) ++g;
			grp[j] = (uint8_t)(2 * g + slow);
			++group_n[g];
			group_lanes_n[g] += !slow;
		}
		std::stable_sort(perm.begin(), perm.begin() + n_conn_streams, [&](uint32_t a, uint32_t b) { return grp[a] < grp[b]; });
		std::stable_sort(perm.begin() + n_conn_streams, perm.end(), [&](uint32_t a, uint32_t b) { return grp[a] < grp[b]; });
		std::vector<StreamJob> pj(jobs.size());
		std::vector<uint32_t> pn(nbytes.size());
		std::vector<uint64_t> po(offs.size());
		for (size_t j = 0; j < jobs.size(); ++j) { pj[j] = jobs[perm[j]]; pn[j] = nbytes[perm[j]]; po[j] = offs[perm[j]]; }
		po[jobs.size()] = offs[jobs.size()];
		jobs.swap(pj); nbytes.swap(pn); offs.swap(po);
	}
	HIP_OK(hipMemcpyAsync(cx.d_init.p, tabs.data(), tabs.size() * 4, hipMemcpyHostToDevice, cx.stream));
	if (payload_bytes) HIP_OK(hipMemcpyAsync(cx.d_cout.p, payload, split_upload ? conn_bytes : payload_bytes, hipMemcpyHostToDevice, cx.stream));
	if (nstreams) {
		HIP_OK(hipMemcpyAsync(cx.d_cjobs.p, jobs.data(), jobs.size() * sizeof(StreamJob), hipMemcpyHostToDevice, cx.stream));
		HIP_OK(hipMemcpyAsync(cx.d_csizes.p, nbytes.data(), nbytes.size() * 4, hipMemcpyHostToDevice, cx.stream));
	}
	HIP_OK(hipMemcpyAsync(cx.d_coffs.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, cx.stream));
	cx.ensure_magic(max_t0 + std::max(CH, CHC) + 16);
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.timing.h2d_ms = ms_since(t_h2d);
	HRY_MARK(g_t0, "payload on the device");
	// The connectivity streams go first: their planes return to the host for the replay, which then runs while the
	// attribute streams (the bulk of the payload) are still being decoded on the device.
	// ... and the attribute streams start at the same time on a stream of their own (every stream is one wavefront: the two
	// launches share the device without noticing each other); everything later on the main stream waits for them.
	if (!cx.stream3) {
		HIP_OK(hipStreamCreateWithFlags(&cx.stream3, hipStreamNonBlocking));
		for (auto &e : cx.ev_x) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		cx.attr_stream[0] = cx.stream3;
		for (int g = 1; g < Context::kAttrGroups; ++g) HIP_OK(hipStreamCreateWithFlags(&cx.attr_stream[g], hipStreamNonBlocking));
		for (auto &e : cx.attr_ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		HIP_OK(hipEventCreateWithFlags(&cx.ev_payload, hipEventDisableTiming));
	}
	HIP_OK(hipEventRecord(cx.ev_x[0], cx.stream));          // payload (its connectivity part at least), jobs and tables are on the device
	HIP_OK(hipEventRecord(cx.ev[1], cx.stream));
	// A launch takes the lane-per-stream kernel for the streams it can where that is the faster of the two (HRY_DECODE_LANES: 0
	// never, 1 always).  Measured on MI355X: a wavefront alone on its SIMD issues an instruction every ~5.5 cycles, so a lane-per-
	// stream wave takes ~1 900 cycles per step of 64 symbols however many waves there are (up to one per SIMD), and the wave-per-
	// stream kernel ~420 cycles per symbol of a stream, ~190 per symbol and SIMD once several waves share a SIMD (scalar port):
	// many short streams (the 53 000 streams of 16 Ki symbols of configs[3] under the default chunk policy, chunked.cpp) go to the
	// lanes, a few thousand long ones (a container written with 128 Ki-symbol chunks: 5 200 attribute streams) stay with a wave each.
	static const int lanes_mode = [] { const char *e = getenv("HRY_DECODE_LANES"); return e ? atoi(e) : -1; }();
	auto decode_streams = [&](hipStream_t st, uint32_t first, uint32_t n, uint32_t n_for_lanes) {
		bool lanes = lanes_mode > 0;
		// (16-bit counts where no stream's total passes 65535: 32 KB a lane-wave, five of them a compute unit; HRY_DECODE_COUNTS32: never)
		static const bool wide_only = getenv("HRY_DECODE_COUNTS32") != nullptr;
		bool counts16 = !wide_only;
		for (uint32_t j = first; j < first + n_for_lanes && counts16; ++j) counts16 = (uint64_t)jobs[j].t0 + jobs[j].n <= 65535u;
		if (lanes_mode < 0 && n_for_lanes >= 256u) {
			uint64_t total = 0, longest = 0;
			for (uint32_t j = first; j < first + n_for_lanes; ++j) { total += jobs[j].n; longest = std::max<uint64_t>(longest, jobs[j].n); }
			const double simds = 1024.0, lane_waves = (n_for_lanes + 63) / 64;
			const double lane_slots = counts16 ? 1280.0 : 512.0;   // 64 KB of counts per lane-wave: two of them in a compute unit's 160 KB of LDS (32 KB: five)
			const double t_waves = std::max((double)longest * 420.0, (double)total * 190.0 / simds);
			const double t_lanes = (double)longest * 1900.0 * std::ceil(lane_waves / lane_slots);
			lanes = t_lanes < t_waves;
		}
		const uint32_t nl = lanes ? n_for_lanes : 0u;
		if (nl) launch_chunk_decode_lanes(st, cx.d_cjobs.as<StreamJob>() + first, nl, cx.d_init.as<uint32_t>(), cx.d_magic.as<MagicEnt>(), cx.d_cout.as<uint8_t>(),
		                                  cx.d_coffs.as<uint64_t>() + first, cx.d_csizes.as<uint32_t>() + first, counts16);
		if (n > nl) launch_chunk_decode(st, cx.d_cjobs.as<StreamJob>() + first + nl, n - nl, cx.d_init.as<uint32_t>(), cx.d_magic.as<MagicEnt>(), cx.d_cout.as<uint8_t>(),
		                                cx.d_coffs.as<uint64_t>() + first + nl, cx.d_csizes.as<uint32_t>() + first + nl);
	};
	decode_streams(cx.stream, 0, n_conn_streams, conn_lanes_n);
	HIP_OK(hipEventRecord(cx.ev[2], cx.stream));
	// the connectivity planes come down into the context's pinned memory (one block, reused: fresh pageable vectors cost a zero
	// fill, a page fault per 4 KiB and a staged copy -- 7 ms of a 51 ms decode on the configs[3] share)
	PlaneView conn[kConnPlanes];
	{
		size_t at[kConnPlanes + 1] = { 0 };
		for (int k = 0; k < kConnPlanes; ++k) at[k + 1] = at[k] + ((nsym[k] + 63) & ~(size_t)63);
		cx.h_conn.ensure(std::max<size_t>(at[kConnPlanes], 64));
		for (int k = 0; k < kConnPlanes; ++k) {
			uint8_t *dst = cx.h_conn.as<uint8_t>() + at[k];
			conn[k] = PlaneView(dst, nsym[k]);
			if (nsym[k]) HIP_OK(hipMemcpyAsync(dst, cx.d_csyms.as<uint8_t>() + plane_off[k], nsym[k], hipMemcpyDeviceToHost, cx.stream));
		}
	}
	HIP_OK(hipEventRecord(cx.ev_x[2], cx.stream));        // the connectivity planes are on their way to the host
	// the attribute streams wait for the connectivity streams' kernel: launched side by side, the long attribute waves took
	// the SIMD slots the short connectivity waves needed (15 ms instead of 1 ms on a 12 M-triangle mesh), and the host replay
	// -- the critical path -- waits for exactly those
	// ... unless the mesh takes the pipelined decode (one large component, triangles replayed at ~6 ns each): there the device
	// chain, not the replay, ends the decode, and it can start only when the attribute streams are done -- side by side then
	const bool chain_bound = !m->general && restarts.empty() && nsym[7] == 0 && vc == m->nv && unpredict3_covers(ldv);
	// (round 5: not by default there either -- beside the attribute waves the 28 M-triangle torus' 30 MB of connectivity planes came
	// down in 7.4 ms instead of 1.5, in front of the replay, and its chain waits for the replay most of the time: decode 213 -> 206 ms,
	// the 1 M-triangle torus the same either way; HRY_ATTR_SIDE_BY_SIDE=1: the old order)
	(void)chain_bound;
	const bool side_by_side = getenv("HRY_ATTR_SIDE_BY_SIDE") ? atoi(getenv("HRY_ATTR_SIDE_BY_SIDE")) != 0 : false;
	// ... and for the planes' copy to the host: beside 10^5 attribute waves the copy of the configs[3] mesh's 110 MB of connectivity
	// planes took 20 ms instead of 5, in front of the replay (HRY_CONN_COPY_FIRST=0: the old order)
	static const bool copy_first = !getenv("HRY_CONN_COPY_FIRST") || atoi(getenv("HRY_CONN_COPY_FIRST")) != 0;
	hipEvent_t attr_after = side_by_side ? cx.ev_x[0] : copy_first ? cx.ev_x[2] : cx.ev[2];
	if (split_upload) {   // the rest of the payload, beside the connectivity streams' kernel (on the uploads' stream)
		cx.ensure_second_stream();
		HIP_OK(hipMemcpyAsync(cx.d_cout.as<uint8_t>() + conn_bytes, payload + conn_bytes, payload_bytes - conn_bytes, hipMemcpyHostToDevice, cx.stream2));
		HIP_OK(hipEventRecord(cx.ev_payload, cx.stream2));
		for (int g = 0; g < Context::kAttrGroups; ++g) HIP_OK(hipStreamWaitEvent(cx.attr_stream[g], cx.ev_payload, 0));
		HRY_MARK(g_t0, "attribute streams on the device");
	}
	HIP_OK(hipStreamWaitEvent(cx.stream3, attr_after, 0));
	HIP_OK(hipEventRecord(cx.ev[5], cx.stream3));
	{
		uint32_t first = n_conn_streams;
		for (int g = 0; g < Context::kAttrGroups; ++g) {
			hipStream_t st = cx.attr_stream[g];
			if (g) HIP_OK(hipStreamWaitEvent(st, attr_after, 0));
			decode_streams(st, first, group_n[g], group_lanes_n[g]);
			HIP_OK(hipEventRecord(cx.attr_ev[g], st));
			first += group_n[g];
		}
		// everything joins on stream3: "all attribute planes decoded"
		for (int g = 1; g < Context::kAttrGroups; ++g) HIP_OK(hipStreamWaitEvent(cx.stream3, cx.attr_ev[g], 0));
	}
	HIP_OK(hipEventRecord(cx.ev[6], cx.stream3));
	HIP_OK(hipEventRecord(cx.ev_x[1], cx.stream3));
	if (trace_on()) { HIP_OK(hipEventSynchronize(cx.ev[2])); HRY_MARK(g_t0, "connectivity streams decoded"); }
	HIP_OK(hipStreamSynchronize(cx.stream));
	const bool take_pipeline = !m->general && pipelined_decode_applicable(*m, restarts, conn, ldv, vc);
	if (!take_pipeline) HIP_OK(hipStreamWaitEvent(cx.stream, cx.ev_x[1], 0));   // attribute planes before anything that reads them (the pipelined decode waits group by group)

	HRY_MARK(g_t0, "connectivity planes on the host");
	// ---- replay the cut-border machine on the host
	auto t_walk = Clock::now();
	OrderVec order_v;
	std::vector<uint32_t> seg_start, seg_level;
	bool pipelined = false;
	if (m->general) {
		cut_border_replay(*m, conn, restarts, rcounters, order_v, seg_start, seg_level, nullptr, &snaps);
		cx.timing.host_walk_ms = ms_since(t_walk);
		HRY_MARK(g_t0, "replay done");
		general_planes_decode(cx, *m, order_v, seg_start, seg_level, cx.d_csyms.as<uint8_t>(), plane_off, nsym, (uint32_t)kConnPlanes);
	} else if (take_pipeline) {
		pipelined = true;
		uint32_t attr_upto[Context::kAttrGroups];
		for (int g = 0; g < Context::kAttrGroups; ++g) attr_upto[g] = (uint32_t)std::min<uint64_t>(kGroupEnd[g], 0xffffffffull);
		decode_pipelined(cx, *m, conn, cx.d_csyms.as<uint8_t>() + plane_off[kConnPlanes], cx.d_csyms.as<uint8_t>() + plane_off[kConnPlanes + ldv.nplanes], ldv, ldf, order_v, attr_upto, snaps);
		cx.timing.host_walk_ms = cx.timing.host_walk_ms - std::chrono::duration<double, std::milli>(t_walk - g_t0).count();
	} else {
		bool conn_resident = false;
		std::unique_ptr<ChainBatches> batches;
		if (!restarts.empty() && rcounters.size() == restarts.size() && m->nf >= (1u << 20) && !getenv("HRY_NO_SPAN_UPLOAD")) {
			// float / 32-bit vertex components: their chains start beside the replay, batch by batch (ChainBatches)
			// ... where that pays: a batch is a launch of its own on the main stream, and a launch takes as long as its longest
			// chain (one wavefront, one component: 7 - 14 ms for the 49 000 vertices of a configs[3] component), so three batches of a
			// mesh whose chains all fit on the device at once (the 12.6 M-triangle share: 384 chains) took 36 ms where one launch takes 15;
			// the 100 M-triangle mesh's 3 072 chains need two rounds anyway and finish 25 ms earlier in batches.
			// HRY_CHAIN_BATCH_MIN_VERTICES: the threshold (tests run small meshes in batches).
			const char *bm = getenv("HRY_CHAIN_BATCH_MIN_VERTICES");
			const uint32_t batch_min = bm ? (uint32_t)strtoul(bm, nullptr, 10) : 20000000u;
			const bool in_batches = ldv.nplanes && unpredict2_applicable(ldv) && !unpredict3_wanted(ldv) && vc == m->nv && vc >= batch_min && !getenv("HRY_NO_CHAIN_BATCHES");
			if (in_batches) batches.reset(new ChainBatches(cx, ldv, cx.d_csyms.as<uint8_t>() + plane_off[kConnPlanes], vc, nsym[0]));
			SpanUploader up(cx, *m, order_v, batches.get(), cx.ev_x[1]);
			cut_border_replay(*m, conn, restarts, rcounters, order_v, seg_start, seg_level, &up, &snaps);
			conn_resident = up.finish();
			if (up.error) std::rethrow_exception(up.error);
			if (batches && conn_resident) { up.finish_edge_faces(cx.stream); batches->conn_adopted = true; }
		} else cut_border_replay(*m, conn, restarts, rcounters, order_v, seg_start, seg_level, nullptr, &snaps);
		cx.timing.host_walk_ms = ms_since(t_walk);
		HRY_MARK(g_t0, "replay done");
		if (order_v.size() != vc && ldv.nplanes) throw Error(HRY_E_FORMAT, "vertex plane length does not match the connectivity");
		reconstruct_attributes(cx, *m, order_v, seg_start, seg_level, cx.d_csyms.as<uint8_t>() + plane_off[kConnPlanes],
		                       cx.d_csyms.as<uint8_t>() + plane_off[kConnPlanes + ldv.nplanes], ldv, ldf, conn_resident, nullptr, batches.get());
	}
	if (cx.keep_stages) {
		cx.stage_put("dec_syms", cx.d_csyms.p, total_syms);
		cx.stage_put_host("dec_nsym", nsym.data(), nsym.size() * 4);
	}
	cx.timing.k_entropy_ms = cx.elapsed(1, 2) + cx.elapsed(5, 6);
	if (!pipelined) cx.timing.k_predict_ms = cx.elapsed(3, 4);
	cx.timing.device_ms = cx.timing.k_entropy_ms + cx.timing.k_predict_ms;
	cx.timing.n_symbols = total_syms;
	cx.timing.payload_bytes = payload_bytes;
	cx.timing.total_ms = ms_since(t_all);
	m->device_token = 0;   // the resident copy belongs to this context only until the next upload
	return m.release();
}

// General bindings (general.cpp) whose vertices all carry a private record of ONE list: that list is the vertex list of the PLY
// layout in everything that matters to the reconstruction chains (record i belongs to the i-th coded vertex, candidates are the
// parallelograms of the fan), so it takes them.  The mesh lends its connectivity and the list for the duration of the call.
bool reconstruct_vertex_list_fast(Context &cx, Mesh &m, int l, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                                  const std::vector<uint32_t> &seg_level, const std::vector<uint8_t> &vplanes, const uint8_t *d_vplanes)
{
	const ListDesc ldv = make_list_desc(m.lists[l]);
	if (!ldv.nplanes || !unpredict2_applicable(ldv) || m.lists[l].count < order_v.size()) return false;
	Mesh t;
	t.nv = m.nv; t.nf = m.nf; t.declared_ne = m.declared_ne; t.have_degree = m.have_degree;
	t.face_off.swap(m.face_off); t.org.swap(m.org); t.twin.swap(m.twin);
	std::swap(t.lists[1], m.lists[l]);
	struct Back {   // returned on every path
		Mesh &m, &t; int l;
		~Back() { t.face_off.swap(m.face_off); t.org.swap(m.org); t.twin.swap(m.twin); std::swap(t.lists[1], m.lists[l]); }
	} back{ m, t, l };
	if (!d_vplanes) {   // planes from the host (reference stream); a chunked container's are in HBM already
		cx.d_csyms.ensure(std::max<size_t>(vplanes.size() + 64, 16));
		if (!vplanes.empty()) HIP_OK(hipMemcpyAsync(cx.d_csyms.p, vplanes.data(), vplanes.size(), hipMemcpyHostToDevice, cx.stream));
		HIP_OK(hipStreamSynchronize(cx.stream));
		d_vplanes = cx.d_csyms.as<uint8_t>();
	}
	reconstruct_attributes(cx, t, order_v, seg_start, seg_level, d_vplanes, nullptr, ldv, make_list_desc(t.lists[0]));
	return true;
}

// The same for a caller that keeps working on the mesh meanwhile (general_planes_decode: the host's bookkeeping of the other lists
// runs beside the vertex chain): `t` holds the sizes and the vertex list's records (moved in by the caller, moved back by it
// afterwards); the connectivity is read from `conn` -- lent, not copied: both threads only read it.  Runs on the calling thread,
// which may be a helper.
long long trace_origin_ns() { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(g_t0.time_since_epoch()).count(); }   // of the calling thread's decode
bool vertex_list_fast_applicable(const Mesh &m, int l, size_t n_order)
{
	const ListDesc ldv = make_list_desc(m.lists[l]);
	return ldv.nplanes && unpredict2_applicable(ldv) && m.lists[l].count >= n_order;
}
void reconstruct_vertex_list_detached(Context &cx, Mesh &t, const Mesh &conn, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                                      const std::vector<uint32_t> &seg_level, const uint8_t *d_vplanes, long long trace_origin)
{
	HIP_OK(hipSetDevice(cx.device));
	g_t0 = Clock::time_point(std::chrono::duration_cast<Clock::duration>(std::chrono::nanoseconds(trace_origin)));   // (this thread's copy: HRY_TRACE's timeline)
	reconstruct_attributes(cx, t, order_v, seg_start, seg_level, d_vplanes, nullptr, make_list_desc(t.lists[1]), make_list_desc(t.lists[0]), false, &conn);
}

// Reference format (.hry v0.1): the single adaptive stream is decoded and replayed on a host core (the format makes
// both serial, compat_read.cpp); the residual planes then take the same device reconstruction as above.
Mesh *decode_compat(Context &cx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> m)
{
	auto t_all = Clock::now();
	g_t0 = t_all;
	cx.timing = hry_timing{};
	const ListDesc ldv = make_list_desc(m->lists[1]), ldf = make_list_desc(m->lists[0]);
	for (int l = 0; l < 2; ++l)
		for (int c = 0; c < m->lists[l].ncomp(); ++c)
			if (m->lists[l].stype(c) == C_DOUBLE) throw Error(HRY_E_UNSUPPORTED, "lossless double components are outside the supported subset");
	auto t_walk = Clock::now();
	OrderVec order_v;
	std::vector<uint32_t> seg_start, seg_level;
	std::vector<uint8_t> vplanes, fplanes;
	read_compat_stream(p + hdr, n - hdr, *m, order_v, seg_start, seg_level, vplanes, fplanes);
	cx.timing.host_walk_ms = ms_since(t_walk);
	auto t_h2d = Clock::now();
	cx.d_csyms.ensure(std::max<size_t>(vplanes.size() + fplanes.size() + 64, 16));
	uint8_t *d_v = cx.d_csyms.as<uint8_t>(), *d_f = d_v + vplanes.size();
	if (!vplanes.empty()) HIP_OK(hipMemcpyAsync(d_v, vplanes.data(), vplanes.size(), hipMemcpyHostToDevice, cx.stream));
	if (!fplanes.empty()) HIP_OK(hipMemcpyAsync(d_f, fplanes.data(), fplanes.size(), hipMemcpyHostToDevice, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.timing.h2d_ms = ms_since(t_h2d);
	reconstruct_attributes(cx, *m, order_v, seg_start, seg_level, d_v, d_f, ldv, ldf);
	cx.timing.k_predict_ms = cx.elapsed(3, 4);
	cx.timing.device_ms = cx.timing.k_predict_ms;
	cx.timing.n_symbols = vplanes.size() + fplanes.size();
	cx.timing.payload_bytes = n - hdr;
	cx.timing.total_ms = ms_since(t_all);
	m->device_token = 0;
	return m.release();
}

}   // namespace hry
