// Chunked profile (.hry v0.2) pipeline: encode and decode.
//
// Container (after the v0.1-compatible header with minor version 2):
//     u32 chunk_syms, u32 conn_chunk_syms, u32 n_planes, n_planes x u32 n_symbols, n_planes x static prior, restart points,
//     n_streams x u32 n_bytes, streams back to back
// Plane order: iop, elem[4], part[2], vertid[4], numtri[2], op class[8], vertex data bytes, face data bytes.
// Every (plane, chunk of chunk_syms symbols) is one stream: fresh adaptive model (the reference's initial counts,
// models.h:197-218 / model.h:38-55), fresh coder with 32-bit registers (arith::Encoder<uint32_t>), 32-bit flush (arith/coder.h).  Symbols that carry no
// information are not stored (reg_face / reg_vtx with a single region, attr_type == DATA, numtri with one degree).
// The symbols themselves are those of the compat stream, so the two profiles transcode losslessly.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "codec_math.hpp"
#include "context.hpp"
#include "kernels.hpp"

namespace hry {

using namespace dev;
typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
// HRY_TRACE=1: wall-clock marks of the encode on stderr (development aid; the decode's are in unchunk.cpp)
static bool trace_on() { static const bool on = getenv("HRY_TRACE") != nullptr; return on; }
#define HRY_MARK(t0, what) do { if (trace_on()) fprintf(stderr, "[hry enc] %8.3f ms  %s\n", ms_since(t0), what); } while (0)

void build_init_tables(const Mesh &m, std::vector<uint32_t> &tabs)
{
	tabs.assign((size_t)INIT_KINDS * 256, 0);
	for (int i = 0; i < 256; ++i) tabs[INIT_ONES * 256 + i] = 1;
	for (int i = 0; i < 9; ++i) tabs[INIT_IOP * 256 + i] = 1;
	for (size_t d = 3; d < m.have_degree.size(); ++d)
		if (m.have_degree[d]) { ++tabs[INIT_NT0 * 256 + ((d - 2) & 0xff)]; ++tabs[INIT_NT1 * 256 + ((d - 2) >> 8)]; }
	for (int i = 0; i < 7; ++i) tabs[INIT_OP * 256 + i] = 1;
	if (m.general) {   // models.h:201-203,212-217
		for (int r = 0; r < m.bind.nregs_vtx() && r < 256; ++r) tabs[INIT_REGV * 256 + r] = 1;
		for (int r = 0; r < m.bind.nregs_face() && r < 256; ++r) tabs[INIT_REGF * 256 + r] = 1;
	}
	tabs[INIT_TYPE2 * 256 + 0] = tabs[INIT_TYPE2 * 256 + 1] = 1;
	tabs[INIT_TYPE3 * 256 + 0] = tabs[INIT_TYPE3 * 256 + 1] = tabs[INIT_TYPE3 * 256 + 2] = 1;
}

static constexpr int kDefaultChunk = 8192;   // with static priors a fresh table per chunk costs little: short chunks = short serial chains
static constexpr uint32_t kMaxChunk = 1u << 20;   // totals stay below 2^21: far inside the 32-bit coder's t <= 2^30 (oracle: same clamp)


// the plane list of a mesh in container order; device pointers are filled by the caller
static const int kConnPlanes = 1 + 4 + 2 + 4 + 2 + 8;
static int plane_init_kind(int conn_index)
{
	if (conn_index == 0) return INIT_IOP;
	if (conn_index == 11) return INIT_NT0;
	if (conn_index == 12) return INIT_NT1;
	if (conn_index >= 13) return INIT_OP;
	return INIT_ONES;
}
static uint32_t stream_words(uint32_t n, uint32_t t0)
{
	uint32_t tmax = t0 + n, lg = 0;
	while ((2u << lg) <= tmax) ++lg;   // floor(log2(tmax))
	uint64_t bits = (uint64_t)n * (lg + 2) + 64;
	return (uint32_t)(bits / 32 + 4);
}

// faces from which the components are analysed on the device (HRY_DEVICE_ANALYSIS_MIN_FACES; 0xffffffff = never): below, the
// sequential walk of the first component and the host's passes over the rest cost less than the launches and round trips
static uint32_t device_analysis_min_faces()
{
	const char *e = getenv("HRY_DEVICE_ANALYSIS_MIN_FACES");   // (read per call: the tests change it)
	return e ? (uint32_t)strtoul(e, nullptr, 10) : (4u << 20);
}

namespace dev { void launch_scatter_u32(hipStream_t st, const uint32_t *pairs, uint32_t n, uint32_t *dst); }

// ---------------------------------------------------------------------------------------------------------
// The device side of an encode BESIDE the walk (round 5).  A mesh of many components is walked on the host threads group by
// group (cbm_walk.cpp: walk_components_parallel), and what a finished group has coded is final: its runs of the coding order, the
// polygons' triangle counts, the twins it repaired (cbm/encoder.h:150,193-198).  The attribute pass of the reference runs after
// the whole walk (writer.cc:210-212) but reads only "is vertex x coded before vertex v" (attrcode.h:119,218) -- rank[x] < rank[v],
// and the fan of v never leaves v's group.  So the walk's output arrays are registered with the runtime for the length of the
// walk (the copy engines read them where they lie: 30 ms for 500 MB, on the sending thread, beside the first walks), and
//   * the walking thread that finishes a group notes the group's runs in the open SLOT of a few pinned staging slots: LONG runs
//     (a component's thousands of vertices / faces) by where they begin, SHORT ones (the slivers around it: a few entries each,
//     45 000 runs on the configs[3] mesh) with their entries copied back to back into the slot, and the repaired twins as
//     (half-edge, twin) pairs;
//   * a thread of its own sends every closed slot up -- a copy per long run straight from the walk's arrays, one per slot region
//     -- scatters the pairs into the resident twins and runs k_rank, k_predict_vtx, k_face_planes and the triangle counts' byte
//     split over the slot's runs (kernels.hip: *_runs) on a stream of its own.
// When the walk returns, the vertex and face planes are complete but for the open slot; what is left behind the walk is what
// needs the WHOLE planes: their histograms (the static priors), the streams, the container.  At 100 M triangles: uploads 12 ms
// + planes 15 ms of the 66 ms that followed the walk.
// The walk is bound by the 16 CPUs the box grants (2 CPU-seconds), so whatever the pipeline makes a CPU do comes back as walk
// time.  Versions on the way: one copy per run, short ones too -- 195 ms of copy calls for a batch; every run gathered by the
// sending thread -- 500 MB through a 17th busy thread, the walks 122 -> 140 ms; gathered by the walkers themselves from their
// caches -- 137 ms: the copy is 0.2 CPU-seconds wherever it runs, and ate what the pipeline saved.
// A group with more short runs or repaired twins than a slot holds, and every group that finds no slot free, is handed to the
// sending thread as a list of runs; it takes it in slot-sized pieces, every vertex ranked before the first fan is walked.
// ---------------------------------------------------------------------------------------------------------
struct EncodePipeline : WalkProgress {
	Context &cx;
	const ConnView cv;
	const ListDesc ldv, ldf;
	const uint32_t vc, fc, dev_nv;
	std::function<void()> arrays_ready;        // a shard coded in place: returns when its intervals are in HBM (called once, before the first kernel)
	const uint32_t *ov = nullptr, *of = nullptr, *nt = nullptr;
	bool began = false, want_f = false;
	Clock::time_point t0;

	// ---- slots.  Tables: G = runs whose entries are gathered in the slot, D = runs copied from where they lie; v / f
	uint32_t elems = 0;                        // entries a gathered region holds
	uint32_t direct_min = 2048;                // a run of this many entries is copied from where it lies (registered arrays only)
	static constexpr uint32_t kRuns = 1u << 15, kPairWords = 1u << 15;
	enum { TGV = 0, TDV, TGF, TDF, kTables };
	size_t off_start[kTables] = {}, off_first[kTables] = {}, off_pairs = 0, off_a = 0, off_b = 0, off_c = 0, slot_words = 0;
	enum { FREE = 0, OPEN, CLOSED, INFLIGHT };
	enum { NORMAL = 0, RANK_ONLY, PLANES_ONLY };
	struct Slot { int state = FREE, mode = NORMAL; uint32_t n[kTables] = {}, runs[kTables] = {}, npairs = 0, writers = 0; uint64_t seq = 0; };
	Slot slot[Context::kPipeSlots];
	int open_slot = -1;
	uint64_t next_seq = 0;
	struct Big { std::vector<uint32_t> v, f, pairs; };   // a group for the sending thread: its runs
	std::deque<Big> big;
	uint64_t min_batch = 0;
	std::mutex mu;
	std::condition_variable cond_sender;
	bool closing = false;
	std::thread th;
	std::exception_ptr err;
	bool kernels_ok = false, registered[3] = { false, false, false };
	uint32_t n_sent = 0;

	EncodePipeline(Context &c, const ListDesc &v, const ListDesc &f, uint32_t n_v, uint32_t n_f, uint32_t nv_dev, std::function<void()> ready, Clock::time_point origin)
	    : cx(c), cv(c.conn_view()), ldv(v), ldf(f), vc(n_v), fc(n_f), dev_nv(nv_dev), arrays_ready(std::move(ready)), t0(origin) {}
	~EncodePipeline()
	{
		if (th.joinable()) { { std::lock_guard<std::mutex> g(mu); closing = true; } cond_sender.notify_all(); th.join(); }
		if (registered[0] || registered[1] || registered[2]) (void)hipStreamSynchronize(cx.pipe_stream);   // (an error path: no copy may still read the arrays)
		unregister_arrays();
	}
	static bool wanted() { const char *e = getenv("HRY_NO_ENCODE_PIPELINE"); return !(e && *e && *e != '0'); }

	void prepare_slots()
	{
		const char *e = getenv("HRY_ENCODE_PIPELINE_BATCH");   // vertices per batch at least (tests: 1)
		min_batch = e ? strtoull(e, nullptr, 10) : std::max<uint64_t>(1u << 16, vc / 48);
		const char *se = getenv("HRY_ENCODE_PIPELINE_SLOT");   // entries per gathered region (tests: small, so that groups outgrow it)
		elems = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(se ? strtoull(se, nullptr, 10) : (1u << 18), 64), 2u << 20);
		const char *de = getenv("HRY_ENCODE_PIPELINE_DIRECT");   // shortest run copied from where it lies (tests: 1 = every run, 0 = none)
		if (de) direct_min = (uint32_t)strtoul(de, nullptr, 10);
		if (direct_min == 0) direct_min = 0xffffffffu;
		size_t at = 0;
		for (int t = 0; t < kTables; ++t) { off_start[t] = at; at += kRuns + 2; off_first[t] = at; at += kRuns; }
		off_pairs = at; off_a = off_pairs + kPairWords; off_b = off_a + elems; off_c = off_b + elems; slot_words = off_c + elems;
		cx.h_pipe.ensure(slot_words * 4 * Context::kPipeSlots);
		cx.d_pipe.ensure(slot_words * 4 * Context::kPipeSlots);
		for (auto &ev : cx.pipe_slot_ev) if (!ev) HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	}
	uint32_t *host_slot(int i) const { return cx.h_pipe.as<uint32_t>() + (size_t)i * slot_words; }
	uint32_t *dev_slot(int i) const { return cx.d_pipe.as<uint32_t>() + (size_t)i * slot_words; }

	void begin(const uint32_t *order_v, size_t n_v, const uint32_t *order_f, size_t n_f, const uint32_t *numtri) override
	{
		if (n_v != vc || n_f != fc) throw Error(HRY_E_INTERNAL, "encode pipeline: the walk codes other counts than the analysis announced");
		ov = order_v; of = order_f; nt = numtri;
		want_f = ldf.nplanes > 0 || nt != nullptr;
		began = true;
		th = std::thread([this] { try { run(); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); } });
	}
	// the walk's arrays become readable by the copy engines where they lie (sending thread, first thing); where the runtime
	// refuses, every run is gathered
	void register_arrays()
	{
		if (direct_min == 0xffffffffu) return;
		const void *p[3] = { ov, ldf.nplanes ? of : nullptr, nt };
		const size_t n[3] = { (size_t)vc * 4, (size_t)fc * 4, (size_t)fc * 4 };
		bool ok = true;
		for (int i = 0; i < 3 && ok; ++i)
			if (p[i] && n[i]) { ok = hipHostRegister(const_cast<void*>(p[i]), n[i], hipHostRegisterPortable) == hipSuccess; registered[i] = ok; }
		if (!ok) { (void)hipGetLastError(); unregister_arrays(); std::lock_guard<std::mutex> g(mu); direct_min = 0xffffffffu; }
		if (trace_on()) fprintf(stderr, "[hry enc] %8.3f ms    pipeline: the walk's arrays %s\n", ms_since(t0), ok ? "registered" : "could not be registered: every run is gathered");
	}
	void unregister_arrays()
	{
		const void *p[3] = { ov, of, nt };
		for (int i = 0; i < 3; ++i) if (registered[i]) { (void)hipHostUnregister(const_cast<void*>(p[i])); registered[i] = false; }
	}

	// what a list of runs asks of a slot: per table the runs and (gathered tables) the entries
	struct Need { uint32_t runs[kTables] = {}, n[kTables] = {}, pairs = 0; };
	Need need_of(const uint32_t *v_runs, uint32_t n_v_runs, const uint32_t *f_runs, uint32_t n_f_runs, uint32_t pair_words, uint32_t dmin) const
	{
		Need q;
		for (uint32_t i = 0; i < n_v_runs; ++i) { const int t = v_runs[2 * i + 1] >= dmin ? TDV : TGV; ++q.runs[t]; q.n[t] += v_runs[2 * i + 1]; }
		for (uint32_t i = 0; i < n_f_runs; ++i) { const int t = f_runs[2 * i + 1] >= dmin ? TDF : TGF; ++q.runs[t]; q.n[t] += f_runs[2 * i + 1]; }
		q.pairs = pair_words;
		return q;
	}
	bool fits(const Slot &s, const Need &q) const
	{
		for (int t = 0; t < kTables; ++t) if (s.runs[t] + q.runs[t] > kRuns) return false;
		return s.n[TGV] + q.n[TGV] <= elems && s.n[TGF] + q.n[TGF] <= elems && (uint64_t)s.n[TDV] + q.n[TDV] < (1ull << 31) && (uint64_t)s.n[TDF] + q.n[TDF] < (1ull << 31) &&
		       s.npairs + q.pairs <= kPairWords;
	}
	bool fits_empty(const Need &q) const { return fits(Slot(), q); }
	// (under mu) the open slot, with room for the request; closes a slot that lacks it and opens a free one; -1: none free
	int slot_with_room(const Need &q, int mode)
	{
		for (;;) {
			if (err) std::rethrow_exception(err);
			if (open_slot >= 0) {
				Slot &s = slot[open_slot];
				if (s.mode == mode && fits(s, q)) return open_slot;
				s.state = CLOSED; open_slot = -1;
				cond_sender.notify_one();
			}
			for (int i = 0; i < Context::kPipeSlots; ++i)
				if (slot[i].state == FREE) { slot[i] = Slot(); slot[i].state = OPEN; slot[i].mode = mode; slot[i].seq = next_seq++; open_slot = i; break; }
			if (open_slot < 0) return -1;   // every slot is on its way (or the sender is still waiting for a shard's arrays): a walker never waits
		}
	}
	// (under mu) the request's place in slot sl
	struct Reservation { int slot; uint32_t run0[kTables], n0[kTables], p0; };
	Reservation reserve(int sl, const Need &q)
	{
		Slot &s = slot[sl];
		Reservation r;
		r.slot = sl; r.p0 = s.npairs;
		for (int t = 0; t < kTables; ++t) { r.run0[t] = s.runs[t]; r.n0[t] = s.n[t]; s.runs[t] += q.runs[t]; s.n[t] += q.n[t]; }
		s.npairs += q.pairs;
		return r;
	}
	// the runs into the slot's tables, short ones with their entries (payload: false for a pass that needs the tables only)
	void copy_runs(const Reservation &r, const uint32_t *v_runs, uint32_t n_v_runs, const uint32_t *f_runs, uint32_t n_f_runs, const uint32_t *pairs, uint32_t pair_words,
	               uint32_t dmin, bool payload_v, bool payload_f)
	{
		uint32_t *h = host_slot(r.slot);
		uint32_t run[kTables], at[kTables];
		for (int t = 0; t < kTables; ++t) { run[t] = r.run0[t]; at[t] = r.n0[t]; }
		for (uint32_t i = 0; i < n_v_runs; ++i) {
			const uint32_t first = v_runs[2 * i], n = v_runs[2 * i + 1];
			const int t = n >= dmin ? TDV : TGV;
			h[off_start[t] + run[t]] = at[t]; h[off_first[t] + run[t]] = first;
			if (t == TGV && payload_v) memcpy(h + off_a + at[t], ov + first, (size_t)n * 4);
			at[t] += n; ++run[t];
		}
		for (uint32_t i = 0; i < n_f_runs; ++i) {
			const uint32_t first = f_runs[2 * i], n = f_runs[2 * i + 1];
			const int t = n >= dmin ? TDF : TGF;
			h[off_start[t] + run[t]] = at[t]; h[off_first[t] + run[t]] = first;
			if (t == TGF && payload_f) {
				if (ldf.nplanes) memcpy(h + off_b + at[t], of + first, (size_t)n * 4);
				if (nt) memcpy(h + off_c + at[t], nt + first, (size_t)n * 4);
			}
			at[t] += n; ++run[t];
		}
		if (pair_words) memcpy(h + off_pairs + r.p0, pairs, (size_t)pair_words * 4);
	}
	void group_done(const uint32_t *v_runs, uint32_t n_v_runs, const uint32_t *f_runs, uint32_t n_f_runs, const uint32_t *twin_pairs, uint32_t n_pairs) override
	{
		if (!want_f) n_f_runs = 0;
		const uint32_t pw = 2 * n_pairs;
		auto hand_over = [&] {   // the sender takes the group from the walk's arrays, in pieces, when it gets to it
			Big bg;
			bg.v.assign(v_runs, v_runs + 2 * (size_t)n_v_runs); bg.f.assign(f_runs, f_runs + 2 * (size_t)n_f_runs); bg.pairs.assign(twin_pairs, twin_pairs + pw);
			{ std::lock_guard<std::mutex> g(mu); big.push_back(std::move(bg)); }
			cond_sender.notify_one();
		};
		Reservation r;
		uint32_t dmin;
		{
			std::unique_lock<std::mutex> g(mu);
			dmin = direct_min;
			const Need q = need_of(v_runs, n_v_runs, f_runs, n_f_runs, pw, dmin);
			const int sl = fits_empty(q) ? slot_with_room(q, NORMAL) : -1;
			if (sl < 0) { g.unlock(); hand_over(); return; }   // larger than a slot, or no slot free: the walk goes on
			r = reserve(sl, q);
			Slot &s = slot[sl];
			++s.writers;
			if ((uint64_t)s.n[TGV] + s.n[TDV] >= min_batch) { s.state = CLOSED; open_slot = -1; }   // (this writer's copy is still to come: the sender waits for writers == 0)
		}
		copy_runs(r, v_runs, n_v_runs, f_runs, n_f_runs, twin_pairs, pw, dmin, true, true);
		bool wake;
		{ std::lock_guard<std::mutex> g(mu); Slot &s = slot[r.slot]; --s.writers; wake = s.state == CLOSED && s.writers == 0; }
		if (wake) cond_sender.notify_one();
	}
	// after the walk: the open slot and whatever is queued, then the main stream waits for the pipeline's
	void finish()
	{
		if (began) {
			{ std::lock_guard<std::mutex> g(mu); closing = true; if (open_slot >= 0) { slot[open_slot].state = CLOSED; open_slot = -1; } }
			cond_sender.notify_all();
			th.join();
			unregister_arrays();
			if (err) std::rethrow_exception(err);
		}
		HIP_OK(hipEventRecord(cx.pipe_ev, cx.pipe_stream));
		HIP_OK(hipStreamWaitEvent(cx.stream, cx.pipe_ev, 0));
	}

	// ---- the sending thread
	void need_arrays() { if (!kernels_ok) { if (arrays_ready) arrays_ready(); kernels_ok = true; } }   // (a shard in place: its connectivity and records are on their way up)
	// a closed slot without writers: copies + kernels, then the slot is on its way (INFLIGHT until its event)
	void send(int sl)
	{
		hipStream_t st = cx.pipe_stream;
		const Slot s = slot[sl];
		uint32_t *h = host_slot(sl), *d = dev_slot(sl);
		for (int t = 0; t < kTables; ++t) {
			if (!s.runs[t]) continue;
			h[off_start[t] + s.runs[t]] = s.n[t];
			HIP_OK(hipMemcpyAsync(d + off_start[t], h + off_start[t], ((size_t)s.runs[t] + 1) * 4, hipMemcpyHostToDevice, st));
			HIP_OK(hipMemcpyAsync(d + off_first[t], h + off_first[t], (size_t)s.runs[t] * 4, hipMemcpyHostToDevice, st));
		}
		// the short runs' entries; the long runs from where they lie
		if (s.n[TGV] && s.mode != PLANES_ONLY) HIP_OK(hipMemcpyAsync(d + off_a, h + off_a, (size_t)s.n[TGV] * 4, hipMemcpyHostToDevice, st));
		if (s.n[TGF] && s.mode != RANK_ONLY) {
			if (ldf.nplanes) HIP_OK(hipMemcpyAsync(d + off_b, h + off_b, (size_t)s.n[TGF] * 4, hipMemcpyHostToDevice, st));
			if (nt) HIP_OK(hipMemcpyAsync(d + off_c, h + off_c, (size_t)s.n[TGF] * 4, hipMemcpyHostToDevice, st));
		}
		if (s.mode != PLANES_ONLY)
			for (uint32_t i = 0; i < s.runs[TDV]; ++i) {
				const uint32_t first = h[off_first[TDV] + i], n = h[off_start[TDV] + i + 1] - h[off_start[TDV] + i];
				HIP_OK(hipMemcpyAsync(cx.d_order_v.as<uint32_t>() + first, ov + first, (size_t)n * 4, hipMemcpyHostToDevice, st));
			}
		if (s.mode != RANK_ONLY)
			for (uint32_t i = 0; i < s.runs[TDF]; ++i) {
				const uint32_t first = h[off_first[TDF] + i], n = h[off_start[TDF] + i + 1] - h[off_start[TDF] + i];
				if (ldf.nplanes) HIP_OK(hipMemcpyAsync(cx.d_order_f.as<uint32_t>() + first, of + first, (size_t)n * 4, hipMemcpyHostToDevice, st));
				if (nt) HIP_OK(hipMemcpyAsync(cx.d_nt_val.as<uint32_t>() + first, nt + first, (size_t)n * 4, hipMemcpyHostToDevice, st));
			}
		if (s.npairs) HIP_OK(hipMemcpyAsync(d + off_pairs, h + off_pairs, (size_t)s.npairs * 4, hipMemcpyHostToDevice, st));
		need_arrays();
		if (s.npairs) dev::launch_scatter_u32(st, d + off_pairs, s.npairs / 2, cx.d_twin.as<uint32_t>());   // (before anything walks a fan of these groups)
		uint32_t *dov = cx.d_order_v.as<uint32_t>(), *dof = cx.d_order_f.as<uint32_t>();
		if (s.mode != PLANES_ONLY) {   // every vertex of the slot has its rank before the first fan is walked
			launch_rank_runs(st, d + off_start[TGV], d + off_first[TGV], s.runs[TGV], s.n[TGV], d + off_a, dov, cv.org, cx.d_rank.as<uint32_t>());
			launch_rank_runs(st, d + off_start[TDV], d + off_first[TDV], s.runs[TDV], s.n[TDV], nullptr, dov, cv.org, cx.d_rank.as<uint32_t>());
		}
		if (s.mode != RANK_ONLY) {
			for (int t : { TGV, TDV })
				launch_predict_vtx_runs(st, cv, d + off_start[t], d + off_first[t], s.runs[t], s.n[t], dov, vc, cx.d_rank.as<uint32_t>(), cx.d_rec[1].as<uint8_t>(), ldv, cx.d_vplanes.as<uint8_t>());
			for (int t : { TGF, TDF }) {
				if (ldf.nplanes) launch_face_planes_runs(st, cv, d + off_start[t], d + off_first[t], s.runs[t], s.n[t], t == TGF ? d + off_b : nullptr, dof, fc, cx.d_rec[0].as<uint8_t>(), ldf, cx.d_fplanes.as<uint8_t>());
				if (nt) launch_split_bytes_runs(st, d + off_start[t], d + off_first[t], s.runs[t], s.n[t], t == TGF ? d + off_c : nullptr, cx.d_nt_val.as<uint32_t>(), fc, kGroupBytes[G_NUMTRI], cx.d_nt_planes.as<uint8_t>());
			}
		}
		HIP_OK(hipEventRecord(cx.pipe_slot_ev[sl], st));
		++n_sent;
		if (trace_on()) fprintf(stderr, "[hry enc] %8.3f ms    pipeline: slot %u, vertices %u in %u short runs + %u in %u long ones, faces %u in %u + %u in %u, %u repaired twins%s\n", ms_since(t0), n_sent,
		                        s.n[TGV], s.runs[TGV], s.n[TDV], s.runs[TDV], s.n[TGF], s.runs[TGF], s.n[TDF], s.runs[TDF], s.npairs / 2,
		                        s.mode == RANK_ONLY ? " (ranks of a group taken in pieces)" : s.mode == PLANES_ONLY ? " (planes of a group taken in pieces)" : "");
	}
	// a group handed over as a list of runs.  One that fits a slot joins the open slot like a walker's (the sender may wait for a
	// slot); a larger one goes in slot-sized pieces: first every vertex' rank (and the repaired twins), then the planes
	void send_big(const Big &bg)
	{
		uint32_t dmin;
		{ std::lock_guard<std::mutex> g(mu); dmin = direct_min; }
		const uint32_t nrv0 = (uint32_t)(bg.v.size() / 2), nrf0 = (uint32_t)(bg.f.size() / 2), pw0 = (uint32_t)bg.pairs.size();
		const Need q0 = need_of(bg.v.data(), nrv0, bg.f.data(), nrf0, pw0, dmin);
		if (fits_empty(q0)) {
			Reservation r;
			for (;;) {
				{
					std::unique_lock<std::mutex> g(mu);
					const int sl = slot_with_room(q0, NORMAL);
					if (sl >= 0) {
						r = reserve(sl, q0);
						Slot &s = slot[sl];
						++s.writers;
						if ((uint64_t)s.n[TGV] + s.n[TDV] >= min_batch) { s.state = CLOSED; open_slot = -1; }
						break;
					}
				}
				flush_closed();
				free_oldest(true);
			}
			copy_runs(r, bg.v.data(), nrv0, bg.f.data(), nrf0, bg.pairs.data(), pw0, dmin, true, true);
			{ std::lock_guard<std::mutex> g(mu); --slot[r.slot].writers; }
			return;
		}
		for (int pass = 0; pass < 2; ++pass) {
			const int mode = pass == 0 ? RANK_ONLY : PLANES_ONLY;
			size_t iv = 0, jf = 0, ip = 0;
			uint32_t tv = 0, tf = 0;   // entries of run iv / jf that earlier pieces took
			const size_t nrv = bg.v.size() / 2, nrf = pass == 1 ? bg.f.size() / 2 : 0, npw = pass == 0 ? bg.pairs.size() : 0;
			while (iv < nrv || jf < nrf || ip < npw) {
				// as many runs as a slot takes (a short run's entries count against the gathered region; a long run may be cut anywhere)
				std::vector<uint32_t> pv, pf;
				Need q;
				auto take = [&](const std::vector<uint32_t> &runs, size_t &i, uint32_t &taken, std::vector<uint32_t> &piece, int tg, int td) {
					while (i < runs.size() / 2) {
						const uint32_t left = runs[2 * i + 1] - taken;
						const bool direct = left >= dmin;
						const int t = direct ? td : tg;
						if (q.runs[t] >= kRuns) break;
						uint32_t n = left;
						if (!direct) { if (q.n[tg] >= elems) break; n = std::min(left, elems - q.n[tg]); }
						else n = std::min<uint32_t>(left, (1u << 30) - std::min<uint32_t>(q.n[td], 1u << 30));
						if (!n) break;
						piece.push_back(runs[2 * i] + taken); piece.push_back(n);
						++q.runs[t]; q.n[t] += n;
						if (taken + n == runs[2 * i + 1]) { ++i; taken = 0; } else taken += n;
					}
				};
				take(bg.v, iv, tv, pv, TGV, TDV);
				if (pass == 1) take(bg.f, jf, tf, pf, TGF, TDF);
				const uint32_t pw = (uint32_t)std::min<size_t>(npw - ip, kPairWords & ~1u);
				q.pairs = pw;
				// (a cut run may have changed sides: short <-> long; the tables are filled by the same rule as they were sized)
				q = need_of(pv.data(), (uint32_t)(pv.size() / 2), pf.data(), (uint32_t)(pf.size() / 2), pw, dmin);
				if (!fits_empty(q)) throw Error(HRY_E_INTERNAL, "encode pipeline: a piece outgrew its slot");
				int sl;
				Reservation r;
				{
					std::unique_lock<std::mutex> g(mu);
					if (open_slot >= 0) { slot[open_slot].state = CLOSED; open_slot = -1; }   // (the walkers' open slot goes first: this piece takes a slot of its own)
					g.unlock();
					flush_closed();
					g.lock();
					sl = take_free_slot(g, mode);
					r = reserve(sl, q);
					slot[sl].state = CLOSED;
				}
				copy_runs(r, pv.data(), (uint32_t)(pv.size() / 2), pf.data(), (uint32_t)(pf.size() / 2), bg.pairs.data() + ip, pw, dmin, pass == 0, pass == 1);
				ip += pw;
				send(sl);
				{ std::lock_guard<std::mutex> g(mu); slot[sl].state = INFLIGHT; }
			}
		}
	}
	// (under mu) a free slot for the sender itself
	int take_free_slot(std::unique_lock<std::mutex> &g, int mode)
	{
		for (;;) {
			for (int i = 0; i < Context::kPipeSlots; ++i)
				if (slot[i].state == FREE) { slot[i] = Slot(); slot[i].state = OPEN; slot[i].mode = mode; slot[i].seq = next_seq++; return i; }
			g.unlock();
			flush_closed();      // (a slot a walker has just finished copying into)
			free_oldest(true);
			g.lock();
		}
	}
	// slots whose kernels have run are free again; wait: block for the oldest one on its way
	void free_oldest(bool wait)
	{
		for (;;) {
			int oldest = -1;
			{
				std::lock_guard<std::mutex> g(mu);
				for (int i = 0; i < Context::kPipeSlots; ++i) if (slot[i].state == INFLIGHT && (oldest < 0 || slot[i].seq < slot[oldest].seq)) oldest = i;
			}
			if (oldest < 0) return;
			if (wait) HIP_OK(hipEventSynchronize(cx.pipe_slot_ev[oldest]));
			else if (hipEventQuery(cx.pipe_slot_ev[oldest]) != hipSuccess) return;
			{ std::lock_guard<std::mutex> g(mu); slot[oldest].state = FREE; }
			wait = false;
		}
	}
	// every closed slot whose writers are done, oldest first
	void flush_closed()
	{
		for (;;) {
			int pick = -1;
			bool empty = true;
			{
				std::lock_guard<std::mutex> g(mu);
				for (int i = 0; i < Context::kPipeSlots; ++i)
					if (slot[i].state == CLOSED && slot[i].writers == 0 && (pick < 0 || slot[i].seq < slot[pick].seq)) pick = i;
				if (pick >= 0) { for (int t = 0; t < kTables; ++t) empty &= slot[pick].n[t] == 0; empty &= slot[pick].npairs == 0; }
			}
			if (pick < 0) return;
			if (!empty) send(pick);
			std::lock_guard<std::mutex> g(mu);
			slot[pick].state = empty ? FREE : INFLIGHT;
		}
	}
	void run()
	{
		HIP_OK(hipSetDevice(cx.device));
		register_arrays();
		for (;;) {
			Big bg;
			bool have_big = false, last = false;
			{
				std::unique_lock<std::mutex> g(mu);
				auto ready = [&] {
					if (closing || !big.empty()) return true;
					for (int i = 0; i < Context::kPipeSlots; ++i) if (slot[i].state == CLOSED && slot[i].writers == 0) return true;
					return false;
				};
				// (slots on their way are looked at every now and then: a walker that finds none free hands its group over instead of waiting)
				while (!ready()) {
					if (cond_sender.wait_for(g, std::chrono::microseconds(500)) == std::cv_status::timeout) {
						bool inflight = false;
						for (int i = 0; i < Context::kPipeSlots; ++i) inflight |= slot[i].state == INFLIGHT;
						if (inflight) break;
					}
				}
				if (!big.empty()) { bg = std::move(big.front()); big.pop_front(); have_big = true; }
				else if (closing) {
					if (open_slot >= 0) { slot[open_slot].state = CLOSED; open_slot = -1; }   // (a group handed over late opened one more)
					bool pending = false;   // (a walker may still be copying into a closed slot)
					for (int i = 0; i < Context::kPipeSlots; ++i) pending |= slot[i].state == CLOSED || slot[i].state == OPEN;
					last = !pending;
				}
			}
			flush_closed();
			if (have_big) send_big(bg);
			free_oldest(false);
			if (last) break;
		}
		HIP_OK(hipStreamSynchronize(cx.pipe_stream));
	}
};

// ---------------------------------------------------------------------------------------------------------
void encode_chunked(Context &cx, Mesh &m, int chunk_syms, ByteSink &out, const InPlaceShard *in_place)
{
	HIP_OK(hipSetDevice(cx.device));
	auto t_all = Clock::now();
	cx.timing = hry_timing{};
	check_codable(m);
	if (m.general) check_general(m);
	uint32_t CH = chunk_syms > 0 ? std::min<uint32_t>((uint32_t)chunk_syms, kMaxChunk) : (uint32_t)kDefaultChunk;
	for (auto &L : m.lists) if (!L.have_bounds && L.ncomp()) { device_bounds(cx, m); break; }
	for (auto &L : m.lists) if (!L.have_bounds) { L.bmin.assign(L.stride(), 0); L.bmax.assign(L.stride(), 0); L.have_bounds = true; }
	if (in_place && (m.general || !m.shard.active())) throw Error(HRY_E_INTERNAL, "only a shard of a mesh in the PLY layout is coded in place");
	if (m.general) upload_general(cx, m);
	else if (in_place) {}   // (the caller filled the whole mesh's arrays over the shard's intervals)
	else if (m.device_token == 0 || m.device_token != cx.resident_token) cx.upload_mesh(m);
	const uint32_t dev_nv = in_place ? in_place->whole->nv : m.nv;   // vertex-indexed device arrays follow the numbering of the arrays in HBM

	// a shard of a larger mesh writes one segment of a sharded container (.hry v0.3, host/shard.cpp): the header of the whole
	// mesh, then its runs and an ordinary v0.2 body of the shard in its own numbering
	const bool sharded = m.shard.active();
	out.clear();
	{
		std::vector<uint8_t> hdr;
		write_hry_header(m, sharded ? 3 : 2, hdr);
		out.append(hdr.data(), hdr.data() + hdr.size());
	}
	size_t seg_len_at = 0, seg_begin = 0;
	if (sharded && m.nf == 0) {   // a rank without a group to code contributes no segment
		const uint32_t none = 0;
		out.append((const uint8_t*)&none, (const uint8_t*)&none + 4);
		cx.timing.total_ms = ms_since(t_all);
		return;
	}
	if (sharded) {
		auto put32 = [&](uint32_t v) { out.append((const uint8_t*)&v, (const uint8_t*)&v + 4); };
		put32(1);
		seg_len_at = out.size();
		put32(0); put32(0);
		seg_begin = out.size();
		put32((uint32_t)m.shard.runs.size());
		static_assert(sizeof(ShardRun) == 24, "runs are written as they lie in memory");
		// general bindings: every run is followed by its place in the record numbering of every list (first record, records)
		const size_t nl2 = m.general ? 2 * m.lists.size() : 0;
		if (m.shard.run_records.size() != nl2 * m.shard.runs.size()) throw Error(HRY_E_ARG, "shard without its record ranges");
		for (size_t j = 0; j < m.shard.runs.size(); ++j) {
			const uint8_t *rp = (const uint8_t*)&m.shard.runs[j];
			out.append(rp, rp + sizeof(ShardRun));
			if (nl2) { const uint8_t *qp = (const uint8_t*)(m.shard.run_records.data() + j * nl2); out.append(qp, qp + 4 * nl2); }
		}
	}
	HRY_MARK(t_all, "mesh resident, header written");
	auto t_walk = Clock::now();
	const ListDesc ldv = m.general ? ListDesc{} : make_list_desc(m.lists[1]), ldf = m.general ? ListDesc{} : make_list_desc(m.lists[0]);   // (general bindings: general_planes_encode)
	WalkResult w;
	w.numtri_positions = false;     // (places in ONE symbol sequence: the chunked planes have none)
	w.snapshot_faces = snapshot_spacing(m.nf);   // restart points inside large components (host.hpp BorderSnapshot; a shard: its own faces)
	bool walked = false, walk_again = false;
	// the device side beside the walk (EncodePipeline above): for walks on several threads whose sizes are known before they start
	std::unique_ptr<EncodePipeline> pipe;
	auto start_pipeline = [&](const ComponentAnalysis &A, std::function<void()> ready) {
		if (m.general || !EncodePipeline::wanted() || host_threads() < 2 || cx.keep_stages) return;
		uint64_t nvc = 0, nfc = 0;
		for (uint32_t k = 0; k < A.ncomp; ++k) { nvc += A.fresh[k]; nfc += A.n_faces[k]; }
		if (nvc == 0 || nvc >= (1ull << 32) || nfc >= (1ull << 32)) return;
		int ndeg = 0;
		for (uint8_t d : m.have_degree) ndeg += d ? 1 : 0;
		if (!cx.pipe_stream) { HIP_OK(hipStreamCreateWithFlags(&cx.pipe_stream, hipStreamNonBlocking)); HIP_OK(hipEventCreateWithFlags(&cx.pipe_ev, hipEventDisableTiming)); }
		// (every allocation before the walk starts: hipMalloc waits for the device)
		cx.d_order_v.ensure(std::max<size_t>((size_t)nvc * 4, 16));
		cx.d_order_f.ensure(std::max<size_t>((size_t)nfc * 4, 16));
		cx.d_rank.ensure(std::max<size_t>((size_t)dev_nv * 4, 16));
		cx.d_vplanes.ensure(std::max<size_t>((size_t)nvc * ldv.nplanes, 16));
		cx.d_fplanes.ensure(std::max<size_t>((size_t)nfc * ldf.nplanes, 16));
		if (ndeg > 1) { cx.d_nt_val.ensure(std::max<size_t>((size_t)nfc * 4, 16)); cx.d_nt_planes.ensure(std::max<size_t>((size_t)nfc * kGroupBytes[G_NUMTRI], 16)); }
		HIP_OK(hipMemsetAsync(cx.d_rank.p, 0xff, (size_t)dev_nv * 4, cx.pipe_stream));
		pipe.reset(new EncodePipeline(cx, ldv, ldf, (uint32_t)nvc, (uint32_t)nfc, dev_nv, std::move(ready), t_all));
		pipe->prepare_slots();
		w.progress = pipe.get();
	};
	// a shard in place: its intervals are on their way up on the caller's thread; whoever needs them first waits for them, once
	std::once_flag arrays_once;
	std::exception_ptr arrays_err;
	auto shard_arrays_ready = [&] {
		if (!(in_place && in_place->arrays_ready)) return;
		std::call_once(arrays_once, [&] { try { in_place->arrays_ready(); } catch (...) { arrays_err = std::current_exception(); } });
		if (arrays_err) std::rethrow_exception(arrays_err);
	};
	if (in_place) {
		if (in_place->before_walk) in_place->before_walk();   // (its turn among the executor's workers, and the turn's thread budget)
		struct After { const std::function<void()> &f; ~After() { if (f) f(); } } after{ in_place->after_walk };
		start_pipeline(*in_place->part, shard_arrays_ready);
		cut_border_walk_in_place(*in_place->whole, *in_place->part, in_place->eface, *in_place->marks, w);
		walked = true;
	}
	else if (!m.general && m.shard.seeds.empty() && m.nf >= device_analysis_min_faces() && host_threads() > 1) {
		// A large mesh: its components -- labels, coding order, sizes, new vertices, ties -- are found on the device, where the
		// connectivity is resident (analysis.cpp: 1.9 CPU-seconds of host passes at 100 M triangles), and all of them are walked on
		// the host threads where they lie, from the first one on; a mesh of ONE component takes the sequential loop as before
		ComponentAnalysis A;
		// (beside it, on a host thread of its own: what the walk loops need whatever the analysis says.  The device calls stay on
		// this thread: a short-lived thread that has used the runtime leaves the next pageable copy of this one 25 ms slower)
		int ud = 0;
		const bool uniform = m.uniform_degree(ud) && (ud == 3 || ud == 4);
		BigVec<uint32_t> eface;   // mixed degrees: the face of every half-edge
		std::unique_ptr<WalkState> marks;
		std::exception_ptr failed;
		const unsigned nt = host_threads();
		std::thread tables([&] {
			try {
				set_thread_budget(nt);
				if (!uniform) {
					eface.resize(m.ne());
					parallel_for(nt, [&](unsigned t) {
						const uint32_t b = (uint32_t)((uint64_t)m.nf * t / nt), e = (uint32_t)((uint64_t)m.nf * (t + 1) / nt);
						for (uint32_t f = b; f < e; ++f) for (uint32_t h = m.face_off[f]; h < m.face_off[f + 1]; ++h) eface[h] = f;
					});
				}
				marks.reset(new WalkState(m.nv, m.nf, nt));
			} catch (...) { failed = std::current_exception(); }
		});
		try { device_component_analysis(cx, m, A); } catch (...) { tables.join(); throw; }
		tables.join();
		if (failed) std::rethrow_exception(failed);
		if (A.ncomp > 1) {
			try {
				start_pipeline(A, nullptr);
				cut_border_walk_in_place(m, A, uniform ? nullptr : eface.data(), *marks, w);
				walked = true;
				HRY_MARK(t_all, "  walk returned");
			} catch (const WalkMismatch &) {
				// (host.hpp WalkMismatch: a repaired twin cut a component in two) everything of the attempt is dropped -- what the pipeline
				// has sent, the result, the twins as the attempt left them -- and the mesh takes the walk below, which ends on one thread
				if (pipe) { try { pipe->finish(); } catch (...) {} pipe.reset(); }
				WalkResult fresh;
				fresh.numtri_positions = false; fresh.snapshot_faces = w.snapshot_faces;
				w = std::move(fresh);
				build_twins(m);
				walk_again = true;
				HRY_MARK(t_all, "  a repaired twin split a component: walked again");
			}
		}
	}
	if (!walked) cut_border_walk(m, w, false);   // operation planes carry symbol + order class; no model evaluation needed
	if (walk_again) { w.twins_changed = true; w.twin_patches.clear(); }   // (the device's twins are those of the dropped attempt: the whole array goes up again)
	cx.timing.host_walk_ms = ms_since(t_walk);
	HRY_MARK(t_all, "walked");
	shard_arrays_ready();
	const bool piped = pipe && pipe->began;
	if (pipe) { pipe->finish(); w.progress = nullptr; }
	if (piped) HRY_MARK(t_all, "  pipeline drained");
	// directory, part one: the restart points of the connectivity replay come from the walk's marks alone -- for a mesh of many
	// components on a thread of its own from here on (17 ms for the 151 741 components of the configs[3] mesh: beside the stream
	// kernels until round 5, which the carry kernels' rewrite left shorter than that)
	std::vector<RestartCounters> rcounters;
	std::vector<RestartPoint> restarts;
	std::vector<uint32_t> counter_dir;   // per restart point: n, then n x (vertex, counter)
	std::vector<uint8_t> snap_dir;       // the border snapshots' section (host/header.cpp: write_snapshot_section), empty without any
	auto select_restarts = [&] {
		std::vector<RestartCounters> scounters;
		restarts = select_restart_points(w.marks, w.named, rcounters, &w.snapshots, &scounters);
		for (const RestartCounters &cs : rcounters) {
			counter_dir.push_back((uint32_t)cs.size());
			for (const auto &c : cs) { counter_dir.push_back(c.first); counter_dir.push_back(c.second); }
		}
		if (!w.snapshots.empty()) write_snapshot_section(w.snapshot_faces, w.snapshots, scounters, snap_dir);
	};
	struct Helper {   // (joined on every way out)
		std::thread th; std::exception_ptr failed;
		void wait() { if (th.joinable()) th.join(); if (failed) { std::exception_ptr e = failed; failed = nullptr; std::rethrow_exception(e); } }
		~Helper() { if (th.joinable()) th.join(); }
	} dir_helper;
	const bool dir_beside = w.marks.size() >= 4096 && host_threads() > 1;
	if (dir_beside) dir_helper.th = std::thread([&] { try { select_restarts(); } catch (...) { dir_helper.failed = std::current_exception(); } });

	const uint32_t vc = (uint32_t)w.order_v.size(), fc = (uint32_t)w.order_f.size();
	if (piped && (vc != pipe->vc || fc != pipe->fc)) throw Error(HRY_E_INTERNAL, "encode pipeline: sizes changed under the walk");
	if (chunk_syms <= 0) {
		// default policy: 8 Ki symbols per chunk; larger meshes get larger chunks as long as some thousands of streams remain to
		// fill the 1024 SIMDs -- up to 32 Ki symbols: beyond a few thousand streams the decoder runs a stream per LANE
		// (k_chunk_decode_lanes), 512 wavefronts of them at a time, and its time is the longest stream's (32 Ki symbols: 26 ms).
		// The 100 M-triangle configs[3] mesh in one piece had 128 Ki-symbol chunks until round 4 (5 200 attribute streams of a
		// wavefront each, 100 ms); every smaller mesh keeps the size it had
		uint64_t total = (uint64_t)w.n_conn + (uint64_t)vc * ldv.nplanes + (uint64_t)fc * ldf.nplanes;
		if (m.general) for (const AttrList &L : m.lists) total += (uint64_t)L.count * L.coded_bytes();
		while (CH < (1u << 15) && total / CH > 8192) CH <<= 1;
	}

	// ---- connectivity planes on the host side: 5 groups (split into bytes on the device) + 8 operation planes
	auto t_h2d = Clock::now();
	// operations arrive as one byte each (symbol | order class << 3); the device sorts them into one plane per class
	// (k_split_classes: a stable partition), the walk has counted the classes
	size_t op_plane_n[8];
	for (int k = 0; k < 8; ++k) op_plane_n[k] = w.n_op_class[k];
	size_t ngrp = 0, conn_plane_bytes = 0, nopb = w.op_sc.size();
	for (int g = 0; g < G_COUNT; ++g) { ngrp += w.grp_val[g].size(); conn_plane_bytes += w.grp_val[g].size() * kGroupBytes[g]; }
	cx.d_order_v.ensure(std::max<size_t>((size_t)vc * 4, 16));
	cx.d_order_f.ensure(std::max<size_t>((size_t)fc * 4, 16));
	cx.d_rank.ensure(std::max<size_t>((size_t)dev_nv * 4, 16));
	cx.d_grp_val.ensure(std::max<size_t>(ngrp * 4, 16));
	cx.d_connplanes.ensure(std::max<size_t>(conn_plane_bytes + 2 * nopb + 64, 16));   // ... + operation planes + the raw operation bytes
	cx.d_vplanes.ensure(std::max<size_t>((size_t)vc * ldv.nplanes, 16));
	cx.d_fplanes.ensure(std::max<size_t>((size_t)fc * ldf.nplanes, 16));
	// (piped: the pipeline has brought the orders, the triangle counts and the repaired twins up beside the walk)
	const bool nt_piped = piped && pipe->nt != nullptr;
	if (vc && !piped) HIP_OK(hipMemcpyAsync(cx.d_order_v.p, w.order_v.data(), (size_t)vc * 4, hipMemcpyHostToDevice, cx.stream));
	if (fc && (ldf.nplanes || m.general) && !piped) HIP_OK(hipMemcpyAsync(cx.d_order_f.p, w.order_f.data(), (size_t)fc * 4, hipMemcpyHostToDevice, cx.stream));   // only the face planes read it
	// the resident copy of the twins is current unless the walk repaired some (non-manifold edges, consumed neighbours)
	if (!piped) upload_repaired_twins(cx, in_place ? *in_place->whole : m, w, in_place != nullptr);
	if (in_place) cx.inplace_twin_patches.insert(cx.inplace_twin_patches.end(), w.twin_patches.begin(), w.twin_patches.end());
	size_t goff[G_COUNT + 1] = { 0 };
	for (int g = 0; g < G_COUNT; ++g) {
		size_t n = w.grp_val[g].size();
		goff[g + 1] = goff[g] + n;
		if (n && !(g == G_NUMTRI && nt_piped)) HIP_OK(hipMemcpyAsync(cx.d_grp_val.as<uint32_t>() + goff[g], w.grp_val[g].data(), n * 4, hipMemcpyHostToDevice, cx.stream));
	}
	uint8_t *d_opplanes = cx.d_connplanes.as<uint8_t>() + conn_plane_bytes;
	uint8_t *d_opraw = d_opplanes + nopb;
	if (nopb) HIP_OK(hipMemcpyAsync(d_opraw, w.op_sc.data(), nopb, hipMemcpyHostToDevice, cx.stream));

	// ---- plane list in container order
	std::vector<PlaneRef> planes;
	{
		size_t poff = 0;
		int ci = 0;
		for (int g = 0; g < G_COUNT; ++g) {
			uint32_t n = (uint32_t)w.grp_val[g].size();
			const uint8_t *at = g == G_NUMTRI && nt_piped ? cx.d_nt_planes.as<uint8_t>() : cx.d_connplanes.as<uint8_t>() + poff;   // (the pipeline split the triangle counts into planes of their own)
			for (int b = 0; b < kGroupBytes[g]; ++b, ++ci) planes.push_back(PlaneRef{ at + (size_t)b * n, n, plane_init_kind(ci) });
			poff += (size_t)n * kGroupBytes[g];
		}
		size_t o = 0;
		for (int k = 0; k < 8; ++k, ++ci) { planes.push_back(PlaneRef{ d_opplanes + o, (uint32_t)op_plane_n[k], INIT_OP }); o += op_plane_n[k]; }
		for (int p = 0; p < ldv.nplanes; ++p) planes.push_back(PlaneRef{ cx.d_vplanes.as<uint8_t>() + (size_t)p * vc, vc, INIT_ONES });
		for (int p = 0; p < ldf.nplanes; ++p) planes.push_back(PlaneRef{ cx.d_fplanes.as<uint8_t>() + (size_t)p * fc, fc, INIT_ONES });
	}
	// general bindings: which record every element names is settled on the host, the residuals of the records coded as data are
	// computed by general.hip; the planes join the list like any other (kernels are enqueued here, behind the uploads above)
	if (m.general) general_planes_encode(cx, m, w, planes);
	std::vector<uint32_t> kind_tabs;
	build_init_tables(m, kind_tabs);
	const uint32_t CHC = std::min(CH, std::max(CH / 8, 512u));
	const uint32_t npl = (uint32_t)planes.size();
	// slices of the planes for the histogram pass
	std::vector<HistSlice> slices;
	for (uint32_t pi = 0; pi < npl; ++pi)
		for (uint32_t f = 0; f < planes[pi].n; f += 16384) slices.push_back(HistSlice{ planes[pi].dptr + f, std::min(16384u, planes[pi].n - f), pi });
	cx.d_small.ensure(std::max<size_t>(slices.size() * sizeof(HistSlice) + (size_t)npl * 1024 + 64, 64));
	uint32_t *d_hist = cx.d_small.as<uint32_t>();
	HistSlice *d_slices = (HistSlice*)(cx.d_small.as<uint8_t>() + (size_t)npl * 1024);
	if (!slices.empty()) HIP_OK(hipMemcpyAsync(d_slices, slices.data(), slices.size() * sizeof(HistSlice), hipMemcpyHostToDevice, cx.stream));
	HIP_OK(hipMemsetAsync(d_hist, 0, (size_t)npl * 1024, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.timing.h2d_ms = ms_since(t_h2d);
	HRY_MARK(t_all, "walk's outputs on the device");

	// ---- device: prediction + residuals + planes, then the planes' histograms
	ConnView cv = cx.conn_view();
	HIP_OK(hipEventRecord(cx.ev[1], cx.stream));
	if (!m.general && !piped) {
		HIP_OK(hipMemsetAsync(cx.d_rank.p, 0xff, (size_t)dev_nv * 4, cx.stream));
		launch_rank(cx.stream, cx.d_order_v.as<uint32_t>(), vc, cv.org, cx.d_rank.as<uint32_t>());
		launch_predict_vtx(cx.stream, cv, cx.d_order_v.as<uint32_t>(), vc, cx.d_rank.as<uint32_t>(), cx.d_rec[1].as<uint8_t>(), ldv, cx.d_vplanes.as<uint8_t>());
		launch_face_planes(cx.stream, cv, cx.d_order_f.as<uint32_t>(), fc, cx.d_rec[0].as<uint8_t>(), ldf, cx.d_fplanes.as<uint8_t>());
	}
	{
		size_t poff = 0;
		for (int g = 0; g < G_COUNT; ++g) {
			uint32_t n = (uint32_t)w.grp_val[g].size();
			if (!(g == G_NUMTRI && nt_piped)) launch_split_bytes(cx.stream, cx.d_grp_val.as<uint32_t>() + goff[g], n, kGroupBytes[g], cx.d_connplanes.as<uint8_t>() + poff);
			poff += (size_t)n * kGroupBytes[g];
		}
	}
	{
		uint32_t base[8], o = 0;
		for (int k = 0; k < 8; ++k) { base[k] = o; o += (uint32_t)op_plane_n[k]; }
		cx.d_op.ensure(((nopb + kSplitUnit - 1) / kSplitUnit + 1) * 8 * 4 + 64);
		launch_split_classes(cx.stream, d_opraw, (uint32_t)nopb, base, cx.d_op.as<uint32_t>(), d_opplanes);
	}
	launch_plane_hist(cx.stream, d_slices, (uint32_t)slices.size(), d_hist);
	HIP_OK(hipEventRecord(cx.ev[2], cx.stream));
	std::vector<uint32_t> hist((size_t)npl * 256);
	if (npl) HIP_OK(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	HRY_MARK(t_all, "planes and their histograms");

	// ---- initial table of every plane: its static prior, or the reference's initial counts for short planes
	std::vector<uint32_t> inits((size_t)npl * 256);
	std::vector<uint32_t> totals(npl, 0);
	std::vector<uint8_t> prior_dir;
	for (uint32_t pi = 0; pi < npl; ++pi) {
		uint32_t *tab = inits.data() + (size_t)pi * 256;
		const bool use = plane_prior_from_hist(hist.data() + (size_t)pi * 256, planes[pi].n, tab);
		if (!use) memcpy(tab, kind_tabs.data() + (size_t)planes[pi].init * 256, 1024);
		for (int i = 0; i < 256; ++i) totals[pi] += tab[i];
		write_prior(prior_dir, use, tab);
	}
	std::vector<StreamJob> jobs;
	uint64_t words = 0;
	uint64_t nsym_total = 0;
	uint32_t max_t0 = 256;
	// connectivity planes are cut shorter: the decoder needs them first and a stream is a serial chain (container
	// description: oracle/hry_oracle.cc "chunked profile")
	for (uint32_t pi = 0; pi < npl; ++pi) {
		const PlaneRef &pl = planes[pi];
		nsym_total += pl.n;
		max_t0 = std::max(max_t0, totals[pi]);
		for (uint32_t f = 0, step; f < pl.n; f += step) {
			step = pi < (uint32_t)kConnPlanes ? CHC : attr_chunk_len(f, CH);
			uint32_t n = std::min(step, pl.n - f);
			if (words >= (1ull << 32) - (1u << 24)) throw Error(HRY_E_UNSUPPORTED, "chunked stream accumulator exceeds 2^32 words");
			jobs.push_back(StreamJob{ pl.dptr + f, n, pi, totals[pi], (uint32_t)words });
			words += stream_words(n, totals[pi]);
		}
	}
	const uint32_t ns = (uint32_t)jobs.size();
	const uint32_t nw = (uint32_t)words + 2;
	HRY_MARK(t_all, "  priors and stream jobs (host)");
	cx.d_init.ensure(std::max<size_t>(inits.size() * 4, 16));
	if (!inits.empty()) HIP_OK(hipMemcpyAsync(cx.d_init.p, inits.data(), inits.size() * 4, hipMemcpyHostToDevice, cx.stream));
	cx.d_cjobs.ensure(std::max<size_t>((size_t)ns * sizeof(StreamJob), 16));
	if (ns) HIP_OK(hipMemcpyAsync(cx.d_cjobs.p, jobs.data(), (size_t)ns * sizeof(StreamJob), hipMemcpyHostToDevice, cx.stream));
	cx.ensure_magic(max_t0 + CH + 16);
	cx.d_acc.ensure((size_t)nw * 8);
	// (no folded copy of the accumulators: the carry kernels fold where they read, kernels.hip)
	cx.d_summary.ensure(((size_t)nw / 1024 + 8) * 4);   // the carry kernels' block pairs and, behind them, their marks of used ranges
	cx.d_bytes.ensure((size_t)nw * 4);
	cx.d_csizes.ensure(std::max<size_t>((size_t)ns * 8, 16));   // bits | nbytes
	cx.d_coffs.ensure(((size_t)ns + 1) * 8);
	// ---- device: one wavefront per stream
	HIP_OK(hipMemsetAsync(cx.d_acc.p, 0, (size_t)nw * 8, cx.stream));
	uint32_t *d_bits = cx.d_csizes.as<uint32_t>(), *d_nbytes = d_bits + ns;
	// thousands of streams: the model a wavefront per stream, the range registers a lane per stream (chunked.hip: k_chunk_model,
	// k_chunk_ranges; the one kernel is bound by the compute units' scalar units: 0.04 ns a symbol over all streams, and by its
	// longest stream at 0.09 us a symbol; the two take 0.0057 ns a symbol for the model and 0.21 us a symbol of the longest stream
	// for the ranges -- 100 M triangles: 28 -> 11.8 ms, the 12.6 M-triangle share 4.7 -> 3.9, the 1 M-triangle torus 0.76 -> 1.64).
	// HRY_ENCODE_SPLIT_MIN_STREAMS: from how many streams instead of by that estimate (0 = never)
	bool split_kernels;
	{
		uint32_t longest = 0;
		for (uint32_t i = 0; i < ns; ++i) longest = std::max(longest, jobs[i].n);
		const double one = std::max(longest * 0.09e-3, (double)nsym_total * 0.04e-6), two = longest * 0.207e-3 + (double)nsym_total * 0.0057e-6;   // ms
		const char *e = getenv("HRY_ENCODE_SPLIT_MIN_STREAMS");
		split_kernels = e ? (strtoul(e, nullptr, 10) != 0 && ns >= strtoul(e, nullptr, 10)) : (ns >= 64 && two < 0.9 * one);
	}
	for (uint32_t i = 0; i < ns && split_kernels; ++i) split_kernels = (uint64_t)jobs[i].t0 + jobs[i].n <= 65535u;   // (its records hold 16-bit counts)
	std::vector<uint64_t> tab;   // (lives until the stream is waited for below)
	if (split_kernels) {
		tab.resize((size_t)ns + ((size_t)ns + 1) / 2);   // rec_off (u64 each), then the order (u32 each)
		uint32_t *order = (uint32_t*)(tab.data() + ns);
		uint64_t off = 0;
		for (uint32_t i = 0; i < ns; ++i) { tab[i] = off; off += jobs[i].n; order[i] = i; }
		std::stable_sort(order, order + ns, [&](uint32_t a, uint32_t b) { return jobs[a].n > jobs[b].n; });
		cx.d_rec_sym.ensure(std::max<size_t>((size_t)off * 8, 16));
		cx.d_split.ensure(tab.size() * 8);
		HIP_OK(hipMemcpyAsync(cx.d_split.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, cx.stream));
		HIP_OK(hipEventRecord(cx.ev[3], cx.stream));
		launch_chunk_encode_split(cx.stream, cx.d_cjobs.as<StreamJob>(), ns, cx.d_init.as<uint32_t>(), cx.d_magic.as<MagicEnt>(), cx.d_acc.as<uint64_t>(), d_bits,
		                          cx.d_split.as<uint64_t>(), cx.d_rec_sym.p, (const uint32_t*)(cx.d_split.as<uint64_t>() + ns));
		HIP_OK(hipEventRecord(cx.ev[4], cx.stream));
	} else {
	HIP_OK(hipEventRecord(cx.ev[3], cx.stream));
	launch_chunk_encode(cx.stream, cx.d_cjobs.as<StreamJob>(), ns, cx.d_init.as<uint32_t>(), cx.d_magic.as<MagicEnt>(), cx.d_acc.as<uint64_t>(), d_bits);
	HIP_OK(hipEventRecord(cx.ev[4], cx.stream));
	}
	launch_carry(cx.stream, cx.d_acc.as<uint64_t>(), nw, cx.d_v.as<uint64_t>(), cx.d_summary.as<uint32_t>(), cx.d_bytes.as<uint8_t>(), cx.d_cjobs.as<StreamJob>(), d_bits, ns);
	launch_stream_pack(cx.stream, cx.d_cjobs.as<StreamJob>(), ns, d_bits, nullptr, d_nbytes, cx.d_coffs.as<uint64_t>(), nullptr, false);
	// (into pinned memory: a copy into a pageable variable returned only when the streams were coded, and the restart points below
	// -- 17 ms of host work at 100 M triangles -- were selected after the kernels instead of beside them)
	cx.h_small.ensure(64);
	volatile uint64_t *total_bytes_p = cx.h_small.as<uint64_t>();
	*total_bytes_p = 0;
	HIP_OK(hipMemcpyAsync((void*)total_bytes_p, cx.d_coffs.as<uint64_t>() + ns, 8, hipMemcpyDeviceToHost, cx.stream));
	HRY_MARK(t_all, "  stream kernels launched");
	// directory: chunk sizes, plane lengths, restart points of the connectivity replay (while the device codes the streams, unless
	// a thread has been at them since the walk), stream lengths
	if (dir_beside) dir_helper.wait(); else select_restarts();
	const uint32_t nrs = (uint32_t)restarts.size();
	HRY_MARK(t_all, "  restart points selected");
	HIP_OK(hipStreamSynchronize(cx.stream));
	const uint64_t total_bytes = *total_bytes_p;
	HRY_MARK(t_all, "streams coded");
	cx.d_cout.ensure(std::max<size_t>(total_bytes, 16));
	launch_stream_pack(cx.stream, cx.d_cjobs.as<StreamJob>(), ns, d_bits, cx.d_bytes.as<uint8_t>(), d_nbytes, cx.d_coffs.as<uint64_t>(), cx.d_cout.as<uint8_t>(), true);
	HIP_OK(hipEventRecord(cx.ev[5], cx.stream));

	// ---- container
	static_assert(sizeof(RestartPoint) == kRestartWords * 4, "restart points are written as they lie in memory");
	const size_t dir_prior = 12 + 4 * planes.size();
	const size_t dir_restart = dir_prior + prior_dir.size();
	const size_t dir_counters = dir_restart + 4 + sizeof(RestartPoint) * (size_t)nrs;
	const size_t dir_snaps = dir_counters + 4 * counter_dir.size();   // (round 6: the border snapshots, announced by the top bit of the restart points' count)
	const size_t dir_streams = dir_snaps + snap_dir.size();
	size_t dir = dir_streams + 4 * (size_t)ns;
	size_t base = out.size();
	out.resize(base + dir + total_bytes);
	uint8_t *o = out.data() + base;
	uint32_t np = (uint32_t)planes.size();
	memcpy(o, &CH, 4); memcpy(o + 4, &CHC, 4); memcpy(o + 8, &np, 4);
	for (size_t i = 0; i < planes.size(); ++i) memcpy(o + 12 + 4 * i, &planes[i].n, 4);
	if (!prior_dir.empty()) memcpy(o + dir_prior, prior_dir.data(), prior_dir.size());
	{ const uint32_t nrs_word = nrs | (snap_dir.empty() ? 0u : 0x80000000u); memcpy(o + dir_restart, &nrs_word, 4); }
	if (nrs) memcpy(o + dir_restart + 4, restarts.data(), sizeof(RestartPoint) * (size_t)nrs);
	if (!counter_dir.empty()) memcpy(o + dir_counters, counter_dir.data(), 4 * counter_dir.size());
	if (!snap_dir.empty()) memcpy(o + dir_snaps, snap_dir.data(), snap_dir.size());
	if (ns) HIP_OK(hipMemcpyAsync(o + dir_streams, d_nbytes, (size_t)ns * 4, hipMemcpyDeviceToHost, cx.stream));
	fetch_to_host(cx, o + dir, cx.d_cout.p, total_bytes);   // (returns when everything on the stream has happened)
	HRY_MARK(t_all, "container on the host");
	if (sharded) { const uint64_t seg_len = out.size() - seg_begin; memcpy(out.data() + seg_len_at, &seg_len, 8); }

	if (cx.keep_stages) {
		cx.stage_put_host("order_v", w.order_v.data(), (size_t)vc * 4);
		cx.stage_put_host("order_f", w.order_f.data(), (size_t)fc * 4);
		cx.stage_put("vplanes", cx.d_vplanes.p, (size_t)vc * ldv.nplanes);
		cx.stage_put("fplanes", cx.d_fplanes.p, (size_t)fc * ldf.nplanes);
		cx.stage_put("d_twin", cx.d_twin.p, (size_t)(in_place ? in_place->whole->ne() : m.ne()) * 4);   // the twins the prediction's fan walks followed
	}
	cx.timing.k_predict_ms = cx.elapsed(1, 2);
	cx.timing.k_entropy_ms = cx.elapsed(3, 4);
	cx.timing.device_ms = cx.elapsed(1, 5);
	cx.timing.n_symbols = nsym_total;
	cx.timing.payload_bytes = total_bytes;
	cx.timing.total_ms = ms_since(t_all);
}

// ---------------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------------
Mesh *decode_chunked(Context &cx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> m);
Mesh *decode_compat(Context &cx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> m);

Mesh *decode_any(Context &cx, const uint8_t *p, size_t n, int shard_index, int shard_count, bool allow_partial)
{
	HIP_OK(hipSetDevice(cx.device));
	std::unique_ptr<Mesh> m(new Mesh());
	int minor = 0;
	const bool sharded = n >= 6 && p[4] == 0 && p[5] == 3;   // the whole mesh's records are filled segment by segment: no zero fill first
	size_t hdr = read_hry_header(p, n, *m, minor, !sharded);
	if (minor == 3) { Context *one = &cx; return decode_sharded(&one, 1, p, n, hdr, std::move(m), shard_index, shard_count, allow_partial || shard_count > 1, nullptr); }
	if (shard_count > 1) throw Error(HRY_E_ARG, "only a sharded container (.hry v0.3) decodes segment by segment");
	if (minor == 2) return decode_chunked(cx, p, n, hdr, std::move(m));
	if (m->general) return decode_general(cx, p, n, hdr, std::move(m));
	return decode_compat(cx, p, n, hdr, std::move(m));
}

}   // namespace hry
