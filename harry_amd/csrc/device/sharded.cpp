// One mesh over several GPUs from ONE process (north_star: "meshes shard by independent connected
// component across the 8 GPUs of one node ... only for final stream concatenation").
//
// The reference has a single entry (main.cc:93-123 -> hry::writer::write, formats/hry/writer.cc:200-214) and one thread.  What
// scales here is "host thread + device context": per million triangles an encode is ~7 ms of sequential cut-border walk on a
// host core and ~1 ms of kernels, so the unit of parallelism is a worker thread that owns a context -- N of them in one
// process over N devices (or several per device).  Nothing in the data path is exchanged between the workers:
//
//   encode   plan once (host/shard.cpp: components, coding order, groups, exclusive scans)
//            workers: extract their shards, upload, k_bounds per shard                               (phase A)
//            caller:  bounds of the whole mesh = the shards' bounds combined with the scan's own tie rule (combine_shard_bounds)
//            workers: quantisation + chunked encode of their shards -> one-segment containers        (phase B)
//            caller:  concatenation into ONE .hry v0.3 (merge_containers) -- the segments meet in host memory
//   decode   directory parsed and validated once (parse_sharded_directory), whole-mesh arrays allocated once
//            workers: decode their segments (ordinary v0.2 bodies) and place them run by run into the whole numbering
//
// A worker runs on the CPUs of the memory node its device hangs on and limits the helper threads of its host phases to its
// share of the process's CPUs.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <thread>

#include "context.hpp"
#include "kernels.hpp"

namespace hry {

typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

Mesh *decode_chunked(Context &cx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> m);

namespace {

// CPUs of the memory node the device's PCI function belongs to (nullptr: unknown or a single node)
const void *device_cpus(int device)
{
	char bus[64] = {};
	if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device) != hipSuccess) return nullptr;
	for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
	char path[160];
	snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
	FILE *f = fopen(path, "r");
	if (!f) return nullptr;
	int node = -1;
	if (fscanf(f, "%d", &node) != 1) node = -1;
	fclose(f);
	return node_cpus(node);
}

// helper threads a worker may start in its host phases: its share of the CPUs it is confined to (the memory node of its device,
// shared with the other workers on that node; the whole process's CPUs when the node is unknown)
std::vector<unsigned> worker_thread_budgets(const std::vector<const void*> &cpus)
{
	const unsigned allowed = cpu_allowance();   // affinity mask and control-group quota
	const unsigned cap = host_threads();
	const unsigned n = (unsigned)std::max<size_t>(1, cpus.size());
	std::vector<unsigned> out(cpus.size(), 1);
	for (size_t w = 0; w < cpus.size(); ++w) {
		unsigned sharing = 0;
		for (size_t x = 0; x < cpus.size(); ++x) sharing += cpus[x] == cpus[w];
		const unsigned avail = cpus[w] ? (unsigned)CPU_COUNT((const cpu_set_t*)cpus[w]) : allowed;
		// a node's CPUs among the workers on it, and never more busy threads over all workers than the process may run
		out[w] = std::max(1u, std::min(cap, std::min(avail / std::max(1u, sharing), (allowed + n - 1) / n)));
	}
	return out;
}

// body(w) for every worker w on a thread of its own (one worker: the caller's thread); the first exception is rethrown here
template <typename F> void run_workers(Context *const *cxs, int n, F &&body)
{
	if (n == 1) { body(0); return; }
	std::vector<const void*> cpus((size_t)n);
	for (int w = 0; w < n; ++w) cpus[w] = device_cpus(cxs[w]->device);
	const std::vector<unsigned> budget = worker_thread_budgets(cpus);
	std::vector<std::thread> th;
	std::exception_ptr err;
	std::mutex mu;
	for (int w = 0; w < n; ++w)
		th.emplace_back([&, w] {
			try {
				stay_on_node(cpus[w]);
				set_thread_budget(budget[w]);
				body(w);
			} catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
		});
	for (auto &t : th) t.join();
	if (err) std::rethrow_exception(err);
}

// Workers that share something take turns at it, in the order of their shards: the CPUs of a memory node (the walks) and the
// link to a device (the interval copies).  A walk is bound by the CPUs it gets and a copy by its link, so sharing them evenly
// makes every worker finish at the same time -- and the device phases that follow (planes, streams, container) pile up behind
// the last walk.  One after the other, each with all of the shared resource, the total is the same and the first shard's
// device phase starts after an eighth of it: the workers' device phases hide behind each other's walks.
struct Turnstile {
	std::mutex mu;
	std::condition_variable cv;
	std::vector<int> order;   // the shards that pass here, ascending
	size_t next = 0;
	bool open = false;        // a worker failed: nobody waits any more
	void enter(int s)
	{
		std::unique_lock<std::mutex> g(mu);
		cv.wait(g, [&] { return open || (next < order.size() && order[next] == s); });
	}
	void leave(int s)
	{
		{ std::lock_guard<std::mutex> g(mu); if (next < order.size() && order[next] == s) ++next; }
		cv.notify_all();
	}
	void abort() { { std::lock_guard<std::mutex> g(mu); open = true; } cv.notify_all(); }
};

// segments of a merged container (a shard without a group contributes none: its part holds a zero count)
uint32_t merged_segments(const ByteSink &out)
{
	Mesh tmp;
	int minor = 0;
	const size_t hdr = read_hry_header(out.data(), out.size(), tmp, minor, false);
	uint32_t n = 0;
	if (minor == 3 && hdr + 4 <= out.size()) memcpy(&n, out.data() + hdr, 4);
	return n;
}

void check_contexts(Context *const *cxs, int n)
{
	if (!cxs || n <= 0) throw Error(HRY_E_ARG, "need at least one context");
	for (int i = 0; i < n; ++i) {
		if (!cxs[i]) throw Error(HRY_E_ARG, "null context");
		for (int j = 0; j < i; ++j) if (cxs[j] == cxs[i]) throw Error(HRY_E_ARG, "a context may appear only once: its streams and buffers serve one worker");
	}
}

}   // namespace

// ---------------------------------------------------------------------------------------------------------------------
// The shards coded WHERE THEY LIE (round 4).  Until then a shard was a sub-mesh of its own (shard_extract: every face, half-edge
// and vertex renumbered and copied, 0.1 s per 100 M triangles on top of a plan that indexed every element, and the bounds of
// the whole mesh met behind a barrier between two phases): 8 contexts took 1.08 s for what one context did in 1.0 s.  Now
//   * the plan is the analysis alone (components, coding order, groups, scans; host/shard.cpp: light);
//   * beside it the first context scans the records of the WHOLE mesh for the bounds (one k_bounds: no combination, no barrier);
//   * a worker fills its context's arrays -- sized for the whole mesh, in the whole mesh's numbering -- over the index intervals
//     its shard's components lie in (a few long copies; what lies between two nearby intervals travels along, unused), and
//     walks its components over the whole mesh's host arrays, all workers on ONE set of marks (components of different
//     groups touch different faces and vertices);
//   * the kernels neither know nor care: half-edge and vertex numbers are names, the stream holds none of them (explicit
//     vertex names are decode-order indices, counted from the shard's first component).
// The segments are byte for byte what the virtual ranks write from extracted sub-meshes (tests/test_gpu_shard.py).
static void encode_sharded_in_place(Context *const *cxs, int n_ctx, Mesh &m, const hry_quant *q, size_t nq, bool clear, int n_shards, int chunk_syms,
                                    ByteSink &out, hry_shard_timing &st, bool store_bounds)
{
	const auto t_all = Clock::now();
	const bool trace = getenv("HRY_TRACE") != nullptr;
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry shards] %8.2f ms  %s\n", ms_since(t_all), what); };
	auto t0 = Clock::now();
	const int nl = (int)m.lists.size();
	std::vector<char> had(nl, 0);
	bool need_bounds = false;
	for (int l = 0; l < nl; ++l) { had[l] = m.lists[l].have_bounds || m.lists[l].ncomp() == 0; need_bounds |= !had[l]; }
	Mesh bm;   // formats of the lists + (after the scan) the bounds; the records stay where they are
	auto bounds_formats = [&] {
		bm.lists.resize((size_t)nl);
		for (int l = 0; l < nl; ++l) {
			AttrList &D = bm.lists[l];
			const AttrList &L = m.lists[l];
			D.target = L.target; D.type = L.type; D.quant = L.quant; D.offset = L.offset; D.count = L.count;
			D.bmin = L.bmin; D.bmax = L.bmax; D.have_bounds = L.have_bounds;
		}
	};
	ShardPlan plan;
	std::unique_ptr<WalkState> marks_p;
	// ---- Round 5: THE PLAN ON THE FIRST CONTEXT'S DEVICE for a large mesh.  The whole mesh goes up there once (four long copies at
	// the link's rate: 54 ms for the 2.9 GB of configs[3]; a freshly read mesh went there anyway, for its twins), its bounds are
	// one k_bounds, its components -- labels, coding order, sizes, new vertices, ties, index intervals -- one device analysis
	// (analysis.cpp: 43 ms), while the host threads build what the walks need whatever the analysis says (the half-edge -> face
	// table, the marks).  About what the host's analysis takes on 16 CPUs, but it leaves the mesh resident: the contexts of
	// THAT device read the first context's arrays (an allocation belongs to the device, not to a stream) instead of bringing their
	// intervals up again -- eight contexts rehearsing on one device had 2.9 GB of interval copies from pageable memory beside
	// their walks (140 - 170 ms, every pin a round of TLB shootdowns for sixteen walking threads).  Contexts on other devices
	// bring their intervals up as before, over their own links.  HRY_SHARD_HOST_PLAN=1: the host's analysis (round 4).
	bool device_plan = false;
	{
		const char *e = getenv("HRY_DEVICE_ANALYSIS_MIN_FACES");
		const uint32_t min_faces = e ? (uint32_t)strtoul(e, nullptr, 10) : (4u << 20);
		device_plan = m.nf >= min_faces && getenv("HRY_SHARD_HOST_PLAN") == nullptr && host_threads() > 1;
	}
	double bounds_ms = 0;
	bool bounds_done = false;
	if (device_plan) {
		Context &cx0 = *cxs[0];
		HIP_OK(hipSetDevice(cx0.device));
		if (m.twins_pending || m.device_token == 0 || m.device_token != cx0.resident_token) cx0.upload_mesh(m);   // (matches the twins of a freshly read mesh)
		st.twins_ms = ms_since(t0);
		mark("the whole mesh on the first context's device");
		BigVec<uint32_t> eface_tab;
		std::exception_ptr failed;
		const unsigned nt = host_threads();
		int ud0 = 0;
		const bool uniform = m.uniform_degree(ud0) && (ud0 == 3 || ud0 == 4);
		std::thread tables([&] {
			try {
				set_thread_budget(nt);
				if (!uniform) {
					eface_tab.resize(m.ne());
					parallel_for(nt, [&](unsigned t) {
						const uint32_t fb = (uint32_t)((uint64_t)m.nf * t / nt), fe = (uint32_t)((uint64_t)m.nf * (t + 1) / nt);
						for (uint32_t f = fb; f < fe; ++f) for (uint32_t h = m.face_off[f]; h < m.face_off[f + 1]; ++h) eface_tab[h] = f;
					});
				}
				marks_p.reset(new WalkState(m.nv, m.nf, nt));
			} catch (...) { failed = std::current_exception(); }
		});
		ComponentAnalysis A;
		try {
			if (need_bounds) {
				const auto tb = Clock::now();
				bounds_formats();
				device_bounds(cx0, bm, &m);   // (the records are resident: the scan alone)
				bounds_ms = ms_since(tb);
				bounds_done = true;
			}
			device_component_analysis(cx0, m, A);
		} catch (...) { tables.join(); throw; }
		tables.join();
		if (failed) std::rethrow_exception(failed);
		if (A.ncomp >= 2) {
			shard_plan_from_analysis(m, (uint32_t)n_shards, std::move(A), plan);
			plan.A.eface = std::move(eface_tab);
		} else device_plan = false;   // (one component: nothing to split; the host's plan says so in its own words)
	}
	if (!device_plan) {
	if (m.twins_pending) cxs[0]->upload_mesh(m, false);   // a freshly read mesh: its half-edge twins are matched on the first context's device
	ensure_twins(m);
	st.twins_ms = ms_since(t0);
	// ---- the bounds of the whole mesh on the first context, beside the plan on the host threads
	std::exception_ptr bounds_err;
	std::thread bounds_thread;
	if (need_bounds && !bounds_done) {
		bounds_thread = std::thread([&] {
			try {
				const auto tb = Clock::now();
				Context &cx = *cxs[0];
				HIP_OK(hipSetDevice(cx.device));
				bounds_formats();
				device_bounds(cx, bm, &m);
				bounds_ms = ms_since(tb);
			} catch (...) { bounds_err = std::current_exception(); }
		});
	}
	std::exception_ptr plan_err;
	try { shard_plan(m, (uint32_t)n_shards, plan, true); } catch (...) { plan_err = std::current_exception(); }
	if (bounds_thread.joinable()) bounds_thread.join();
	if (plan_err) std::rethrow_exception(plan_err);
	if (bounds_err) std::rethrow_exception(bounds_err);
	}
	st.plan_ms = ms_since(t0);
	st.bounds_ms = bounds_ms;
	mark(device_plan ? "plan (device analysis), bounds of the whole mesh" : "plan, bounds of the whole mesh");
	st.n_shards = (uint32_t)n_shards; st.n_contexts = (uint32_t)n_ctx; st.n_components = plan.A.ncomp;
	for (uint32_t k = 0; k < plan.A.ncomp; ++k) st.n_groups += plan.A.group[k] == k;
	if (need_bounds && store_bounds)   // what the reference's reader leaves in the mesh (ply/reader.cc:428)
		for (int l = 0; l < nl; ++l) if (!had[l]) { AttrList &L = m.lists[l]; L.bmin = bm.lists[l].bmin; L.bmax = bm.lists[l].bmax; L.bmin_at.clear(); L.bmax_at.clear(); L.have_bounds = true; }
	const uint32_t ne = m.ne();
	const uint32_t *eface = plan.udeg == 3 || plan.udeg == 4 ? nullptr : plan.A.eface.data();
	if (!(plan.udeg == 3 || plan.udeg == 4) && plan.A.eface.size() != ne) {
		// (a uniform degree other than 3 or 4: the analysis divides, the walk wants the table)
		plan.A.eface.resize(ne);
		for (uint32_t f = 0; f < m.nf; ++f) for (uint32_t h = m.face_off[f]; h < m.face_off[f + 1]; ++h) plan.A.eface[h] = f;
		eface = plan.A.eface.data();
	}
	if (!marks_p) marks_p.reset(new WalkState(m.nv, m.nf, host_threads()));
	WalkState &marks = *marks_p;
	mark("marks");

	std::unique_ptr<ByteSink[]> parts(new ByteSink[(size_t)n_shards]);
	std::vector<double> w_upload(n_ctx, 0.0), w_encode(n_ctx, 0.0);
	// ---- turns (Turnstile above): the walks of the workers on one memory node, the copies to one device.  Measured on the
	// configs[3] mesh, eight contexts on one device and 16 CPUs, and NOT the default: a shard's walk with all sixteen threads takes
	// 20 - 25 ms where an eighth of the parallel walks would be 17 (128 groups over sixteen threads leave some idle at the end, and
	// every walk has its sequential ends), the eight of them 194 ms against 133 side by side -- more than the device phases that hide
	// behind them (encode 317 against 262 ms).  HRY_SHARD_TURNS=1 takes turns.
	const bool turns = n_ctx > 1 && getenv("HRY_SHARD_TURNS") != nullptr;
	// (the copies in turn were measured on one device: each shard's 360 MB of intervals took 30 ms alone -- pageable memory that
	// sixteen walking threads are reading, every pin a round of TLB shootdowns -- 240 ms one after the other against 150 ms for
	// eight staging threads at once; HRY_SHARD_COPY_TURNS=1 for the comparison)
	const bool copy_turns = turns && getenv("HRY_SHARD_COPY_TURNS") != nullptr;
	std::vector<int> walk_group(n_ctx, 0), copy_group(n_ctx, 0);
	std::vector<unsigned> walk_budget(n_ctx, 0);
	std::vector<std::unique_ptr<Turnstile>> walk_turn, copy_turn;
	if (turns) {
		std::vector<const void*> cpus((size_t)n_ctx);
		for (int w = 0; w < n_ctx; ++w) cpus[w] = device_cpus(cxs[w]->device);
		const unsigned allowed = cpu_allowance(), cap = host_threads();
		std::vector<const void*> wkeys;
		std::vector<int> ckeys;
		for (int w = 0; w < n_ctx; ++w) {
			size_t g = std::find(wkeys.begin(), wkeys.end(), cpus[w]) - wkeys.begin();
			if (g == wkeys.size()) { wkeys.push_back(cpus[w]); walk_turn.emplace_back(new Turnstile()); }
			walk_group[w] = (int)g;
			size_t c = std::find(ckeys.begin(), ckeys.end(), cxs[w]->device) - ckeys.begin();
			if (c == ckeys.size()) { ckeys.push_back(cxs[w]->device); copy_turn.emplace_back(new Turnstile()); }
			copy_group[w] = (int)c;
		}
		for (int w = 0; w < n_ctx; ++w) {
			unsigned members = 0;
			for (int x = 0; x < n_ctx; ++x) members += walk_group[x] == walk_group[w];
			const unsigned avail = cpus[w] ? (unsigned)CPU_COUNT((const cpu_set_t*)cpus[w]) : allowed;
			// the node's CPUs, and the group's share of what the process may keep busy
			walk_budget[w] = std::max(1u, std::min(cap, std::min(avail, std::max(1u, allowed * members / (unsigned)n_ctx))));
		}
		for (int sh = 0; sh < n_shards; ++sh) { walk_turn[walk_group[sh % n_ctx]]->order.push_back(sh); copy_turn[copy_group[sh % n_ctx]]->order.push_back(sh); }
	}
	auto abort_turns = [&] { for (auto &t : walk_turn) t->abort(); for (auto &t : copy_turn) t->abort(); };
	// ---- (device plan) the whole mesh is resident on the first context: quantisation happens there once, over whole lists -- the
	// intervals of different shards overlap where they were merged, and a record must be quantised exactly once
	if (device_plan && (nq || clear)) {
		Context &cx0 = *cxs[0];
		Mesh whole_sk;
		for (int l = 0; l < 2; ++l) {
			const AttrList &L = m.lists[l];
			AttrList &D = whole_sk.lists[l];
			D.target = L.target; D.type = L.type; D.quant = L.quant; D.offset = L.offset;
			D.interp_off = L.interp_off; D.interp_len = L.interp_len; D.interp_name = L.interp_name;
			D.count = L.count;
			const AttrList &B = had[l] ? L : bm.lists[l];
			D.bmin = B.bmin; D.bmax = B.bmax; D.have_bounds = true;
		}
		const std::vector<std::vector<uint8_t>> to = requant_targets(whole_sk, q, nq, clear);
		const auto tq = Clock::now();
		HIP_OK(hipSetDevice(cx0.device));
		for (int l = 0; l < 2; ++l) {
			AttrList &L = whole_sk.lists[l];
			if (to[l] == L.quant || !L.count) continue;
			dev::launch_requant(cx0.stream, cx0.d_rec[l].as<uint8_t>(), L.count, L.stride(), requant_plan(L, to[l]));
		}
		HIP_OK(hipStreamSynchronize(cx0.stream));
		st.quant_ms = ms_since(tq);   // (contexts on other devices quantise their intervals behind their copies: part of extract_ms)
		cx0.resident_token = 0;   // (the resident records are no longer the host mesh's)
	}
	t0 = Clock::now();
	for (int w = 0; w < n_ctx; ++w) cxs[w]->inplace_twin_patches.clear();
	// (HRY_SHARD_FOREIGN_CONTEXTS=1, for tests on a one-GPU box: every context but the first behaves as one on another device --
	// it brings its own intervals up and repairs its own copy of the twins)
	const bool foreign_ctx = getenv("HRY_SHARD_FOREIGN_CONTEXTS") != nullptr;
	auto on_first_device = [&](int w) { return w == 0 || (!foreign_ctx && cxs[w]->device == cxs[0]->device); };
	run_workers(cxs, n_ctx, [&](int w) {
	  // (device plan) a context on the first context's device reads that context's arrays: lent for the length of this call
	  const bool shares0 = device_plan && on_first_device(w);
	  struct Lent { DevBuf *b; void *p; size_t cap; };
	  std::vector<Lent> lent;
	  struct GiveBack { std::vector<Lent> &l; ~GiveBack() { for (const Lent &x : l) { x.b->p = x.p; x.b->cap = x.cap; } } } give_back{ lent };
	  try {
		Context &cx = *cxs[w];
		HIP_OK(hipSetDevice(cx.device));
		hry_timing acc{};
		bool arrays_ready = false;
		const unsigned own_budget = host_threads();   // (run_workers: this worker's share)
		for (int s = w; s < n_shards; s += n_ctx) {
			// ---- the shard: its components, its runs, the intervals it lies in
			ComponentAnalysis part;
			Mesh sk;   // the skeleton encode_chunked describes the segment with
			shard_components(plan, (uint32_t)s, part, sk.shard);
			std::vector<std::pair<uint32_t, uint32_t>> fiv, viv;
			shard_intervals(plan, (uint32_t)s, 4096, fiv, viv);
			for (const ShardRun &r : sk.shard.runs) { sk.nf += r.n_faces; sk.nv += r.n_vertices; }
			sk.have_degree = m.have_degree;
			for (int l = 0; l < 2; ++l) {
				const AttrList &L = m.lists[l];
				AttrList &D = sk.lists[l];
				D.target = L.target; D.type = L.type; D.quant = L.quant; D.offset = L.offset;
				D.interp_off = L.interp_off; D.interp_len = L.interp_len; D.interp_name = L.interp_name;
				D.count = l == 0 ? sk.nf : sk.nv;
				const AttrList &B = had[l] ? L : bm.lists[l];
				D.bmin = B.bmin; D.bmax = B.bmax; D.have_bounds = true;
			}
			// ---- the whole mesh's arrays on this device, filled over the shard's intervals
			auto t = Clock::now();
			if (!arrays_ready && shares0) {
				if (w != 0) {
					Context &c0 = *cxs[0];
					auto lend = [&](DevBuf &dst, DevBuf &src) { lent.push_back(Lent{ &dst, dst.p, dst.cap }); dst.p = src.p; dst.cap = src.cap; };
					lend(cx.d_org, c0.d_org); lend(cx.d_twin, c0.d_twin); lend(cx.d_foff, c0.d_foff); lend(cx.d_eface, c0.d_eface);
					for (int l = 0; l < 2; ++l) lend(cx.d_rec[l], c0.d_rec[l]);
					cx.res_has_eface = c0.res_has_eface; cx.res_udeg = c0.res_udeg;
					cx.res_nv = m.nv; cx.res_nf = m.nf; cx.res_ne = ne;
					cx.resident_token = 0;
				}
				arrays_ready = true;
			}
			if (!arrays_ready) {
				cx.d_org.ensure(std::max<size_t>((size_t)ne * 4, 16)); cx.d_twin.ensure(std::max<size_t>((size_t)ne * 4, 16));
				cx.d_foff.ensure(((size_t)m.nf + 1) * 4);
				for (int l = 0; l < 2; ++l) cx.d_rec[l].ensure(std::max<size_t>(m.lists[l].data.size(), 16));
				int ud = 0;
				cx.res_has_eface = !m.uniform_degree(ud);
				cx.res_udeg = (uint32_t)ud;
				if (cx.res_has_eface) cx.d_eface.ensure(std::max<size_t>((size_t)ne * 4, 16));
				cx.res_nv = m.nv; cx.res_nf = m.nf; cx.res_ne = ne;
				cx.resident_token = 0;
				arrays_ready = true;
			}
			// (the copies run on a thread of their own, on a stream of their own: from pageable memory every copy keeps its caller
			// until it is staged -- 126 ms per shard of the configs[3] mesh with eight workers on one link -- and the walk does not
			// need the device; the main stream waits for the event before the kernels of this shard, see encode_chunked)
			const size_t st0 = (size_t)m.lists[0].stride(), st1 = (size_t)m.lists[1].stride();
			cx.ensure_second_stream();
			// quantisation of the records happens on the device, over the same intervals, behind the copies; the skeleton announces it
			struct ListPlan { int l; size_t stride; dev::RequantPlan plan; };
			std::vector<ListPlan> rplans;
			if (nq || clear) {
				const std::vector<std::vector<uint8_t>> to = requant_targets(sk, q, nq, clear);
				for (int l = 0; l < 2; ++l) {
					AttrList &L = sk.lists[l];
					if (to[l] == L.quant) continue;
					rplans.push_back(ListPlan{ l, (size_t)L.stride(), requant_plan(L, to[l]) });
					L.quant = to[l];
				}
			}
			if (shares0) rplans.clear();   // (quantised above, once for the whole lists)
			std::exception_ptr up_err;
			double up_ms = 0;
			std::thread uploader;
			if (!shares0) uploader = std::thread([&] {
				struct Turn { Turnstile *t; int s; ~Turn() { if (t) t->leave(s); } } turn{ copy_turns ? copy_turn[copy_group[w]].get() : nullptr, s };
				try {
					if (turn.t) turn.t->enter(s);   // (the link to this device: one shard's intervals after the other)
					const auto tu = Clock::now();
					HIP_OK(hipSetDevice(cx.device));
					hipStream_t us = cx.stream2;
					// (the twins go up while this worker's and the others' walks repair some of them in the host array: whichever value
					// a repaired entry arrives with, every repaired entry is sent again as a (half-edge, twin) pair once its group is
					// walked -- by the encode's pipeline, or by upload_repaired_twins(patches_only) -- and both wait for these copies)
					for (const auto &iv : fiv) {
						const size_t h0 = m.face_off[iv.first], h1 = m.face_off[iv.second];
						HIP_OK(hipMemcpyAsync(cx.d_foff.as<uint32_t>() + iv.first, m.face_off.data() + iv.first, ((size_t)iv.second - iv.first + 1) * 4, hipMemcpyHostToDevice, us));
						if (h1 > h0) {
							HIP_OK(hipMemcpyAsync(cx.d_org.as<uint32_t>() + h0, m.org.data() + h0, (h1 - h0) * 4, hipMemcpyHostToDevice, us));
							HIP_OK(hipMemcpyAsync(cx.d_twin.as<uint32_t>() + h0, m.twin.data() + h0, (h1 - h0) * 4, hipMemcpyHostToDevice, us));
						}
						if (st0) HIP_OK(hipMemcpyAsync(cx.d_rec[0].as<uint8_t>() + iv.first * st0, m.lists[0].data.data() + iv.first * st0, ((size_t)iv.second - iv.first) * st0, hipMemcpyHostToDevice, us));
						if (cx.res_has_eface) dev::launch_edge_faces(us, cx.d_foff.as<uint32_t>(), iv.second, cx.d_eface.as<uint32_t>(), iv.first);
					}
					if (st1) for (const auto &iv : viv)
						HIP_OK(hipMemcpyAsync(cx.d_rec[1].as<uint8_t>() + iv.first * st1, m.lists[1].data.data() + iv.first * st1, ((size_t)iv.second - iv.first) * st1, hipMemcpyHostToDevice, us));
					for (const ListPlan &rp : rplans)
						for (const auto &iv : rp.l == 0 ? fiv : viv)
							dev::launch_requant(us, cx.d_rec[rp.l].as<uint8_t>() + iv.first * rp.stride, iv.second - iv.first, (int)rp.stride, rp.plan);
					if (turn.t) { turn.t->leave(s); turn.t = nullptr; }   // (every copy is staged: the next shard's may start while the last of these land)
					HIP_OK(hipStreamSynchronize(us));
					up_ms = ms_since(tu);
				} catch (...) { up_err = std::current_exception(); }
			});
			struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{ uploader };
			t = Clock::now();
			bool walk_entered = false, walk_left = false;
			Turnstile *wt = turns ? walk_turn[walk_group[w]].get() : nullptr;
			const InPlaceShard ip{ &m, &part, eface, &marks, &fiv, [&] {
				if (uploader.joinable()) uploader.join();
				if (up_err) std::rethrow_exception(up_err);
				w_upload[w] += up_ms;
			}, [&] {   // before the walk: this shard's turn at the node's CPUs, all of them
				if (!wt) return;
				wt->enter(s); walk_entered = true;
				set_thread_budget(walk_budget[w]);
			}, [&] {   // after it
				if (!wt) return;
				set_thread_budget(own_budget);
				wt->leave(s); walk_left = true;
			} };
			struct Pass { Turnstile *t; int s; bool &left; ~Pass() { if (t && !left) { t->enter(s); t->leave(s); } } } pass{ wt, s, walk_left };   // (a shard that never walked passes its turn on)
			encode_chunked(cx, sk, chunk_syms, parts[s], &ip);
			(void)walk_entered;
			w_encode[w] += ms_since(t);
			const hry_timing &tm = cx.timing;
			acc.host_walk_ms += tm.host_walk_ms; acc.h2d_ms += tm.h2d_ms; acc.device_ms += tm.device_ms; acc.d2h_ms += tm.d2h_ms;
			acc.k_predict_ms += tm.k_predict_ms; acc.k_entropy_ms += tm.k_entropy_ms; acc.n_symbols += tm.n_symbols; acc.payload_bytes += tm.payload_bytes;
			acc.total_ms += tm.total_ms;
		}
		cx.timing = acc;
	  } catch (...) { abort_turns(); throw; }
	});
	// (device plan) the whole mesh stays resident on the first context.  Workers on its device repaired their twins in its arrays;
	// workers on OTHER devices repaired the host array and their own copies only -- their (half-edge, twin) pairs go to the resident
	// copy now, or a later hry_encode of the resident mesh would predict from stale twins (its walk finds nothing left to repair)
	if (device_plan && cxs[0]->resident_token != 0 && cxs[0]->resident_token == m.device_token) {
		Context &cx0 = *cxs[0];
		WalkResult foreign;
		for (int w = 1; w < n_ctx; ++w) if (!on_first_device(w)) {
			const std::vector<uint32_t> &tp = cxs[w]->inplace_twin_patches;
			foreign.twin_patches.insert(foreign.twin_patches.end(), tp.begin(), tp.end());
		}
		if (!foreign.twin_patches.empty()) {
			foreign.twins_changed = true;
			HIP_OK(hipSetDevice(cx0.device));
			upload_repaired_twins(cx0, m, foreign, true);
			HIP_OK(hipStreamSynchronize(cx0.stream));
		}
	}
	for (int w = 0; w < n_ctx; ++w) { cxs[w]->inplace_twin_patches.clear(); cxs[w]->inplace_twin_patches.shrink_to_fit(); }
	st.phase_b_ms = ms_since(t0);
	mark("segments");
	t0 = Clock::now();
	std::vector<const uint8_t*> pp;
	std::vector<size_t> ps;
	for (int s = 0; s < n_shards; ++s) { pp.push_back(parts[s].data()); ps.push_back(parts[s].size()); }
	merge_containers(pp.data(), ps.data(), pp.size(), out);
	st.merge_ms = ms_since(t0);
	auto mx = [](const std::vector<double> &v) { double x = 0; for (double y : v) x = std::max(x, y); return x; };
	st.extract_ms = mx(w_upload); st.encode_ms = mx(w_encode);   // (quant_ms: set where the whole lists were quantised on the first context)
	for (int w = 0; w < n_ctx; ++w) st.host_walk_ms = std::max(st.host_walk_ms, cxs[w]->timing.host_walk_ms);
	st.n_segments = merged_segments(out);
	st.total_ms = ms_since(t_all);
	mark("one container");
}

void encode_sharded(Context *const *cxs, int n_ctx, Mesh &m, const hry_quant *q, size_t nq, bool clear, int n_shards, int chunk_syms,
                    ByteSink &out, hry_shard_timing &st, bool store_bounds)
{
	const auto t_all = Clock::now();
	st = hry_shard_timing{};
	check_contexts(cxs, n_ctx);
	if (n_shards <= 0) n_shards = n_ctx;
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh");
	if (!m.general && !getenv("HRY_SHARD_EXTRACT")) { encode_sharded_in_place(cxs, n_ctx, m, q, nq, clear, n_shards, chunk_syms, out, st, store_bounds); return; }
	// ---- plan, once
	auto t0 = Clock::now();
	// a freshly read mesh: its half-edge twins are matched on the first context's device (twins.hip: 0.3 ms per million triangles
	// + the connectivity's trip over PCIe) rather than by the host's hash buckets
	if (m.twins_pending && !m.general) cxs[0]->upload_mesh(m, false);
	ensure_twins(m);
	st.twins_ms = ms_since(t0);
	ShardPlan plan;
	shard_plan(m, (uint32_t)n_shards, plan);
	st.plan_ms = ms_since(t0);
	st.n_shards = (uint32_t)n_shards; st.n_contexts = (uint32_t)n_ctx; st.n_components = plan.A.ncomp;
	for (uint32_t k = 0; k < plan.A.ncomp; ++k) st.n_groups += plan.A.group[k] == k;
	const int nl = (int)m.lists.size();
	std::vector<char> had(nl, 0);
	bool need_bounds = false;
	for (int l = 0; l < nl; ++l) { had[l] = m.lists[l].have_bounds || m.lists[l].ncomp() == 0; need_bounds |= !had[l]; }

	// ---- phase A: extract (+ bounds of the shard)
	std::vector<std::unique_ptr<Mesh>> shards((size_t)n_shards);
	std::vector<double> w_extract(n_ctx, 0.0), w_bounds(n_ctx, 0.0), w_quant(n_ctx, 0.0), w_encode(n_ctx, 0.0);
	t0 = Clock::now();
	run_workers(cxs, n_ctx, [&](int w) {
		for (int s = w; s < n_shards; s += n_ctx) {
			auto t = Clock::now();
			shards[s].reset(shard_extract(m, plan, (uint32_t)s));
			w_extract[w] += ms_since(t);
			if (need_bounds) {
				t = Clock::now();
				device_bounds(*cxs[w], *shards[s]);   // uploads the shard: it stays resident for phase B when the worker has one shard
				w_bounds[w] += ms_since(t);
			}
		}
	});
	st.phase_a_ms = ms_since(t0);
	// ---- bounds of the whole mesh
	t0 = Clock::now();
	if (need_bounds) {
		std::vector<const Mesh*> view;
		for (auto &s : shards) view.push_back(s.get());
		for (int l = 0; l < nl; ++l) {
			std::vector<uint8_t> bmin, bmax;
			if (had[l]) { bmin = m.lists[l].bmin; bmax = m.lists[l].bmax; }
			else combine_shard_bounds(view, l, bmin, bmax);
			for (auto &s : shards) {
				AttrList &L = s->lists[l];
				L.bmin = bmin; L.bmax = bmax; L.bmin_at.clear(); L.bmax_at.clear(); L.have_bounds = true;
			}
			if (!had[l] && store_bounds) {   // what the reference's reader leaves in the mesh (ply/reader.cc:428)
				AttrList &L = m.lists[l];
				L.bmin = bmin; L.bmax = bmax; L.bmin_at.clear(); L.bmax_at.clear(); L.have_bounds = true;
			}
		}
	}
	st.combine_ms = ms_since(t0);
	// ---- phase B: quantisation + encode, one segment per shard
	std::unique_ptr<ByteSink[]> parts(new ByteSink[(size_t)n_shards]);
	t0 = Clock::now();
	run_workers(cxs, n_ctx, [&](int w) {
		hry_timing acc{};
		for (int s = w; s < n_shards; s += n_ctx) {
			Context &cx = *cxs[w];
			auto t = Clock::now();
			if (nq || clear) device_requant(cx, *shards[s], q, nq, clear);
			w_quant[w] += ms_since(t);
			t = Clock::now();
			encode_chunked(cx, *shards[s], chunk_syms, parts[s]);
			w_encode[w] += ms_since(t);
			const hry_timing &tm = cx.timing;
			acc.host_walk_ms += tm.host_walk_ms; acc.h2d_ms += tm.h2d_ms; acc.device_ms += tm.device_ms; acc.d2h_ms += tm.d2h_ms;
			acc.k_predict_ms += tm.k_predict_ms; acc.k_entropy_ms += tm.k_entropy_ms; acc.n_symbols += tm.n_symbols; acc.payload_bytes += tm.payload_bytes;
			acc.total_ms += tm.total_ms;
			shards[s].reset();
		}
		cxs[w]->timing = acc;
	});
	st.phase_b_ms = ms_since(t0);
	// ---- one container
	t0 = Clock::now();
	std::vector<const uint8_t*> pp;
	std::vector<size_t> ps;
	for (int s = 0; s < n_shards; ++s) { pp.push_back(parts[s].data()); ps.push_back(parts[s].size()); }
	merge_containers(pp.data(), ps.data(), pp.size(), out);
	st.merge_ms = ms_since(t0);
	auto mx = [](const std::vector<double> &v) { double x = 0; for (double y : v) x = std::max(x, y); return x; };
	st.extract_ms = mx(w_extract); st.bounds_ms = mx(w_bounds); st.quant_ms = mx(w_quant); st.encode_ms = mx(w_encode);
	for (int w = 0; w < n_ctx; ++w) st.host_walk_ms = std::max(st.host_walk_ms, cxs[w]->timing.host_walk_ms);
	st.n_segments = merged_segments(out);
	st.total_ms = ms_since(t_all);
}

// ---------------------------------------------------------------------------------------------------------------------
// decode of a sharded container: the segments s with s % shard_count == shard_index (shard_count <= 1: all), spread over the
// contexts.  allow_partial: the caller accepts a mesh that holds only a share (its own segments, or a container that is itself
// one rank's part); otherwise every face and half-edge must end up decoded.
Mesh *decode_sharded(Context *const *cxs, int n_ctx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> g, int shard_index, int shard_count,
                     bool allow_partial, hry_shard_timing *st_out)
{
	const auto t_all = Clock::now();
	hry_shard_timing st{};
	check_contexts(cxs, n_ctx);
	if (shard_count < 0 || shard_index < 0 || (shard_count > 0 && shard_index >= shard_count)) throw Error(HRY_E_ARG, "invalid shard selection");
	const uint32_t gnv = g->nv, gnf = g->nf, gne = g->declared_ne;
	auto t0 = Clock::now();
	ShardedDirectory dir;
	const bool general = g->general;
	const size_t nl = g->lists.size();
	std::vector<uint32_t> list_counts;
	for (const AttrList &L : g->lists) list_counts.push_back(L.count);
	parse_sharded_directory(p, n, hdr, gnv, gnf, gne, dir, true, general ? &list_counts : nullptr);
	const uint32_t nseg = (uint32_t)dir.segments.size();
	std::vector<uint32_t> mine;
	for (uint32_t si = 0; si < nseg; ++si) if (shard_count <= 1 || (int)(si % (uint32_t)shard_count) == shard_index) mine.push_back(si);
	const bool everything = dir.complete && mine.size() == nseg;
	if (!everything && !allow_partial)
		throw Error(HRY_E_FORMAT, dir.complete ? "a share of the segments decodes into a partial mesh: ask for it (HRY_FLAG_PARTIAL)"
		                                       : "sharded container does not cover the mesh (missing segments); HRY_FLAG_PARTIAL decodes what is there");
	st.plan_ms = ms_since(t0);
	st.n_segments = (uint32_t)mine.size(); st.n_contexts = (uint32_t)n_ctx; st.n_shards = nseg;
	// ---- the whole mesh's arrays.  BigVec does not fill on resize: covered ranges are written by the placement below, the rest
	// by the filler pass after it.
	g->face_off.resize((size_t)gnf + 1);
	g->org.resize(gne); g->twin.resize(gne);
	for (size_t l = 0; l < nl; ++l) {
		AttrList &L = g->lists[l];
		if (general) L.data.assign((size_t)L.count * L.stride(), 0);   // (OBJ-sized; a record no run creates stays zero)
		else L.data.resize((size_t)L.count * L.stride());
	}
	if (general) {   // element -> region and element x slot -> record, for the whole mesh (filler: region 0, record 0)
		Bindings &b = g->bind;
		b.face_reg.assign(gnf, 0); b.vtx_reg.assign(gnv, 0);
		b.face_attr.assign((size_t)gnf * b.nb_face, 0); b.vtx_attr.assign((size_t)gnv * b.nb_vtx, 0); b.corner_attr.assign((size_t)gne * b.nb_corner, 0);
	}
	g->covered.clear(); g->covered_records.clear();
	const size_t vstride = general ? 0 : (size_t)g->lists[1].stride(), fstride = general ? 0 : (size_t)g->lists[0].stride();

	std::vector<double> w_decode(n_ctx, 0.0), w_place(n_ctx, 0.0);
	t0 = Clock::now();
	run_workers(cxs, n_ctx, [&](int w) {
		Context &cx = *cxs[w];
		hry_timing acc{};
		const unsigned nt = std::max(1u, host_threads());
		for (size_t k = (size_t)w; k < mine.size(); k += (size_t)n_ctx) {
			const ShardedDirectory::Segment &sg = dir.segments[mine[k]];
			const std::vector<ShardRun> &runs = sg.runs;
			const uint32_t nr = (uint32_t)runs.size();
			const uint8_t *sp = p + sg.offset;
			std::vector<uint32_t> cv(nr + 1, 0), cf(nr + 1, 0), ch(nr + 1, 0);
			for (uint32_t j = 0; j < nr; ++j) { cv[j + 1] = cv[j] + runs[j].n_vertices; cf[j + 1] = cf[j] + runs[j].n_faces; ch[j + 1] = ch[j] + runs[j].n_halfedges; }
			const uint32_t lnv = sg.nv, lnf = sg.nf, lne = sg.ne;
			// the shard as a mesh of its own: formats and bounds of the whole, sizes of the runs
			auto t = Clock::now();
			std::unique_ptr<Mesh> lm(new Mesh());
			lm->nv = lnv; lm->nf = lnf; lm->declared_ne = lne;
			lm->have_degree = g->have_degree;
			if (general) {   // the regions of the whole mesh; every list with the records this segment creates
				lm->general = true;
				lm->lists.assign(nl, AttrList());
				Bindings &lb = lm->bind;
				const Bindings &b = g->bind;
				lb.reg_facelist = b.reg_facelist; lb.reg_vtxlist = b.reg_vtxlist; lb.reg_cornerlist = b.reg_cornerlist;
				lb.off_facelist = b.off_facelist; lb.off_vtxlist = b.off_vtxlist; lb.off_cornerlist = b.off_cornerlist;
				lb.nb_face = b.nb_face; lb.nb_vtx = b.nb_vtx; lb.nb_corner = b.nb_corner;
				lb.face_reg.assign(lnf, 0); lb.vtx_reg.assign(lnv, 0);   // (what the header reader allocates for an unsharded file)
				lb.face_attr.assign((size_t)lnf * lb.nb_face, 0); lb.vtx_attr.assign((size_t)lnv * lb.nb_vtx, 0);
			}
			for (size_t l = 0; l < nl; ++l) {
				const AttrList &L = g->lists[l];
				AttrList &D = lm->lists[l];
				D.target = L.target; D.type = L.type; D.quant = L.quant; D.offset = L.offset;
				D.interp_off = L.interp_off; D.interp_len = L.interp_len; D.interp_name = L.interp_name;
				D.bmin = L.bmin; D.bmax = L.bmax; D.have_bounds = true;
				D.count = general ? sg.nrec[l] : l == 0 ? lm->nf : lm->nv;
				D.data.assign((size_t)D.count * D.stride(), 0);
			}
			std::unique_ptr<Mesh> dm(decode_chunked(cx, sp + sg.body_at, sg.bytes - sg.body_at, 0, std::move(lm)));
			w_decode[w] += ms_since(t);
			const hry_timing &tm = cx.timing;
			acc.host_walk_ms += tm.host_walk_ms; acc.h2d_ms += tm.h2d_ms; acc.device_ms += tm.device_ms; acc.d2h_ms += tm.d2h_ms;
			acc.k_predict_ms += tm.k_predict_ms; acc.k_entropy_ms += tm.k_entropy_ms; acc.k_chain_ms += tm.k_chain_ms;
			acc.n_symbols += tm.n_symbols; acc.payload_bytes += tm.payload_bytes;
			if (dm->ne() != lne || dm->nf != lnf || dm->nv != lnv) throw Error(HRY_E_FORMAT, "corrupt segment (sizes do not match its runs)");
			// ---- into the numbering of the whole mesh
			t = Clock::now();
			BigVec<uint32_t> l2g;
			l2g.resize((size_t)lnv);
			struct Task { uint32_t run, kind, b, e; };   // kind 0 vertices, 1 faces, 2 half-edges
			std::vector<Task> tasks;
			const uint32_t grain = 1u << 16;
			for (uint32_t j = 0; j < nr; ++j) {
				for (uint32_t b = 0; b < runs[j].n_vertices; b += grain) tasks.push_back(Task{ j, 0, b, std::min(runs[j].n_vertices, b + grain) });
				for (uint32_t b = 0; b < runs[j].n_faces; b += grain) tasks.push_back(Task{ j, 1, b, std::min(runs[j].n_faces, b + grain) });
			}
			const size_t n_vf_tasks = tasks.size();
			for (uint32_t j = 0; j < nr; ++j)
				for (uint32_t b = 0; b < runs[j].n_halfedges; b += grain) tasks.push_back(Task{ j, 2, b, std::min(runs[j].n_halfedges, b + grain) });
			std::atomic<bool> bad{ false };
			auto run_tasks = [&](size_t from, size_t to) {
				std::atomic<size_t> next{ from };
				parallel_for((unsigned)std::max<size_t>(1, std::min<size_t>(nt, (to - from + 3) / 4)), [&](unsigned) {
					for (;;) {
						const size_t ti = next.fetch_add(1, std::memory_order_relaxed);
						if (ti >= to) break;
						const Task &tk = tasks[ti];
						const ShardRun &r = runs[tk.run];
						if (tk.kind == 0) {
							for (uint32_t i = tk.b; i < tk.e; ++i) l2g[cv[tk.run] + i] = r.first_vertex + i;
							if (vstride) memcpy(g->lists[1].data.data() + ((size_t)r.first_vertex + tk.b) * vstride, dm->lists[1].data.data() + ((size_t)cv[tk.run] + tk.b) * vstride, (size_t)(tk.e - tk.b) * vstride);
						} else if (tk.kind == 1) {
							const uint32_t shift = r.first_halfedge - ch[tk.run];   // modulo 2^32: local half-edge + shift = half-edge of the whole mesh
							for (uint32_t i = tk.b; i < tk.e; ++i) {
								const uint32_t lo = dm->face_off[(size_t)cf[tk.run] + i], hi = dm->face_off[(size_t)cf[tk.run] + i + 1];
								if (lo > hi || hi > ch[tk.run + 1] || lo < ch[tk.run]) { bad.store(true, std::memory_order_relaxed); continue; }
								g->face_off[(size_t)r.first_face + i + 1] = hi + shift;
							}
							if (tk.b == 0 && dm->face_off[cf[tk.run]] != ch[tk.run]) bad.store(true, std::memory_order_relaxed);
							if (tk.e == r.n_faces && dm->face_off[(size_t)cf[tk.run] + r.n_faces] != ch[tk.run + 1]) bad.store(true, std::memory_order_relaxed);
							if (fstride) memcpy(g->lists[0].data.data() + ((size_t)r.first_face + tk.b) * fstride, dm->lists[0].data.data() + ((size_t)cf[tk.run] + tk.b) * fstride, (size_t)(tk.e - tk.b) * fstride);
						} else {
							const uint32_t lo = ch[tk.run], hi = ch[tk.run + 1], shift = r.first_halfedge - lo;
							for (uint32_t i = tk.b; i < tk.e; ++i) {
								const uint32_t h = lo + i, tw = dm->twin[h], v = dm->org[h];
								if (tw < lo || tw >= hi || v >= lnv) { bad.store(true, std::memory_order_relaxed); g->org[(size_t)r.first_halfedge + i] = 0; g->twin[(size_t)r.first_halfedge + i] = r.first_halfedge + i; continue; }
								g->org[(size_t)r.first_halfedge + i] = l2g[v];
								g->twin[(size_t)r.first_halfedge + i] = tw + shift;
							}
						}
					}
				});
			};
			run_tasks(0, n_vf_tasks);               // the vertex map first: half-edges of one run may name vertices of another
			run_tasks(n_vf_tasks, tasks.size());
			if (bad.load()) throw Error(HRY_E_FORMAT, "corrupt segment (runs do not match the connectivity)");
			if (general) {
				// the records of every list into the numbering of the whole (creation order over all components: run j's records
				// of list l sit at its first_record), then every element's region and slots
				const Bindings &db = dm->bind;
				Bindings &b = g->bind;
				if (db.face_reg.size() != lnf || db.vtx_reg.size() != lnv || db.face_attr.size() != (size_t)lnf * b.nb_face || db.vtx_attr.size() != (size_t)lnv * b.nb_vtx ||
				    db.corner_attr.size() != (size_t)lne * b.nb_corner)
					throw Error(HRY_E_FORMAT, "corrupt segment (binding tables)");
				std::vector<std::vector<uint32_t>> rl2g(nl);
				for (size_t l = 0; l < nl; ++l) {
					const AttrList &S = dm->lists[l];
					AttrList &D = g->lists[l];
					if (S.count != sg.nrec[l]) throw Error(HRY_E_FORMAT, "corrupt segment (records of a list do not match its runs)");
					rl2g[l].resize(S.count);
					const size_t st = (size_t)D.stride();
					uint32_t at = 0;
					for (uint32_t j = 0; j < nr; ++j) {
						const uint32_t first = sg.run_records[(size_t)j * 2 * nl + 2 * l], cnt = sg.run_records[(size_t)j * 2 * nl + 2 * l + 1];
						for (uint32_t i = 0; i < cnt; ++i) rl2g[l][at + i] = first + i;
						if (st && cnt) memcpy(D.data.data() + (size_t)first * st, S.data.data() + (size_t)at * st, (size_t)cnt * st);
						at += cnt;
					}
				}
				auto rec = [&](int l, uint32_t r) -> uint32_t { if ((size_t)l >= nl || r >= rl2g[l].size()) throw Error(HRY_E_FORMAT, "corrupt segment (record index)"); return rl2g[l][r]; };
				for (uint32_t j = 0; j < nr; ++j) {
					const ShardRun &r = runs[j];
					for (uint32_t i = 0; i < r.n_vertices; ++i) {
						const uint32_t lv = cv[j] + i, gv2 = r.first_vertex + i;
						const int reg = db.vtx_reg[lv];
						if (reg >= b.nregs_vtx()) throw Error(HRY_E_FORMAT, "corrupt segment (vertex region)");
						b.vtx_reg[gv2] = (uint16_t)reg;
						for (int a = 0; a < b.nvtxlists(reg); ++a) b.vtx_attr[(size_t)gv2 * b.nb_vtx + a] = rec(b.vtxlist(reg, a), db.vtx_attr[(size_t)lv * b.nb_vtx + a]);
					}
					for (uint32_t i = 0; i < r.n_faces; ++i) {
						const uint32_t lf = cf[j] + i, gf = r.first_face + i;
						const int reg = db.face_reg[lf];
						if (reg >= b.nregs_face()) throw Error(HRY_E_FORMAT, "corrupt segment (face region)");
						b.face_reg[gf] = (uint16_t)reg;
						for (int a = 0; a < b.nfacelists(reg); ++a) b.face_attr[(size_t)gf * b.nb_face + a] = rec(b.facelist(reg, a), db.face_attr[(size_t)lf * b.nb_face + a]);
						const uint32_t shift = r.first_halfedge - ch[j];
						for (uint32_t h = dm->face_off[lf]; h < dm->face_off[(size_t)lf + 1]; ++h)
							for (int a = 0; a < b.ncornerlists(reg); ++a)
								b.corner_attr[(size_t)(h + shift) * b.nb_corner + a] = rec(b.cornerlist(reg, a), db.corner_attr[(size_t)h * b.nb_corner + a]);
					}
				}
			}
			w_place[w] += ms_since(t);
		}
		cx.timing = acc;
	});
	st.phase_b_ms = ms_since(t0);
	// ---- what no decoded run covers gets filler: faces without half-edges (the last face of a gap takes the gap's half-edges, so
	// the offsets stay monotone and inside the arrays), half-edges that are borders at vertex 0, zero records.  In a complete decode
	// that is exactly the vertices no face references (the reference never codes them: zero records).
	t0 = Clock::now();
	for (uint32_t si : mine) {
		g->covered.insert(g->covered.end(), dir.segments[si].runs.begin(), dir.segments[si].runs.end());
		g->covered_records.insert(g->covered_records.end(), dir.segments[si].run_records.begin(), dir.segments[si].run_records.end());
	}
	std::vector<ShardRun> byf;
	for (const ShardRun &r : g->covered) if (r.n_faces) byf.push_back(r);
	std::sort(byf.begin(), byf.end(), [](const ShardRun &a, const ShardRun &b) { return a.first_face < b.first_face; });
	{
		uint64_t f_at = 0, h_at = 0;
		auto gap = [&](uint64_t f_to, uint64_t h_to) {   // faces [f_at, f_to), half-edges [h_at, h_to) belong to nobody
			for (uint64_t f = f_at; f < f_to; ++f) g->face_off[(size_t)f] = (uint32_t)h_at;
			for (uint64_t h = h_at; h < h_to; ++h) { g->org[(size_t)h] = 0; g->twin[(size_t)h] = (uint32_t)h; }
			if (fstride && f_to > f_at) memset(g->lists[0].data.data() + (size_t)f_at * fstride, 0, (size_t)(f_to - f_at) * fstride);
		};
		for (const ShardRun &r : byf) {
			gap(r.first_face, r.first_halfedge);
			g->face_off[r.first_face] = r.first_halfedge;
			f_at = (uint64_t)r.first_face + r.n_faces; h_at = (uint64_t)r.first_halfedge + r.n_halfedges;
		}
		gap(gnf, gne);
		g->face_off[gnf] = gne;
		if (h_at < gne && f_at == gnf && gnf) {   // half-edges behind the last face: nothing may own them
			throw Error(HRY_E_FORMAT, "corrupt sharded container (half-edges outside every face)");
		}
		std::vector<std::pair<uint32_t, uint32_t>> vr;
		for (const ShardRun &r : g->covered) if (r.n_vertices) vr.push_back({ r.first_vertex, r.n_vertices });
		std::sort(vr.begin(), vr.end());
		uint64_t at = 0;
		auto zero = [&](uint64_t b, uint64_t e) { if (e > b && vstride) memset(g->lists[1].data.data() + b * vstride, 0, (size_t)(e - b) * vstride); };
		for (const auto &r : vr) { zero(at, r.first); at = (uint64_t)r.first + r.second; }
		zero(at, gnv);
	}
	st.merge_ms = ms_since(t0);
	g->partial = !everything;
	auto mx = [](const std::vector<double> &v) { double x = 0; for (double y : v) x = std::max(x, y); return x; };
	st.encode_ms = mx(w_decode); st.extract_ms = mx(w_place);
	for (int w = 0; w < n_ctx; ++w) st.host_walk_ms = std::max(st.host_walk_ms, cxs[w]->timing.host_walk_ms);
	st.total_ms = ms_since(t_all);
	for (int w = 0; w < n_ctx; ++w) cxs[w]->timing.total_ms = st.total_ms;
	if (st_out) *st_out = st;
	g->device_token = 0;
	return g.release();
}

}   // namespace hry
