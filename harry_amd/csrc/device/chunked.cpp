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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
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
	WalkResult w;
	w.numtri_positions = false;     // (places in ONE symbol sequence: the chunked planes have none)
	bool walked = false;
	if (in_place) { cut_border_walk_in_place(*in_place->whole, *in_place->part, in_place->eface, *in_place->marks, w); walked = true; }
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
			cut_border_walk_in_place(m, A, uniform ? nullptr : eface.data(), *marks, w);
			walked = true;
		}
	}
	if (!walked) cut_border_walk(m, w, false);   // operation planes carry symbol + order class; no model evaluation needed
	cx.timing.host_walk_ms = ms_since(t_walk);
	HRY_MARK(t_all, "walked");
	if (in_place && in_place->arrays_ready) in_place->arrays_ready();

	const uint32_t vc = (uint32_t)w.order_v.size(), fc = (uint32_t)w.order_f.size();
	const ListDesc ldv = m.general ? ListDesc{} : make_list_desc(m.lists[1]), ldf = m.general ? ListDesc{} : make_list_desc(m.lists[0]);   // (general bindings: general_planes_encode)
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
	if (vc) HIP_OK(hipMemcpyAsync(cx.d_order_v.p, w.order_v.data(), (size_t)vc * 4, hipMemcpyHostToDevice, cx.stream));
	if (fc && (ldf.nplanes || m.general)) HIP_OK(hipMemcpyAsync(cx.d_order_f.p, w.order_f.data(), (size_t)fc * 4, hipMemcpyHostToDevice, cx.stream));   // only the face planes read it
	// the resident copy of the twins is current unless the walk repaired some (non-manifold edges, consumed neighbours)
	upload_repaired_twins(cx, in_place ? *in_place->whole : m, w);
	size_t goff[G_COUNT + 1] = { 0 };
	for (int g = 0; g < G_COUNT; ++g) {
		size_t n = w.grp_val[g].size();
		goff[g + 1] = goff[g] + n;
		if (n) HIP_OK(hipMemcpyAsync(cx.d_grp_val.as<uint32_t>() + goff[g], w.grp_val[g].data(), n * 4, hipMemcpyHostToDevice, cx.stream));
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
			for (int b = 0; b < kGroupBytes[g]; ++b, ++ci) planes.push_back(PlaneRef{ cx.d_connplanes.as<uint8_t>() + poff + (size_t)b * n, n, plane_init_kind(ci) });
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
	if (!m.general) {
		HIP_OK(hipMemsetAsync(cx.d_rank.p, 0xff, (size_t)dev_nv * 4, cx.stream));
		launch_rank(cx.stream, cx.d_order_v.as<uint32_t>(), vc, cv.org, cx.d_rank.as<uint32_t>());
		launch_predict_vtx(cx.stream, cv, cx.d_order_v.as<uint32_t>(), vc, cx.d_rank.as<uint32_t>(), cx.d_rec[1].as<uint8_t>(), ldv, cx.d_vplanes.as<uint8_t>());
		launch_face_planes(cx.stream, cv, cx.d_order_f.as<uint32_t>(), fc, cx.d_rec[0].as<uint8_t>(), ldf, cx.d_fplanes.as<uint8_t>());
	}
	{
		size_t poff = 0;
		for (int g = 0; g < G_COUNT; ++g) {
			uint32_t n = (uint32_t)w.grp_val[g].size();
			launch_split_bytes(cx.stream, cx.d_grp_val.as<uint32_t>() + goff[g], n, kGroupBytes[g], cx.d_connplanes.as<uint8_t>() + poff);
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
	cx.d_init.ensure(std::max<size_t>(inits.size() * 4, 16));
	if (!inits.empty()) HIP_OK(hipMemcpyAsync(cx.d_init.p, inits.data(), inits.size() * 4, hipMemcpyHostToDevice, cx.stream));
	cx.d_cjobs.ensure(std::max<size_t>((size_t)ns * sizeof(StreamJob), 16));
	if (ns) HIP_OK(hipMemcpyAsync(cx.d_cjobs.p, jobs.data(), (size_t)ns * sizeof(StreamJob), hipMemcpyHostToDevice, cx.stream));
	cx.ensure_magic(max_t0 + CH + 16);
	cx.d_acc.ensure((size_t)nw * 8);
	cx.d_v.ensure((size_t)nw * 8);
	cx.d_summary.ensure(((size_t)nw / 1024 + 2) * 4);
	cx.d_bytes.ensure((size_t)nw * 4);
	cx.d_csizes.ensure(std::max<size_t>((size_t)ns * 8, 16));   // bits | nbytes
	cx.d_coffs.ensure(((size_t)ns + 1) * 8);
	// ---- device: one wavefront per stream
	HIP_OK(hipMemsetAsync(cx.d_acc.p, 0, (size_t)nw * 8, cx.stream));
	uint32_t *d_bits = cx.d_csizes.as<uint32_t>(), *d_nbytes = d_bits + ns;
	HIP_OK(hipEventRecord(cx.ev[3], cx.stream));
	launch_chunk_encode(cx.stream, cx.d_cjobs.as<StreamJob>(), ns, cx.d_init.as<uint32_t>(), cx.d_magic.as<MagicEnt>(), cx.d_acc.as<uint64_t>(), d_bits);
	HIP_OK(hipEventRecord(cx.ev[4], cx.stream));
	launch_carry(cx.stream, cx.d_acc.as<uint64_t>(), nw, cx.d_v.as<uint64_t>(), cx.d_summary.as<uint32_t>(), cx.d_bytes.as<uint8_t>());
	launch_stream_pack(cx.stream, cx.d_cjobs.as<StreamJob>(), ns, d_bits, nullptr, d_nbytes, cx.d_coffs.as<uint64_t>(), nullptr, false);
	uint64_t total_bytes = 0;
	HIP_OK(hipMemcpyAsync(&total_bytes, cx.d_coffs.as<uint64_t>() + ns, 8, hipMemcpyDeviceToHost, cx.stream));
	// (while the device codes the streams: 17 ms of host work for the 151 741 components of the configs[3] mesh)
	// directory: chunk sizes, plane lengths, restart points of the connectivity replay, stream lengths
	std::vector<RestartCounters> rcounters;
	const std::vector<RestartPoint> restarts = select_restart_points(w.marks, w.named, rcounters);
	const uint32_t nrs = (uint32_t)restarts.size();
	std::vector<uint32_t> counter_dir;   // per restart point: n, then n x (vertex, counter)
	for (const RestartCounters &cs : rcounters) {
		counter_dir.push_back((uint32_t)cs.size());
		for (const auto &c : cs) { counter_dir.push_back(c.first); counter_dir.push_back(c.second); }
	}
	HRY_MARK(t_all, "  restart points selected");
	HIP_OK(hipStreamSynchronize(cx.stream));
	HRY_MARK(t_all, "streams coded");
	cx.d_cout.ensure(std::max<size_t>(total_bytes, 16));
	launch_stream_pack(cx.stream, cx.d_cjobs.as<StreamJob>(), ns, d_bits, cx.d_bytes.as<uint8_t>(), d_nbytes, cx.d_coffs.as<uint64_t>(), cx.d_cout.as<uint8_t>(), true);
	HIP_OK(hipEventRecord(cx.ev[5], cx.stream));

	// ---- container
	static_assert(sizeof(RestartPoint) == kRestartWords * 4, "restart points are written as they lie in memory");
	const size_t dir_prior = 12 + 4 * planes.size();
	const size_t dir_restart = dir_prior + prior_dir.size();
	const size_t dir_counters = dir_restart + 4 + sizeof(RestartPoint) * (size_t)nrs;
	const size_t dir_streams = dir_counters + 4 * counter_dir.size();
	size_t dir = dir_streams + 4 * (size_t)ns;
	size_t base = out.size();
	out.resize(base + dir + total_bytes);
	uint8_t *o = out.data() + base;
	uint32_t np = (uint32_t)planes.size();
	memcpy(o, &CH, 4); memcpy(o + 4, &CHC, 4); memcpy(o + 8, &np, 4);
	for (size_t i = 0; i < planes.size(); ++i) memcpy(o + 12 + 4 * i, &planes[i].n, 4);
	if (!prior_dir.empty()) memcpy(o + dir_prior, prior_dir.data(), prior_dir.size());
	memcpy(o + dir_restart, &nrs, 4);
	if (nrs) memcpy(o + dir_restart + 4, restarts.data(), sizeof(RestartPoint) * (size_t)nrs);
	if (!counter_dir.empty()) memcpy(o + dir_counters, counter_dir.data(), 4 * counter_dir.size());
	if (ns) HIP_OK(hipMemcpyAsync(o + dir_streams, d_nbytes, (size_t)ns * 4, hipMemcpyDeviceToHost, cx.stream));
	if (total_bytes) HIP_OK(hipMemcpyAsync(o + dir, cx.d_cout.p, total_bytes, hipMemcpyDeviceToHost, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	HRY_MARK(t_all, "container on the host");
	if (sharded) { const uint64_t seg_len = out.size() - seg_begin; memcpy(out.data() + seg_len_at, &seg_len, 8); }

	if (cx.keep_stages) {
		cx.stage_put_host("order_v", w.order_v.data(), (size_t)vc * 4);
		cx.stage_put_host("order_f", w.order_f.data(), (size_t)fc * 4);
		cx.stage_put("vplanes", cx.d_vplanes.p, (size_t)vc * ldv.nplanes);
		cx.stage_put("fplanes", cx.d_fplanes.p, (size_t)fc * ldf.nplanes);
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
