// Cut-border replay from entropy-decoded symbol planes (chunked profile).  See cbm_replay.hpp.
#include "cbm_replay.hpp"

#include <cstring>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace hry {
namespace {
struct Planes {
	const PlaneView *pl;   // container order: iop, elem[4], part[2], vertid[4], numtri[2], op[8]
	size_t cur[21] = { 0 };
	int fixed_numtri;
	uint32_t byte(int plane)
	{
		const PlaneView &v = pl[plane];
		if (cur[plane] >= v.size()) throw Error(HRY_E_FORMAT, "corrupt stream (connectivity plane exhausted)");
		return v[cur[plane]++];
	}
	uint32_t iop() { return byte(0); }
	uint32_t u32(int first) { uint32_t v = byte(first); v |= byte(first + 1) << 8; v |= byte(first + 2) << 16; v |= byte(first + 3) << 24; return v; }
	int elem() { uint32_t z = u32(1); return (int)((z >> 1) ^ ((z & 1) ? 0xffffffffu : 0u)); }   // transform.h:31-36
	int part() { uint32_t v = byte(5); v |= byte(6) << 8; return (int)v; }
	uint32_t vertid() { return u32(7); }
	int numtri() { if (fixed_numtri >= 0) return fixed_numtri; uint32_t v = byte(11); v |= byte(12) << 8; return (int)v; }
	uint32_t op(int order) { int k = order - 1; if (k > 7) k = 7; if (k < 0) k = 0; return byte(13 + k); }
};

}   // namespace

// planes: 21 connectivity planes in container order.  Fills m.face_off / org / twin and returns the decode order
// (one half-edge per vertex; vertex ids are assigned in this order, cbm/decoder.h:48-75,145).
//
// Restart points of the container directory cut the replay into spans that start at a component boundary with a known
// state: plane cursors, next vertex id / face / half-edge, and the order counters of the older vertices the span names
// (shared non-manifold vertices across the cut).  A span writes only the faces, half-edges and vertices it creates and counts
// older vertices on its private copy of their counters, so every span runs on its own host thread.
void cut_border_replay(Mesh &m, const PlaneView *conn_planes, const std::vector<RestartPoint> &restarts,
                       const std::vector<RestartCounters> &counters,
                       OrderVec &order_v, std::vector<uint32_t> &seg_start, std::vector<uint32_t> &seg_level, SpanDone *on_span)
{
	using replay_detail::NONE32;
	int ndeg = 0, onlydeg = 0;
	for (size_t d = 0; d < m.have_degree.size(); ++d) if (m.have_degree[d]) { ++ndeg; onlydeg = (int)d; }
	const int fixed_numtri = ndeg <= 1 ? onlydeg - 2 : -1;
	const unsigned n_threads = host_threads();
	if (restarts.empty() || n_threads < 2 || m.nf < parallel_min_faces() || counters.size() != restarts.size()) {
		if (fixed_numtri == 1 && !getenv("HRY_GENERIC_REPLAY")) {
			// triangles only: the lean loop (cbm_replay.hpp: replay_triangles), same results
			m.face_off.resize((size_t)m.nf + 1); m.face_off[0] = 0;
			m.org.resize(m.declared_ne);
			m.twin.resize(m.declared_ne);
			order_v.assign(m.nv, 0);
			BigVec<uint16_t> seen(m.nv, 0);
			ReplayCursor cur;
			std::vector<std::pair<uint32_t, uint32_t>> refs;
			seg_start.clear();
			PerfCounters pc;
			const bool count = getenv("HRY_PERF") != nullptr;
			if (count) pc.start();
			replay_triangles<false>(m, conn_planes, seen.data(), order_v.data(), cur, seg_start, refs, nullptr);
			if (count) { pc.stop(); pc.report("cut-border replay (triangles)", (double)cur.he - 2.0 * cur.face); }
			if (cur.face != m.nf) throw Error(HRY_E_FORMAT, "corrupt stream (face count)");
			if (cur.he != m.declared_ne) throw Error(HRY_E_FORMAT, "corrupt stream (polygon edge count)");
			order_v.resize(cur.next_id);
			replay_levels(seg_start, refs, seg_level);
			seg_start.push_back(cur.next_id);
			return;
		}
		Planes rd{ conn_planes, { 0 }, fixed_numtri };
		cut_border_replay_with(m, rd, order_v, seg_start, seg_level);
		return;
	}
	// spans: [start state, stop face); the directory must be strictly increasing in faces and consistent in every counter
	const size_t ns = restarts.size() + 1;
	struct Span { ReplayCursor cur; uint32_t stop_face; Planes rd; std::vector<uint32_t> first; std::vector<std::pair<uint32_t, uint32_t>> refs; bool eom = false; };
	std::vector<Span> spans(ns);
	for (size_t k = 0; k < ns; ++k) {
		Span &sp = spans[k];
		sp.rd = Planes{ conn_planes, { 0 }, fixed_numtri };
		if (k > 0) {
			const RestartPoint &r = restarts[k - 1];
			const uint32_t prev_face = k > 1 ? restarts[k - 2].first_face : 0u;
			if (r.first_face <= prev_face || r.first_face >= m.nf || r.first_vertex > m.nv || r.first_halfedge > m.declared_ne)
				throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart points)");
			for (const auto &c : counters[k - 1]) if (c.first >= r.first_vertex) throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart counters)");
			sp.cur.next_id = r.first_vertex; sp.cur.face = r.first_face; sp.cur.he = r.first_halfedge;
			static const int first_plane[G_COUNT] = { 0, 1, 5, 7, 11 };
			for (int g = 0; g < G_COUNT; ++g) for (int b = 0; b < kGroupBytes[g]; ++b) sp.rd.cur[first_plane[g] + b] = r.n_grp[g];
			for (int i = 0; i < 8; ++i) sp.rd.cur[13 + i] = r.n_op[i];
		}
		sp.stop_face = k + 1 < ns ? restarts[k].first_face : NONE32;
	}
	const bool trace = getenv("HRY_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry replay] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	if (trace) fprintf(stderr, "[hry replay] %zu spans\n", ns);
	m.face_off.resize((size_t)m.nf + 1); m.face_off[0] = 0;   // every entry is written before it is read (BigVec: no fill)
	m.org.resize(m.declared_ne);
	m.twin.resize(m.declared_ne);
	// (pooled arrays, zeroed by the helper threads: the two fills were 3 of the 4.3 ms in front of the spans on the 12.6 M-triangle share)
	order_v.clear(); order_v.resize(m.nv);
	BigVec<uint16_t> seen;
	seen.resize(m.nv);
	parallel_for(n_threads, [&](unsigned t) {
		const size_t b = (size_t)m.nv * t / n_threads, e = (size_t)m.nv * (t + 1) / n_threads;
		if (e > b) { memset(order_v.data() + b, 0, (e - b) * 4); memset(seen.data() + b, 0, (e - b) * 2); }
	});
	const RestartCounters none;
	mark("allocated");
	// a span must end exactly where the next one starts, in every counter
	auto check_end = [&](size_t k) {
		const Span &sp = spans[k];
		if (k + 1 == ns) {
			if (!sp.eom || sp.cur.face != m.nf || sp.cur.he != m.declared_ne) throw Error(HRY_E_FORMAT, "corrupt stream (face count)");
			return;
		}
		const RestartPoint &r = restarts[k];
		bool ok = !sp.eom && sp.cur.face == r.first_face && sp.cur.next_id == r.first_vertex && sp.cur.he == r.first_halfedge;
		for (int p = 0; p < 21 && ok; ++p) {
			size_t want = p == 0 ? r.n_grp[0] : p < 5 ? r.n_grp[1] : p < 7 ? r.n_grp[2] : p < 11 ? r.n_grp[3] : p < 13 ? r.n_grp[4] : r.n_op[p - 13];
			if (fixed_numtri >= 0 && (p == 11 || p == 12)) continue;   // numtri is not stored for a single polygon degree
			ok = sp.rd.cur[p] == want;
		}
		if (!ok) throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart point does not match the stream)");
	};
	std::atomic<size_t> next{ 0 };
	parallel_for((unsigned)std::min<size_t>(n_threads, ns), [&](unsigned) {
		for (;;) {
			size_t k = next.fetch_add(1, std::memory_order_relaxed);
			if (k >= ns) break;
			Span &sp = spans[k];
			const uint32_t f0 = sp.cur.face, h0 = sp.cur.he, v0 = sp.cur.next_id;
			sp.eom = replay_span(m, sp.rd, seen.data(), order_v.data(), sp.cur, sp.stop_face, k ? sp.cur.next_id : 0u, k ? counters[k - 1] : none, sp.first, sp.refs);
			check_end(k);
			if (on_span) on_span->span(f0, sp.cur.face, h0, sp.cur.he, v0, sp.cur.next_id);
		}
	});
	mark("spans replayed");
	// the component table and the dependency levels of the reconstruction
	seg_start.clear();
	std::vector<std::pair<uint32_t, uint32_t>> refs;
	for (size_t k = 0; k < ns; ++k) {
		const uint32_t base = (uint32_t)seg_start.size();
		seg_start.insert(seg_start.end(), spans[k].first.begin(), spans[k].first.end());
		for (const auto &r : spans[k].refs) refs.push_back({ base + r.first, r.second });
	}
	replay_levels(seg_start, refs, seg_level);
	const uint32_t n_ids = spans.back().cur.next_id;
	order_v.resize(n_ids);
	seg_start.push_back(n_ids);
}

}   // namespace hry
