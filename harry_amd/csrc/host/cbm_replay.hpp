// Host-side cut-border replay of the decoder, generic over the symbol source: rebuilds the connectivity either from
// already entropy-decoded symbol planes (chunked profile) or from a live arithmetic decoder (reference v0.1 stream).
// Behavioural contract: cbm/decoder.h:27-211, cbm/cutborder.h:49-333, formats/hry/io.h:168-231.
// Same flat node pool as the encoder-side walk (cbm_walk.cpp).
#pragma once
#include "host.hpp"
#include "perf_counters.hpp"

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <unordered_map>
#include <mutex>
#include <thread>

namespace hry {
namespace replay_detail {

constexpr uint32_t NONE32 = 0xffffffffu;
enum InitOp { I_INIT, I_TRI100, I_TRI010, I_TRI001, I_TRI110, I_TRI101, I_TRI011, I_TRI111, I_EOM };
enum Op { O_BORDER, O_CONNBWD, O_SPLIT, O_UNION, O_NM, O_NEWVTX, O_CONNFWD, O_CLOSE };

struct Ring {
	struct Node { uint32_t v, a; int32_t prev, next; };
	struct Part { int32_t head, tail; uint32_t size; bool edge_begin; };
	std::vector<Node> pool;
	int32_t free_head = -1;          // dropped nodes, chained through .next
	std::vector<Part> parts;
	uint16_t *on_border = nullptr;   // optional: how often every vertex currently occurs on the border (ReplayLive)
	Part &top() { return parts.back(); }
	int32_t make(uint32_t v, uint32_t a)
	{
		if (on_border) ++on_border[v];
		int32_t i = free_head;
		if (i >= 0) free_head = pool[i].next;
		else { i = (int32_t)pool.size(); pool.push_back(Node()); }
		pool[i] = Node{ v, a, -1, -1 };
		return i;
	}
	void drop(int32_t i) { if (on_border) --on_border[pool[i].v]; pool[i].next = free_head; free_head = i; }
	void append(Part &p, int32_t i)
	{
		pool[i].prev = p.tail; pool[i].next = -1;
		if (p.tail >= 0) pool[p.tail].next = i; else p.head = i;
		p.tail = i; ++p.size;
	}
	void prepend(Part &p, int32_t i)
	{
		pool[i].next = p.head; pool[i].prev = -1;
		if (p.head >= 0) pool[p.head].prev = i; else p.tail = i;
		p.head = i; ++p.size;
	}
	int32_t unlink_tail(Part &p)
	{
		int32_t i = p.tail;
		p.tail = pool[i].prev;
		if (p.tail >= 0) pool[p.tail].next = -1; else p.head = -1;
		--p.size;
		return i;
	}
	int32_t unlink_head(Part &p)
	{
		int32_t i = p.head;
		p.head = pool[i].next;
		if (p.head >= 0) pool[p.head].prev = -1; else p.tail = -1;
		--p.size;
		return i;
	}
	void discard_top()
	{
		for (int32_t i = top().head; i >= 0;) { int32_t nx = pool[i].next; drop(i); i = nx; }   // (drop overwrites .next: read it first)
		parts.pop_back();
	}
	Op border()   // cutborder.h:217-248
	{
		Part &p = top();
		if (p.size - (p.edge_begin ? 0 : 1) == 1) { discard_top(); return O_BORDER; }
		bool rename = !p.edge_begin;
		int32_t t = unlink_tail(p);
		if (!p.edge_begin) drop(unlink_head(p));
		prepend(p, t);
		p.edge_begin = false;
		return rename ? O_CONNFWD : O_BORDER;
	}
	// element addressed by a transmitted offset (cutborder.h:114-123): i > 0 from the front (1-based), i <= 0 from the back
	int32_t at(int i, int p, uint32_t &before)
	{
		if ((size_t)p >= parts.size()) throw Error(HRY_E_FORMAT, "corrupt stream (part index)");
		Part &pt = parts[parts.size() - 1 - (size_t)p];
		int32_t n;
		if (i > 0) {
			if ((uint32_t)i > pt.size) throw Error(HRY_E_FORMAT, "corrupt stream (element offset)");
			n = pt.head;
			for (int k = 1; k < i; ++k) n = pool[n].next;
			before = (uint32_t)(i - 1);
		} else {
			if ((uint32_t)(-i) >= pt.size) throw Error(HRY_E_FORMAT, "corrupt stream (element offset)");
			n = pt.tail;
			for (int k = 0; k < -i; ++k) n = pool[n].prev;
			before = pt.size - 1 - (uint32_t)(-i);
		}
		return n;
	}
};

}   // namespace replay_detail

// One span of the replay: components from the state (next_id, face, he) up to the component that would start at face
// stop_face (NONE32: up to the end-of-mesh symbol).  The connectivity arrays, order_v and seen are preallocated and shared
// between spans; a span writes only the faces / half-edges / vertices it creates (every index is checked against the sizes
// announced by the header, so a corrupt stream or directory cannot write outside them).
//   own_first     : first vertex id of the span.  Older vertices may be named only if the span brings their order counters
//                   along (old_counts: the restart point's snapshot); the span then counts on its private copy, so that spans
//                   never touch each other's counters
//   comp_first    : out, first vertex id of every component of this span
//   refs          : out, (component of this span, older vertex it names) pairs -- the caller derives the dependency levels of
//                   the reconstruction from them (replay_levels) once every span is known
// RD provides: iop(), vertid(), elem(), part(), numtri(), op(order)
struct ReplayCursor { uint32_t next_id = 0, face = 0, he = 0; };

// A span of the replay that starts INSIDE a component (round 6): the border comes from a snapshot of the container's directory
// (host.hpp SnapshotPoint).  The snapshot has no half-edges: element j (parts from the bottom of the stack, head -> tail) carries the
// placeholder sym_base + j -- a number above every half-edge of the mesh -- and the twin array has room up there: whatever the span
// links such an edge to is noted at the placeholder's own entry, and when the span BEFORE this one has finished, its last border
// says which half-edges the placeholders were (join_spans in cbm_unwalk.cpp).  A border edge only ever becomes the twin of an edge
// the span creates (decoder.h:179-197), so nothing else of the replay looks at these numbers.
struct BorderSeed { const SnapshotPoint *snap = nullptr; uint32_t sym_base = 0; };
// the border a span stopped with inside a component, in the snapshot's order (a: half-edges, or placeholders of the span's own seed
// for elements it never touched; seen: the triangle counts, clamped like the snapshot's)
struct BorderEnd { std::vector<uint32_t> parts, vtx, a; std::vector<uint8_t> seen; };
// the triangle counts of ONE span that starts inside a component (cbm_unwalk.cpp)
struct SeenOfSpan {
	uint16_t *p = nullptr; size_t bytes = 0;
	explicit SeenOfSpan(uint32_t nv);
	~SeenOfSpan();
	SeenOfSpan(const SeenOfSpan&) = delete;
	SeenOfSpan &operator=(const SeenOfSpan&) = delete;
};
struct ReplayLive;
struct SpanJoiner {   // (cbm_unwalk.cpp) the spans' placeholders resolved one span after the other, in stream order
	Mesh &m;
	std::vector<uint32_t> real_prev;   // what the placeholders of the span before were
	explicit SpanJoiner(Mesh &mesh) : m(mesh) {}
	void step(size_t k, const SnapshotPoint *seed, const SnapshotPoint *seed_before, uint32_t sym_base, uint32_t sym_base_before, const BorderEnd &end_before, ReplayLive *live);
};
void join_spans(Mesh &m, const std::vector<const SnapshotPoint*> &seeds, const std::vector<uint32_t> &sym_base, const std::vector<BorderEnd> &ends, ReplayLive *live);
// A triangle mesh whose replay publishes its progress (unchunk.cpp: the pipelined decode) and whose directory holds border
// snapshots: the caller replays the stretch up to the first snapshot itself, publishing as it goes; the stretches behind the
// snapshots run on helper threads meanwhile (cbm_unwalk.cpp).  No restart points, no explicitly named vertices (the pipelined
// decode's own conditions).
struct SnapshotSpans {
	Mesh &m;
	const PlaneView *conn;
	const std::vector<SnapshotPoint> &snaps;
	uint32_t *order_v;
	size_t n_spans = 0;
	uint64_t n_sym = 0;
	struct Span { ReplayCursor cur; size_t cur0[21], cur1[21], cur_end[21]; uint32_t stop_face = 0xffffffffu; bool stop_mid = false, eom = false, done = false; BorderSeed seed; std::vector<uint32_t> first; std::vector<std::pair<uint32_t, uint32_t>> refs; };
	std::vector<Span> spans;
	std::vector<const SnapshotPoint*> seeds;
	std::vector<uint32_t> sym_base;
	std::vector<BorderEnd> ends;
	std::vector<std::thread> helpers;
	std::atomic<size_t> next{ 1 };
	std::mutex mu;
	std::condition_variable cv;
	std::exception_ptr failed;
	// checks the snapshots against the header's sizes, sizes m.twin for the placeholders (m.org / m.twin / m.face_off are
	// allocated by the caller before) -- start() then sets the helpers off
	SnapshotSpans(Mesh &mesh, const PlaneView *planes, const std::vector<SnapshotPoint> &points, uint32_t *ov);
	~SnapshotSpans();
	void start(unsigned n_threads);
	// the caller's stretch has stopped at the first snapshot with border `end0` and cursors `cur` / plane cursors `cur_end0`: waits
	// for the helpers, checks every stretch's end against the next snapshot, joins them (twin links into published half-edges become
	// patches of `live`) and leaves the last stretch's cursor in `cur`
	void finish(ReplayCursor &cur, const size_t *cur_end0, BorderEnd &&end0, bool eom0, ReplayLive *live);
	ReplayLive *announce_to = nullptr;   // (set before start(): finished stretches are announced there)
	// (set before start(), or nullptr) pinned mirrors of face offsets (nf + 1 words), origins, twins (declared_ne words each) and the
	// decode order (nv words): a helper copies its finished stretch there itself -- the consumer then only starts the transfers
	uint32_t *mirror_foff = nullptr, *mirror_org = nullptr, *mirror_twin = nullptr, *mirror_order = nullptr;
};
constexpr uint32_t kContinues = 0xffffffffu;   // refs of a span that starts inside a component: "the component the span before me ended in"

// Progress of a running replay, published for a consumer thread that uploads the finished part of the connectivity and
// starts the attribute reconstruction of the vertices that can no longer change (unchunk.cpp, pipelined decode).
// A vertex is COMPLETE when it has left the cut-border for good: every face around it exists, and so does every face around
// its older neighbours once all smaller ids are complete too.  Without explicitly named vertices (NM operations, TRIxxx
// starts: the vertid planes are empty) a vertex never returns to the border, so "all ids < upto are complete" is simply the
// smallest id still on the border.  Twins of edges that lie below the last published half-edge count may already have been
// copied by the consumer: later links of such edges are recorded as patches (idempotent index/value pairs).
struct ReplayLive {
	struct Pub { uint64_t seq = 0, n_pub = 0; uint32_t faces = 0, he = 0, upto = 0; bool done = false, failed = false, joined = false; };   // seq: announcements of any kind; n_pub: publications of the replaying thread; joined: behind the replaying thread's own stretch (nothing of it is in that thread's cache: no lag)
	std::mutex mu;
	std::atomic<uint64_t> announced{ 0 };      // pub.seq, readable without the lock: the consumer polls it (a condition variable
	                                           // costs the replay a futex wake per publication, 3 us each, 7 % of its time)
	Pub pub;                                   // guarded by mu
	std::vector<uint32_t> patches;             // guarded by mu: (half-edge, twin) pairs since the consumer last took them
	// Round 6 (SnapshotSpans): a stretch of the replay that ran on a helper thread, finished -- its faces' offsets, its half-edges'
	// origins and twins (placeholders of border edges included: patched when the stretches are joined) and its vertices' decode
	// order are where they will stay, the consumer may copy them while the other stretches are still running
	struct Range { uint32_t f0, f1, h0, h1, v0, v1; bool mirrored; };   // mirrored: the stretch lies in the pinned mirrors too (SnapshotSpans)
	std::vector<Range> ranges;                 // guarded by mu: since the consumer last took them
	void range_done(const Range &r)
	{
		std::lock_guard<std::mutex> g(mu);
		ranges.push_back(r);
		++pub.seq;
		announced.store(pub.seq, std::memory_order_release);
	}
	// producer side
	BigVec<uint16_t> on_border;
	std::vector<uint32_t> pending;             // patches since the last publication
	uint32_t interval = 8192, face_pub = 0, he_pub = 0, min_open = 0;
	void link(uint32_t a, uint32_t b)
	{
		if (a < he_pub) { pending.push_back(a); pending.push_back(b); }
		if (b < he_pub) { pending.push_back(b); pending.push_back(a); }
	}
	double t_publish_ms = 0, t_lock_ms = 0; uint32_t n_publish = 0;
	// (round 6) a stretch of a helper thread has been joined with the ones before it: faces / half-edges up to its end are final
	// (patches pending), vertices below `upto` have every face
	void publish_at(uint32_t face, uint32_t he, uint32_t upto)
	{
		std::lock_guard<std::mutex> g(mu);
		++pub.seq; ++pub.n_pub; pub.faces = face; pub.he = he; pub.upto = upto; pub.done = false; pub.failed = false; pub.joined = true;
		patches.insert(patches.end(), pending.begin(), pending.end());
		pending.clear();
		face_pub = face; he_pub = he;
		announced.store(pub.seq, std::memory_order_release);
	}
	void publish(uint32_t face, uint32_t he, uint32_t next_id, bool done, bool failed = false)
	{
		auto t0 = std::chrono::steady_clock::now();
		++n_publish;
		struct Fin { ReplayLive &l; std::chrono::steady_clock::time_point t0; ~Fin() { l.t_publish_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } fin{ *this, t0 };
		{   // first vertex that is still on a border: four counters per step while none of them is set
			const uint16_t *ob = on_border.data();
			while (min_open + 4 <= next_id) { uint64_t w; memcpy(&w, ob + min_open, 8); if (w) break; min_open += 4; }
			while (min_open < next_id && ob[min_open] == 0) ++min_open;
		}
		{
			std::lock_guard<std::mutex> g(mu);
			++pub.seq; ++pub.n_pub; pub.faces = face; pub.he = he; pub.upto = done ? next_id : min_open; pub.done = done; pub.failed = failed; pub.joined = false;
			patches.insert(patches.end(), pending.begin(), pending.end());
			announced.store(pub.seq, std::memory_order_release);   // (inside the lock since round 6: helper threads announce their stretches too)
		}
		pending.clear();
		face_pub = face; he_pub = he;
	}
};

template <class RD>
bool replay_span(Mesh &m, RD &rd, uint16_t *seen_shared, uint32_t *order_v, ReplayCursor &cur, uint32_t stop_face, uint32_t own_first,
                 const RestartCounters &old_counts, std::vector<uint32_t> &comp_first, std::vector<std::pair<uint32_t, uint32_t>> &refs,
                 ReplayLive *live = nullptr)
{
	using namespace replay_detail;
	const uint32_t nv = m.nv, nf = m.nf, ne_max = (uint32_t)m.org.size();
	Ring cb;
	if (live) cb.on_border = live->on_border.data();
	uint32_t next_id = cur.next_id, face = cur.face, he = cur.he;
	auto new_face = [&](int ne) {
		if (ne < 3 || ne > 255) throw Error(HRY_E_FORMAT, "corrupt stream (polygon degree)");
		if (face >= nf) throw Error(HRY_E_FORMAT, "corrupt stream (too many faces)");
		if ((uint64_t)he + (uint32_t)ne > ne_max) throw Error(HRY_E_FORMAT, "corrupt stream (too many polygon edges)");
		uint32_t o = he;
		he += (uint32_t)ne;
		m.face_off[++face] = he;
		for (int i = 0; i < ne; ++i) m.twin[o + i] = o + i;
		for (int i = 3; i < ne; ++i) m.org[o + i] = 0;   // the first three origins are assigned by the caller right away
		return o;
	};

	auto link = [&](uint32_t a, uint32_t b) { m.twin[a] = b; m.twin[b] = a; if (live) live->link(a, b); };
	// order counters: the span's own vertices in the shared array, older ones in a private map seeded by the restart point
	std::unordered_map<uint32_t, uint16_t> old_seen;
	for (const auto &c : old_counts) old_seen.emplace(c.first, (uint16_t)c.second);
	struct Seen {
		uint16_t *shared; std::unordered_map<uint32_t, uint16_t> &old; uint32_t own_first;
		uint16_t &operator[](uint32_t v) { return v >= own_first ? shared[v] : old.find(v)->second; }
	} seen{ seen_shared, old_seen, own_first };
	auto chk = [&](uint32_t v) {
		if (v >= nv) throw Error(HRY_E_FORMAT, "corrupt stream (vertex id)");
		if (v < own_first && !old_seen.count(v)) throw Error(HRY_E_FORMAT, "corrupt stream (restart point names an older vertex without its counter)");
		return v;
	};
	auto fresh = [&]() { if (next_id >= nv) throw Error(HRY_E_FORMAT, "corrupt stream (vertex count)"); return next_id++; };
	bool eom = false;

	for (;;) {
		if (stop_face != NONE32 && face >= stop_face) break;
		uint32_t iop = rd.iop();
		if (iop == I_EOM) { eom = true; break; }
		uint32_t a = 0, b = 0, c = 0;
		// a component is an independent reconstruction chain unless it touches vertices coded before it started
		// (TRIxxx start, or an NM operation naming an older vertex)
		const uint32_t seg_first_id = next_id;
		comp_first.push_back(seg_first_id);
		const uint32_t comp_idx = (uint32_t)comp_first.size() - 1;
		auto depends_on = [&](uint32_t vid) { if (vid < seg_first_id) refs.push_back({ comp_idx, vid }); };
		switch (iop) {   // decoder.h:46-77
		case I_INIT: a = fresh(); b = fresh(); c = fresh(); break;
		case I_TRI100: a = rd.vertid(); b = fresh(); c = fresh(); break;
		case I_TRI010: c = fresh(); b = rd.vertid(); a = fresh(); break;
		case I_TRI001: a = fresh(); b = fresh(); c = rd.vertid(); break;
		case I_TRI110: a = rd.vertid(); b = rd.vertid(); c = fresh(); break;
		case I_TRI101: c = rd.vertid(); b = fresh(); a = rd.vertid(); break;
		case I_TRI011: a = fresh(); b = rd.vertid(); c = rd.vertid(); break;
		case I_TRI111: a = rd.vertid(); b = rd.vertid(); c = rd.vertid(); break;
		default: throw Error(HRY_E_FORMAT, "corrupt stream (init op)");
		}
		chk(a); chk(b); chk(c);
		depends_on(a); depends_on(b); depends_on(c);
		int ntri = rd.numtri(), curtri = 1;
		++seen[a]; ++seen[b]; ++seen[c];
		uint32_t base = new_face(ntri + 2);
		uint32_t e0 = base, e1 = base + 1, e2 = base + 2, fend = base + (uint32_t)ntri + 2;
		m.org[e0] = a; m.org[e1] = b; m.org[e2] = c;
		// vertex ids are handed out in decode order, so order_v is indexed by the id (decoder.h:86-110)
		switch (iop) {
		case I_INIT: order_v[a] = e0; order_v[b] = e1; order_v[c] = e2; break;
		case I_TRI100: order_v[b] = e1; order_v[c] = e2; break;
		case I_TRI010: order_v[c] = e2; order_v[a] = e0; break;
		case I_TRI001: order_v[a] = e0; order_v[b] = e1; break;
		case I_TRI110: order_v[c] = e2; break;
		case I_TRI101: order_v[b] = e1; break;
		case I_TRI011: order_v[a] = e0; break;
		default: break;
		}
		cb.parts.push_back(Ring::Part{ -1, -1, 0, true });
		cb.append(cb.top(), cb.make(a, e0));
		cb.append(cb.top(), cb.make(b, e1));
		cb.append(cb.top(), cb.make(c, e2));

		while (!cb.parts.empty()) {
			// between two operations the border and the faces agree; publish when the current polygon is complete as well
			if (live && curtri == ntri && face - live->face_pub >= live->interval) live->publish(face, he, next_id, false);
			Ring::Part &pt = cb.top();
			const uint32_t v0 = cb.pool[pt.tail].v, v1 = cb.pool[pt.head].v;
			const uint32_t gate = cb.pool[pt.tail].a;
			if (pt.size < 2) throw Error(HRY_E_FORMAT, "corrupt stream (border part)");
			const uint32_t gateprev = cb.pool[cb.pool[pt.tail].prev].a;
			const uint32_t gatenext = cb.pool[pt.head].a;
			const uint32_t op = rd.op(seen[v1]);
			const bool seq_first = curtri == ntri;
			uint32_t v2 = NONE32;
			int32_t first = -1, second = -1;
			uint32_t realop = op;
			switch (op) {   // decoder.h:133-166
			case O_CONNFWD: {
				Ring::Part &p = cb.top();
				if (!p.edge_begin) { realop = cb.border(); break; }   // renamed border (cutborder.h:177-179)
				if (p.size < 2) throw Error(HRY_E_FORMAT, "corrupt stream (connect forward)");
				v2 = cb.pool[cb.pool[p.head].next].v;
				if (p.size == 3) { cb.discard_top(); realop = O_CLOSE; }
				else { cb.drop(cb.unlink_head(p)); first = p.tail; realop = O_CONNFWD; }
				break;
			}
			case O_CONNBWD: {
				Ring::Part &p = cb.top();
				if (p.size < 2) throw Error(HRY_E_FORMAT, "corrupt stream (connect backward)");
				cb.drop(cb.unlink_tail(p));
				first = p.tail;
				v2 = cb.pool[p.tail].v;
				break;
			}
			case O_SPLIT: {
				int i = rd.elem();
				uint32_t before;
				int32_t hit = cb.at(i, 0, before);
				size_t oi = cb.parts.size() - 1;
				int32_t g = cb.unlink_tail(cb.parts[oi]);
				if (hit == g) throw Error(HRY_E_FORMAT, "corrupt stream (split at the gate)");
				Ring::Part np{ -1, -1, 0, true };
				if (before > 0) {
					Ring::Part &old = cb.parts[oi];
					int32_t last = cb.pool[hit].prev;
					np.head = old.head; np.tail = last; np.size = before;
					cb.pool[last].next = -1; cb.pool[hit].prev = -1;
					old.head = hit; old.size -= before;
				}
				cb.append(cb.parts[oi], g);
				second = cb.make(cb.pool[hit].v, cb.pool[hit].a);
				cb.append(np, second);
				np.edge_begin = cb.parts[oi].edge_begin;
				cb.parts[oi].edge_begin = true;
				cb.parts.push_back(np);
				first = g;
				v2 = cb.pool[hit].v;
				break;
			}
			case O_UNION: {
				int i = rd.elem();
				int p = rd.part();
				if (p <= 0) throw Error(HRY_E_FORMAT, "corrupt stream (union with the current part)");
				uint32_t before;
				int32_t hit = cb.at(i, p, before);
				size_t ci = cb.parts.size() - 1, oi = ci - (size_t)p;
				Ring::Part other = cb.parts[oi];
				Ring::Part &cur = cb.parts[ci];
				first = cur.tail;
				if (hit != other.head) {
					cb.pool[other.tail].next = other.head; cb.pool[other.head].prev = other.tail;
					int32_t last = cb.pool[hit].prev;
					cb.pool[last].next = -1; cb.pool[hit].prev = -1;
					other.head = hit; other.tail = last;
				}
				cb.pool[cur.tail].next = other.head; cb.pool[other.head].prev = cur.tail;
				cur.tail = other.tail; cur.size += other.size;
				second = cb.make(cb.pool[hit].v, cb.pool[hit].a);
				cb.append(cur, second);
				v2 = cb.pool[hit].v;
				cb.parts.erase(cb.parts.begin() + (long)oi);
				break;
			}
			case O_NEWVTX: case O_NM: {
				v2 = op == O_NEWVTX ? fresh() : rd.vertid();
				chk(v2);
				if (op == O_NM) depends_on(v2);
				Ring::Part &p = cb.top();
				first = p.tail;
				second = cb.make(v2, 0);
				cb.append(p, second);
				break;
			}
			case O_BORDER: cb.border(); break;
			default: throw Error(HRY_E_FORMAT, "corrupt stream (op)");
			}
			if (v2 == NONE32) continue;
			uint32_t f0;
			if (seq_first) {
				ntri = rd.numtri();
				curtri = 0;
				base = new_face(ntri + 2);
				fend = base + (uint32_t)ntri + 2;
				e0 = base; e1 = base + 1; e2 = base + 2;
				m.org[e0] = v1; m.org[e1] = v0; m.org[e2] = v2;
			} else {
				e1 = e1 + 1 == fend ? base : e1 + 1;
				e2 = e1 + 1 == fend ? base : e1 + 1;
				m.org[e2] = v2;
			}
			f0 = base;
			const bool seq_last = curtri + 1 == ntri;
			switch (realop) {   // decoder.h:182-197
			case O_CONNFWD: cb.pool[first].a = e1; break;
			case O_CONNBWD: cb.pool[first].a = e2; break;
			case O_SPLIT: case O_UNION: case O_NEWVTX: case O_NM: cb.pool[first].a = e1; cb.pool[second].a = e2; break;
			default: break;
			}
			++seen[v0]; ++seen[v1]; ++seen[v2];
			if (op == O_NEWVTX) order_v[v2] = f0 + (uint32_t)curtri + 2;
			++curtri;
			if (seq_first) link(gate, e0);
			if (op == O_CONNFWD) {
				if (seq_last && realop != O_BORDER) link(gatenext, e2);
				if (realop == O_CLOSE) link(gateprev, e1);
			} else if (op == O_CONNBWD) link(gateprev, e1);
		}
	}
	cur.next_id = next_id; cur.face = face; cur.he = he;
	return eom;
}

// ---------------------------------------------------------------------------------------------------------
// The same replay for the case that carries the headline workload: every polygon is a triangle, one span from the start of
// the stream, symbols in planes.  Hardware counters on the generic loop (EPYC 9575F, HRY_PERF=1) showed what bounds it:
// 274 instructions per triangle at 5.8 instructions per cycle, 0.004 branch misses and 0.03 last-level misses per
// triangle -- instruction count, nothing else.  So this loop is written for few instructions: a triangle per operation (no
// fan bookkeeping, the face offsets are 3 f and filled up front), the operation planes end in a sentinel that the switch
// rejects (no bounds check per symbol), gate neighbours are loaded only by the operations that use them, nodes and the top
// part through bare pointers, the order counters in the shared array only.  Same checks against a corrupt stream as
// replay_span, same results (the tests run both on the same inputs).
// ---------------------------------------------------------------------------------------------------------
// A span of it (round 6): cur0 / cur1 = where the 21 planes' cursors stand at its start and (at most) at its end (nullptr: the planes'
// own ends); stop_face / stop_mid: it ends in front of the component that would start at face stop_face, or -- stop_mid -- inside a
// component, between two operations, as soon as stop_face faces exist (`end` receives the border); seed: it starts inside a
// component (`seen` then is the span's own array: the counts at the border's vertices are entered here).
template <bool LIVE>
bool replay_triangles(Mesh &m, const PlaneView *conn, uint16_t *seen, uint32_t *order_v, ReplayCursor &cur,
                      std::vector<uint32_t> &comp_first, std::vector<std::pair<uint32_t, uint32_t>> &refs, ReplayLive *live,
                      const size_t *cur0 = nullptr, const size_t *cur1 = nullptr, uint32_t stop_face = replay_detail::NONE32, bool stop_mid = false,
                      const BorderSeed *seed = nullptr, BorderEnd *end = nullptr, size_t *cur_out = nullptr)
{
	using namespace replay_detail;
	struct Node { uint32_t v, a; int32_t prev, next; };
	struct Part { int32_t head, tail; uint32_t size, edge_begin; };
	const uint32_t nv = m.nv, nf = m.nf;
	const uint64_t ne_max = m.org.size();
	uint32_t *const org = m.org.data(), *const twin = m.twin.data();
	{   // a triangle mesh: face f owns the half-edges 3 f .. 3 f + 2 (the counts are checked against the header at the end)
		uint32_t *fo = m.face_off.data();
		const size_t last = std::min<uint64_t>(stop_face != NONE32 ? (uint64_t)stop_face : (uint64_t)nf, std::min<uint64_t>((uint64_t)nf, ne_max / 3));
		for (size_t i = (size_t)cur.face + 1; i <= last; ++i) fo[i] = (uint32_t)(3 * i);   // (entry cur.face: the caller's 0, or the span's before this one)
	}
	// operation planes with a sentinel behind the last symbol (of the span)
	std::vector<uint8_t> opl[8];
	const uint8_t *opc[8];
	for (int k = 0; k < 8; ++k) {
		const size_t b = cur0 ? cur0[13 + k] : 0, e = cur1 ? cur1[13 + k] : conn[13 + k].size();
		if (b > e || e > conn[13 + k].size()) throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart points)");
		opl[k].reserve(e - b + 1);
		opl[k].assign(conn[13 + k].begin() + b, conn[13 + k].begin() + e);
		opl[k].push_back(0xff);
		opc[k] = opl[k].data();
	}
	// the other connectivity planes are read a few times per mesh: checked cursors
	size_t rc[13] = { 0 };
	if (cur0) for (int k = 0; k < 13; ++k) rc[k] = cur0[k];
	auto rbyte = [&](int p) -> uint32_t { if (rc[p] >= conn[p].size()) throw Error(HRY_E_FORMAT, "corrupt stream (connectivity plane exhausted)"); return conn[p][rc[p]++]; };
	auto ru32 = [&](int first) -> uint32_t { uint32_t v = rbyte(first); v |= rbyte(first + 1) << 8; v |= rbyte(first + 2) << 16; v |= rbyte(first + 3) << 24; return v; };
	auto r_elem = [&]() -> int { uint32_t z = ru32(1); return (int)((z >> 1) ^ ((z & 1) ? 0xffffffffu : 0u)); };
	auto r_part = [&]() -> int { uint32_t v = rbyte(5); v |= rbyte(6) << 8; return (int)v; };
	auto r_vertid = [&]() -> uint32_t { return ru32(7); };

	std::vector<Node> pool;
	pool.reserve(1 << 14);
	Node *P = pool.data();
	int32_t free_head = -1;
	std::vector<Part> parts;
	uint16_t *const onb = LIVE ? live->on_border.data() : nullptr;
	auto make = [&](uint32_t v, uint32_t a) -> int32_t {
		if (LIVE) ++onb[v];
		int32_t i = free_head;
		if (i >= 0) free_head = P[i].next;
		else { i = (int32_t)pool.size(); pool.push_back(Node()); P = pool.data(); }
		P[i].v = v; P[i].a = a; P[i].prev = -1; P[i].next = -1;
		return i;
	};
	auto drop = [&](int32_t i) { if (LIVE) --onb[P[i].v]; P[i].next = free_head; free_head = i; };
	auto append = [&](Part &p, int32_t i) { P[i].prev = p.tail; P[i].next = -1; if (p.tail >= 0) P[p.tail].next = i; else p.head = i; p.tail = i; ++p.size; };
	auto prepend = [&](Part &p, int32_t i) { P[i].next = p.head; P[i].prev = -1; if (p.head >= 0) P[p.head].prev = i; else p.tail = i; p.head = i; ++p.size; };
	auto unlink_tail = [&](Part &p) -> int32_t { int32_t i = p.tail; p.tail = P[i].prev; if (p.tail >= 0) P[p.tail].next = -1; else p.head = -1; --p.size; return i; };
	auto unlink_head = [&](Part &p) -> int32_t { int32_t i = p.head; p.head = P[i].next; if (p.head >= 0) P[p.head].prev = -1; else p.tail = -1; --p.size; return i; };
	auto discard_top = [&]() { for (int32_t i = parts.back().head; i >= 0;) { int32_t nx = P[i].next; drop(i); i = nx; } parts.pop_back(); };
	auto border = [&]() -> uint32_t {   // cutborder.h:217-248
		Part &p = parts.back();
		if (p.size - (p.edge_begin ? 0u : 1u) == 1u) { discard_top(); return O_BORDER; }
		const bool rename = !p.edge_begin;
		int32_t t = unlink_tail(p);
		if (!p.edge_begin) drop(unlink_head(p));
		prepend(p, t);
		p.edge_begin = 0;
		return rename ? O_CONNFWD : O_BORDER;
	};
	auto at = [&](int i, int pi, uint32_t &before) -> int32_t {   // cutborder.h:114-123
		if ((size_t)pi >= parts.size()) throw Error(HRY_E_FORMAT, "corrupt stream (part index)");
		Part &pt = parts[parts.size() - 1 - (size_t)pi];
		int32_t n;
		if (i > 0) {
			if ((uint32_t)i > pt.size) throw Error(HRY_E_FORMAT, "corrupt stream (element offset)");
			n = pt.head;
			for (int k = 1; k < i; ++k) n = P[n].next;
			before = (uint32_t)(i - 1);
		} else {
			if ((uint32_t)(-i) >= pt.size) throw Error(HRY_E_FORMAT, "corrupt stream (element offset)");
			n = pt.tail;
			for (int k = 0; k < -i; ++k) n = P[n].prev;
			before = pt.size - 1 - (uint32_t)(-i);
		}
		return n;
	};

	uint32_t next_id = cur.next_id, face = cur.face, he = cur.he;
	uint32_t he_pub = LIVE ? live->he_pub : 0u, face_pub = LIVE ? live->face_pub : 0u;
	const uint32_t interval = LIVE ? live->interval : 0u;
	const uint32_t stop_at = stop_mid ? stop_face : NONE32;   // (NONE32: `face` never gets there)
	const uint32_t own_first = cur.next_id;                   // a span names older vertices only with their counters (chk below)
	bool eom = false, resume = false, older_known = false;
	std::vector<uint32_t> older;   // vertices older than the span that it may name (chk below)
	if (seed) {
		// the border of the snapshot: nodes in its order, placeholders for the half-edges, the counts at its vertices
		const SnapshotPoint &S = *seed->snap;
		size_t j = 0;
		for (const uint32_t pt : S.parts) {
			parts.push_back(Part{ -1, -1, 0, pt & 1u });
			for (uint32_t q = 0, nq = pt >> 1; q < nq; ++q, ++j) {
				const int32_t n = make(S.vtx[j], seed->sym_base + (uint32_t)j);
				append(parts.back(), n);
				seen[S.vtx[j]] = S.seen[j];
				twin[seed->sym_base + j] = seed->sym_base + (uint32_t)j;   // (nothing linked yet)
			}
		}
		for (const auto &c : S.counters) seen[c.first] = (uint16_t)c.second;
		resume = true;
	}
	for (;;) {
		uint32_t seg_first_id = next_id, comp_idx = kContinues;
		auto depends_on = [&](uint32_t vid) { if (vid < seg_first_id) refs.push_back({ comp_idx, vid }); };
		auto chk = [&](uint32_t v) {
			if (v >= nv) throw Error(HRY_E_FORMAT, "corrupt stream (vertex id)");
			if (v < own_first) {   // (rare: a vertex named explicitly) older than the span: only with its counter -- on the snapshot's border, or listed with it
				if (!older_known) {
					older_known = true;
					if (seed) { older.assign(seed->snap->vtx.begin(), seed->snap->vtx.end()); for (const auto &c : seed->snap->counters) older.push_back(c.first); }
					std::sort(older.begin(), older.end());
				}
				if (!std::binary_search(older.begin(), older.end(), v)) throw Error(HRY_E_FORMAT, "corrupt stream (restart point names an older vertex without its counter)");
			}
			return v;
		};
		auto fresh = [&]() { if (next_id >= nv) throw Error(HRY_E_FORMAT, "corrupt stream (vertex count)"); return next_id++; };
		if (resume) resume = false;   // (inside the component the span before this one ended in)
		else {
		if (!stop_mid && stop_face != NONE32 && face >= stop_face) break;
		const uint32_t iop = rbyte(0);
		if (iop == I_EOM) { eom = true; break; }
		comp_first.push_back(seg_first_id);
		comp_idx = (uint32_t)comp_first.size() - 1;
		uint32_t a = 0, b = 0, c = 0;
		switch (iop) {   // decoder.h:46-77
		case I_INIT: a = fresh(); b = fresh(); c = fresh(); break;
		case I_TRI100: a = r_vertid(); b = fresh(); c = fresh(); break;
		case I_TRI010: c = fresh(); b = r_vertid(); a = fresh(); break;
		case I_TRI001: a = fresh(); b = fresh(); c = r_vertid(); break;
		case I_TRI110: a = r_vertid(); b = r_vertid(); c = fresh(); break;
		case I_TRI101: c = r_vertid(); b = fresh(); a = r_vertid(); break;
		case I_TRI011: a = fresh(); b = r_vertid(); c = r_vertid(); break;
		case I_TRI111: a = r_vertid(); b = r_vertid(); c = r_vertid(); break;
		default: throw Error(HRY_E_FORMAT, "corrupt stream (init op)");
		}
		chk(a); chk(b); chk(c);
		depends_on(a); depends_on(b); depends_on(c);
		++seen[a]; ++seen[b]; ++seen[c];
		if (face >= nf) throw Error(HRY_E_FORMAT, "corrupt stream (too many faces)");
		if ((uint64_t)he + 3 > ne_max) throw Error(HRY_E_FORMAT, "corrupt stream (too many polygon edges)");
		{
			const uint32_t e0 = he, e1 = he + 1, e2 = he + 2;
			he += 3; ++face;
			twin[e0] = e0; twin[e1] = e1; twin[e2] = e2;
			org[e0] = a; org[e1] = b; org[e2] = c;
			switch (iop) {   // decoder.h:86-110: vertex ids are handed out in decode order, order_v is indexed by the id
			case I_INIT: order_v[a] = e0; order_v[b] = e1; order_v[c] = e2; break;
			case I_TRI100: order_v[b] = e1; order_v[c] = e2; break;
			case I_TRI010: order_v[c] = e2; order_v[a] = e0; break;
			case I_TRI001: order_v[a] = e0; order_v[b] = e1; break;
			case I_TRI110: order_v[c] = e2; break;
			case I_TRI101: order_v[b] = e1; break;
			case I_TRI011: order_v[a] = e0; break;
			default: break;
			}
			parts.push_back(Part{ -1, -1, 0, 1 });
			append(parts.back(), make(a, e0));
			append(parts.back(), make(b, e1));
			append(parts.back(), make(c, e2));
		}
		}

		while (!parts.empty()) {
			if (face >= stop_at) {   // the span ends here, inside the component: its border goes to whoever joins the spans
				if (end) {
					end->parts.clear(); end->vtx.clear(); end->a.clear(); end->seen.clear();
					for (const Part &q : parts) {
						end->parts.push_back(q.size << 1 | (q.edge_begin ? 1u : 0u));
						for (int32_t i = q.head; i >= 0; i = P[i].next) { end->vtx.push_back(P[i].v); end->a.push_back(P[i].a); end->seen.push_back((uint8_t)std::min<uint32_t>(seen[P[i].v], 9u)); }
					}
				}
				goto stopped;
			}
			if (LIVE && face - face_pub >= interval) { live->publish(face, he, next_id, false); he_pub = live->he_pub; face_pub = live->face_pub; }
			Part *T = &parts.back();
			if (T->size < 2) throw Error(HRY_E_FORMAT, "corrupt stream (border part)");
			const int32_t tn = T->tail, hn = T->head;
			const uint32_t v0 = P[tn].v, gate = P[tn].a, v1 = P[hn].v;
			uint32_t k = seen[v1];
			k = k == 0 ? 0u : k > 8u ? 7u : k - 1u;   // models.h:101-105
			const uint32_t op = *opc[k];
			opc[k] += op != 0xffu;                      // (the sentinel is never passed)
			uint32_t v2, realop = op, lk_next = NONE32, lk_prev = NONE32;   // lk_*: cut-border edges the new triangle closes (decoder.h:182-197)
			int32_t first = -1, second = -1;
			switch (op) {   // decoder.h:133-166
			case O_NEWVTX: {
				if (next_id >= nv) throw Error(HRY_E_FORMAT, "corrupt stream (vertex count)");
				v2 = next_id++;
				first = tn;
				second = make(v2, 0);
				T = &parts.back();
				P[second].prev = tn; P[tn].next = second; T->tail = second; ++T->size;   // append
				break;
			}
			case O_CONNFWD: {
				if (!T->edge_begin) { border(); continue; }   // renamed border (cutborder.h:177-179): no triangle
				v2 = P[P[hn].next].v;
				lk_next = P[hn].a;
				if (T->size == 3) { lk_prev = P[P[tn].prev].a; discard_top(); realop = O_CLOSE; }
				else { drop(unlink_head(*T)); first = T->tail; }
				break;
			}
			case O_CONNBWD: {
				lk_prev = P[P[tn].prev].a;
				drop(unlink_tail(*T));
				first = T->tail;
				v2 = P[first].v;
				break;
			}
			case O_NM: {
				v2 = r_vertid();
				chk(v2);
				depends_on(v2);
				first = tn;
				second = make(v2, 0);
				T = &parts.back();
				P[second].prev = tn; P[tn].next = second; T->tail = second; ++T->size;
				break;
			}
			case O_SPLIT: {
				const int i = r_elem();
				uint32_t before;
				const int32_t hit = at(i, 0, before);
				const size_t oi = parts.size() - 1;
				const int32_t g = unlink_tail(parts[oi]);
				if (hit == g) throw Error(HRY_E_FORMAT, "corrupt stream (split at the gate)");
				Part np{ -1, -1, 0, 1 };
				if (before > 0) {
					Part &old = parts[oi];
					const int32_t last = P[hit].prev;
					np.head = old.head; np.tail = last; np.size = before;
					P[last].next = -1; P[hit].prev = -1;
					old.head = hit; old.size -= before;
				}
				append(parts[oi], g);
				second = make(P[hit].v, P[hit].a);
				append(np, second);
				np.edge_begin = parts[oi].edge_begin;
				parts[oi].edge_begin = 1;
				parts.push_back(np);
				first = g;
				v2 = P[hit].v;
				break;
			}
			case O_UNION: {
				const int i = r_elem();
				const int pp = r_part();
				if (pp <= 0) throw Error(HRY_E_FORMAT, "corrupt stream (union with the current part)");
				uint32_t before;
				const int32_t hit = at(i, pp, before);
				const size_t ci = parts.size() - 1, oi = ci - (size_t)pp;
				Part other = parts[oi];
				Part &cp = parts[ci];
				first = cp.tail;
				if (hit != other.head) {
					P[other.tail].next = other.head; P[other.head].prev = other.tail;
					const int32_t last = P[hit].prev;
					P[last].next = -1; P[hit].prev = -1;
					other.head = hit; other.tail = last;
				}
				P[cp.tail].next = other.head; P[other.head].prev = cp.tail;
				cp.tail = other.tail; cp.size += other.size;
				second = make(P[hit].v, P[hit].a);
				append(parts[ci], second);
				v2 = P[hit].v;
				parts.erase(parts.begin() + (long)oi);
				break;
			}
			case O_BORDER: border(); continue;
			default: throw Error(HRY_E_FORMAT, op == 0xffu ? "corrupt stream (connectivity plane exhausted)" : "corrupt stream (op)");
			}
			// the new triangle (gate's twin, v1 -> v0 -> v2)
			if (face >= nf) throw Error(HRY_E_FORMAT, "corrupt stream (too many faces)");
			if ((uint64_t)he + 3 > ne_max) throw Error(HRY_E_FORMAT, "corrupt stream (too many polygon edges)");
			const uint32_t e0 = he, e1 = he + 1, e2 = he + 2;
			he += 3; ++face;
			org[e0] = v1; org[e1] = v0; org[e2] = v2;
			twin[e0] = gate; twin[gate] = e0;             // decoder.h:179 merge(gate, e0)
			twin[e1] = e1; twin[e2] = e2;
			if (LIVE && gate < he_pub) { live->pending.push_back(gate); live->pending.push_back(e0); }
			switch (realop) {   // decoder.h:182-197
			case O_CONNFWD: P[first].a = e1; break;
			case O_CONNBWD: P[first].a = e2; break;
			case O_CLOSE: break;
			default: P[first].a = e1; P[second].a = e2; break;   // SPLIT, UNION, NEWVTX, NM
			}
			++seen[v0]; ++seen[v1]; ++seen[v2];
			if (op == O_NEWVTX) order_v[v2] = e2;
			if (lk_next != NONE32) {   // CONNFWD / CLOSE: the triangle's last edge meets the next cut-border edge
				twin[lk_next] = e2; twin[e2] = lk_next;
				if (LIVE && lk_next < he_pub) { live->pending.push_back(lk_next); live->pending.push_back(e2); }
			}
			if (lk_prev != NONE32) {   // CONNBWD / CLOSE: its second edge meets the previous one
				twin[lk_prev] = e1; twin[e1] = lk_prev;
				if (LIVE && lk_prev < he_pub) { live->pending.push_back(lk_prev); live->pending.push_back(e1); }
			}
		}
	}
stopped:
	cur.next_id = next_id; cur.face = face; cur.he = he;
	if (cur_out) {
		for (int k = 0; k < 13; ++k) cur_out[k] = rc[k];
		for (int k = 0; k < 8; ++k) cur_out[13 + k] = (cur0 ? cur0[13 + k] : 0) + (size_t)(opc[k] - opl[k].data());
	}
	return eom;
}

// Dependency levels of the attribute reconstruction: level 0 = the component names no older vertex; else 1 + the highest
// level among the components that own a vertex it names.  seg_start: first vertex id of every component (ascending).
// A component that created no vertex shares its first id with its successor: upper_bound then lands on the last such entry,
// which is at least as late as the true owner -- its level is >= the owner's, so the bound stays valid.
inline void replay_levels(const std::vector<uint32_t> &seg_start, const std::vector<std::pair<uint32_t, uint32_t>> &refs, std::vector<uint32_t> &seg_level)
{
	seg_level.assign(seg_start.size(), 0);
	for (const auto &r : refs) {   // refs are in component order
		const uint32_t comp = r.first, vid = r.second;
		size_t owner = (size_t)(std::upper_bound(seg_start.begin(), seg_start.begin() + comp + 1, vid) - seg_start.begin());
		if (owner == 0) continue;
		--owner;
		if (owner >= comp) owner = comp ? comp - 1 : 0;
		if (comp) seg_level[comp] = std::max(seg_level[comp], seg_level[owner] + 1);
	}
}

// the whole connectivity as one span (reference v0.1 streams, and v0.2 containers without restart points)
template <class RD>
void cut_border_replay_with(Mesh &m, RD &rd, OrderVec &order_v, std::vector<uint32_t> &seg_start, std::vector<uint32_t> &seg_level)
{
	m.face_off.resize((size_t)m.nf + 1); m.face_off[0] = 0;   // every entry is written before it is read (BigVec: no fill)
	m.org.resize(m.declared_ne);
	m.twin.resize(m.declared_ne);
	order_v.assign(m.nv, 0);
	BigVec<uint16_t> seen(m.nv, 0);
	ReplayCursor cur;
	const RestartCounters none;
	std::vector<std::pair<uint32_t, uint32_t>> refs;
	seg_start.clear();
	if (getenv("HRY_PERF")) {
		PerfCounters pc;
		pc.start();
		replay_span(m, rd, seen.data(), order_v.data(), cur, replay_detail::NONE32, 0, none, seg_start, refs);
		pc.stop();
		pc.report("cut-border replay", (double)cur.he - 2.0 * cur.face);
	} else replay_span(m, rd, seen.data(), order_v.data(), cur, replay_detail::NONE32, 0, none, seg_start, refs);
	if (cur.face != m.nf) throw Error(HRY_E_FORMAT, "corrupt stream (face count)");
	if (cur.he != m.declared_ne) throw Error(HRY_E_FORMAT, "corrupt stream (polygon edge count)");
	order_v.resize(cur.next_id);
	replay_levels(seg_start, refs, seg_level);
	seg_start.push_back(cur.next_id);
}

}   // namespace hry
