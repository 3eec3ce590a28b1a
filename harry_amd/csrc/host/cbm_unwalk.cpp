// Cut-border replay from entropy-decoded symbol planes (chunked profile).  See cbm_replay.hpp.
#include "cbm_replay.hpp"

#include <cstring>
#include <sys/mman.h>
#include <sched.h>

#include <atomic>
#include <unordered_map>
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


// One span of the replay for meshes with polygons, written like replay_triangles (cbm_replay.hpp) for few instructions -- the
// generic replay_span measured 310 instructions per triangle on the configs[3] share (78 cycles at 4 per cycle; DESIGN.md
// section 4b).  The operation planes and the two triangle-count planes are read through bare cursors with an end pointer, the
// other planes (a few symbols per component) through the checked reader; nodes and the top part through bare pointers; the
// gate's neighbours are loaded only by the operations that use them; a polygon's half-edges are initialised when its first
// triangle arrives and its fan is continued by counting (cbm/decoder.h:133-197).  Same checks against a corrupt stream, same
// results as replay_span (the tests run both).
// Round 6: stop_mid -- the span ends inside a component, between two operations with the polygon in hand complete, as soon as
// stop_face faces exist (`end` receives the border); seed -- it starts inside one, from a border snapshot of the directory
// (cbm_replay.hpp BorderSeed; seen_shared then is the span's OWN array, for every vertex it touches).
bool replay_polygons(Mesh &m, Planes &rd, uint16_t *seen_shared, uint32_t *order_v, ReplayCursor &cur, uint32_t stop_face, uint32_t own_first,
                     const RestartCounters &old_counts, std::vector<uint32_t> &comp_first, std::vector<std::pair<uint32_t, uint32_t>> &refs,
                     bool stop_mid = false, const BorderSeed *seed = nullptr, BorderEnd *end = nullptr)
{
	using namespace replay_detail;
	struct Node { uint32_t v, a; int32_t prev, next; };
	struct Part { int32_t head, tail; uint32_t size, edge_begin; };
	const uint32_t nv = m.nv, nf = m.nf;
	const uint64_t ne_max = m.org.size();
	uint32_t *const org = m.org.data(), *const twin = m.twin.data(), *const foff = m.face_off.data();
	const uint8_t *opc[8], *ope[8];
	for (int k = 0; k < 8; ++k) {
		const PlaneView &v = rd.pl[13 + k];
		if (rd.cur[13 + k] > v.size()) throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart points)");
		opc[k] = v.data() + rd.cur[13 + k]; ope[k] = v.data() + v.size();
	}
	const int fixed = rd.fixed_numtri;
	const uint8_t *nt0 = nullptr, *nt1 = nullptr, *nt0e = nullptr, *nt1e = nullptr;
	if (fixed < 0) {
		if (rd.cur[11] > rd.pl[11].size() || rd.cur[12] > rd.pl[12].size()) throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart points)");
		nt0 = rd.pl[11].data() + rd.cur[11]; nt0e = rd.pl[11].data() + rd.pl[11].size();
		nt1 = rd.pl[12].data() + rd.cur[12]; nt1e = rd.pl[12].data() + rd.pl[12].size();
	}
	auto numtri = [&]() -> uint32_t {   // io.h:228-231
		if (fixed >= 0) return (uint32_t)fixed;
		if (nt0 == nt0e || nt1 == nt1e) throw Error(HRY_E_FORMAT, "corrupt stream (connectivity plane exhausted)");
		return (uint32_t)*nt0++ | (uint32_t)*nt1++ << 8;
	};
	// order counters: the span's own vertices in the shared array, older ones in a private map seeded by the restart point
	std::unordered_map<uint32_t, uint16_t> old_seen;
	for (const auto &c : old_counts) old_seen.emplace(c.first, (uint16_t)c.second);
	const bool own_array = seed != nullptr;   // (a span that starts inside a component counts in an array of its own; old_seen: which older vertices it may name)
	if (seed) {
		for (size_t j = 0; j < seed->snap->vtx.size(); ++j) { old_seen.emplace(seed->snap->vtx[j], 0); seen_shared[seed->snap->vtx[j]] = seed->snap->seen[j]; }
		for (const auto &c : seed->snap->counters) { old_seen.emplace(c.first, 0); seen_shared[c.first] = (uint16_t)c.second; }
	}
	auto seen = [&](uint32_t v) -> uint16_t& { return own_array || v >= own_first ? seen_shared[v] : old_seen.find(v)->second; };
	auto chk = [&](uint32_t v) {
		if (v >= nv) throw Error(HRY_E_FORMAT, "corrupt stream (vertex id)");
		if (v < own_first && !old_seen.count(v)) throw Error(HRY_E_FORMAT, "corrupt stream (restart point names an older vertex without its counter)");
		return v;
	};

	std::vector<Node> pool;
	pool.reserve(1 << 14);
	Node *P = pool.data();
	int32_t free_head = -1;
	std::vector<Part> parts;
	auto make = [&](uint32_t v, uint32_t a) -> int32_t {
		int32_t i = free_head;
		if (i >= 0) free_head = P[i].next;
		else { i = (int32_t)pool.size(); pool.push_back(Node()); P = pool.data(); }
		P[i].v = v; P[i].a = a; P[i].prev = -1; P[i].next = -1;
		return i;
	};
	auto drop = [&](int32_t i) { P[i].next = free_head; free_head = i; };
	auto append = [&](Part &p, int32_t i) { P[i].prev = p.tail; P[i].next = -1; if (p.tail >= 0) P[p.tail].next = i; else p.head = i; p.tail = i; ++p.size; };
	auto prepend = [&](Part &p, int32_t i) { P[i].next = p.head; P[i].prev = -1; if (p.head >= 0) P[p.head].prev = i; else p.tail = i; p.head = i; ++p.size; };
	auto unlink_tail = [&](Part &p) -> int32_t { int32_t i = p.tail; p.tail = P[i].prev; if (p.tail >= 0) P[p.tail].next = -1; else p.head = -1; --p.size; return i; };
	auto unlink_head = [&](Part &p) -> int32_t { int32_t i = p.head; p.head = P[i].next; if (p.head >= 0) P[p.head].prev = -1; else p.tail = -1; --p.size; return i; };
	auto discard_top = [&]() { for (int32_t i = parts.back().head; i >= 0;) { int32_t nx = P[i].next; drop(i); i = nx; } parts.pop_back(); };
	auto border = [&]() {   // cutborder.h:217-248
		Part &p = parts.back();
		if (p.size - (p.edge_begin ? 0u : 1u) == 1u) { discard_top(); return; }
		int32_t t = unlink_tail(p);
		if (!p.edge_begin) drop(unlink_head(p));
		prepend(p, t);
		p.edge_begin = 0;
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
	// a polygon of nt triangles: its half-edges, each its own twin until linked; origins beyond the first three are set as the
	// fan reaches them (zero until then, like the generic replay)
	auto new_face = [&](uint32_t nt) -> uint32_t {
		const uint32_t ne = nt + 2;
		if (ne < 3 || ne > 255) throw Error(HRY_E_FORMAT, "corrupt stream (polygon degree)");
		if (face >= nf) throw Error(HRY_E_FORMAT, "corrupt stream (too many faces)");
		if ((uint64_t)he + ne > ne_max) throw Error(HRY_E_FORMAT, "corrupt stream (too many polygon edges)");
		const uint32_t o = he;
		he += ne;
		foff[++face] = he;
		for (uint32_t i = 0; i < ne; ++i) twin[o + i] = o + i;
		for (uint32_t i = 3; i < ne; ++i) org[o + i] = 0;
		return o;
	};
	bool eom = false, resume = false;
	const uint32_t stop_at = stop_mid ? stop_face : NONE32;
	uint32_t ntri = 0, curtri = 0, base = 0;
	if (seed) {   // the border of the snapshot, placeholders for its half-edges
		size_t j = 0;
		for (const uint32_t pt : seed->snap->parts) {
			parts.push_back(Part{ -1, -1, 0, pt & 1u });
			for (uint32_t q = 0, nq = pt >> 1; q < nq; ++q, ++j) {
				append(parts.back(), make(seed->snap->vtx[j], seed->sym_base + (uint32_t)j));
				twin[seed->sym_base + j] = seed->sym_base + (uint32_t)j;
			}
		}
		resume = true;   // (between two polygons: curtri == ntri)
	}
	for (;;) {
		const uint32_t seg_first_id = next_id;
		uint32_t comp_idx = kContinues;
		auto depends_on = [&](uint32_t vid) { if (vid < seg_first_id) refs.push_back({ comp_idx, vid }); };
		auto fresh = [&]() { if (next_id >= nv) throw Error(HRY_E_FORMAT, "corrupt stream (vertex count)"); return next_id++; };
		if (resume) resume = false;
		else {
		if (!stop_mid && stop_face != NONE32 && face >= stop_face) break;
		const uint32_t iop = rd.iop();
		if (iop == I_EOM) { eom = true; break; }
		comp_first.push_back(seg_first_id);
		comp_idx = (uint32_t)comp_first.size() - 1;
		uint32_t a = 0, b = 0, c = 0;
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
		ntri = numtri(); curtri = 1;
		++seen(a); ++seen(b); ++seen(c);
		base = new_face(ntri);
		{
			const uint32_t e0 = base, e1 = base + 1, e2 = base + 2;
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
			if (face >= stop_at && curtri == ntri) {   // the span ends here, inside the component
				if (end) {
					end->parts.clear(); end->vtx.clear(); end->a.clear(); end->seen.clear();
					for (const Part &q : parts) {
						end->parts.push_back(q.size << 1 | (q.edge_begin ? 1u : 0u));
						for (int32_t i = q.head; i >= 0; i = P[i].next) { end->vtx.push_back(P[i].v); end->a.push_back(P[i].a); end->seen.push_back((uint8_t)std::min<uint32_t>(seen(P[i].v), 9u)); }
					}
				}
				goto stopped;
			}
			Part *T = &parts.back();
			if (T->size < 2) throw Error(HRY_E_FORMAT, "corrupt stream (border part)");
			const int32_t tn = T->tail, hn = T->head;
			const uint32_t v0 = P[tn].v, gate = P[tn].a, v1 = P[hn].v;
			uint32_t k = seen(v1);
			k = k == 0 ? 0u : k > 8u ? 7u : k - 1u;   // models.h:101-105
			if (opc[k] == ope[k]) throw Error(HRY_E_FORMAT, "corrupt stream (connectivity plane exhausted)");
			const uint32_t op = *opc[k]++;
			uint32_t v2, realop = op, lk_next = NONE32, lk_prev = NONE32;   // lk_*: cut-border edges the new triangle may close (decoder.h:182-197)
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
				v2 = rd.vertid();
				chk(v2);
				depends_on(v2);
				first = tn;
				second = make(v2, 0);
				T = &parts.back();
				P[second].prev = tn; P[tn].next = second; T->tail = second; ++T->size;
				break;
			}
			case O_SPLIT: {
				const int i = rd.elem();
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
				const int i = rd.elem();
				const int pp = rd.part();
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
			default: throw Error(HRY_E_FORMAT, "corrupt stream (op)");
			}
			// the new triangle: the first one of a polygon enters through the gate (v1 -> v0 -> v2), the others continue its fan
			uint32_t e1, e2;
			const bool first_tri = curtri == ntri;
			if (first_tri) {
				ntri = numtri();
				curtri = 0;
				base = new_face(ntri);
				e1 = base + 1; e2 = base + 2;
				org[base] = v1; org[e1] = v0; org[e2] = v2;
				twin[base] = gate; twin[gate] = base;             // decoder.h:179 merge(gate, e0)
			} else {
				e1 = base + curtri + 1; e2 = e1 + 1;
				org[e2] = v2;
			}
			switch (realop) {   // decoder.h:182-197
			case O_CONNFWD: P[first].a = e1; break;
			case O_CONNBWD: P[first].a = e2; break;
			case O_CLOSE: break;
			default: P[first].a = e1; P[second].a = e2; break;   // SPLIT, UNION, NEWVTX, NM
			}
			++seen(v0); ++seen(v1); ++seen(v2);
			if (op == O_NEWVTX) order_v[v2] = e2;
			++curtri;
			if (lk_next != NONE32 && curtri == ntri) { twin[lk_next] = e2; twin[e2] = lk_next; }   // CONNFWD / CLOSE on the polygon's last triangle: its last edge meets the next cut-border edge
			if (lk_prev != NONE32) { twin[lk_prev] = e1; twin[e1] = lk_prev; }                       // CONNBWD / CLOSE: its second edge meets the previous one
		}
	}
stopped:
	cur.next_id = next_id; cur.face = face; cur.he = he;
	for (int k = 0; k < 8; ++k) rd.cur[13 + k] = (size_t)(opc[k] - rd.pl[13 + k].data());
	if (fixed < 0) { rd.cur[11] = (size_t)(nt0 - rd.pl[11].data()); rd.cur[12] = (size_t)(nt1 - rd.pl[12].data()); }
	return eom;
}
}   // namespace

// ---- spans that start inside a component (round 6: border snapshots of the directory) --------------------------------------------
// An array of triangle counts for ONE such span: it counts at the vertices of the snapshot's border (older than the span, and
// still being counted at by the spans before it, which run at the same time) and at its own.  The whole index range is mapped,
// pages come into being where the span touches them -- its own vertices and the border's, which mostly follow each other.
SeenOfSpan::SeenOfSpan(uint32_t nv)
{
	bytes = ((size_t)nv * 2 + 4095) & ~(size_t)4095;
	if (bytes == 0) bytes = 4096;
	void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (q == MAP_FAILED) throw Error(HRY_E_INTERNAL, "replay: no address space for a span's counters");
	(void)madvise(q, bytes, MADV_NOHUGEPAGE);   // (a touched page is a page, not two megabytes of zeros)
	p = (uint16_t*)q;
}
SeenOfSpan::~SeenOfSpan() { if (p) munmap(p, bytes); }

// Span k started from snapshot k with placeholders for the half-edges of the border's elements, span k - 1 stopped there with the
// border it had -- the same border (checked: parts, vertices, counts), whose half-edges are what the placeholders stood for
// (possibly placeholders of span k - 1's own start, resolved a step earlier).  What span k linked a placeholder to sits in the twin
// array at the placeholder's index: the link goes to the real half-edge, both ways.  One span after the other, in stream order:
// step(k) when spans k - 1 and k have finished.  live: links into the part of the arrays a consumer may have copied are noted as
// patches (a patch of an entry that is copied later is harmless: the copy carries the final value too).
void SpanJoiner::step(size_t k, const SnapshotPoint *seed, const SnapshotPoint *seed_before, uint32_t sym_base, uint32_t sym_base_before, const BorderEnd &end_before, ReplayLive *live)
{
	const uint32_t ne = m.declared_ne;
	uint32_t *twin = m.twin.data();
	std::vector<uint32_t> real;
	if (seed) {
		if (k == 0) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
		const SnapshotPoint &S = *seed;
		const BorderEnd &E = end_before;
		if (E.parts != S.parts || E.vtx != S.vtx || E.seen != S.seen) throw Error(HRY_E_FORMAT, "corrupt chunked directory (a border snapshot does not match the stream)");
		real.resize(S.vtx.size());
		for (size_t j = 0; j < real.size(); ++j) {
			uint32_t a = E.a[j];
			if (a >= ne) {   // an element the span before never touched: what it was at ITS start
				const uint32_t i = a - sym_base_before;
				if (!seed_before || a < sym_base_before || i >= real_prev.size()) throw Error(HRY_E_INTERNAL, "replay: stray placeholder");
				a = real_prev[i];
			}
			real[j] = a;
		}
		for (size_t j = 0; j < real.size(); ++j) {
			const uint32_t sym = sym_base + (uint32_t)j, t = twin[sym];
			if (t == sym) continue;   // still on the border when the span ended (or closed onto itself: never -- a border edge meets an edge of a new face)
			if (t >= ne) throw Error(HRY_E_INTERNAL, "replay: a placeholder linked to a placeholder");
			twin[real[j]] = t; twin[t] = real[j];
			if (live) { live->pending.push_back(real[j]); live->pending.push_back(t); live->pending.push_back(t); live->pending.push_back(real[j]); }
		}
	}
	real_prev.swap(real);
}
void join_spans(Mesh &m, const std::vector<const SnapshotPoint*> &seeds, const std::vector<uint32_t> &sym_base, const std::vector<BorderEnd> &ends, ReplayLive *live)
{
	SpanJoiner J(m);
	static const BorderEnd none;
	for (size_t k = 0; k < seeds.size(); ++k) J.step(k, seeds[k], k ? seeds[k - 1] : nullptr, sym_base[k], k ? sym_base[k - 1] : 0u, k ? ends[k - 1] : none, live);
}

SnapshotSpans::SnapshotSpans(Mesh &mesh, const PlaneView *planes, const std::vector<SnapshotPoint> &points, uint32_t *ov) : m(mesh), conn(planes), snaps(points), order_v(ov)
{
	n_spans = snaps.size() + 1;
	spans.resize(n_spans); seeds.assign(n_spans, nullptr); sym_base.assign(n_spans, 0); ends.resize(n_spans);
	static const int first_plane[G_COUNT] = { 0, 1, 5, 7, 11 };
	auto cursors_of = [&](const RestartPoint &r, size_t *cur) {
		for (int g = 0; g < G_COUNT; ++g) for (int b = 0; b < kGroupBytes[g]; ++b) cur[first_plane[g] + b] = r.n_grp[g];
		for (int i = 0; i < 8; ++i) cur[13 + i] = r.n_op[i];
	};
	for (size_t k = 0; k < n_spans; ++k) {
		Span &sp = spans[k];
		for (int p = 0; p < 21; ++p) { sp.cur0[p] = 0; sp.cur1[p] = conn[p].size(); sp.cur_end[p] = 0; }
		if (k > 0) {
			const SnapshotPoint &S = snaps[k - 1];
			const RestartPoint &r = S.at;
			const uint32_t prev_face = k > 1 ? snaps[k - 2].at.first_face : 0u;
			if (r.first_face <= prev_face || r.first_face > m.nf || r.first_vertex > m.nv || r.first_halfedge > m.declared_ne || !S.counters.empty())
				throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshots)");
			sp.cur.next_id = r.first_vertex; sp.cur.face = r.first_face; sp.cur.he = r.first_halfedge;
			cursors_of(r, sp.cur0);
			for (int p = 0; p < 21; ++p) if (sp.cur0[p] > conn[p].size() && !((p == 11 || p == 12) && conn[p].empty())) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshots)");
			seeds[k] = &S;
			sym_base[k] = (uint32_t)(m.declared_ne + n_sym);
			n_sym += S.vtx.size();
			sp.seed = BorderSeed{ &S, sym_base[k] };
		}
		if (k + 1 < n_spans) { sp.stop_face = snaps[k].at.first_face; sp.stop_mid = true; cursors_of(snaps[k].at, sp.cur1); }
	}
	if ((uint64_t)m.declared_ne + n_sym >= 0xffffffffull) throw Error(HRY_E_UNSUPPORTED, "border snapshots: the placeholders do not fit behind the half-edges");
	m.twin.resize((size_t)m.declared_ne + n_sym);
}
void SnapshotSpans::start(unsigned n_threads)
{
	const void *node = callers_node_cpus();
	const bool trace = getenv("HRY_TRACE") != nullptr;
	const auto t_start = std::chrono::steady_clock::now();
	if (trace) fprintf(stderr, "[hry replay]   the caller sets the helpers off on cpu %d\n", sched_getcpu());
	auto work = [this, trace, t_start] {
		try {
			for (;;) {
				{ std::lock_guard<std::mutex> g(mu); if (failed) return; }
				const size_t k = next.fetch_add(1, std::memory_order_relaxed);
				if (k >= n_spans) return;
				Span &sp = spans[k];
				const auto t_a = std::chrono::steady_clock::now();
				SeenOfSpan own(m.nv);
				const ReplayCursor c0 = sp.cur;
				PerfCounters pc;
				const bool count = getenv("HRY_PERF") != nullptr;
				if (count) pc.start();
				sp.eom = replay_triangles<false>(m, conn, own.p, order_v, sp.cur, sp.first, sp.refs, nullptr, sp.cur0, sp.cur1, sp.stop_face, sp.stop_mid, &sp.seed, &ends[k], sp.cur_end);
				if (count) { pc.stop(); char what[64]; snprintf(what, sizeof what, "replay, stretch %zu on a helper thread", k); pc.report(what, (double)(sp.cur.face - c0.face)); }
				if (announce_to) {
					const bool mir = mirror_org != nullptr;
					if (mir) {   // (placeholders in the twins included: the join's patches follow)
						if (sp.cur.face > c0.face) memcpy(mirror_foff + c0.face + 1, m.face_off.data() + c0.face + 1, ((size_t)sp.cur.face - c0.face) * 4);
						if (sp.cur.he > c0.he) { memcpy(mirror_org + c0.he, m.org.data() + c0.he, ((size_t)sp.cur.he - c0.he) * 4); memcpy(mirror_twin + c0.he, m.twin.data() + c0.he, ((size_t)sp.cur.he - c0.he) * 4); }
						if (sp.cur.next_id > c0.next_id) memcpy(mirror_order + c0.next_id, order_v + c0.next_id, ((size_t)sp.cur.next_id - c0.next_id) * 4);
					}
					announce_to->range_done(ReplayLive::Range{ c0.face, sp.cur.face, c0.he, sp.cur.he, c0.next_id, sp.cur.next_id, mir });
				}
				{ std::lock_guard<std::mutex> g(mu); sp.done = true; }
				cv.notify_all();
				if (trace) fprintf(stderr, "[hry replay]   stretch %zu: %.3f .. %.3f ms after the helpers were set off, on cpu %d\n", k, std::chrono::duration<double, std::milli>(t_a - t_start).count(),
				                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), sched_getcpu());
			}
		} catch (...) { { std::lock_guard<std::mutex> g(mu); if (!failed) failed = std::current_exception(); } cv.notify_all(); }
	};
	const size_t nt = std::min<size_t>(std::max(1u, n_threads), n_spans - 1);
	helpers.reserve(nt);
	for (size_t t = 0; t < nt; ++t) helpers.emplace_back([work, node] { stay_on_node(node); work(); });
}
SnapshotSpans::~SnapshotSpans() { next.store(n_spans, std::memory_order_relaxed); for (auto &h : helpers) if (h.joinable()) h.join(); }
void SnapshotSpans::finish(ReplayCursor &cur, const size_t *cur_end0, BorderEnd &&end0, bool eom0, ReplayLive *live)
{
	spans[0].cur = cur; spans[0].eom = eom0;
	for (int p = 0; p < 21; ++p) spans[0].cur_end[p] = cur_end0[p];
	ends[0] = std::move(end0);
	SpanJoiner J(m);
	static const BorderEnd none;
	J.step(0, nullptr, nullptr, 0, 0, none, live);
	// one stretch after the other, as they finish: a stretch ends exactly where the next one starts, in every counter; joined with
	// the stretches before it, its vertices below the smallest one on its last border have every face (no vertex is named twice
	// here: none returns to a border) -- published at once, the consumer goes on while the later stretches are waited for
	for (size_t k = 0; k < n_spans; ++k) {
		if (k > 0) {
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return spans[k].done || failed; });
			if (failed) { lk.unlock(); for (auto &h : helpers) if (h.joinable()) h.join(); std::rethrow_exception(failed); }
		}
		const Span &sp = spans[k];
		if (k > 0) {
			const Span &sb = spans[k - 1];
			const RestartPoint &r = snaps[k - 1].at;
			bool ok = !sb.eom && sb.cur.face == r.first_face && sb.cur.next_id == r.first_vertex && sb.cur.he == r.first_halfedge && !ends[k - 1].parts.empty();
			for (int p = 0; p < 21 && ok; ++p) if (p != 11 && p != 12) ok = sb.cur_end[p] == sp.cur0[p];   // (triangles: no counts of triangles per polygon)
			if (!ok) throw Error(HRY_E_FORMAT, "corrupt chunked directory (a border snapshot does not match the stream)");
			J.step(k, seeds[k], seeds[k - 1], sym_base[k], sym_base[k - 1], ends[k - 1], live);
		}
		if (k + 1 == n_spans) {
			if (!sp.eom || sp.cur.face != m.nf || sp.cur.he != m.declared_ne) throw Error(HRY_E_FORMAT, "corrupt stream (face count)");
		} else if (live && k > 0) {
			uint32_t upto = sp.cur.next_id;
			for (const uint32_t v : ends[k].vtx) upto = std::min(upto, v);
			live->publish_at(sp.cur.face, sp.cur.he, upto);
		}
	}
	for (auto &h : helpers) if (h.joinable()) h.join();
	m.twin.resize(m.declared_ne);
	cur = spans.back().cur;
}

// planes: 21 connectivity planes in container order.  Fills m.face_off / org / twin and returns the decode order
// (one half-edge per vertex; vertex ids are assigned in this order, cbm/decoder.h:48-75,145).
//
// Restart points of the container directory cut the replay into spans that start at a component boundary with a known
// state: plane cursors, next vertex id / face / half-edge, and the order counters of the older vertices the span names
// (shared non-manifold vertices across the cut).  A span writes only the faces, half-edges and vertices it creates and counts
// older vertices on its private copy of their counters, so every span runs on its own host thread.
void cut_border_replay(Mesh &m, const PlaneView *conn_planes, const std::vector<RestartPoint> &restarts,
                       const std::vector<RestartCounters> &counters,
                       OrderVec &order_v, std::vector<uint32_t> &seg_start, std::vector<uint32_t> &seg_level, SpanDone *on_span,
                       const std::vector<SnapshotPoint> *snaps_in)
{
	using replay_detail::NONE32;
	static const std::vector<SnapshotPoint> no_snaps;
	const std::vector<SnapshotPoint> &snaps = snaps_in && !getenv("HRY_NO_SNAPSHOT_REPLAY") && !getenv("HRY_GENERIC_REPLAY") ? *snaps_in : no_snaps;
	int ndeg = 0, onlydeg = 0;
	for (size_t d = 0; d < m.have_degree.size(); ++d) if (m.have_degree[d]) { ++ndeg; onlydeg = (int)d; }
	const int fixed_numtri = ndeg <= 1 ? onlydeg - 2 : -1;
	unsigned n_threads = host_threads();
	if (const char *e = getenv("HRY_REPLAY_THREADS")) { const int v = atoi(e); if (v > 0) n_threads = (unsigned)v; }   // (development: the spans' threads beside the uploaders')
	if ((restarts.empty() && snaps.empty()) || n_threads < 2 || m.nf < parallel_min_faces() || counters.size() != restarts.size()) {
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
		if (!getenv("HRY_GENERIC_REPLAY")) {   // polygons: the lean loop as one span from the start of the stream
			m.face_off.resize((size_t)m.nf + 1); m.face_off[0] = 0;
			m.org.resize(m.declared_ne);
			m.twin.resize(m.declared_ne);
			order_v.assign(m.nv, 0);
			BigVec<uint16_t> seen(m.nv, 0);
			ReplayCursor cur;
			const RestartCounters none;
			std::vector<std::pair<uint32_t, uint32_t>> refs;
			seg_start.clear();
			PerfCounters pc;
			const bool count = getenv("HRY_PERF") != nullptr;
			if (count) pc.start();
			replay_polygons(m, rd, seen.data(), order_v.data(), cur, NONE32, 0, none, seg_start, refs);
			if (count) { pc.stop(); pc.report("cut-border replay (polygons)", (double)cur.he - 2.0 * cur.face); }
			if (cur.face != m.nf) throw Error(HRY_E_FORMAT, "corrupt stream (face count)");
			if (cur.he != m.declared_ne) throw Error(HRY_E_FORMAT, "corrupt stream (polygon edge count)");
			order_v.resize(cur.next_id);
			replay_levels(seg_start, refs, seg_level);
			seg_start.push_back(cur.next_id);
			return;
		}
		cut_border_replay_with(m, rd, order_v, seg_start, seg_level);
		return;
	}
	// spans: [start state, stop face); the points of both kinds in stream order -- faces ascending, and of a snapshot and a restart
	// point at the same face the snapshot comes first (it lies inside the component whose last operations make no face)
	struct Point { const RestartPoint *r; const RestartCounters *c; const SnapshotPoint *s; };
	std::vector<Point> pts;
	{
		size_t i = 0, j = 0;
		while (i < restarts.size() || j < snaps.size()) {
			const bool take_snap = j < snaps.size() && (i == restarts.size() || snaps[j].at.first_face <= restarts[i].first_face);
			if (take_snap) { pts.push_back(Point{ &snaps[j].at, &snaps[j].counters, &snaps[j] }); ++j; }
			else { pts.push_back(Point{ &restarts[i], &counters[i], nullptr }); ++i; }
		}
	}
	const size_t ns = pts.size() + 1;
	struct Span {
		ReplayCursor cur; uint32_t stop_face; bool stop_mid = false; Planes rd; size_t cur1[21]; std::vector<uint32_t> first; std::vector<std::pair<uint32_t, uint32_t>> refs; bool eom = false;
		BorderSeed seed; uint32_t own_first = 0, f0 = 0, h0 = 0, v0 = 0;
	};
	std::vector<Span> spans(ns);
	std::vector<const SnapshotPoint*> seeds(ns, nullptr);
	std::vector<uint32_t> sym_base(ns, 0);
	std::vector<BorderEnd> ends(ns);
	uint64_t n_sym = 0;
	static const int first_plane[G_COUNT] = { 0, 1, 5, 7, 11 };
	auto cursors_of = [&](const RestartPoint &r, size_t *cur) {
		for (int g = 0; g < G_COUNT; ++g) for (int b = 0; b < kGroupBytes[g]; ++b) cur[first_plane[g] + b] = r.n_grp[g];
		for (int i = 0; i < 8; ++i) cur[13 + i] = r.n_op[i];
	};
	for (size_t k = 0; k < ns; ++k) {
		Span &sp = spans[k];
		sp.rd = Planes{ conn_planes, { 0 }, fixed_numtri };
		if (k > 0) {
			const RestartPoint &r = *pts[k - 1].r;
			const uint32_t prev_face = k > 1 ? pts[k - 2].r->first_face : 0u;
			// (a snapshot may share its face count with the restart point behind it, never with another snapshot or the point before it)
			const bool ordered = pts[k - 1].s ? r.first_face > prev_face : (r.first_face > prev_face || (k > 1 && pts[k - 2].s && r.first_face == prev_face));
			if (!ordered || r.first_face == 0 || r.first_face > m.nf || (r.first_face == m.nf && !pts[k - 1].s) || r.first_vertex > m.nv || r.first_halfedge > m.declared_ne)
				throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart points)");
			for (const auto &c : *pts[k - 1].c) if (c.first >= r.first_vertex) throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart counters)");
			sp.cur.next_id = r.first_vertex; sp.cur.face = r.first_face; sp.cur.he = r.first_halfedge;
			sp.own_first = r.first_vertex;
			cursors_of(r, sp.rd.cur);
			if (pts[k - 1].s) {
				seeds[k] = pts[k - 1].s;
				sym_base[k] = (uint32_t)(m.declared_ne + n_sym);
				n_sym += pts[k - 1].s->vtx.size();
				sp.seed = BorderSeed{ pts[k - 1].s, sym_base[k] };
			}
		}
		sp.stop_face = k + 1 < ns ? pts[k].r->first_face : NONE32;
		sp.stop_mid = k + 1 < ns && pts[k].s != nullptr;
		if (k + 1 < ns) cursors_of(*pts[k].r, sp.cur1);
		else for (int p = 0; p < 21; ++p) sp.cur1[p] = conn_planes[p].size();
	}
	if ((uint64_t)m.declared_ne + n_sym >= 0xffffffffull) throw Error(HRY_E_UNSUPPORTED, "border snapshots: the placeholders do not fit behind the half-edges");
	const bool trace = getenv("HRY_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry replay] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	if (trace) fprintf(stderr, "[hry replay] %zu spans\n", ns);
	m.face_off.resize((size_t)m.nf + 1); m.face_off[0] = 0;   // every entry is written before it is read (BigVec: no fill)
	m.org.resize(m.declared_ne);
	m.twin.resize((size_t)m.declared_ne + n_sym);   // (behind the half-edges: the placeholders of the spans that start inside a component)
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
		const RestartPoint &r = *pts[k].r;
		bool ok = !sp.eom && sp.cur.face == r.first_face && sp.cur.next_id == r.first_vertex && sp.cur.he == r.first_halfedge;
		ok = ok && (ends[k].parts.empty() == (pts[k].s == nullptr));   // (it stopped inside a component exactly where a snapshot says so)
		for (int p = 0; p < 21 && ok; ++p) {
			size_t want = p == 0 ? r.n_grp[0] : p < 5 ? r.n_grp[1] : p < 7 ? r.n_grp[2] : p < 11 ? r.n_grp[3] : p < 13 ? r.n_grp[4] : r.n_op[p - 13];
			if (fixed_numtri >= 0 && (p == 11 || p == 12)) continue;   // numtri is not stored for a single polygon degree
			ok = sp.rd.cur[p] == want;
		}
		if (!ok) throw Error(HRY_E_FORMAT, "corrupt chunked directory (restart point does not match the stream)");
	};
	std::atomic<size_t> next{ 0 };
	const bool lean = !getenv("HRY_GENERIC_REPLAY");
	parallel_for((unsigned)std::min<size_t>(n_threads, ns), [&](unsigned) {
		for (;;) {
			size_t k = next.fetch_add(1, std::memory_order_relaxed);
			if (k >= ns) break;
			Span &sp = spans[k];
			const uint32_t f0 = sp.cur.face, h0 = sp.cur.he, v0 = sp.cur.next_id;
			sp.f0 = f0; sp.h0 = h0; sp.v0 = v0;
			const RestartCounters &old_counts = k && !seeds[k] ? *pts[k - 1].c : none;
			std::unique_ptr<SeenOfSpan> own;   // (a span that starts inside a component counts in an array of its own)
			if (seeds[k]) own.reset(new SeenOfSpan(m.nv));
			uint16_t *sn = own ? own->p : seen.data();
			// triangles: the lean loop where the span needs no counters of older vertices from a table (none named, or all of them in its own array)
			if (lean && fixed_numtri == 1 && old_counts.empty()) {
				size_t cur_end[21];
				sp.eom = replay_triangles<false>(m, conn_planes, sn, order_v.data(), sp.cur, sp.first, sp.refs, nullptr, sp.rd.cur, sp.cur1, sp.stop_face, sp.stop_mid,
				                                 seeds[k] ? &sp.seed : nullptr, &ends[k], cur_end);
				for (int p = 0; p < 21; ++p) sp.rd.cur[p] = cur_end[p];
			}
			else if (lean) sp.eom = replay_polygons(m, sp.rd, sn, order_v.data(), sp.cur, sp.stop_face, sp.own_first, old_counts, sp.first, sp.refs, sp.stop_mid, seeds[k] ? &sp.seed : nullptr, &ends[k]);
			else sp.eom = replay_span(m, sp.rd, seen.data(), order_v.data(), sp.cur, sp.stop_face, sp.own_first, old_counts, sp.first, sp.refs);
			check_end(k);
			if (trace) fprintf(stderr, "[hry replay] %8.2f ms    span %zu done (faces %u .. %u%s)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), k, f0, sp.cur.face, seeds[k] ? ", from a border snapshot" : "");
			// (a span next to a snapshot is not final before the spans are joined: nobody is told)
			if (on_span && !seeds[k] && !sp.stop_mid) on_span->span((uint32_t)k, (uint32_t)ns, f0, sp.cur.face, h0, sp.cur.he, v0, sp.cur.next_id, sp.first.data(), (uint32_t)sp.first.size(), false);
		}
	});
	mark("spans replayed");
	if (n_sym) {
		join_spans(m, seeds, sym_base, ends, nullptr);
		m.twin.resize(m.declared_ne);
		mark("spans joined");
		// ... and now the spans next to a snapshot are final too
		if (on_span) for (size_t k = 0; k < ns; ++k) {
			const Span &sp = spans[k];
			if (seeds[k] || sp.stop_mid) on_span->span((uint32_t)k, (uint32_t)ns, sp.f0, sp.cur.face, sp.h0, sp.cur.he, sp.v0, sp.cur.next_id, sp.first.data(), (uint32_t)sp.first.size(), sp.stop_mid);
		}
	}
	// the component table and the dependency levels of the reconstruction
	seg_start.clear();
	std::vector<std::pair<uint32_t, uint32_t>> refs;
	for (size_t k = 0; k < ns; ++k) {
		const uint32_t base = (uint32_t)seg_start.size();
		seg_start.insert(seg_start.end(), spans[k].first.begin(), spans[k].first.end());
		for (const auto &r : spans[k].refs) {
			if (r.first != kContinues) { refs.push_back({ base + r.first, r.second }); continue; }
			// named inside the component the span before ended in: a dependency only if the vertex is older than that component
			if (base == 0) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
			if (r.second < seg_start[base - 1]) refs.push_back({ base - 1, r.second });
		}
	}
	replay_levels(seg_start, refs, seg_level);
	const uint32_t n_ids = spans.back().cur.next_id;
	order_v.resize(n_ids);
	seg_start.push_back(n_ids);
}

}   // namespace hry
