// Host-side cut-border walk of the encoder.
//
// This is the inherently sequential part of the .hry path that stays on the CPU (SURVEY.md section 8 row a16,
// north_star): it fixes the traversal order, repairs the half-edge twins exactly as the reference does, and
// emits the connectivity symbols as byte planes for the device coder.  Behavioural contract:
//   cbm/encoder.h:54-217 (walk), cbm/cutborder.h:49-333 (border operations), formats/hry/writer.cc:28-58
//   (start-face / neighbour choice), formats/hry/io.h:31-88,141-165 (symbols), models.h:49-120 (op model).
// Data layout is MI355X-pipeline oriented rather than the reference's std::list/std::deque: one node pool
// with index links for all parts of the border, and symbol planes + global positions instead of an immediate
// call into the arithmetic coder.
#include "host.hpp"
#include "perf_counters.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_set>

namespace hry {
namespace {

constexpr uint32_t NONE32 = 0xffffffffu;
enum InitOp { I_INIT, I_TRI100, I_TRI010, I_TRI001, I_TRI110, I_TRI101, I_TRI011, I_TRI111, I_EOM };
enum Op { O_BORDER, O_CONNBWD, O_SPLIT, O_UNION, O_NM, O_NEWVTX, O_CONNFWD, O_CLOSE };

struct Border {
	struct Node { uint32_t v, a; int32_t prev, next; };
	struct Part { int32_t head, tail; uint32_t size; bool edge_begin; };
	std::vector<Node> pool;
	Node *P = nullptr;               // == pool.data(): the hot loop indexes through it (refreshed when the pool grows)
	int32_t free_head = -1;          // dropped nodes, chained through .next
	std::vector<Part> parts;
	BigVec<OnCount> &on;  // how many border elements reference a vertex (cutborder.h:69); shared per-vertex array

	explicit Border(BigVec<OnCount> &on_) : on(on_) { pool.reserve(1 << 16); P = pool.data(); }
	Part &top() { return parts.back(); }
	Node &N(int32_t i) { return P[i]; }

	int32_t make(uint32_t v, uint32_t a)
	{
		int32_t i = free_head;
		if (i >= 0) free_head = P[i].next;
		else { i = (int32_t)pool.size(); pool.push_back(Node()); P = pool.data(); }
		P[i] = Node{ v, a, -1, -1 };
		++on[v];
		return i;
	}
	void drop(int32_t i) { --on[P[i].v]; P[i].next = free_head; free_head = i; }
	void append(Part &p, int32_t i)
	{
		P[i].prev = p.tail; P[i].next = -1;
		if (p.tail >= 0) P[p.tail].next = i; else p.head = i;
		p.tail = i; ++p.size;
	}
	void prepend(Part &p, int32_t i)
	{
		P[i].next = p.head; P[i].prev = -1;
		if (p.head >= 0) P[p.head].prev = i; else p.tail = i;
		p.head = i; ++p.size;
	}
	int32_t unlink_tail(Part &p)
	{
		int32_t i = p.tail;
		p.tail = P[i].prev;
		if (p.tail >= 0) P[p.tail].next = -1; else p.head = -1;
		--p.size;
		return i;
	}
	int32_t unlink_head(Part &p)
	{
		int32_t i = p.head;
		p.head = P[i].next;
		if (p.head >= 0) P[p.head].prev = -1; else p.tail = -1;
		--p.size;
		return i;
	}
	void discard_top()
	{
		Part &p = top();
		for (int32_t i = p.head; i >= 0;) { int32_t nx = P[i].next; drop(i); i = nx; }
		parts.pop_back();
	}

	void start(uint32_t a, uint32_t ea, uint32_t b, uint32_t eb, uint32_t c, uint32_t ec)
	{
		parts.push_back(Part{ -1, -1, 0, true });
		append(top(), make(a, ea));
		append(top(), make(b, eb));
		append(top(), make(c, ec));
	}
	// cutborder.h:217-248
	Op border()
	{
		Part &p = top();
		uint32_t edges = p.size - (p.edge_begin ? 0 : 1);
		if (edges == 1) { discard_top(); return O_BORDER; }
		bool rename = !p.edge_begin;
		int32_t t = unlink_tail(p);
		if (!p.edge_begin) drop(unlink_head(p));
		prepend(p, t);
		p.edge_begin = false;
		return rename ? O_CONNFWD : O_BORDER;
	}
	// cutborder.h:124-155: two-ended search, front hit tested first; i > 0 counts from the front (1-based),
	// i <= 0 counts back from the tail; p = depth in the stack of parts
	int32_t locate(uint32_t v, int &i, int &p)
	{
		size_t pi = parts.size() - 1;
		int32_t fw = parts[pi].head, bw = parts[pi].tail;
		i = 0; p = 0;
		for (;;) {
			if (P[fw].v == v) { ++i; return fw; }
			if (P[bw].v == v) { i = -i; return bw; }
			if (bw == fw || P[bw].next == fw) {
				++p; --pi;
				fw = parts[pi].head; bw = parts[pi].tail;
				i = 0;
			} else { fw = P[fw].next; bw = P[bw].prev; ++i; }
		}
	}
	// cutborder.h:250-268. Returns (gate node, copy-of-hit node): their .a are filled by the caller.
	void split(int32_t hit, int i, int32_t &gate_node, int32_t &copy_node)
	{
		size_t oi = parts.size() - 1;
		uint32_t S = parts[oi].size;
		uint32_t before = i > 0 ? (uint32_t)(i - 1) : S - 1 - (uint32_t)(-i);
		int32_t g = unlink_tail(parts[oi]);
		Part np{ -1, -1, 0, true };
		if (before > 0) {   // move [head, hit) to the new part
			Part &old = parts[oi];
			int32_t last = P[hit].prev;
			np.head = old.head; np.tail = last; np.size = before;
			P[last].next = -1;
			P[hit].prev = -1;
			old.head = hit;
			old.size -= before;
		}
		append(parts[oi], g);
		copy_node = make(P[hit].v, P[hit].a);
		append(np, copy_node);
		np.edge_begin = parts[oi].edge_begin;
		parts[oi].edge_begin = true;
		parts.push_back(np);
		gate_node = g;
	}
	// cutborder.h:274-297
	void unite(int32_t hit, int p, int32_t &gate_node, int32_t &copy_node)
	{
		size_t ci = parts.size() - 1, oi = ci - (size_t)p;
		Part other = parts[oi];
		Part &cur = parts[ci];
		gate_node = cur.tail;
		if (hit != other.head) {   // rotate the other part so that it starts at the hit
			P[other.tail].next = other.head;
			P[other.head].prev = other.tail;
			int32_t last = P[hit].prev;
			P[last].next = -1;
			P[hit].prev = -1;
			other.head = hit; other.tail = last;
		}
		P[cur.tail].next = other.head;
		P[other.head].prev = cur.tail;
		cur.tail = other.tail;
		cur.size += other.size;
		copy_node = make(P[hit].v, P[hit].a);
		append(cur, copy_node);
		parts.erase(parts.begin() + (long)oi);
	}
};

// Start faces: face 0, then the first unvisited face in the iteration order of a std::unordered_set<uint32_t> that
// received 0..F-1 in order (writer.cc:28-46).  The order is a function of F and of libstdc++'s
// hashtable only, so it is derived instead of building a 100 M-node table: with the identity hash, load factor <= 1 and
// keys 0..k-1, every insertion lands in an empty bucket and is linked at the FRONT of the node list, and every rehash
// walks the list front to back re-linking each node at the front, i.e. reverses it (bits/hashtable.h: _M_insert_bucket_begin,
// _M_rehash_aux).  Rehash points come from libstdc++'s own policy object, so they follow the installed library.
struct StartFaces {
	struct Block { uint32_t first, last; };   // consecutive keys in list order: ascending if first <= last, else descending
	uint32_t nf;
	BigVec<Gone> &gone;
	std::vector<Block> blocks;
	size_t bi = 0;
	uint32_t pos = 0;
	bool have_order = false;
	StartFaces(uint32_t n, BigVec<Gone> &gone_) : nf(n), gone(gone_) {}
	void derive_order()
	{
		// between two rehash checks every insertion extends the descending block at the front of the list, and the policy only
		// looks at the count when it passes _M_next_resize: whole runs of keys are handled at once (a handful of steps for 10^8 faces)
		std::__detail::_Prime_rehash_policy pol;
		std::size_t nbkt = 1;
		std::vector<Block> list;   // front ... back
		uint32_t k = 0;
		while (k < nf) {
			std::pair<bool, std::size_t> rh = pol._M_need_rehash(nbkt, k, 1);
			if (rh.first) {
				nbkt = rh.second;
				std::reverse(list.begin(), list.end());
				for (Block &b : list) std::swap(b.first, b.last);
			}
			// keys k .. k_end - 1 arrive without another look at the policy (it compares the count with _M_next_resize first)
			const uint64_t quiet = (uint64_t)pol._M_next_resize;
			const uint32_t k_end = (uint32_t)std::min<uint64_t>(nf, std::max<uint64_t>((uint64_t)k + 1, quiet));
			if (!list.empty() && k > 0 && list.front().first == k - 1 && list.front().first >= list.front().last) list.front().first = k_end - 1;   // extend the descending front block
			else list.insert(list.begin(), Block{ k_end - 1, k });
			k = k_end;
		}
		blocks.swap(list);
		have_order = true;
		bi = 0; pos = 0;
	}
	// position of a face in the sequence (blocks sorted by their smallest key, binary search)
	struct Span { uint32_t lo, hi, first_pos; bool asc; };
	std::vector<Span> spans;
	void index_blocks()
	{
		spans.clear();
		uint32_t pos0 = 0;
		for (const Block &b : blocks) {
			bool asc = b.first <= b.last;
			uint32_t lo = asc ? b.first : b.last, hi = asc ? b.last : b.first;
			spans.push_back(Span{ lo, hi, pos0, asc });
			pos0 += hi - lo + 1;
		}
		std::sort(spans.begin(), spans.end(), [](const Span &x, const Span &y) { return x.lo < y.lo; });
	}
	uint32_t position(uint32_t f) const
	{
		size_t a = 0, b = spans.size();
		while (b - a > 1) { size_t mid = (a + b) / 2; if (spans[mid].lo <= f) a = mid; else b = mid; }
		const Span &s = spans[a];
		return s.first_pos + (s.asc ? f - s.lo : s.hi - f);
	}
	// the same for ascending faces: `a` is the caller's cursor into the spans (start it at 0; it only moves forward)
	uint32_t position_from(uint32_t f, size_t &a) const
	{
		while (a + 1 < spans.size() && spans[a + 1].lo <= f) ++a;
		const Span &s = spans[a];
		return s.first_pos + (s.asc ? f - s.lo : s.hi - f);
	}
	uint32_t at_cursor() const
	{
		const Block &b = blocks[bi];
		return b.first <= b.last ? b.first + pos : b.first - pos;
	}
	void advance()
	{
		const Block &b = blocks[bi];
		uint32_t len = (b.first <= b.last ? b.last - b.first : b.first - b.last) + 1;
		if (++pos == len) { ++bi; pos = 0; }
	}
	const std::vector<uint32_t> *seeds = nullptr;   // a shard brings its start faces along, in coding order (mesh.hpp ShardInfo)
	size_t seed_pos = 0;
	uint32_t next()
	{
		if (seeds) {
			while (seed_pos < seeds->size() && gone[(*seeds)[seed_pos]] != Gone::no) ++seed_pos;
			if (seed_pos == seeds->size()) throw Error(HRY_E_ARG, "shard: component without a seed face");
			return (*seeds)[seed_pos];
		}
		uint32_t f = 0;
		if (gone[0] != Gone::no) {
			if (!have_order) derive_order();
			while (gone[at_cursor()] != Gone::no) advance();
			f = at_cursor();
		}
		return f;
	}
};

struct Emitter {
	WalkResult &w;
	uint32_t n = 0;
	// order-conditioned operation model (models.h:49-120), evaluated here because it is connectivity-sized
	uint64_t plain[5] = { 1, 1, 1, 1, 1 }, c_all = 2, c_new[8], c_fwd[8];
	bool eval_model = true;   // the compat stream needs (l, h, t) of every operation; the chunked planes only symbol + class
	uint32_t n_op[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // operations per order class so far
	uint32_t min_ref = NONE32;                        // smallest vertex index named explicitly since the last mark
	uint32_t halfedges = 0;                           // half-edges of the faces coded so far
	// The per-triangle outputs go through bare cursors into arrays sized for the worst case up front (a push_back per symbol
	// re-reads and re-writes the vector's end pointer through memory: the byte stores of the loop may alias anything):
	// operations (symbol | class << 3, one byte), coded vertices, coded faces.  detach() trims the arrays to what was written.
	OpByte *op_cur = nullptr, *op_begin = nullptr;
	uint32_t *ov_cur = nullptr, *ov_begin = nullptr, *of_cur = nullptr, *of_begin = nullptr;
	// the triangle count of every coded polygon (only with more than one polygon degree): one per coded face, in the order of
	// order_f -- the same kind of cursor; ntp_*: its position in the symbol sequence (nullptr: not wanted, WalkResult::numtri_positions)
	uint32_t *nt_cur = nullptr, *nt_begin = nullptr, *ntp_cur = nullptr, *ntp_begin = nullptr;
	explicit Emitter(WalkResult &r) : w(r) { for (int i = 0; i < 8; ++i) c_new[i] = c_fwd[i] = 1; }
	// cap_*: upper bounds of what the walk can still emit on top of what the arrays hold
	void attach(size_t cap_ops, size_t cap_v, size_t cap_f)
	{
		const size_t no = w.op_sc.size(), nv = w.order_v.size(), nf = w.order_f.size();
		w.op_sc.resize(no + cap_ops); w.order_v.resize(nv + cap_v); w.order_f.resize(nf + cap_f);
		op_begin = w.op_sc.data(); op_cur = op_begin + no;
		ov_begin = w.order_v.data(); ov_cur = ov_begin + nv;
		of_begin = w.order_f.data(); of_cur = of_begin + nf;
		if (w.numtri_coded) {
			const size_t nn = w.grp_val[G_NUMTRI].size();
			w.grp_val[G_NUMTRI].resize(nn + cap_f);
			nt_begin = w.grp_val[G_NUMTRI].data(); nt_cur = nt_begin + nn;
			if (w.numtri_positions) { w.grp_pos[G_NUMTRI].resize(nn + cap_f); ntp_begin = w.grp_pos[G_NUMTRI].data(); ntp_cur = ntp_begin + nn; }
		}
	}
	void detach()
	{
		if (!op_begin) return;
		w.op_sc.resize((size_t)(op_cur - op_begin)); w.order_v.resize((size_t)(ov_cur - ov_begin)); w.order_f.resize((size_t)(of_cur - of_begin));
		op_begin = op_cur = nullptr;
		if (nt_begin) {
			w.grp_val[G_NUMTRI].resize((size_t)(nt_cur - nt_begin));
			if (ntp_begin) w.grp_pos[G_NUMTRI].resize((size_t)(ntp_cur - ntp_begin));
			nt_begin = nt_cur = ntp_begin = ntp_cur = nullptr;
		}
	}
	uint32_t faces_coded() const { return (uint32_t)(of_cur - of_begin); }
	// ---- border snapshots (host.hpp BorderSnapshot): restart points inside a component
	uint32_t snaps_in_component = 0;
	// the border as it stands between two operations: parts from the bottom of the stack, elements head -> tail; vertices in the
	// decoder's numbering (sent), triangle counts clamped at 9 (seen == nullptr: the mesh's own vertex numbers into s.orig instead,
	// for the thread that holds the counts)
	static void snapshot_border(const Border &cb, const uint32_t *sent, const uint16_t *seen, BorderSnapshot &s)
	{
		size_t n = 0;
		for (const Border::Part &q : cb.parts) n += q.size;
		s.parts.clear(); s.vtx.clear(); s.seen.clear(); s.orig.clear();
		s.parts.reserve(cb.parts.size()); s.vtx.reserve(n);
		if (seen) s.seen.reserve(n); else s.orig.reserve(n);
		for (const Border::Part &q : cb.parts) {
			s.parts.push_back(q.size << 1 | (q.edge_begin ? 1u : 0u));
			for (int32_t i = q.head; i >= 0; i = cb.P[i].next) {
				const uint32_t v = cb.P[i].v;
				s.vtx.push_back(sent[v]);
				if (seen) s.seen.push_back((uint8_t)std::min<uint32_t>(seen[v], 9u)); else s.orig.push_back(v);
			}
		}
	}
	// the cursors of a snapshot, RELATIVE to the mark of the component in hand (finish_snapshots makes them absolute when the marks
	// are final: a walk on several threads numbers them afterwards).  ops_in_component: operations per class since that mark;
	// nt_now: the triangle counts' cursor where the caller keeps it in a local
	void snapshot_cursors(BorderSnapshot &s, uint32_t next_id, uint32_t faces_in_component, uint32_t halfedges_in_component, const uint32_t *ops_in_component, const uint32_t *nt_now)
	{
		const ComponentMark &mk = w.marks.back();
		s.mark = (uint32_t)w.marks.size() - 1;
		for (int g = 0; g < G_COUNT; ++g) s.n_grp[g] = (uint32_t)w.grp_val[g].size() - mk.n_grp[g];
		s.n_grp[G_NUMTRI] = (uint32_t)(nt_now - nt_begin) - mk.n_grp[G_NUMTRI];
		for (int i = 0; i < 8; ++i) s.n_op[i] = ops_in_component[i];
		s.first_vertex = next_id - mk.first_vertex; s.first_face = faces_in_component; s.first_halfedge = halfedges_in_component;
		++snaps_in_component;
	}
	void mark_component(uint32_t next_id)
	{
		snaps_in_component = 0;
		if (!w.marks.empty()) w.marks.back().min_ref = min_ref;
		ComponentMark k;
		for (int g = 0; g < G_COUNT; ++g) k.n_grp[g] = (uint32_t)w.grp_val[g].size();
		k.n_grp[G_NUMTRI] = (uint32_t)(nt_cur - nt_begin);
		for (int i = 0; i < 8; ++i) k.n_op[i] = n_op[i];
		k.first_vertex = next_id; k.first_face = faces_coded(); k.first_halfedge = halfedges;
		k.min_ref = NONE32;
		w.marks.push_back(k);
		min_ref = NONE32;
	}
	void finish_marks() { if (!w.marks.empty()) w.marks.back().min_ref = min_ref; }
	void group(int g, uint32_t v) { w.grp_val[g].push_back(v); w.grp_pos[g].push_back(n); n += kGroupBytes[g]; }
	void iop(uint32_t s) { group(G_IOP, s); }
	void vert(uint32_t v, uint32_t count)
	{
		group(G_VERT, v);
		if (v < min_ref) min_ref = v;
		if (!w.marks.empty()) w.named.push_back(NamedVertex{ (uint32_t)w.marks.size() - 1, v, count, snaps_in_component });
	}
	void elem(int i) { uint32_t c = (uint32_t)i; group(G_ELEM, (c << 1) ^ ((c >> 31) ? 0xffffffffu : 0u)); }   // transform.h:25-30
	void part(int p) { group(G_PART, (uint32_t)(uint16_t)p); }
	void numtri(int nt)                                                                                          // io.h:162-165
	{
		if (nt == 0 || !w.numtri_coded) return;
		*nt_cur++ = (uint32_t)(uint16_t)nt;
		if (ntp_cur) *ntp_cur++ = n;
		n += kGroupBytes[G_NUMTRI];
	}
	void op(uint32_t s, int order)
	{
		int k = order - 1;   // models.h:101-105; order >= 1 because the gate's front vertex lies on a coded triangle
		if (k > 7) k = 7;
		if (k < 0) k = 0;
		++n_op[k];
		*op_cur++ = (OpByte)(s | ((uint32_t)k << 3));
		if (!eval_model) { ++n; return; }
		uint64_t nv = c_new[k] * c_all / (c_new[k] + c_fwd[k]);
		uint64_t f[7] = { plain[0], plain[1], plain[2], plain[3], plain[4], nv, c_all - nv };
		uint64_t l = 0;
		for (uint32_t x = 0; x < s; ++x) l += f[x];
		uint64_t t = plain[0] + plain[1] + plain[2] + plain[3] + plain[4] + c_all;
		w.op_l.push_back((uint32_t)l); w.op_h.push_back((uint32_t)(l + f[s])); w.op_t.push_back((uint32_t)t);
		w.op_pos.push_back(n++);
		if (s == O_NEWVTX) { ++c_all; ++c_new[k]; }
		else if (s == O_CONNFWD) { ++c_all; ++c_fwd[k]; }
		else ++plain[s];
	}
};

// One connected component, starting at face f (encoder.h:68-214).  DEG > 0: every polygon has DEG edges and the face of a
// half-edge is a division by a compile-time constant; DEG == 0: mixed degrees, table lookup.
template <int DEG>
static void walk_component(Mesh &m, WalkState &st, const uint32_t *eface_tab, uint32_t f, Border &cb, Emitter &em, uint32_t &next_id, uint32_t &consumed)
{
	static const bool kWalkPrefetch = [] { const char *e = getenv("HRY_WALK_PREFETCH"); return !e || atoi(e) != 0; }();
	WalkResult &w = em.w;
	const uint32_t *foff = m.face_off.data();
	const uint32_t *org = m.org.data();
	uint32_t *twin = m.twin.data();
	Gone *gone = st.gone.data();
	uint32_t *sent = st.sent.data();
	uint16_t *seen = st.seen.data();
	auto face_of = [&](uint32_t e) -> uint32_t { return DEG ? e / (uint32_t)(DEG ? DEG : 1) : eface_tab[e]; };
	auto nxt = [&](uint32_t e) -> uint32_t {
		if (DEG) { uint32_t k = e % (uint32_t)(DEG ? DEG : 1); return k + 1 == (uint32_t)DEG ? e - k : e + 1; }
		uint32_t fe = eface_tab[e];
		return e + 1 == foff[fe + 1] ? foff[fe] : e + 1;
	};
	auto link = [&](uint32_t a, uint32_t b) { twin[a] = b; twin[b] = a; w.twins_changed = true; w.twin_patches.push_back(a); w.twin_patches.push_back(b); };
	auto record_vertex = [&](uint32_t e) { *em.ov_cur++ = e; sent[org[e]] = next_id++; };
	auto take = [&](uint32_t face) { gone[face] = Gone::yes; ++consumed; em.halfedges += foff[face + 1] - foff[face]; };

	em.mark_component(next_id);
	const uint32_t consumed0 = consumed, halfedges0 = em.halfedges;
	const uint32_t snap_every = w.snapshot_faces;
	uint64_t next_snap = snap_every ? snap_every : ~0ull;   // (faces of this component at which the border is noted: host.hpp BorderSnapshot)
	take(f);
	uint32_t e0 = foff[f], e1 = nxt(e0), e2 = nxt(e1);
	uint32_t a = org[e0], b = org[e1], c = org[e2];
	int ntri = (int)(foff[f + 1] - foff[f]) - 2, curtri = 1;
	unsigned mask = (sent[a] != NONE32 ? 4u : 0u) | (sent[b] != NONE32 ? 2u : 0u) | (sent[c] != NONE32 ? 1u : 0u);
	switch (mask) {
	case 7: em.iop(I_TRI111); em.vert(sent[a], seen[a]); em.vert(sent[b], seen[b]); em.vert(sent[c], seen[c]); em.numtri(ntri); break;
	case 6: em.iop(I_TRI110); em.vert(sent[a], seen[a]); em.vert(sent[b], seen[b]); em.numtri(ntri); record_vertex(e2); break;
	case 3: em.iop(I_TRI011); em.vert(sent[b], seen[b]); em.vert(sent[c], seen[c]); em.numtri(ntri); record_vertex(e0); break;
	case 5: em.iop(I_TRI101); em.vert(sent[c], seen[c]); em.vert(sent[a], seen[a]); em.numtri(ntri); record_vertex(e1); break;
	case 4: em.iop(I_TRI100); em.vert(sent[a], seen[a]); em.numtri(ntri); record_vertex(e1); record_vertex(e2); break;
	case 2: em.iop(I_TRI010); em.vert(sent[b], seen[b]); em.numtri(ntri); record_vertex(e2); record_vertex(e0); break;
	case 1: em.iop(I_TRI001); em.vert(sent[c], seen[c]); em.numtri(ntri); record_vertex(e0); record_vertex(e1); break;
	default: em.iop(I_INIT); em.numtri(ntri); record_vertex(e0); record_vertex(e1); record_vertex(e2); break;
	}
	*em.of_cur++ = e0;
	++seen[a]; ++seen[b]; ++seen[c];
	cb.start(a, e0, b, e1, c, e2);

	// ---- grow until the border of this component is exhausted (encoder.h:133-214)
	while (!cb.parts.empty()) {
		if (consumed - consumed0 >= next_snap && curtri == ntri) {
			next_snap += snap_every;
			uint32_t ops[8];
			for (int i = 0; i < 8; ++i) ops[i] = em.n_op[i] - w.marks.back().n_op[i];
			w.snapshots.emplace_back();
			Emitter::snapshot_border(cb, sent, seen, w.snapshots.back());
			em.snapshot_cursors(w.snapshots.back(), next_id, consumed - consumed0, em.halfedges - halfedges0, ops, em.nt_cur);
		}
		Border::Part &pt = cb.top();
		const uint32_t v0 = cb.N(pt.tail).v, v1 = cb.N(pt.head).v;
		const uint32_t gate = cb.N(pt.tail).a;
		const uint32_t gateprev = cb.N(cb.N(pt.tail).prev).a;
		const uint32_t gatenext = cb.N(pt.head).a;
		const bool seq_first = curtri == ntri;
		const int order = seen[v1];
		if (seq_first) {
			uint32_t t = twin[gate];
			if (t == gate || gone[face_of(t)] != Gone::no) {   // writer.cc:48-58: mesh border or neighbour already consumed
				Op bop = cb.border();
				if (t != gate) { twin[gate] = gate; w.twins_changed = true; w.twin_patches.push_back(gate); }   // one-sided split (writer.cc:81-84)
				em.op(bop, order);
				continue;
			}
			take(face_of(t));
			e0 = t;
			f = face_of(e0);
			ntri = (int)(foff[f + 1] - foff[f]) - 2;
			curtri = 0;
			e1 = nxt(e0);
			// the faces behind this polygon's other edges are the next gates' neighbours: their lines (twins, origins, the
			// half-edge -> face table) are asked for now -- a component is walked once, every line of it is a miss the first time
			// (hardware counters on the configs[3] share: 0.44 last-level misses per triangle, IPC 1.6)
			if (kWalkPrefetch) {
				for (uint32_t h = foff[f], he = foff[f + 1]; h < he; ++h) {
					const uint32_t o = twin[h];
					__builtin_prefetch(twin + o); __builtin_prefetch(org + o);
					if (!DEG) __builtin_prefetch(eface_tab + o);
				}
			}
		} else e1 = nxt(e1);
		e2 = nxt(e1);
		const uint32_t v2 = org[e2];
		const bool seq_last = curtri + 1 == ntri;
		const int nt = seq_first ? ntri : 0;

		bool fresh = sent[v2] == NONE32;
		if (fresh || cb.on[v2] == 0) {
			// NEWVTX, or a vertex that was coded before but left the border (non-manifold): encoder.h:167-181
			Border::Part &p = cb.top();
			cb.N(p.tail).a = e1;
			cb.append(p, cb.make(v2, e2));
			if (fresh) { em.op(O_NEWVTX, order); em.numtri(nt); record_vertex(e2); }
			else { em.op(O_NM, order); em.vert(sent[v2], seen[v2]); em.numtri(nt); }
		} else {
			int i, p;
			int32_t hit = cb.locate(v2, i, p);
			if (p > 0) {
				int32_t g, cp;
				cb.unite(hit, p, g, cp);
				em.op(O_UNION, order); em.elem(i); em.part(p); em.numtri(nt);
				cb.N(g).a = e1; cb.N(cp).a = e2;
			} else {
				Border::Part &tp = cb.top();
				if (tp.edge_begin && cb.N(cb.N(tp.head).next).v == v2) {
					bool close = tp.size == 3;   // edge_begin && 3 elements: the part is exactly this triangle
					if (seq_last && twin[gatenext] != e2) link(gatenext, e2);
					if (close && twin[gateprev] != e1) link(gateprev, e1);
					if (close) cb.discard_top();
					else { cb.drop(cb.unlink_head(tp)); cb.N(tp.tail).a = e1; }
					em.op(O_CONNFWD, order); em.numtri(nt);
				} else if (cb.N(cb.N(tp.tail).prev).v == v2) {
					if (twin[gateprev] != e1) link(gateprev, e1);
					cb.drop(cb.unlink_tail(tp));
					cb.N(tp.tail).a = e2;
					em.op(O_CONNBWD, order); em.numtri(nt);
				} else {
					int32_t g, cp;
					cb.split(hit, i, g, cp);
					em.op(O_SPLIT, order); em.elem(i); em.numtri(nt);
					cb.N(g).a = e1; cb.N(cp).a = e2;
				}
			}
		}
		++seen[v0]; ++seen[v1]; ++seen[v2];
		if (seq_first) *em.of_cur++ = e0;
		++curtri;
	}
}

// The same component walk for the case that carries the headline workload: every polygon is a triangle and the operation model
// is not evaluated (chunked profile).  Hardware counters on the generic loop (EPYC 9575F, HRY_PERF=1): 228 instructions per
// triangle at 3.9 per cycle, 0.005 branch misses, 0.4 last-level misses -- mostly instruction count.  This loop keeps its
// cursors and counters in locals (the mark arrays and the operation stream are not character types, so their stores do not
// force reloads), derives the triangle's edges from one division, loads the gate's neighbours only where an operation needs
// them, tests connect-forward / -backward before searching the border, and writes the Emitter back once per component.
template <bool MODEL>   // MODEL: the operation model of the reference stream is evaluated per operation (compat profile), through the Emitter
static void walk_component_tri(Mesh &m, WalkState &st, uint32_t f, Border &cb, Emitter &em, uint32_t &next_id_io, uint32_t &consumed_io)
{
	WalkResult &w = em.w;
	const uint32_t *org = m.org.data();
	uint32_t *twin = m.twin.data();
	Gone *gone = st.gone.data();
	uint32_t *sent = st.sent.data();
	uint16_t *seen = st.seen.data();
	OnCount *on = st.on.data();
	uint32_t next_id = next_id_io, consumed = consumed_io;
	OpByte *opc = em.op_cur;
	uint32_t *ovc = em.ov_cur, *ofc = em.of_cur;
	uint32_t n_op[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, n_ops = 0;
	bool changed = false;
	std::vector<uint32_t> &tp = w.twin_patches;   // half-edges whose twin this walk changes (rare: non-manifold edges, consumed neighbours)
	// HRY_WALK_PREFETCH: 0 none, 1 the twins of the triangle's other edges, 2 (default) + the faces behind them, 3 + their marks
	// (1 M triangles: 7.8 -> 7.4 ms; 28 M, beyond the caches: 293 -> 237 ms)
	static const int pf_level = [] { const char *e = getenv("HRY_WALK_PREFETCH"); return e ? atoi(e) : 2; }();
	auto emit = [&](uint32_t s, uint32_t order) {
		if (MODEL) { em.op(s, (int)order); return; }
		uint32_t k = order == 0 ? 0u : order > 8u ? 7u : order - 1u;   // models.h:101-105
		++n_op[k]; ++n_ops;
		*opc++ = (OpByte)(s | (k << 3));
	};
	// rare symbols go through the Emitter (its symbol counter is brought up to date first)
	auto sync_n = [&] { em.n += n_ops; n_ops = 0; };

	em.mark_component(next_id);
	gone[f] = Gone::yes; ++consumed;
	{
		const uint32_t e0 = 3 * f, e1 = e0 + 1, e2 = e0 + 2;
		const uint32_t a = org[e0], b = org[e1], c = org[e2];
		auto rec = [&](uint32_t e) { *ovc++ = e; sent[org[e]] = next_id++; };
		unsigned mask = (sent[a] != NONE32 ? 4u : 0u) | (sent[b] != NONE32 ? 2u : 0u) | (sent[c] != NONE32 ? 1u : 0u);
		switch (mask) {   // encoder.h:68-131 (numtri is not coded: one polygon degree)
		case 7: em.iop(I_TRI111); em.vert(sent[a], seen[a]); em.vert(sent[b], seen[b]); em.vert(sent[c], seen[c]); break;
		case 6: em.iop(I_TRI110); em.vert(sent[a], seen[a]); em.vert(sent[b], seen[b]); rec(e2); break;
		case 3: em.iop(I_TRI011); em.vert(sent[b], seen[b]); em.vert(sent[c], seen[c]); rec(e0); break;
		case 5: em.iop(I_TRI101); em.vert(sent[c], seen[c]); em.vert(sent[a], seen[a]); rec(e1); break;
		case 4: em.iop(I_TRI100); em.vert(sent[a], seen[a]); rec(e1); rec(e2); break;
		case 2: em.iop(I_TRI010); em.vert(sent[b], seen[b]); rec(e2); rec(e0); break;
		case 1: em.iop(I_TRI001); em.vert(sent[c], seen[c]); rec(e0); rec(e1); break;
		default: em.iop(I_INIT); rec(e0); rec(e1); rec(e2); break;
		}
		*ofc++ = e0;
		++seen[a]; ++seen[b]; ++seen[c];
		cb.start(a, e0, b, e1, c, e2);
	}
	Border::Node *P = cb.P;
	const uint32_t snap_every = MODEL ? 0u : w.snapshot_faces;
	uint64_t next_snap = snap_every ? (uint64_t)consumed_io + snap_every : ~0ull;   // (host.hpp BorderSnapshot)
	while (!cb.parts.empty()) {
		if (consumed >= next_snap) {
			next_snap += snap_every;
			w.snapshots.emplace_back();
			Emitter::snapshot_border(cb, sent, seen, w.snapshots.back());
			em.snapshot_cursors(w.snapshots.back(), next_id, consumed - consumed_io, 3u * (consumed - consumed_io), n_op, em.nt_cur);
		}
		Border::Part &pt = cb.parts.back();
		const int32_t tn = pt.tail, hn = pt.head;
		const uint32_t v0 = P[tn].v, gate = P[tn].a, v1 = P[hn].v;
		const uint32_t order = seen[v1];
		const uint32_t t = twin[gate];
		uint32_t fc = t / 3u;
		if (t == gate || gone[fc] != Gone::no) {   // writer.cc:48-58: mesh border or neighbour already consumed
			const Op bop = cb.border();
			P = cb.P;
			if (t != gate) { twin[gate] = gate; changed = true; tp.push_back(gate); }   // one-sided split (writer.cc:81-84)
			emit(bop, order);
			continue;
		}
		gone[fc] = Gone::yes; ++consumed;
		const uint32_t base = 3u * fc, kk = t - base;
		const uint32_t e0 = t, e1 = base + (kk == 2u ? 0u : kk + 1u), e2 = base + (kk == 0u ? 2u : kk - 1u);
		const uint32_t v2 = org[e2];
		if (pf_level >= 1) {
			// the next gate is one of this triangle's other two edges: their twins (this face's line of the twin array) and,
			// one step further, the faces behind them
			const uint32_t t1 = twin[e1], t2 = twin[e2];
			if (pf_level >= 2) { __builtin_prefetch(org + t1); __builtin_prefetch(org + t2); }
			if (pf_level >= 3) { __builtin_prefetch(gone + t1 / 3u); __builtin_prefetch(gone + t2 / 3u); }
		}
		const bool fresh = sent[v2] == NONE32;
		if (fresh || on[v2] == 0) {
			// NEWVTX, or a vertex that was coded before but left the border (non-manifold): encoder.h:167-181
			P[tn].a = e1;
			const int32_t nn = cb.make(v2, e2);
			P = cb.P;
			cb.append(cb.parts.back(), nn);
			if (fresh) { emit(O_NEWVTX, order); *ovc++ = e2; sent[v2] = next_id++; }
			else { emit(O_NM, order); sync_n(); em.vert(sent[v2], seen[v2]); }
		} else if (pt.edge_begin && P[P[hn].next].v == v2) {
			// connect forward (or close: the part is exactly this triangle); the triangle's last edge meets the next border edge
			const bool close = pt.size == 3;
			const uint32_t gatenext = P[hn].a;
			if (twin[gatenext] != e2) { twin[gatenext] = e2; twin[e2] = gatenext; changed = true; tp.push_back(gatenext); tp.push_back(e2); }
			if (close) {
				const uint32_t gateprev = P[P[tn].prev].a;
				if (twin[gateprev] != e1) { twin[gateprev] = e1; twin[e1] = gateprev; changed = true; tp.push_back(gateprev); tp.push_back(e1); }
				cb.discard_top();
			} else { cb.drop(cb.unlink_head(pt)); P[pt.tail].a = e1; }
			emit(O_CONNFWD, order);
		} else if (P[P[tn].prev].v == v2) {
			const uint32_t gateprev = P[P[tn].prev].a;
			if (twin[gateprev] != e1) { twin[gateprev] = e1; twin[e1] = gateprev; changed = true; tp.push_back(gateprev); tp.push_back(e1); }
			cb.drop(cb.unlink_tail(pt));
			P[pt.tail].a = e2;
			emit(O_CONNBWD, order);
		} else {
			int i, p;
			const int32_t hit = cb.locate(v2, i, p);
			int32_t g, cp;
			if (p > 0) {
				cb.unite(hit, p, g, cp);
				P = cb.P;
				emit(O_UNION, order); sync_n(); em.elem(i); em.part(p);
			} else {
				cb.split(hit, i, g, cp);
				P = cb.P;
				emit(O_SPLIT, order); sync_n(); em.elem(i);
			}
			P[g].a = e1; P[cp].a = e2;
		}
		++seen[v0]; ++seen[v1]; ++seen[v2];
		*ofc++ = e0;
	}
	sync_n();
	for (int i = 0; i < 8; ++i) em.n_op[i] += n_op[i];
	em.halfedges += 3 * (consumed - consumed_io);
	if (!MODEL) em.op_cur = opc;
	em.ov_cur = ovc; em.of_cur = ofc;
	if (changed) w.twins_changed = true;
	next_id_io = next_id; consumed_io = consumed;
}

// ---------------------------------------------------------------------------------------------------------------------
// The triangle walk on TWO cores (round 5).  The loop above is bound by its instructions (132 per triangle at 3.5 per cycle, no
// misses to speak of), and a quarter of them decide nothing: the triangle counts per vertex (three read-modify-writes per
// triangle: they only select an operation's model class, models.h:101-105), the operation byte, the coded-vertex and coded-face
// entries, the per-class counters.  So the walking thread (A) keeps what the automaton needs -- the border, the face marks, which
// vertices are coded, the twins -- and writes ONE 8-byte record per operation into a trace: (half-edge the triangle was entered
// through | operation) or (head vertex | border operation), plus the rare operands (the split / union position).  A second thread
// (B) on a core next to it (same last-level cache, not the sibling hardware thread) follows the trace and expands it: the vertices
// of a triangle are the origins of the entered half-edge's face, in order (the entered half-edge runs against the gate: from its
// head to its tail), the model class from its own count table, then the operation byte, order_v / order_f, and everything that
// goes through the Emitter (component marks, initial operations, explicitly named vertices).  The output is what the one-thread
// loop writes, entry for entry (tests/test_host_cpu.py compares them); A alone measured 7.9 -> 6.0 - 6.4 ms per million
// triangles on the boxes' EPYC 9575F before B existed (a ring that stays in its cache; 6.9 with one that does not).
// HRY_WALK_SPLIT=0: the one-thread loop.
struct WalkTrace {
	enum { T_TRI = 0, T_BORDER = 1, T_START = 2, T_NEXTID = 3, T_ELEM = 4, T_PART = 5, T_SNAP = 6 };   // code in bits 8..15 of the high word, operation in bits 0..7
	// a RING of kRing records (8 MB: stays in the two cores' shared cache; until late in round 5 one array for the whole walk,
	// 64 MB per million triangles from the block pool -- whose blocks the decode between two encodes may have taken, and then the
	// walk paid the page faults of a fresh one: one step in twenty took 10 - 18 ms instead of 6).  The walking thread waits when it
	// is a ring ahead (never seen: the expanding thread stays within 10^4 - 10^5 records).
	static constexpr size_t kRing = (size_t)1 << 20;
	size_t ring = kRing, mask = kRing - 1;   // (HRY_WALK_RING: a smaller ring, so that small test meshes go round it many times)
	BigVec<uint64_t> rec;
	alignas(64) std::atomic<size_t> head{ 0 };
	alignas(64) std::atomic<int> done{ 0 };
	alignas(64) std::atomic<size_t> tail{ 0 };   // how far the expanding thread has come (published once per batch it takes: traces, tests)
	std::atomic<int> failed{ 0 };                // the expanding thread has left with an exception: the walking thread must not wait for room any more
	static uint64_t make(uint32_t a, uint32_t code, uint32_t op, uint32_t payload = 0) { return (uint64_t)a | ((uint64_t)(op | (code << 8) | (payload << 16)) << 32); }
};

// one ring is kept between calls (its pages stay where they are)
static std::mutex g_trace_mu;
static std::unique_ptr<WalkTrace> g_trace_spare;
static std::unique_ptr<WalkTrace> take_walk_trace()
{
	std::unique_ptr<WalkTrace> t;
	{ std::lock_guard<std::mutex> g(g_trace_mu); t = std::move(g_trace_spare); }
	size_t want = WalkTrace::kRing;
	if (const char *e = getenv("HRY_WALK_RING")) { const size_t v = (size_t)strtoull(e, nullptr, 10); want = 1024; while (want < v && want < WalkTrace::kRing) want <<= 1; }
	if (t && t->ring != want) t.reset();
	if (!t) { t.reset(new WalkTrace()); t->ring = want; t->mask = want - 1; t->rec.resize(want); }
	t->head.store(0, std::memory_order_relaxed); t->tail.store(0, std::memory_order_relaxed); t->done.store(0, std::memory_order_relaxed); t->failed.store(0, std::memory_order_relaxed);
	return t;
}
static void keep_walk_trace(std::unique_ptr<WalkTrace> t)
{
	std::lock_guard<std::mutex> g(g_trace_mu);
	if (!g_trace_spare) g_trace_spare = std::move(t);
}

// A: the automaton (the loop of walk_component_tri<false> without what B does)
static void walk_component_tri_a(Mesh &m, WalkState &st, uint32_t f, Border &cb, WalkResult &w, WalkTrace &tr, size_t &at_io, uint32_t &next_id_io, uint32_t &consumed_io)
{
	const uint32_t *org = m.org.data();
	uint32_t *twin = m.twin.data();
	Gone *gone = st.gone.data();
	uint32_t *sent = st.sent.data();
	OnCount *on = st.on.data();
	uint32_t next_id = next_id_io, consumed = consumed_io;
	bool changed = false;
	std::vector<uint32_t> &tp = w.twin_patches;
	static const int pf_level = [] { const char *e = getenv("HRY_WALK_PREFETCH"); return e ? atoi(e) : 2; }();
	uint64_t *rec = tr.rec.data();
	size_t at = at_io, published = at_io;
	const size_t ring = tr.ring, rmask = tr.mask;
	size_t room_upto = tr.tail.load(std::memory_order_acquire) + ring;   // records below it may be written
	auto put = [&](uint64_t r) {
		if (at >= room_upto) {   // (a ring ahead of the expanding thread)
			tr.head.store(at, std::memory_order_release); published = at;
			while (at >= (room_upto = tr.tail.load(std::memory_order_acquire) + ring)) {
				if (tr.failed.load(std::memory_order_acquire)) throw Error(HRY_E_INTERNAL, "walk: the expanding thread has failed");   // (its own exception is reported in front of this one)
				__builtin_ia32_pause();
			}
		}
		rec[at++ & rmask] = r;
		if (at - published >= 256) { tr.head.store(at, std::memory_order_release); published = at; }
	};
	gone[f] = Gone::yes; ++consumed;
	{
		const uint32_t e0 = 3 * f, e1 = e0 + 1, e2 = e0 + 2;
		const uint32_t a = org[e0], b = org[e1], c = org[e2];
		const unsigned mask = (sent[a] != NONE32 ? 4u : 0u) | (sent[b] != NONE32 ? 2u : 0u) | (sent[c] != NONE32 ? 1u : 0u);
		put(WalkTrace::make(f, WalkTrace::T_START, 0, mask));
		put(WalkTrace::make(next_id, WalkTrace::T_NEXTID, 0));
		auto fresh = [&](uint32_t v) { sent[v] = next_id++; };
		switch (mask) {   // the order in which the one-thread loop numbers the new vertices (encoder.h:68-131)
		case 7: break;
		case 6: fresh(c); break;
		case 3: fresh(a); break;
		case 5: fresh(b); break;
		case 4: fresh(b); fresh(c); break;
		case 2: fresh(c); fresh(a); break;
		case 1: fresh(a); fresh(b); break;
		default: fresh(a); fresh(b); fresh(c); break;
		}
		cb.start(a, e0, b, e1, c, e2);
	}
	Border::Node *P = cb.P;
	const uint32_t snap_every = w.snapshot_faces;
	uint64_t next_snap = snap_every ? (uint64_t)consumed_io + snap_every : ~0ull;   // (host.hpp BorderSnapshot)
	while (!cb.parts.empty()) {
		if (consumed >= next_snap) {
			// the border is this thread's, the triangle counts and the cursors are the expanding thread's: it completes the
			// snapshot when it gets to the record (the array does not move: walk_sequential reserved it)
			next_snap += snap_every;
			if (w.snapshots.size() == w.snapshots.capacity()) throw Error(HRY_E_INTERNAL, "walk: more border snapshots than faces allow");
			const uint32_t idx = (uint32_t)w.snapshots.size();
			w.snapshots.emplace_back();
			BorderSnapshot &sn = w.snapshots.back();
			Emitter::snapshot_border(cb, sent, nullptr, sn);
			sn.first_vertex = next_id - next_id_io; sn.first_face = consumed - consumed_io; sn.first_halfedge = 3u * (consumed - consumed_io);
			put(WalkTrace::make(idx, WalkTrace::T_SNAP, 0));
		}
		Border::Part &pt = cb.parts.back();
		const int32_t tn = pt.tail, hn = pt.head;
		const uint32_t gate = P[tn].a;
		const uint32_t t = twin[gate];
		uint32_t fc = t / 3u;
		if (t == gate || gone[fc] != Gone::no) {   // writer.cc:48-58: mesh border or neighbour already consumed
			const uint32_t v1 = P[hn].v;
			const Op bop = cb.border();
			P = cb.P;
			if (t != gate) { twin[gate] = gate; changed = true; tp.push_back(gate); }   // one-sided split (writer.cc:81-84)
			put(WalkTrace::make(v1, WalkTrace::T_BORDER, (uint32_t)bop));
			continue;
		}
		gone[fc] = Gone::yes; ++consumed;
		const uint32_t base = 3u * fc, kk = t - base;
		const uint32_t e1 = base + (kk == 2u ? 0u : kk + 1u), e2 = base + (kk == 0u ? 2u : kk - 1u);
		const uint32_t v2 = org[e2];
		if (pf_level >= 1) {
			const uint32_t t1 = twin[e1], t2 = twin[e2];
			if (pf_level >= 2) { __builtin_prefetch(org + t1); __builtin_prefetch(org + t2); }
			if (pf_level >= 3) { __builtin_prefetch(gone + t1 / 3u); __builtin_prefetch(gone + t2 / 3u); }
		}
		const bool fresh = sent[v2] == NONE32;
		if (fresh || on[v2] == 0) {
			P[tn].a = e1;
			const int32_t nn = cb.make(v2, e2);
			P = cb.P;
			cb.append(cb.parts.back(), nn);
			if (fresh) { sent[v2] = next_id++; put(WalkTrace::make(t, WalkTrace::T_TRI, O_NEWVTX)); }
			else put(WalkTrace::make(t, WalkTrace::T_TRI, O_NM));
		} else if (pt.edge_begin && P[P[hn].next].v == v2) {
			const bool close = pt.size == 3;
			const uint32_t gatenext = P[hn].a;
			if (twin[gatenext] != e2) { twin[gatenext] = e2; twin[e2] = gatenext; changed = true; tp.push_back(gatenext); tp.push_back(e2); }
			if (close) {
				const uint32_t gateprev = P[P[tn].prev].a;
				if (twin[gateprev] != e1) { twin[gateprev] = e1; twin[e1] = gateprev; changed = true; tp.push_back(gateprev); tp.push_back(e1); }
				cb.discard_top();
			} else { cb.drop(cb.unlink_head(pt)); P[pt.tail].a = e1; }
			put(WalkTrace::make(t, WalkTrace::T_TRI, O_CONNFWD));
		} else if (P[P[tn].prev].v == v2) {
			const uint32_t gateprev = P[P[tn].prev].a;
			if (twin[gateprev] != e1) { twin[gateprev] = e1; twin[e1] = gateprev; changed = true; tp.push_back(gateprev); tp.push_back(e1); }
			cb.drop(cb.unlink_tail(pt));
			P[pt.tail].a = e2;
			put(WalkTrace::make(t, WalkTrace::T_TRI, O_CONNBWD));
		} else {
			int i, p;
			const int32_t hit = cb.locate(v2, i, p);
			int32_t g, cp;
			if (p > 0) {
				cb.unite(hit, p, g, cp);
				P = cb.P;
				put(WalkTrace::make(t, WalkTrace::T_TRI, O_UNION));
				put(WalkTrace::make((uint32_t)i, WalkTrace::T_ELEM, 0));
				put(WalkTrace::make((uint32_t)p, WalkTrace::T_PART, 0));
			} else {
				cb.split(hit, i, g, cp);
				P = cb.P;
				put(WalkTrace::make(t, WalkTrace::T_TRI, O_SPLIT));
				put(WalkTrace::make((uint32_t)i, WalkTrace::T_ELEM, 0));
			}
			P[g].a = e1; P[cp].a = e2;
		}
	}
	tr.head.store(at, std::memory_order_release);
	at_io = at;
	if (changed) w.twins_changed = true;
	next_id_io = next_id; consumed_io = consumed;
}

// B: follows the trace until A says it is complete; owns the Emitter and the per-vertex triangle counts meanwhile
static void walk_trace_expand(const Mesh &m, WalkState &st, Emitter &em, WalkTrace &tr)
{
	const uint32_t *org = m.org.data();
	const uint32_t *sent = st.sent.data();   // (read for vertices that were coded before the record was written: final by then)
	uint16_t *seen = st.seen.data();
	const uint64_t *rec = tr.rec.data();
	const size_t rmask = tr.mask;
	BorderSnapshot *const snaps = em.w.snapshots.data();   // (reserved before the walk: the walking thread appends, the array stays where it is)
	OpByte *opc = em.op_cur;
	uint32_t *ovc = em.ov_cur, *ofc = em.of_cur;
	uint32_t n_op[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, n_ops = 0, faces = 0;
	auto emit = [&](uint32_t s, uint32_t order) {
		const uint32_t k = order == 0 ? 0u : order > 8u ? 7u : order - 1u;   // models.h:101-105
		++n_op[k]; ++n_ops;
		*opc++ = (OpByte)(s | (k << 3));
	};
	auto sync_n = [&] { em.n += n_ops; n_ops = 0; };
	size_t pos = 0, tail_said = 0;
	for (;;) {
		size_t h = tr.head.load(std::memory_order_acquire);
		if (h == pos) {
			if (tr.done.load(std::memory_order_acquire)) { h = tr.head.load(std::memory_order_acquire); if (h == pos) break; }
			else { __builtin_ia32_pause(); continue; }
		}
		// (a split / union / start record is followed by its operands: A publishes whole groups only at the end of a component or
		// in blocks of 256, so an operand may still be on its way -- wait for it)
		auto need = [&](size_t k) {
			if (h < k) { tr.tail.store(pos, std::memory_order_release); tail_said = pos; }   // (the record in hand has been read: the walking thread is not kept waiting for room by this wait)
			while (h < k) {
				// (the walking thread has stopped -- with an exception between a record and its operands -- and nothing more will come)
				if (tr.done.load(std::memory_order_acquire) && (h = tr.head.load(std::memory_order_acquire)) < k) throw Error(HRY_E_INTERNAL, "walk: truncated trace");
				__builtin_ia32_pause(); h = tr.head.load(std::memory_order_acquire);
			}
		};
		while (pos < h) {
			if (pos - tail_said >= 4096) { tr.tail.store(pos, std::memory_order_release); tail_said = pos; }
			if (pos + 8 < h) { const uint64_t ahead = rec[(pos + 8) & rmask]; if (((ahead >> 40) & 0xffu) == WalkTrace::T_TRI) __builtin_prefetch(org + 3u * ((uint32_t)ahead / 3u)); }
			const uint64_t r = rec[pos++ & rmask];
			const uint32_t a = (uint32_t)r, hi = (uint32_t)(r >> 32), op = hi & 0xffu, code = (hi >> 8) & 0xffu;
			if (code == WalkTrace::T_TRI) {
				const uint32_t base = 3u * (a / 3u), kk = a - base;
				const uint32_t e1 = base + (kk == 2u ? 0u : kk + 1u), e2 = base + (kk == 0u ? 2u : kk - 1u);
				const uint32_t v1 = org[a], v0 = org[e1], v2 = org[e2];
				const uint32_t order = seen[v1];
				emit(op, order);
				if (op == O_NEWVTX) *ovc++ = e2;
				else if (op == O_NM) { sync_n(); em.vert(sent[v2], seen[v2]); }
				else if (op == O_UNION) {
					need(pos + 2);
					sync_n(); em.elem((int)(int32_t)(uint32_t)rec[pos & rmask]); em.part((int)(uint32_t)rec[(pos + 1) & rmask]);
					pos += 2;
				} else if (op == O_SPLIT) {
					need(pos + 1);
					sync_n(); em.elem((int)(int32_t)(uint32_t)rec[pos & rmask]);
					pos += 1;
				}
				++seen[v0]; ++seen[v1]; ++seen[v2];
				*ofc++ = a;
				++faces;
			} else if (code == WalkTrace::T_BORDER) {
				emit(op, seen[a]);
			} else if (code == WalkTrace::T_START) {
				need(pos + 1);
				const uint32_t next_id = (uint32_t)rec[pos++ & rmask];
				const unsigned mask = (hi >> 16) & 0xffu;
				// the Emitter's own view of the cursors and counters (its mark of the new component reads them)
				sync_n();
				for (int i = 0; i < 8; ++i) { em.n_op[i] += n_op[i]; n_op[i] = 0; }
				em.halfedges += 3 * faces; faces = 0;
				em.op_cur = opc; em.ov_cur = ovc; em.of_cur = ofc;
				em.mark_component(next_id);
				const uint32_t e0 = 3 * a, e1 = e0 + 1, e2 = e0 + 2;
				const uint32_t va = org[e0], vb = org[e1], vc = org[e2];
				auto recv = [&](uint32_t e) { *ovc++ = e; };
				switch (mask) {   // encoder.h:68-131 (numtri is not coded: one polygon degree)
				case 7: em.iop(I_TRI111); em.vert(sent[va], seen[va]); em.vert(sent[vb], seen[vb]); em.vert(sent[vc], seen[vc]); break;
				case 6: em.iop(I_TRI110); em.vert(sent[va], seen[va]); em.vert(sent[vb], seen[vb]); recv(e2); break;
				case 3: em.iop(I_TRI011); em.vert(sent[vb], seen[vb]); em.vert(sent[vc], seen[vc]); recv(e0); break;
				case 5: em.iop(I_TRI101); em.vert(sent[vc], seen[vc]); em.vert(sent[va], seen[va]); recv(e1); break;
				case 4: em.iop(I_TRI100); em.vert(sent[va], seen[va]); recv(e1); recv(e2); break;
				case 2: em.iop(I_TRI010); em.vert(sent[vb], seen[vb]); recv(e2); recv(e0); break;
				case 1: em.iop(I_TRI001); em.vert(sent[vc], seen[vc]); recv(e0); recv(e1); break;
				default: em.iop(I_INIT); recv(e0); recv(e1); recv(e2); break;
				}
				*ofc++ = e0;
				++seen[va]; ++seen[vb]; ++seen[vc];
				++faces;
			} else if (code == WalkTrace::T_SNAP) {
				// a border snapshot of the walking thread: the counts at its vertices and the cursors are this thread's
				BorderSnapshot &sn = snaps[a];
				sn.seen.reserve(sn.orig.size());
				for (const uint32_t v : sn.orig) sn.seen.push_back((uint8_t)std::min<uint32_t>(seen[v], 9u));
				std::vector<uint32_t>().swap(sn.orig);
				if (sn.first_face != faces) throw Error(HRY_E_INTERNAL, "walk trace: a snapshot out of step");
				em.snapshot_cursors(sn, em.w.marks.back().first_vertex + sn.first_vertex, sn.first_face, sn.first_halfedge, n_op, em.nt_cur);
			} else throw Error(HRY_E_INTERNAL, "walk trace: stray record");
		}
		tr.tail.store(pos, std::memory_order_release);   // (everything below pos has been read: the walking thread may write over it)
		tail_said = pos;
	}
	sync_n();
	for (int i = 0; i < 8; ++i) em.n_op[i] += n_op[i];
	em.halfedges += 3 * faces;
	em.op_cur = opc; em.ov_cur = ovc; em.of_cur = ofc;
}

// The component walk for polygons, written like the triangle loop above (round 4; hardware counters of the generic loop on the
// configs[3] share: 240 instructions and 108 cycles per triangle).  A polygon is a fan of triangles around the vertex its gate
// starts at (encoder.h:133-166): the face's half-edge range stays in locals while its triangles are coded, so "next edge" is a
// compare instead of two table lookups per step; cursors and counters are locals; the gate's neighbours are loaded only by the
// operations that use them; the triangle count of a polygon goes through a bare cursor (one per coded face, Emitter::nt_cur).
template <int DEG>
static void walk_component_poly(Mesh &m, WalkState &st, const uint32_t *eface_tab, uint32_t f, Border &cb, Emitter &em, uint32_t &next_id_io, uint32_t &consumed_io)
{
	static const bool kWalkPrefetch = [] { const char *e = getenv("HRY_WALK_PREFETCH"); return !e || atoi(e) != 0; }();
	WalkResult &w = em.w;
	const uint32_t *foff = m.face_off.data();
	const uint32_t *org = m.org.data();
	uint32_t *twin = m.twin.data();
	Gone *gone = st.gone.data();
	uint32_t *sent = st.sent.data();
	uint16_t *seen = st.seen.data();
	OnCount *on = st.on.data();
	uint32_t next_id = next_id_io, consumed = consumed_io, halfedges = em.halfedges;
	OpByte *opc = em.op_cur;
	uint32_t *ovc = em.ov_cur, *ofc = em.of_cur;
	uint32_t n_op[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, n_ops = 0;
	bool changed = false;
	std::vector<uint32_t> &tp = w.twin_patches;   // half-edges whose twin this walk changes (rare: non-manifold edges, consumed neighbours)
	const bool nt_coded = w.numtri_coded;
	auto face_of = [&](uint32_t e) -> uint32_t { return DEG ? e / (uint32_t)(DEG ? DEG : 1) : eface_tab[e]; };
	auto emit = [&](uint32_t s, uint32_t order) {
		uint32_t k = order == 0 ? 0u : order > 8u ? 7u : order - 1u;   // models.h:101-105
		++n_op[k]; ++n_ops;
		*opc++ = (OpByte)(s | (k << 3));
	};
	// rare symbols go through the Emitter (its symbol counter is brought up to date first)
	auto sync_n = [&] { em.n += n_ops; n_ops = 0; };

	em.mark_component(next_id);
	gone[f] = Gone::yes; ++consumed;
	uint32_t fb = foff[f], fe = foff[f + 1];   // the half-edges of the polygon at hand
	halfedges += fe - fb;
	uint32_t ntri = fe - fb - 2, curtri = 1;
	uint32_t e0 = fb, e1 = fb + 1, e2 = fb + 2;
	{
		const uint32_t a = org[e0], b = org[e1], c = org[e2];
		auto rec = [&](uint32_t e) { *ovc++ = e; sent[org[e]] = next_id++; };
		unsigned mask = (sent[a] != NONE32 ? 4u : 0u) | (sent[b] != NONE32 ? 2u : 0u) | (sent[c] != NONE32 ? 1u : 0u);
		switch (mask) {   // encoder.h:68-131
		case 7: em.iop(I_TRI111); em.vert(sent[a], seen[a]); em.vert(sent[b], seen[b]); em.vert(sent[c], seen[c]); em.numtri((int)ntri); break;
		case 6: em.iop(I_TRI110); em.vert(sent[a], seen[a]); em.vert(sent[b], seen[b]); em.numtri((int)ntri); rec(e2); break;
		case 3: em.iop(I_TRI011); em.vert(sent[b], seen[b]); em.vert(sent[c], seen[c]); em.numtri((int)ntri); rec(e0); break;
		case 5: em.iop(I_TRI101); em.vert(sent[c], seen[c]); em.vert(sent[a], seen[a]); em.numtri((int)ntri); rec(e1); break;
		case 4: em.iop(I_TRI100); em.vert(sent[a], seen[a]); em.numtri((int)ntri); rec(e1); rec(e2); break;
		case 2: em.iop(I_TRI010); em.vert(sent[b], seen[b]); em.numtri((int)ntri); rec(e2); rec(e0); break;
		case 1: em.iop(I_TRI001); em.vert(sent[c], seen[c]); em.numtri((int)ntri); rec(e0); rec(e1); break;
		default: em.iop(I_INIT); em.numtri((int)ntri); rec(e0); rec(e1); rec(e2); break;
		}
		*ofc++ = e0;
		++seen[a]; ++seen[b]; ++seen[c];
		cb.start(a, e0, b, e1, c, e2);
	}
	uint32_t *ntc = em.nt_cur, *ntp = em.ntp_cur;
	Border::Node *P = cb.P;
	const uint32_t snap_every = w.snapshot_faces, halfedges0 = em.halfedges;
	uint64_t next_snap = snap_every ? (uint64_t)consumed_io + snap_every : ~0ull;   // (host.hpp BorderSnapshot)
	while (!cb.parts.empty()) {
		if (consumed >= next_snap && curtri == ntri) {
			next_snap += snap_every;
			w.snapshots.emplace_back();
			Emitter::snapshot_border(cb, sent, seen, w.snapshots.back());
			em.snapshot_cursors(w.snapshots.back(), next_id, consumed - consumed_io, halfedges - halfedges0, n_op, ntc);
		}
		Border::Part &pt = cb.parts.back();
		const int32_t tn = pt.tail, hn = pt.head;
		const uint32_t v0 = P[tn].v, gate = P[tn].a, v1 = P[hn].v;
		const uint32_t order = seen[v1];
		const bool first = curtri == ntri;   // the polygon before is finished: the gate leads into the next one
		if (first) {
			const uint32_t t = twin[gate];
			const uint32_t fc = t == gate ? 0u : face_of(t);
			if (t == gate || gone[fc] != Gone::no) {   // writer.cc:48-58: mesh border or neighbour already consumed
				const Op bop = cb.border();
				P = cb.P;
				if (t != gate) { twin[gate] = gate; changed = true; tp.push_back(gate); }   // one-sided split (writer.cc:81-84)
				emit(bop, order);
				continue;
			}
			gone[fc] = Gone::yes; ++consumed;
			fb = foff[fc]; fe = foff[fc + 1];
			halfedges += fe - fb;
			ntri = fe - fb - 2; curtri = 0;
			e0 = t;
			e1 = t + 1 == fe ? fb : t + 1;
			// the faces behind this polygon's other edges are the next gates' neighbours: their lines (twins, origins, the
			// half-edge -> face table) are asked for now -- a component is walked once, every line of it is a miss the first time
			if (kWalkPrefetch) {
				for (uint32_t h = fb; h < fe; ++h) {
					const uint32_t o = twin[h];
					__builtin_prefetch(twin + o); __builtin_prefetch(org + o);
					if (!DEG) __builtin_prefetch(eface_tab + o);
				}
			}
		} else e1 = e1 + 1 == fe ? fb : e1 + 1;
		e2 = e1 + 1 == fe ? fb : e1 + 1;
		const uint32_t v2 = org[e2];
		const bool fresh = sent[v2] == NONE32;
		if (fresh || on[v2] == 0) {
			// NEWVTX, or a vertex that was coded before but left the border (non-manifold): encoder.h:167-181
			P[tn].a = e1;
			const int32_t nn = cb.make(v2, e2);
			P = cb.P;
			cb.append(cb.parts.back(), nn);
			if (fresh) { emit(O_NEWVTX, order); *ovc++ = e2; sent[v2] = next_id++; }
			else { emit(O_NM, order); sync_n(); em.vert(sent[v2], seen[v2]); }
		} else if (pt.edge_begin && P[P[hn].next].v == v2) {
			// connect forward (or close: the part is exactly this triangle)
			const bool close = pt.size == 3;
			if (curtri + 1 == ntri) {   // the polygon's last triangle: its last edge meets the next border edge
				const uint32_t gatenext = P[hn].a;
				if (twin[gatenext] != e2) { twin[gatenext] = e2; twin[e2] = gatenext; changed = true; tp.push_back(gatenext); tp.push_back(e2); }
			}
			if (close) {
				const uint32_t gateprev = P[P[tn].prev].a;
				if (twin[gateprev] != e1) { twin[gateprev] = e1; twin[e1] = gateprev; changed = true; tp.push_back(gateprev); tp.push_back(e1); }
				cb.discard_top();
			} else { cb.drop(cb.unlink_head(pt)); P[pt.tail].a = e1; }
			emit(O_CONNFWD, order);
		} else if (P[P[tn].prev].v == v2) {
			const uint32_t gateprev = P[P[tn].prev].a;
			if (twin[gateprev] != e1) { twin[gateprev] = e1; twin[e1] = gateprev; changed = true; tp.push_back(gateprev); tp.push_back(e1); }
			cb.drop(cb.unlink_tail(pt));
			P[pt.tail].a = e2;
			emit(O_CONNBWD, order);
		} else {
			int i, p;
			const int32_t hit = cb.locate(v2, i, p);
			int32_t g, cp;
			if (p > 0) {
				cb.unite(hit, p, g, cp);
				P = cb.P;
				emit(O_UNION, order); sync_n(); em.elem(i); em.part(p);
			} else {
				cb.split(hit, i, g, cp);
				P = cb.P;
				emit(O_SPLIT, order); sync_n(); em.elem(i);
			}
			P[g].a = e1; P[cp].a = e2;
		}
		if (first) {
			if (nt_coded) {   // io.h:162-165: the triangle count follows the first operation of the polygon and its operands
				*ntc++ = (uint32_t)(uint16_t)ntri;
				if (ntp) { sync_n(); *ntp++ = em.n; }
				n_ops += kGroupBytes[G_NUMTRI];   // (two places in the symbol sequence)
			}
			*ofc++ = e0;
		}
		++seen[v0]; ++seen[v1]; ++seen[v2];
		++curtri;
	}
	sync_n();
	for (int i = 0; i < 8; ++i) em.n_op[i] += n_op[i];
	em.halfedges = halfedges;
	em.op_cur = opc; em.ov_cur = ovc; em.of_cur = ofc;
	em.nt_cur = ntc; em.ntp_cur = ntp;
	if (changed) w.twins_changed = true;
	next_id_io = next_id; consumed_io = consumed;
}

// the walk is over, every mark has its final counts: the snapshots' cursors from "since my component's mark" to absolute
static void finish_snapshots(WalkResult &w)
{
	for (BorderSnapshot &s : w.snapshots) {
		const ComponentMark &mk = w.marks.at(s.mark);
		for (int g = 0; g < G_COUNT; ++g) s.n_grp[g] += mk.n_grp[g];
		for (int i = 0; i < 8; ++i) s.n_op[i] += mk.n_op[i];
		s.first_vertex += mk.first_vertex; s.first_face += mk.first_face; s.first_halfedge += mk.first_halfedge;
		if (!s.orig.empty() || s.seen.size() != s.vtx.size()) throw Error(HRY_E_INTERNAL, "walk: a border snapshot was left incomplete");
	}
}

template <int DEG>
static void walk_rest_parallel(Mesh &m, WalkState &st, const uint32_t *eface_tab, Emitter &em0, uint32_t first_id, unsigned n_threads);
template <int DEG>
static void walk_components_parallel(Mesh &m, WalkState &st, const uint32_t *eface_tab, Emitter &em0, uint32_t first_id, unsigned n_threads, const ComponentAnalysis &A);

template <int DEG>
static void walk_sequential(Mesh &m, WalkResult &w, const uint32_t *eface_tab, bool eval_op_model, unsigned n_threads)
{
	const bool trace = getenv("HRY_TRACE") != nullptr;
	auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry walk] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	WalkState st(m.nv, m.nf);
	mark("(sequential part) state allocated");
	Border cb(st.on);
	StartFaces pool(m.nf, st.gone);
	if (!m.shard.seeds.empty()) {
		for (uint32_t f : m.shard.seeds) if (f >= m.nf) throw Error(HRY_E_ARG, "shard seed face out of range");
		pool.seeds = &m.shard.seeds;
	}
	Emitter em(w);
	em.eval_model = eval_op_model;
	em.attach((size_t)m.ne() + m.ntri() + 16, m.nv, m.nf);   // every half-edge ends at most one border operation, every triangle one other
	if (eval_op_model) w.snapshot_faces = 0;   // (the reference stream has no directory to put them in)
	if (w.snapshot_faces) w.snapshots.reserve(w.snapshots.size() + (size_t)m.nf / w.snapshot_faces + 2);   // (the two-core walk appends while its second thread reads: no growth)
	mark("(sequential part) start faces and output planes");
	uint32_t next_id = 0, consumed = 0;
	const bool count = getenv("HRY_PERF") != nullptr;   // hardware counters of this thread around the first component's walk
	// the triangle walk on two cores (walk_component_tri_a / walk_trace_expand above): large triangle meshes of the chunked profile
	struct Split {
		std::unique_ptr<WalkTrace> trace;
		std::thread expander;
		std::exception_ptr err;
		size_t at = 0;
		void stop()
		{
			if (!expander.joinable()) return;
			const size_t behind = trace->head.load(std::memory_order_relaxed) - trace->tail.load(std::memory_order_relaxed);
			const auto t_stop = std::chrono::steady_clock::now();
			trace->done.store(1, std::memory_order_release);
			expander.join();
			if (getenv("HRY_TRACE")) fprintf(stderr, "[hry walk] the expanding thread was %zu records behind the walk's %zu, joined after %.2f ms\n", behind,
			                                 trace->head.load(std::memory_order_relaxed), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_stop).count());
		}
		~Split() { stop(); if (trace) keep_walk_trace(std::move(trace)); }
	} split;
	const uint32_t split_min_faces = [] { const char *e = getenv("HRY_WALK_SPLIT"); return e ? (atoi(e) > 0 ? (uint32_t)atoi(e) : 0xffffffffu) : (1u << 17); }();   // 0: never; n: from n faces (read per call: the tests change it)
	const bool want_split = DEG == 3 && !eval_op_model && !getenv("HRY_GENERIC_WALK") && !count && n_threads > 1 && m.nf >= split_min_faces;
	do {
		uint32_t f = pool.next();
		const bool lean = DEG == 3 && !getenv("HRY_GENERIC_WALK");
		const bool lean_poly = DEG != 3 && !eval_op_model && !getenv("HRY_GENERIC_WALK");   // (the generic loop evaluates the operation model)
		if (count && consumed == 0) {
			PerfCounters pc;
			pc.start();
			if (lean && eval_op_model) walk_component_tri<true>(m, st, f, cb, em, next_id, consumed);
			else if (lean) walk_component_tri<false>(m, st, f, cb, em, next_id, consumed);
			else if (lean_poly) walk_component_poly<DEG>(m, st, eface_tab, f, cb, em, next_id, consumed);
			else walk_component<DEG>(m, st, eface_tab, f, cb, em, next_id, consumed);
			pc.stop();
			pc.report(eval_op_model ? "cut-border walk (with the operation model)" : "cut-border walk", (double)(em.halfedges - 2.0 * consumed));
			continue;
		}
		if (want_split) {
			if (!split.trace) {
				split.trace = take_walk_trace();   // (the ring of the call before, where there was one)
				const void *near_cpus = callers_neighbour_cpus();   // (this thread's cache domain without its own core)
				WalkTrace *trp = split.trace.get();
				std::exception_ptr *errp = &split.err;
				split.expander = std::thread([&m, &st, &em, trp, errp, near_cpus] {
					try { stay_on_node(near_cpus); walk_trace_expand(m, st, em, *trp); } catch (...) { *errp = std::current_exception(); trp->failed.store(1, std::memory_order_release); }
				});
			}
			try { walk_component_tri_a(m, st, f, cb, w, *split.trace, split.at, next_id, consumed); }
			catch (...) { split.stop(); if (split.err) std::rethrow_exception(split.err); throw; }   // (the expanding thread's failure is the cause, where there is one)
		}
		else if (lean && eval_op_model) walk_component_tri<true>(m, st, f, cb, em, next_id, consumed);
		else if (lean) walk_component_tri<false>(m, st, f, cb, em, next_id, consumed);
		else if (lean_poly) walk_component_poly<DEG>(m, st, eface_tab, f, cb, em, next_id, consumed);
		else walk_component<DEG>(m, st, eface_tab, f, cb, em, next_id, consumed);
		// The operation model of the reference stream adapts across the whole file (models.h:49-120), so a walk that evaluates
		// it is one sequence.  Without it (chunked profile: symbol + order class only) the remaining components are walked on
		// several threads once the first one shows that the mesh has more than one.
		if (n_threads > 1 && !eval_op_model && m.nf - consumed >= parallel_min_faces()) {
			split.stop();   // (the expander owns the Emitter until the trace is complete)
			if (split.err) std::rethrow_exception(split.err);
			em.detach();
			mark("(sequential part) first component walked");
			walk_rest_parallel<DEG>(m, st, eface_tab, em, next_id, n_threads);
			break;
		}
	} while (consumed != m.nf);
	split.stop();
	if (split.err) std::rethrow_exception(split.err);
	mark("(sequential part) back");
	em.detach();
	em.finish_marks();
	em.iop(I_EOM);
	mark("(sequential part) done");
	w.n_conn = em.n;
	for (int i = 0; i < 8; ++i) w.n_op_class[i] = em.n_op[i];
	finish_snapshots(w);
}

// ---- several host threads --------------------------------------------------------
}   // namespace
static thread_local unsigned t_thread_budget = 0;
void set_thread_budget(unsigned n) { t_thread_budget = n; }
unsigned host_threads()
{
	if (t_thread_budget) return t_thread_budget;
	if (const char *e = getenv("HRY_HOST_THREADS")) { int v = atoi(e); return v > 0 ? (unsigned)v : 1u; }
	// what the process may keep busy (affinity mask, control-group quota: thread_pool.cpp), shared with the other ranks of a
	// one-process-per-GPU launch on this node (LOCAL_WORLD_SIZE, torch.distributed.run); up to 32, and at most an eighth of a
	// large unshared node's CPUs: eight processes (one per GPU) share the node
	unsigned hw = cpu_allowance();
	if (const char *e = getenv("LOCAL_WORLD_SIZE")) { int v = atoi(e); if (v > 1) return std::max(1u, std::min(32u, hw / (unsigned)v)); }
	return std::max(1u, std::min(32u, hw >= 128 ? hw / 8 : std::min(16u, hw)));
}
// below this many remaining faces the analysis passes cost more than they save (HRY_PARALLEL_MIN_FACES overrides, tests)
uint32_t parallel_min_faces()
{
	if (const char *e = getenv("HRY_PARALLEL_MIN_FACES")) return (uint32_t)strtoul(e, nullptr, 10);
	return 1u << 16;
}
namespace {
// lock-free union-find on atomics: a root is always the smallest index of its set's links, so links never form a cycle
// big arrays of atomics come from the recycling pool like every other per-call array (fresh pages cost a fault per 4 KiB)
struct AtomicArray {
	std::atomic<uint32_t> *p;
	size_t bytes;
	explicit AtomicArray(size_t n) : bytes(std::max<size_t>(n, 1) * sizeof(std::atomic<uint32_t>))
	{
		static_assert(sizeof(std::atomic<uint32_t>) == 4 && std::is_trivially_destructible<std::atomic<uint32_t>>::value, "plain words");
		p = (std::atomic<uint32_t>*)(bytes >= BlockPool::kMinBytes ? BlockPool::take(bytes) : ::operator new(bytes));
	}
	~AtomicArray() { if (bytes >= BlockPool::kMinBytes) BlockPool::give(p); else ::operator delete(p); }
	AtomicArray(const AtomicArray&) = delete;
	AtomicArray &operator=(const AtomicArray&) = delete;
	std::atomic<uint32_t> &operator[](size_t i) const { return p[i]; }
};
struct AtomicSets {
	AtomicArray parent;
	explicit AtomicSets(size_t n) : parent(n) {}
	uint32_t find(uint32_t x)
	{
		for (;;) {
			uint32_t p = parent[x].load(std::memory_order_relaxed);
			if (p == x) return x;
			uint32_t g = parent[p].load(std::memory_order_relaxed);
			if (g != p) parent[x].compare_exchange_weak(p, g, std::memory_order_relaxed);   // path halving
			x = g;
		}
	}
	void unite(uint32_t a, uint32_t b)
	{
		for (;;) {
			a = find(a); b = find(b);
			if (a == b) return;
			if (a > b) std::swap(a, b);
			uint32_t expect = b;
			if (parent[b].compare_exchange_weak(expect, a, std::memory_order_relaxed)) return;
		}
	}
};
static void atomic_min(std::atomic<uint64_t> &a, uint64_t v)
{
	uint64_t cur = a.load(std::memory_order_relaxed);
	while (v < cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
}
static void atomic_min(std::atomic<uint32_t> &a, uint32_t v)
{
	uint32_t cur = a.load(std::memory_order_relaxed);
	while (v < cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
}
static void atomic_max(std::atomic<uint32_t> &a, uint32_t v)
{
	uint32_t cur = a.load(std::memory_order_relaxed);
	while (v > cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
}

// Component analysis shared by the multi-threaded walk (below) and the shard planner (shard.cpp).  The stream a sequential
// walk produces is a function of (a) which faces form a component, (b) the order of the components = the order of their
// first faces in the start-face sequence (or the explicit seed list of a shard), (c) the index the first new vertex of a
// component gets = the number of vertices the components before it introduce.  All three are computed up front; components
// that share a vertex are tied into one group (their symbols depend on that vertex's index and triangle count).
// gone / sent: faces already consumed and vertices already transmitted by the part walked before (nullptr: none).
template <int DEG>
static void analyse_impl(const Mesh &m, const uint32_t *eface_tab, const uint8_t *gone, const uint32_t *sent, unsigned n_threads, ComponentAnalysis &A)
{
	const uint32_t nf = m.nf, nv = m.nv;
	const uint32_t *org = m.org.data();
	const uint32_t *twin = m.twin.data();
	const uint32_t *foff = m.face_off.data();
	auto face_of = [&](uint32_t e) -> uint32_t { return DEG ? e / (uint32_t)(DEG ? DEG : 1) : eface_tab[e]; };
	auto is_gone = [&](uint32_t f) -> bool { return gone && gone[f]; };
	auto split = [&](uint32_t n, unsigned t, uint32_t &b, uint32_t &e) { b = (uint32_t)((uint64_t)n * t / n_threads); e = (uint32_t)((uint64_t)n * (t + 1) / n_threads); };
	const bool trace = getenv("HRY_TRACE") != nullptr;
	auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry walk] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	// (a) components of the remaining faces
	// Every thread first unites the faces of its own range among themselves, in order and without atomics: a face takes the
	// root of its first smaller neighbour (the face itself needs no search -- it is new -- and a neighbour's root is one or two
	// loads away, every path being compressed when it is walked), further smaller neighbours with another root link the two
	// roots.  Edges that leave the range, and one-sided twins, are noted and united afterwards through the atomics, when every
	// parent is initialised: the faces along the range boundaries on a mesh whose components are contiguous; a thread whose
	// notes outgrow a quarter of its faces (faces in random order) goes over its edges again instead of keeping them.
	// (Round 3 united every edge through two atomic searches: 81 of the 347 ms of the configs[3] host walk on the box's 16 CPUs.)
	AtomicSets sets(nf);
	{
		std::vector<std::vector<std::pair<uint32_t, uint32_t>>> cross(n_threads);
		std::vector<char> messy(n_threads, 0);
		parallel_for(n_threads, [&](unsigned t) {
			uint32_t b, e; split(nf, t, b, e);
			std::vector<std::pair<uint32_t, uint32_t>> &notes = cross[t];
			const size_t cap = (size_t)(e - b) / 4 + 4096;
			bool over = false;
			uint32_t *par = (uint32_t*)sets.parent.p;   // this thread's range only: plain words until the threads meet
			for (uint32_t f = b; f < e; ++f) {
				if (is_gone(f)) { par[f] = f; continue; }
				uint32_t r = NONE32;
				for (uint32_t h = foff[f], he = foff[f + 1]; h < he; ++h) {
					const uint32_t o = twin[h];
					if (o == h) continue;
					const uint32_t fb = face_of(o);
					if (fb == f || is_gone(fb)) continue;
					if (fb > f) { if (twin[o] != h && !over) notes.push_back({ f, fb }); continue; }   // (a two-sided edge is seen from the larger face)
					if (fb < b) { if (!over) notes.push_back({ f, fb }); continue; }
					uint32_t rb = par[fb];
					if (par[rb] != rb) { do rb = par[rb]; while (par[rb] != rb); par[fb] = rb; }
					if (r == NONE32) r = rb;
					else if (rb != r) { if (rb < r) std::swap(rb, r); par[rb] = r; }   // a root is the smallest face of its set
				}
				par[f] = r == NONE32 ? f : r;
				if (notes.size() > cap) { over = true; messy[t] = 1; std::vector<std::pair<uint32_t, uint32_t>>().swap(notes); }
			}
		});
		if (trace) { size_t nn = 0; unsigned nm = 0; for (unsigned t = 0; t < n_threads; ++t) { nn += cross[t].size(); nm += messy[t]; } fprintf(stderr, "[hry walk] %zu edges across the threads' ranges noted, %u of %u ranges go over their edges again\n", nn, nm, n_threads); }
		parallel_for(n_threads, [&](unsigned t) {
			if (!messy[t]) { for (const auto &pr : cross[t]) sets.unite(pr.first, pr.second); return; }
			uint32_t b, e; split(nf, t, b, e);
			for (uint32_t f = b; f < e; ++f) {
				if (is_gone(f)) continue;
				for (uint32_t h = foff[f], he = foff[f + 1]; h < he; ++h) {
					const uint32_t o = twin[h];
					if (o == h) continue;
					const uint32_t fb = face_of(o);
					if (fb != f && !is_gone(fb) && (fb < b || (fb > f && twin[o] != h))) sets.unite(f, fb);
				}
			}
		});
	}
	mark("  faces united");
	// dense component numbers: every face's root once (a thread keeps the roots it owns), the roots numbered in face order --
	// the number goes where the root's parent word was, nobody searches any more -- and every face takes its root's number
	BigVec<uint32_t> &comp = A.comp;
	comp.resize(nf);
	std::vector<std::vector<uint32_t>> roots_of(n_threads);
	parallel_for(n_threads, [&](unsigned t) {
		uint32_t b, e; split(nf, t, b, e);
		for (uint32_t f = b; f < e; ++f) {
			if (is_gone(f)) { comp[f] = NONE32; continue; }
			const uint32_t r = sets.find(f);
			comp[f] = r;
			if (r == f) roots_of[t].push_back(f);
		}
	});
	uint32_t ncomp_count = 0;
	for (unsigned t = 0; t < n_threads; ++t) { for (uint32_t f : roots_of[t]) sets.parent[f].store(ncomp_count++, std::memory_order_relaxed); }
	const uint32_t ncomp = A.ncomp = ncomp_count;
	parallel_for(n_threads, [&](unsigned t) {
		uint32_t b, e; split(nf, t, b, e);
		for (uint32_t f = b; f < e; ++f) if (comp[f] != NONE32) comp[f] = sets.parent[comp[f]].load(std::memory_order_relaxed);
	});
	mark("components labelled");
	// (b) first face of every component in the start-face sequence, and the coding order
	const bool seeded = !m.shard.seeds.empty();
	BigVec<Gone> no_gone;   // StartFaces wants a reference; only its order is used here
	StartFaces seq(nf, no_gone);
	if (!seeded) { seq.derive_order(); seq.index_blocks(); }
	std::unique_ptr<std::atomic<uint64_t>[]> first_key(new std::atomic<uint64_t>[ncomp]);
	std::unique_ptr<std::atomic<uint32_t>[]> nfaces(new std::atomic<uint32_t>[ncomp]), nhe(new std::atomic<uint32_t>[ncomp]);
	std::unique_ptr<std::atomic<uint32_t>[]> flo(new std::atomic<uint32_t>[ncomp]), fhi(new std::atomic<uint32_t>[ncomp]);   // the interval of face indices a component lies in
	for (uint32_t c = 0; c < ncomp; ++c) {
		first_key[c].store(~0ull, std::memory_order_relaxed); nfaces[c].store(0, std::memory_order_relaxed); nhe[c].store(0, std::memory_order_relaxed);
		flo[c].store(NONE32, std::memory_order_relaxed); fhi[c].store(0, std::memory_order_relaxed);
	}
	parallel_for(n_threads, [&](unsigned t) {
		uint32_t b, e; split(nf, t, b, e);
		uint32_t run_c = NONE32, run_n = 0, run_he = 0, run_first = 0, run_last = 0;
		uint64_t run_min = ~0ull;
		size_t span_at = 0;
		auto flush = [&] {
			if (run_c == NONE32) return;
			if (!seeded) atomic_min(first_key[run_c], run_min);
			nfaces[run_c].fetch_add(run_n, std::memory_order_relaxed);
			nhe[run_c].fetch_add(run_he, std::memory_order_relaxed);
			atomic_min(flo[run_c], run_first); atomic_max(fhi[run_c], run_last + 1);
		};
		for (uint32_t f = b; f < e; ++f) {
			if (is_gone(f)) continue;
			uint32_t c = comp[f];
			if (c != run_c) { flush(); run_c = c; run_n = 0; run_he = 0; run_min = ~0ull; run_first = f; }
			run_last = f;
			++run_n;
			run_he += foff[f + 1] - foff[f];
			// the reference takes face 0 first whatever the set's order (writer.cc:40-46)
			if (!seeded) run_min = std::min<uint64_t>(run_min, f == 0 ? 0ull : ((((uint64_t)seq.position_from(f, span_at) + 1) << 32) | f));
		}
		flush();
	});
	if (seeded) {
		// a shard brings the coding order of its components along: one start face per component
		const std::vector<uint32_t> &sd = m.shard.seeds;
		for (size_t i = 0; i < sd.size(); ++i) {
			if (sd[i] >= nf) throw Error(HRY_E_ARG, "shard seed face out of range");
			if (is_gone(sd[i])) continue;
			const uint32_t c = comp[sd[i]];
			if (first_key[c].load(std::memory_order_relaxed) != ~0ull) throw Error(HRY_E_ARG, "shard: two seed faces in one component");
			first_key[c].store(((uint64_t)(i + 1) << 32) | sd[i], std::memory_order_relaxed);
		}
		for (uint32_t c = 0; c < ncomp; ++c) if (first_key[c].load(std::memory_order_relaxed) == ~0ull) throw Error(HRY_E_ARG, "shard: component without a seed face");
	}
	std::vector<uint32_t> &by_rank = A.by_rank, &rank_of = A.rank_of;
	by_rank.resize(ncomp); rank_of.resize(ncomp);
	for (uint32_t c = 0; c < ncomp; ++c) by_rank[c] = c;
	std::sort(by_rank.begin(), by_rank.end(), [&](uint32_t x, uint32_t y) { return first_key[x].load(std::memory_order_relaxed) < first_key[y].load(std::memory_order_relaxed); });
	for (uint32_t k = 0; k < ncomp; ++k) rank_of[by_rank[k]] = k;
	A.seed.resize(ncomp); A.n_faces.resize(ncomp); A.n_halfedges.resize(ncomp);
	for (uint32_t k = 0; k < ncomp; ++k) {
		const uint32_t c = by_rank[k];
		A.seed[k] = (uint32_t)first_key[c].load(std::memory_order_relaxed);
		A.n_faces[k] = nfaces[c].load(std::memory_order_relaxed);
		A.n_halfedges[k] = nhe[c].load(std::memory_order_relaxed);
	}
	A.face_lo.resize(ncomp); A.face_hi.resize(ncomp);
	for (uint32_t k = 0; k < ncomp; ++k) { A.face_lo[k] = flo[by_rank[k]].load(std::memory_order_relaxed); A.face_hi[k] = fhi[by_rank[k]].load(std::memory_order_relaxed); }
	mark("coding order");
	// (c) the first remaining component (in coding order) that touches each vertex: it introduces the vertex unless the part
	// walked before already transmitted it.  Components that touch a common vertex are tied together: the vertex's index,
	// its triangle count (operation class) and its border count make the later one depend on the earlier one.
	// One pass in face order (a face's rank is looked up once; no half-edge -> face table, no second sweep): every corner
	// takes part in an atomic minimum on its vertex' word.  Whoever finds another component's rank there -- above its own (it
	// replaces it) or below (it leaves it) -- notes the pair: every component at a vertex except the first to arrive notes
	// one with a component that was there before it, so the notes connect all of them.
	AtomicArray vfirst(nv);
	parallel_for(n_threads, [&](unsigned t) { uint32_t b, e; split(nv, t, b, e); for (uint32_t v = b; v < e; ++v) vfirst[v].store(NONE32, std::memory_order_relaxed); });
	mark("  vertex words initialised");
	std::vector<std::vector<std::pair<uint32_t, uint32_t>>> tie_notes(n_threads);
	parallel_for(n_threads, [&](unsigned t) {
		uint32_t b, e; split(nf, t, b, e);
		std::vector<std::pair<uint32_t, uint32_t>> &notes = tie_notes[t];
		uint32_t last_c = NONE32, k = 0;
		for (uint32_t f = b; f < e; ++f) {
			if (is_gone(f)) continue;
			const uint32_t c = comp[f];
			if (c != last_c) { k = rank_of[c]; last_c = c; }
			for (uint32_t h = foff[f], he = foff[f + 1]; h < he; ++h) {
				std::atomic<uint32_t> &a = vfirst[org[h]];
				uint32_t cur = a.load(std::memory_order_relaxed);
				if (cur == k) continue;
				while (k < cur && !a.compare_exchange_weak(cur, k, std::memory_order_relaxed)) {}
				// cur: what was there when this corner settled (k < cur: replaced by k; k > cur: stays)
				if (cur != NONE32 && cur != k && (notes.empty() || notes.back() != std::make_pair(cur, k))) notes.push_back({ cur, k });
			}
		}
	});
	mark("  corners");
	AtomicSets ties(ncomp);
	for (uint32_t k = 0; k < ncomp; ++k) ties.parent[k].store(k, std::memory_order_relaxed);
	for (const auto &notes : tie_notes) for (const auto &pr : notes) ties.unite(pr.first, pr.second);
	std::unique_ptr<std::atomic<uint32_t>[]> fresh(new std::atomic<uint32_t>[ncomp]), vlo(new std::atomic<uint32_t>[ncomp]), vhi(new std::atomic<uint32_t>[ncomp]);
	for (uint32_t k = 0; k < ncomp; ++k) { fresh[k].store(0, std::memory_order_relaxed); vlo[k].store(NONE32, std::memory_order_relaxed); vhi[k].store(0, std::memory_order_relaxed); }
	parallel_for(n_threads, [&](unsigned t) {
		uint32_t b, e; split(nv, t, b, e);
		uint32_t run_k = NONE32, run_n = 0, run_first = 0, run_last = 0;
		auto flush = [&] {
			if (run_k == NONE32) return;
			fresh[run_k].fetch_add(run_n, std::memory_order_relaxed);
			atomic_min(vlo[run_k], run_first); atomic_max(vhi[run_k], run_last + 1);   // the interval of vertex indices the component introduces
		};
		for (uint32_t v = b; v < e; ++v) {
			uint32_t first = vfirst[v].load(std::memory_order_relaxed);
			if (first == NONE32 || (sent && sent[v] != NONE32)) continue;
			if (first != run_k) { flush(); run_k = first; run_n = 0; run_first = v; }
			run_last = v;
			++run_n;
		}
		flush();
	});
	mark("  vertices counted");
	A.fresh.resize(ncomp); A.group.resize(ncomp); A.vtx_lo.resize(ncomp); A.vtx_hi.resize(ncomp);
	for (uint32_t k = 0; k < ncomp; ++k) {
		A.fresh[k] = fresh[k].load(std::memory_order_relaxed); A.group[k] = ties.find(k);   // a root is the smallest rank of its group
		A.vtx_lo[k] = vlo[k].load(std::memory_order_relaxed); A.vtx_hi[k] = vhi[k].load(std::memory_order_relaxed);
	}
	if (A.want_vertex_owner) {
		A.vertex_owner.resize(nv);
		parallel_for(n_threads, [&](unsigned t) { uint32_t b, e; split(nv, t, b, e); for (uint32_t v = b; v < e; ++v) A.vertex_owner[v] = vfirst[v].load(std::memory_order_relaxed); });
	}
	mark("vertex bases and groups");
}

// Everything after the first component, on several threads (see analyse_impl); components of one group stay in coding
// order on one thread.
template <int DEG>
static void walk_rest_parallel(Mesh &m, WalkState &st, const uint32_t *eface_tab, Emitter &em0, uint32_t first_id, unsigned n_threads)
{
	const bool trace = getenv("HRY_TRACE") != nullptr;
	auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry walk] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	ComponentAnalysis A;
	const ShardInfo &sh = m.shard;
	const uint32_t shc = (uint32_t)sh.comp_faces.size();
	if (shc >= 2 && sh.seeds.size() == shc && sh.comp_halfedges.size() == shc && sh.comp_fresh.size() == shc && sh.comp_group.size() == shc) {
		// a shard brings its components along (host/shard.cpp): labels, sizes, the vertices each one introduces, the ties -- the planner
		// found them on the whole mesh.  The first one has just been walked; the others keep their order.
		A.ncomp = shc - 1;   // (the walks below need the seeds, the sizes and the ties, not the per-face labels)
		A.by_rank.resize(A.ncomp); A.rank_of.resize(A.ncomp);
		A.seed.resize(A.ncomp); A.n_faces.resize(A.ncomp); A.n_halfedges.resize(A.ncomp); A.fresh.resize(A.ncomp); A.group.resize(A.ncomp);
		uint32_t first_of_group0 = NONE32;   // the first component's group loses its root: the smallest rank left takes over
		for (uint32_t k = 0; k < A.ncomp; ++k) {
			A.by_rank[k] = A.rank_of[k] = k;
			A.seed[k] = sh.seeds[k + 1]; A.n_faces[k] = sh.comp_faces[k + 1]; A.n_halfedges[k] = sh.comp_halfedges[k + 1]; A.fresh[k] = sh.comp_fresh[k + 1];
			const uint32_t g = sh.comp_group[k + 1];
			if (g > k + 1) throw Error(HRY_E_ARG, "shard: a group's root comes after its member");
			if (g == 0) { if (first_of_group0 == NONE32) first_of_group0 = k; A.group[k] = first_of_group0; }
			else A.group[k] = g - 1;
		}
		mark("components taken from the shard's plan");
	} else
	analyse_impl<DEG>(m, eface_tab, (const uint8_t*)st.gone.data(), st.sent.data(), n_threads, A);
	walk_components_parallel<DEG>(m, st, eface_tab, em0, first_id, n_threads, A);
}

// The components A lists (coding order: seed face, faces, half-edges, new vertices, group of every one), walked on several
// threads into em0's WalkResult behind what it holds already; components of one group stay in coding order on one thread.
template <int DEG>
static void walk_components_parallel(Mesh &m, WalkState &st, const uint32_t *eface_tab, Emitter &em0, uint32_t first_id, unsigned n_threads, const ComponentAnalysis &A)
{
	WalkResult &w = em0.w;
	const bool trace = getenv("HRY_TRACE") != nullptr;
	auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry walk] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	const uint32_t ncomp = A.ncomp;
	const std::vector<uint32_t> &group_of = A.group;
	std::vector<uint32_t> id_base(ncomp + 1);
	id_base[0] = first_id;
	for (uint32_t k = 0; k < ncomp; ++k) id_base[k + 1] = id_base[k] + A.fresh[k];
	// work items: groups of tied components (ascending rank inside a group), largest groups first
	std::vector<uint32_t> order(ncomp);
	for (uint32_t k = 0; k < ncomp; ++k) order[k] = k;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return group_of[x] < group_of[y]; });   // ranks stay ascending inside a group
	struct Item { uint32_t begin, end; uint64_t faces; };
	std::vector<Item> items;
	for (uint32_t i = 0; i < ncomp;) {
		uint32_t j = i;
		uint64_t nfc = 0;
		while (j < ncomp && group_of[order[j]] == group_of[order[i]]) { nfc += A.n_faces[order[j]]; ++j; }
		items.push_back(Item{ i, j, nfc });
		i = j;
	}
	std::sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.faces > y.faces; });
	if (trace) fprintf(stderr, "[hry walk] %u components in %zu groups, largest group %llu faces\n", ncomp, items.size(), items.empty() ? 0ull : (unsigned long long)items[0].faces);
	// ---- the walks.  A component's place in the coded vertices, faces and half-edges is known before it is walked (the
	// vertices it introduces, its faces: exclusive scans in coding order), so every walk writes its part of order_v / order_f
	// where it belongs.  What is not known in advance -- how many operations a component takes (a triangle each + the border
	// operations), the rare symbol groups, explicitly named vertices -- goes into the walking thread's own arrays, component
	// after component, and is put in place afterwards.  (Until round 4 every component had a WalkResult of its own, copied and
	// released afterwards: 150 000 of them on the configs[3] mesh, 140 of the 430 ms of its host walk.)
	std::vector<uint64_t> off_v(ncomp + 1), off_f(ncomp + 1), off_he(ncomp + 1);
	off_v[0] = w.order_v.size(); off_f[0] = w.order_f.size(); off_he[0] = em0.halfedges;
	for (uint32_t k = 0; k < ncomp; ++k) { off_v[k + 1] = off_v[k] + A.fresh[k]; off_f[k + 1] = off_f[k] + A.n_faces[k]; off_he[k + 1] = off_he[k] + A.n_halfedges[k]; }
	w.order_v.resize(off_v[ncomp]); w.order_f.resize(off_f[ncomp]);
	// the polygons' triangle counts: one per coded face, so they have their place like order_f (positions: thread-local, moved below)
	const bool generic = getenv("HRY_GENERIC_WALK") != nullptr;
	const size_t nt0 = w.grp_val[G_NUMTRI].size();
	const bool nt_pos = w.numtri_coded && w.numtri_positions;
	if (w.numtri_coded) { w.grp_val[G_NUMTRI].resize(nt0 + (off_f[ncomp] - off_f[0])); if (nt_pos) w.grp_pos[G_NUMTRI].resize(nt0 + (off_f[ncomp] - off_f[0])); }
	struct Piece {   // what component k left in its thread's arrays
		const OpByte *ops; uint32_t n_ops, thread, mark, sym0, n_syms, named0, n_named, snap0, n_snap;
		uint32_t g0[G_COUNT], gn[G_COUNT], n_op[8];
	};
	std::vector<Piece> piece(ncomp);
	struct PerThread {
		WalkResult w;                               // rare groups, marks, named vertices of every component this thread walked
		std::vector<BigVec<OpByte>> blocks;         // operation bytes, a component's in one piece
		OpByte *cur = nullptr; size_t room = 0;
	};
	std::vector<PerThread> per_thread(n_threads);
	std::atomic<size_t> next_item{ 0 };
	WalkProgress *const progress = off_v[0] == 0 && off_f[0] == 0 && nt0 == 0 ? w.progress : nullptr;   // (positions = indices of the arrays)
	if (progress) progress->begin(w.order_v.data(), w.order_v.size(), w.order_f.data(), w.order_f.size(), w.numtri_coded ? w.grp_val[G_NUMTRI].data() : nullptr);
	const uint32_t *const twin_now = m.twin.data();
	parallel_for(n_threads, [&](unsigned t) {
		Border cb(st.on);
		PerThread &T = per_thread[t];
		T.w.numtri_coded = w.numtri_coded;
		T.w.snapshot_faces = w.snapshot_faces;
		Emitter em(T.w);
		em.eval_model = false;
		em.ov_begin = w.order_v.data(); em.of_begin = w.order_f.data();
		std::vector<uint32_t> runs_v, runs_f, pairs;   // (progress) what the finished group has coded
		for (;;) {
			size_t it = next_item.fetch_add(1, std::memory_order_relaxed);
			if (it >= items.size()) break;
			const size_t patches0 = T.w.twin_patches.size();
			for (uint32_t q = items[it].begin; q < items[it].end; ++q) {
				const uint32_t k = order[q];
				const uint32_t nfc = A.n_faces[k];
				const size_t cap = (size_t)2 * A.n_halfedges[k] - 2 * (size_t)nfc + 16;
				if (T.room < cap) {
					T.blocks.emplace_back();
					T.blocks.back().resize(std::max<size_t>(cap, (size_t)4 << 20));
					T.cur = T.blocks.back().data(); T.room = T.blocks.back().size();
				}
				Piece &pc = piece[k];
				pc.thread = t; pc.mark = (uint32_t)T.w.marks.size(); pc.sym0 = em.n; pc.named0 = (uint32_t)T.w.named.size(); pc.ops = T.cur;
				pc.snap0 = (uint32_t)T.w.snapshots.size();
				for (int g = 0; g < G_COUNT; ++g) pc.g0[g] = (uint32_t)T.w.grp_val[g].size();
				uint32_t nop0[8];
				for (int i = 0; i < 8; ++i) nop0[i] = em.n_op[i];
				em.op_begin = em.op_cur = T.cur;
				em.ov_cur = em.ov_begin + off_v[k]; em.of_cur = em.of_begin + off_f[k];
				if (w.numtri_coded) {
					em.nt_begin = w.grp_val[G_NUMTRI].data(); em.nt_cur = em.nt_begin + nt0 + (off_f[k] - off_f[0]);
					if (nt_pos) { em.ntp_begin = w.grp_pos[G_NUMTRI].data(); em.ntp_cur = em.ntp_begin + nt0 + (off_f[k] - off_f[0]); }
				}
				em.halfedges = (uint32_t)off_he[k];
				uint32_t next_id = id_base[k], consumed = 0;
				if (DEG == 3 && !generic) walk_component_tri<false>(m, st, A.seed[k], cb, em, next_id, consumed);
				else if (!generic) walk_component_poly<DEG>(m, st, eface_tab, A.seed[k], cb, em, next_id, consumed);
				else walk_component<DEG>(m, st, eface_tab, A.seed[k], cb, em, next_id, consumed);
				if (next_id != id_base[k + 1] || consumed != nfc || em.ov_cur != em.ov_begin + off_v[k + 1] || em.of_cur != em.of_begin + off_f[k + 1] ||
				    em.halfedges != (uint32_t)off_he[k + 1] || (size_t)(em.op_cur - T.cur) > cap ||
				    (w.numtri_coded && em.nt_cur != em.nt_begin + nt0 + (off_f[k + 1] - off_f[0])))
					throw WalkMismatch();
				pc.n_ops = (uint32_t)(em.op_cur - T.cur);
				T.room -= pc.n_ops; T.cur = em.op_cur;
				pc.n_syms = em.n - pc.sym0; pc.n_named = (uint32_t)T.w.named.size() - pc.named0; pc.n_snap = (uint32_t)T.w.snapshots.size() - pc.snap0;
				for (int g = 0; g < G_COUNT; ++g) pc.gn[g] = (uint32_t)T.w.grp_val[g].size() - pc.g0[g];
				for (int i = 0; i < 8; ++i) pc.n_op[i] = em.n_op[i] - nop0[i];
			}
			if (progress) {
				// the group's components in ascending rank: consecutive ranks lie next to each other in order_v / order_f
				runs_v.clear(); runs_f.clear(); pairs.clear();
				auto add = [](std::vector<uint32_t> &r, uint64_t b, uint64_t e) {
					if (b == e) return;
					if (!r.empty() && (uint64_t)r[r.size() - 2] + r.back() == b) r.back() += (uint32_t)(e - b);
					else { r.push_back((uint32_t)b); r.push_back((uint32_t)(e - b)); }
				};
				for (uint32_t q = items[it].begin; q < items[it].end; ++q) { const uint32_t k = order[q]; add(runs_v, off_v[k], off_v[k + 1]); add(runs_f, off_f[k], off_f[k + 1]); }
				for (size_t i = patches0; i < T.w.twin_patches.size(); ++i) { const uint32_t h = T.w.twin_patches[i]; pairs.push_back(h); pairs.push_back(twin_now[h]); }
				progress->group_done(runs_v.data(), (uint32_t)(runs_v.size() / 2), runs_f.data(), (uint32_t)(runs_f.size() / 2), pairs.data(), (uint32_t)(pairs.size() / 2));
			}
		}
		em.op_begin = em.op_cur = nullptr;
		em.nt_begin = em.nt_cur = em.ntp_begin = em.ntp_cur = nullptr;
		em.finish_marks();
	});
	mark("walks");
	// ---- the operations and the rare groups into coding order
	std::vector<uint64_t> off_sym(ncomp + 1), off_op(ncomp + 1), off_g[G_COUNT];
	for (int g = 0; g < G_COUNT; ++g) off_g[g].resize(ncomp + 1);
	off_sym[0] = em0.n; off_op[0] = w.op_sc.size();
	for (int g = 0; g < G_COUNT; ++g) off_g[g][0] = g == G_NUMTRI ? nt0 : w.grp_val[g].size();
	for (uint32_t k = 0; k < ncomp; ++k) {
		off_sym[k + 1] = off_sym[k] + piece[k].n_syms;
		off_op[k + 1] = off_op[k] + piece[k].n_ops;
		for (int g = 0; g < G_COUNT; ++g) off_g[g][k + 1] = off_g[g][k] + piece[k].gn[g];
		if (w.numtri_coded) off_g[G_NUMTRI][k + 1] = off_g[G_NUMTRI][k] + A.n_faces[k];   // (in place already)
	}
	if (off_sym[ncomp] + 1 >= (1ull << 32)) throw Error(HRY_E_UNSUPPORTED, "more than 2^32 connectivity symbols");
	w.op_sc.resize(off_op[ncomp]);
	for (int g = 0; g < G_COUNT; ++g) if (g != G_NUMTRI) { w.grp_val[g].resize(off_g[g][ncomp]); w.grp_pos[g].resize(off_g[g][ncomp]); }
	// (the marks -- one thread's loop over the components, 4 ms for the 151 741 of the configs[3] mesh -- beside the copies)
	std::exception_ptr marks_failed;
	auto marks_in_order = [&] {
		for (const PerThread &T : per_thread) {
			if (T.w.twins_changed) w.twins_changed = true;
			w.twin_patches.insert(w.twin_patches.end(), T.w.twin_patches.begin(), T.w.twin_patches.end());
		}
		// marks: one per component, with the counts of everything coded before it in the sequence
		em0.finish_marks();
		{
			uint32_t nop[8];
			for (int i = 0; i < 8; ++i) nop[i] = em0.n_op[i];
			const size_t mark0 = w.marks.size();
			w.marks.resize(mark0 + ncomp);
			size_t n_named = w.named.size();
			for (uint32_t k = 0; k < ncomp; ++k) n_named += piece[k].n_named;
			w.named.reserve(n_named);
			for (uint32_t k = 0; k < ncomp; ++k) {
				const Piece &pc = piece[k];
				const WalkResult &tw = per_thread[pc.thread].w;
				for (uint32_t i = 0; i < pc.n_named; ++i) { NamedVertex ev = tw.named[pc.named0 + i]; ev.mark = (uint32_t)(mark0 + k); w.named.push_back(ev); }
				// (border snapshots: relative to their component's mark until finish_snapshots; this loop is the only writer of both)
				for (uint32_t i = 0; i < pc.n_snap; ++i) { w.snapshots.push_back(std::move(per_thread[pc.thread].w.snapshots[pc.snap0 + i])); w.snapshots.back().mark = (uint32_t)(mark0 + k); }
				ComponentMark mk = tw.marks.at(pc.mark);   // first_vertex and min_ref are the walk's
				for (int g = 0; g < G_COUNT; ++g) mk.n_grp[g] = (uint32_t)off_g[g][k];
				for (int i = 0; i < 8; ++i) { mk.n_op[i] = nop[i]; nop[i] += pc.n_op[i]; }
				mk.first_face = (uint32_t)off_f[k];
				mk.first_halfedge = (uint32_t)off_he[k];
				w.marks[mark0 + k] = mk;
			}
			em0.halfedges = (uint32_t)off_he[ncomp];
			for (int i = 0; i < 8; ++i) em0.n_op[i] = nop[i];
		}
	};
	std::thread marks_thread;
	if (ncomp >= (getenv("HRY_PARALLEL_MIN_FACES") ? 1u : 4096u) && n_threads > 1) marks_thread =   // (the tests' switch for "small inputs on threads too")
		 std::thread([&] { try { marks_in_order(); } catch (...) { marks_failed = std::current_exception(); } });
	try {
		parallel_for(n_threads, [&](unsigned t) {
			// ranges of components with about the same number of operation bytes each
			const uint64_t lo = off_op[0] + (off_op[ncomp] - off_op[0]) * t / n_threads, hi = off_op[0] + (off_op[ncomp] - off_op[0]) * (t + 1) / n_threads;
			uint32_t kb = (uint32_t)(std::lower_bound(off_op.begin(), off_op.begin() + ncomp, lo) - off_op.begin());
			uint32_t ke = t + 1 == n_threads ? ncomp : (uint32_t)(std::lower_bound(off_op.begin(), off_op.begin() + ncomp, hi) - off_op.begin());
			for (uint32_t k = kb; k < ke; ++k) {
				const Piece &pc = piece[k];
				if (pc.n_ops) memcpy(w.op_sc.data() + off_op[k], pc.ops, pc.n_ops);
				const WalkResult &tw = per_thread[pc.thread].w;
				const uint32_t add = (uint32_t)off_sym[k] - pc.sym0;   // (modulo 2^32: thread-local position -> position in the sequence)
				if (nt_pos) { uint32_t *q = w.grp_pos[G_NUMTRI].data() + off_g[G_NUMTRI][k]; for (uint32_t i = 0, nn = A.n_faces[k]; i < nn; ++i) q[i] += add; }
				for (int g = 0; g < G_COUNT; ++g) {
					if (!pc.gn[g]) continue;
					memcpy(w.grp_val[g].data() + off_g[g][k], tw.grp_val[g].data() + pc.g0[g], (size_t)pc.gn[g] * 4);
					uint32_t *dst = w.grp_pos[g].data() + off_g[g][k];
					const uint32_t *src = tw.grp_pos[g].data() + pc.g0[g];
					for (uint32_t i = 0; i < pc.gn[g]; ++i) dst[i] = src[i] + add;
				}
			}
		});
	} catch (...) { if (marks_thread.joinable()) marks_thread.join(); throw; }
	mark("operations and rare groups in coding order");
	if (marks_thread.joinable()) marks_thread.join(); else marks_in_order();
	if (marks_failed) std::rethrow_exception(marks_failed);
	em0.n = (uint32_t)off_sym[ncomp];
	em0.min_ref = w.marks.empty() ? NONE32 : w.marks.back().min_ref;   // the caller's finish_marks() writes it back into the last mark
	mark("marks in coding order");
}

template <int DEG>
static void walk_impl(Mesh &m, WalkResult &w, bool eval_op_model, bool one_sequence)
{
	const uint32_t nf = m.nf;
	if (nf == 0) throw Error(HRY_E_UNSUPPORTED, "mesh without faces");
	BigVec<uint32_t> eface_tab;   // (pooled, not value-initialised: 4 bytes per half-edge, every entry written below)
	if (DEG == 0) {
		eface_tab.resize(m.ne());
		const unsigned nt = nf >= (1u << 20) ? host_threads() : 1u;
		parallel_for(nt, [&](unsigned t) {
			const uint32_t b = (uint32_t)((uint64_t)nf * t / nt), e = (uint32_t)((uint64_t)nf * (t + 1) / nt);
			for (uint32_t f = b; f < e; ++f) for (uint32_t h = m.face_off[f]; h < m.face_off[f + 1]; ++h) eface_tab[h] = f;
		});
	}
	int ndeg = 0;
	for (uint8_t d : m.have_degree) ndeg += d ? 1 : 0;
	w.numtri_coded = ndeg > 1;   // one degree => conn_numtri holds a single symbol of count == total: l = 0, h = t, coder state unchanged
	if (eval_op_model) { w.op_l.reserve(m.ntri() + 16); w.op_h.reserve(m.ntri() + 16); w.op_t.reserve(m.ntri() + 16); w.op_pos.reserve(m.ntri() + 16); }
	walk_sequential<DEG>(m, w, eface_tab.data(), eval_op_model, eval_op_model || one_sequence ? 1u : host_threads());
}

}   // namespace

uint32_t snapshot_spacing(uint32_t nf)
{
	if (getenv("HRY_NO_SNAPSHOTS")) return 0;
	if (const char *e = getenv("HRY_SNAPSHOT_FACES")) return (uint32_t)strtoul(e, nullptr, 10);
	uint32_t sp = kSnapshotMinFaces;
	while ((uint64_t)sp * 64u < nf) sp <<= 1;
	return sp;
}

std::vector<RestartPoint> select_restart_points(const std::vector<ComponentMark> &marks, const std::vector<NamedVertex> &named,
                                                std::vector<RestartCounters> &counters, const std::vector<BorderSnapshot> *snaps,
                                                std::vector<RestartCounters> *snap_counters)
{
	std::vector<RestartPoint> out;
	std::vector<uint32_t> span_of_mark(marks.size(), NONE32);   // restart span of every component (none: before the first point)
	uint32_t last_face = 0;
	for (size_t k = 1; k < marks.size(); ++k) {
		if (marks[k].first_face - last_face < kRestartFaces) {
			if (!out.empty() && marks[k].min_ref < out.back().first_vertex) out.back().flags |= 1u;
			if (!out.empty()) span_of_mark[k] = (uint32_t)out.size() - 1;
			continue;
		}
		RestartPoint r;
		for (int g = 0; g < G_COUNT; ++g) r.n_grp[g] = marks[k].n_grp[g];
		for (int i = 0; i < 8; ++i) r.n_op[i] = marks[k].n_op[i];
		r.first_vertex = marks[k].first_vertex; r.first_face = marks[k].first_face; r.first_halfedge = marks[k].first_halfedge;
		r.flags = marks[k].min_ref < marks[k].first_vertex ? 1u : 0u;
		out.push_back(r);
		span_of_mark[k] = (uint32_t)out.size() - 1;
		last_face = marks[k].first_face;
	}
	// the older vertices a span names, with their counters at the first naming inside the span = at the start of the span (a
	// vertex is touched only after the component at hand has named it).  A span ends where the next point lies, of either kind:
	// a naming behind a border snapshot belongs to the span that starts THERE -- which brings the counters of the vertices on its
	// border along already (they have been counted on since: the naming's counter is not the span's start), so only the others
	// are listed with it.  Points and namings are in stream order; a snapshot's place is (component, how many the component
	// had taken before), a restart point's (component, 0), a naming's (component, snapshots its component had taken).
	counters.assign(out.size(), RestartCounters());
	const size_t n_snaps = snaps ? snaps->size() : 0;
	if (snap_counters) snap_counters->assign(n_snaps, RestartCounters());
	std::vector<uint32_t> nth(n_snaps, 0);   // a snapshot's number inside its component, from 1
	for (size_t i = 0; i < n_snaps; ++i) nth[i] = i && (*snaps)[i - 1].mark == (*snaps)[i].mark ? nth[i - 1] + 1 : 1u;
	std::vector<std::unordered_set<uint32_t>> on_border(n_snaps);   // (filled when a snapshot's span names its first older vertex)
	std::unordered_set<uint64_t> taken, taken_snap;
	size_t si = 0;   // snapshots at or before the naming in hand
	for (const NamedVertex &ev : named) {
		if (ev.mark >= marks.size()) continue;
		while (si < n_snaps && ((*snaps)[si].mark < ev.mark || ((*snaps)[si].mark == ev.mark && nth[si] <= ev.snap))) ++si;
		const uint32_t sp = span_of_mark[ev.mark];
		// the restart point's component: the first component of its span
		const bool snap_later = si > 0 && (sp == NONE32 || (*snaps)[si - 1].first_face > out[sp].first_face);
		if (snap_later) {
			const size_t q = si - 1;
			const BorderSnapshot &S = (*snaps)[q];
			if (!snap_counters || ev.id >= S.first_vertex) continue;
			if (on_border[q].empty()) on_border[q].insert(S.vtx.begin(), S.vtx.end());
			if (on_border[q].count(ev.id)) continue;
			if (taken_snap.insert(((uint64_t)q << 32) | ev.id).second) (*snap_counters)[q].push_back({ ev.id, ev.count });
			continue;
		}
		if (sp == NONE32 || ev.id >= out[sp].first_vertex) continue;
		if (taken.insert(((uint64_t)sp << 32) | ev.id).second) counters[sp].push_back({ ev.id, ev.count });
	}
	return out;
}

// the start-face sequence of a mesh of nf faces as spans sorted by their lowest face: (lowest, highest, position of the span's first
// face in the sequence, 1 = ascending) -- what a face's place in the coding order is computed from (StartFaces::position)
void start_face_spans(uint32_t nf, std::vector<uint32_t> &spans)
{
	BigVec<Gone> no_gone;
	StartFaces seq(nf, no_gone);
	seq.derive_order(); seq.index_blocks();
	spans.clear();
	for (const StartFaces::Span &sp : seq.spans) { spans.push_back(sp.lo); spans.push_back(sp.hi); spans.push_back(sp.first_pos); spans.push_back(sp.asc ? 1u : 0u); }
}

void analyse_components(const Mesh &m, ComponentAnalysis &A)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	ensure_twins(m);
	int udeg = 0;
	if (!m.uniform_degree(udeg)) udeg = 0;
	const unsigned nt = m.nf >= (1u << 16) ? host_threads() : 1u;
	if (udeg == 3) { analyse_impl<3>(m, nullptr, nullptr, nullptr, nt, A); return; }
	if (udeg == 4) { analyse_impl<4>(m, nullptr, nullptr, nullptr, nt, A); return; }
	BigVec<uint32_t> &eface_tab = A.eface;   // (pooled, not value-initialised: every entry is written below)
	eface_tab.resize(m.ne());
	parallel_for(nt, [&](unsigned t) {
		const uint32_t b = (uint32_t)((uint64_t)m.nf * t / nt), e = (uint32_t)((uint64_t)m.nf * (t + 1) / nt);
		for (uint32_t f = b; f < e; ++f) for (uint32_t h = m.face_off[f]; h < m.face_off[f + 1]; ++h) eface_tab[h] = f;
	});
	analyse_impl<0>(m, eface_tab.data(), nullptr, nullptr, nt, A);
}

WalkState::WalkState(uint32_t nv, uint32_t nf, unsigned n_threads)
{
	gone.resize(nf); on.resize(nv); sent.resize(nv); seen.resize(nv);
	parallel_for(std::max(1u, n_threads), [&](unsigned t) {
		const unsigned nt = std::max(1u, n_threads);
		const size_t fb = (size_t)nf * t / nt, fe = (size_t)nf * (t + 1) / nt, vb = (size_t)nv * t / nt, ve = (size_t)nv * (t + 1) / nt;
		if (fe > fb) memset((void*)(gone.data() + fb), 0, fe - fb);
		if (ve > vb) { memset(on.data() + vb, 0, (ve - vb) * sizeof(OnCount)); memset(sent.data() + vb, 0xff, (ve - vb) * 4); memset(seen.data() + vb, 0, (ve - vb) * 2); }
	});
}

void cut_border_walk_in_place(Mesh &m, const ComponentAnalysis &part, const uint32_t *eface, WalkState &st, WalkResult &w)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	if (m.twins_pending) throw Error(HRY_E_ARG, "walk in place: the twins must be matched");
	if (st.gone.size() != m.nf || st.sent.size() != m.nv) throw Error(HRY_E_ARG, "walk in place: marks of another mesh");
	const uint32_t nc = part.ncomp;
	if (part.seed.size() != nc || part.n_faces.size() != nc || part.n_halfedges.size() != nc || part.fresh.size() != nc || part.group.size() != nc)
		throw Error(HRY_E_ARG, "walk in place: incomplete component list");
	for (uint32_t k = 0; k < nc; ++k) if (part.seed[k] >= m.nf || part.group[k] > k) throw Error(HRY_E_ARG, "walk in place: component list out of range");
	int udeg = 0, ndeg = 0;
	if (!m.uniform_degree(udeg)) udeg = 0;
	if (udeg != 3 && udeg != 4) { udeg = 0; if (!eface) throw Error(HRY_E_ARG, "walk in place: mixed polygon degrees need the half-edge -> face table"); }
	for (uint8_t d : m.have_degree) ndeg += d ? 1 : 0;
	w.numtri_coded = ndeg > 1;
	Emitter em(w);
	em.eval_model = false;
	const unsigned nt = host_threads();
	switch (udeg) {
	case 3: walk_components_parallel<3>(m, st, nullptr, em, 0, nt, part); break;
	case 4: walk_components_parallel<4>(m, st, nullptr, em, 0, nt, part); break;
	default: walk_components_parallel<0>(m, st, eface, em, 0, nt, part); break;
	}
	em.finish_marks();
	em.iop(I_EOM);
	w.n_conn = em.n;
	for (int i = 0; i < 8; ++i) w.n_op_class[i] = em.n_op[i];
	finish_snapshots(w);
}

void op_position_table(const WalkResult &w, std::vector<uint32_t> &thr, std::vector<uint32_t> &cum)
{
	// merge of the (sorted) position lists of the group kinds; a group of b bytes at position p has p - (bytes of the groups
	// before it) operations in front of it.  Large tables (one triangle count per polygon: 78 M entries for the configs[3] mesh,
	// 0.3 s on one thread) are merged by the host threads, each the groups of its own range of positions.
	size_t total = 0;
	for (int g = 0; g < G_COUNT; ++g) total += w.grp_pos[g].size();
	thr.resize(total); cum.resize(total);
	const unsigned nt = total >= (getenv("HRY_PARALLEL_MIN_FACES") ? (size_t)parallel_min_faces() : (size_t)1 << 20) ? std::max(1u, host_threads()) : 1u;   // (the tests' switch for "small inputs on threads too")
	uint32_t pmax = 0;
	for (int g = 0; g < G_COUNT; ++g) if (!w.grp_pos[g].empty()) pmax = std::max(pmax, w.grp_pos[g].back());
	parallel_for(nt, [&](unsigned t) {
		// positions [lo, hi) (the last range takes everything that is left)
		const uint64_t lo = ((uint64_t)pmax + 1) * t / nt, hi = t + 1 == nt ? (uint64_t)pmax + 1 : ((uint64_t)pmax + 1) * (t + 1) / nt;
		size_t at[G_COUNT], end[G_COUNT], k = 0;
		uint64_t bytes = 0;
		for (int g = 0; g < G_COUNT; ++g) {
			const BigVec<uint32_t> &p = w.grp_pos[g];
			at[g] = (size_t)(std::lower_bound(p.begin(), p.end(), (uint32_t)lo) - p.begin());
			end[g] = t + 1 == nt ? p.size() : (size_t)(std::lower_bound(p.begin(), p.end(), (uint32_t)hi) - p.begin());
			k += at[g];
			bytes += (uint64_t)at[g] * (uint64_t)kGroupBytes[g];
		}
		for (;;) {
			int best = -1;
			for (int g = 0; g < G_COUNT; ++g)
				if (at[g] < end[g] && (best < 0 || w.grp_pos[g][at[g]] < w.grp_pos[best][at[best]])) best = g;
			if (best < 0) break;
			const uint32_t p = w.grp_pos[best][at[best]++];
			thr[k] = p - (uint32_t)bytes;
			bytes += (uint64_t)kGroupBytes[best];
			cum[k] = (uint32_t)bytes;
			++k;
		}
	});
}

void cut_border_walk(Mesh &m, WalkResult &w, bool eval_op_model, bool one_sequence)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	ensure_twins(m);
	int udeg = 0;
	if (!m.uniform_degree(udeg)) udeg = 0;
	auto run = [&](bool one) {
		switch (udeg) {
		case 3: walk_impl<3>(m, w, eval_op_model, one); break;
		case 4: walk_impl<4>(m, w, eval_op_model, one); break;
		default: walk_impl<0>(m, w, eval_op_model, one); break;   // mixed (or unusual uniform) degrees
		}
	};
	try { run(one_sequence); }
	catch (const WalkMismatch &) {
		// (host.hpp WalkMismatch) the twins as the matching leaves them, a fresh result, one thread: the reference's own order
		if (getenv("HRY_TRACE")) fprintf(stderr, "[hry walk] a repaired twin split a component: the mesh is walked again on one thread\n");
		build_twins(m);
		WalkResult fresh;
		fresh.progress = nullptr;   // (whoever listened to the groups of the first attempt has to start over: see chunked.cpp)
		fresh.numtri_positions = w.numtri_positions; fresh.snapshot_faces = w.snapshot_faces;
		w = std::move(fresh);
		w.twins_changed = true;     // (the caller's copy of the twins, if it has one, is that of the first attempt)
		run(true);
		w.twins_changed = true;
		w.twin_patches.clear();     // (... every entry of it: no list of patches)
	}
}

}   // namespace hry
