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

#include <algorithm>
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
	std::vector<int32_t> spare;
	std::vector<Part> parts;
	std::vector<uint8_t> on;   // how many border elements reference a vertex (cutborder.h:69)

	explicit Border(uint32_t nv) : on(nv, 0) { pool.reserve(1024); }
	Part &top() { return parts.back(); }
	Node &N(int32_t i) { return pool[i]; }

	int32_t make(uint32_t v, uint32_t a)
	{
		int32_t i;
		if (!spare.empty()) { i = spare.back(); spare.pop_back(); }
		else { i = (int32_t)pool.size(); pool.push_back(Node()); }
		pool[i] = Node{ v, a, -1, -1 };
		++on[v];
		return i;
	}
	void drop(int32_t i) { --on[pool[i].v]; spare.push_back(i); }
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
		Part &p = top();
		for (int32_t i = p.head; i >= 0;) { int32_t nx = pool[i].next; drop(i); i = nx; }
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
			if (pool[fw].v == v) { ++i; return fw; }
			if (pool[bw].v == v) { i = -i; return bw; }
			if (bw == fw || pool[bw].next == fw) {
				++p; --pi;
				fw = parts[pi].head; bw = parts[pi].tail;
				i = 0;
			} else { fw = pool[fw].next; bw = pool[bw].prev; ++i; }
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
			int32_t last = pool[hit].prev;
			np.head = old.head; np.tail = last; np.size = before;
			pool[last].next = -1;
			pool[hit].prev = -1;
			old.head = hit;
			old.size -= before;
		}
		append(parts[oi], g);
		copy_node = make(pool[hit].v, pool[hit].a);
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
			pool[other.tail].next = other.head;
			pool[other.head].prev = other.tail;
			int32_t last = pool[hit].prev;
			pool[last].next = -1;
			pool[hit].prev = -1;
			other.head = hit; other.tail = last;
		}
		pool[cur.tail].next = other.head;
		pool[other.head].prev = cur.tail;
		cur.tail = other.tail;
		cur.size += other.size;
		copy_node = make(pool[hit].v, pool[hit].a);
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
	std::vector<uint8_t> gone;
	std::vector<Block> blocks;
	size_t bi = 0;
	uint32_t pos = 0;
	bool have_order = false;
	uint32_t left;
	explicit StartFaces(uint32_t n) : nf(n), gone(n, 0), left(n) {}
	void take(uint32_t f) { gone[f] = 1; --left; }
	void derive_order()
	{
		std::__detail::_Prime_rehash_policy pol;
		std::size_t nbkt = 1;
		std::vector<Block> list;   // front ... back
		for (uint32_t k = 0; k < nf; ++k) {
			std::pair<bool, std::size_t> rh = pol._M_need_rehash(nbkt, k, 1);
			if (rh.first) {
				nbkt = rh.second;
				std::reverse(list.begin(), list.end());
				for (Block &b : list) std::swap(b.first, b.last);
			}
			if (!list.empty() && list.front().first == k - 1 && list.front().first >= list.front().last) list.front().first = k;   // extend the descending front block
			else list.insert(list.begin(), Block{ k, k });
		}
		blocks.swap(list);
		have_order = true;
		bi = 0; pos = 0;
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
	uint32_t next()
	{
		uint32_t f = 0;
		if (gone[0]) {
			if (!have_order) derive_order();
			while (gone[at_cursor()]) advance();
			f = at_cursor();
		}
		take(f);
		return f;
	}
};

struct Emitter {
	WalkResult &w;
	uint32_t n = 0;
	// order-conditioned operation model (models.h:49-120), evaluated here because it is connectivity-sized
	uint64_t plain[5] = { 1, 1, 1, 1, 1 }, c_all = 2, c_new[8], c_fwd[8];
	bool eval_model = true;   // the compat stream needs (l, h, t) of every operation; the chunked planes only symbol + class
	explicit Emitter(WalkResult &r) : w(r) { for (int i = 0; i < 8; ++i) c_new[i] = c_fwd[i] = 1; }
	void group(int g, uint32_t v) { w.grp_val[g].push_back(v); w.grp_pos[g].push_back(n); n += kGroupBytes[g]; }
	void iop(uint32_t s) { group(G_IOP, s); }
	void vert(uint32_t v) { group(G_VERT, v); }
	void elem(int i) { uint32_t c = (uint32_t)i; group(G_ELEM, (c << 1) ^ ((c >> 31) ? 0xffffffffu : 0u)); }   // transform.h:25-30
	void part(int p) { group(G_PART, (uint32_t)(uint16_t)p); }
	void numtri(int nt) { if (nt != 0 && w.numtri_coded) group(G_NUMTRI, (uint32_t)(uint16_t)nt); }          // io.h:162-165
	void op(uint32_t s, int order)
	{
		int k = order - 1;   // models.h:101-105; order >= 1 because the gate's front vertex lies on a coded triangle
		if (k > 7) k = 7;
		if (k < 0) k = 0;
		if (!eval_model) {
			w.op_sym.push_back((uint8_t)s); w.op_class.push_back((uint8_t)k);
			++n;
			return;
		}
		uint64_t nv = c_new[k] * c_all / (c_new[k] + c_fwd[k]);
		uint64_t f[7] = { plain[0], plain[1], plain[2], plain[3], plain[4], nv, c_all - nv };
		uint64_t l = 0;
		for (uint32_t x = 0; x < s; ++x) l += f[x];
		uint64_t t = plain[0] + plain[1] + plain[2] + plain[3] + plain[4] + c_all;
		w.op_sym.push_back((uint8_t)s); w.op_class.push_back((uint8_t)k);
		w.op_l.push_back((uint32_t)l); w.op_h.push_back((uint32_t)(l + f[s])); w.op_t.push_back((uint32_t)t);
		w.op_pos.push_back(n++);
		if (s == O_NEWVTX) { ++c_all; ++c_new[k]; }
		else if (s == O_CONNFWD) { ++c_all; ++c_fwd[k]; }
		else ++plain[s];
	}
};

}   // namespace

// DEG > 0: every polygon has DEG edges and the face of a half-edge is a division by a compile-time constant (the runtime
// division this replaces was a third of the walk); DEG == 0: mixed degrees, table lookup
template <int DEG>
static void walk_impl(Mesh &m, WalkResult &w, bool eval_op_model)
{
	const uint32_t nv = m.nv, nf = m.nf;
	if (nf == 0) throw Error(HRY_E_UNSUPPORTED, "mesh without faces");
	const uint32_t *foff = m.face_off.data();
	const uint32_t *org = m.org.data();
	uint32_t *twin = m.twin.data();
	std::vector<uint32_t> eface_tab;
	if (DEG == 0) {
		eface_tab.resize(m.ne());
		for (uint32_t f = 0; f < nf; ++f) for (uint32_t e = foff[f]; e < foff[f + 1]; ++e) eface_tab[e] = f;
	}
	auto face_of = [&](uint32_t e) -> uint32_t { return DEG ? e / (uint32_t)(DEG ? DEG : 1) : eface_tab[e]; };
	auto nxt = [&](uint32_t e) -> uint32_t {
		if (DEG) { uint32_t k = e % (uint32_t)(DEG ? DEG : 1); return k + 1 == (uint32_t)DEG ? e - k : e + 1; }
		uint32_t f = eface_tab[e];
		return e + 1 == foff[f + 1] ? foff[f] : e + 1;
	};
	auto link = [&](uint32_t a, uint32_t b) { twin[a] = b; twin[b] = a; };

	int ndeg = 0;
	for (uint8_t d : m.have_degree) ndeg += d ? 1 : 0;
	w.numtri_coded = ndeg > 1;   // one degree => conn_numtri holds a single symbol of count == total: l = 0, h = t, coder state unchanged
	w.order_v.reserve(nv);
	w.order_f.reserve(nf);
	w.op_sym.reserve(m.ntri() + 16); w.op_class.reserve(m.ntri() + 16);
	if (eval_op_model) { w.op_l.reserve(m.ntri() + 16); w.op_h.reserve(m.ntri() + 16); w.op_t.reserve(m.ntri() + 16); w.op_pos.reserve(m.ntri() + 16); }

	Border cb(nv);
	StartFaces pool(nf);
	Emitter em(w);
	em.eval_model = eval_op_model;
	std::vector<uint32_t> sent(nv, NONE32);   // original vertex -> transmitted index (encoder.h:28-52)
	std::vector<uint16_t> seen(nv, 0);        // triangles seen per vertex (selects the op model class)
	uint32_t next_id = 0;
	auto record_vertex = [&](uint32_t e) { w.order_v.push_back(e); sent[org[e]] = next_id++; };

	do {
		// ---- start a component (encoder.h:68-131)
		uint32_t f = pool.next();
		uint32_t e0 = foff[f], e1 = nxt(e0), e2 = nxt(e1);
		uint32_t a = org[e0], b = org[e1], c = org[e2];
		int ntri = (int)(foff[f + 1] - foff[f]) - 2, curtri = 1;
		unsigned mask = (sent[a] != NONE32 ? 4u : 0u) | (sent[b] != NONE32 ? 2u : 0u) | (sent[c] != NONE32 ? 1u : 0u);
		switch (mask) {
		case 7: em.iop(I_TRI111); em.vert(sent[a]); em.vert(sent[b]); em.vert(sent[c]); em.numtri(ntri); break;
		case 6: em.iop(I_TRI110); em.vert(sent[a]); em.vert(sent[b]); em.numtri(ntri); record_vertex(e2); break;
		case 3: em.iop(I_TRI011); em.vert(sent[b]); em.vert(sent[c]); em.numtri(ntri); record_vertex(e0); break;
		case 5: em.iop(I_TRI101); em.vert(sent[c]); em.vert(sent[a]); em.numtri(ntri); record_vertex(e1); break;
		case 4: em.iop(I_TRI100); em.vert(sent[a]); em.numtri(ntri); record_vertex(e1); record_vertex(e2); break;
		case 2: em.iop(I_TRI010); em.vert(sent[b]); em.numtri(ntri); record_vertex(e2); record_vertex(e0); break;
		case 1: em.iop(I_TRI001); em.vert(sent[c]); em.numtri(ntri); record_vertex(e0); record_vertex(e1); break;
		default: em.iop(I_INIT); em.numtri(ntri); record_vertex(e0); record_vertex(e1); record_vertex(e2); break;
		}
		w.order_f.push_back(e0);
		++seen[a]; ++seen[b]; ++seen[c];
		cb.start(a, e0, b, e1, c, e2);

		// ---- grow until the border of this component is exhausted (encoder.h:133-214)
		while (!cb.parts.empty()) {
			Border::Part &pt = cb.top();
			const uint32_t v0 = cb.N(pt.tail).v, v1 = cb.N(pt.head).v;
			const uint32_t gate = cb.N(pt.tail).a;
			const uint32_t gateprev = cb.N(cb.N(pt.tail).prev).a;
			const uint32_t gatenext = cb.N(pt.head).a;
			const bool seq_first = curtri == ntri;
			const int order = seen[v1];
			if (seq_first) {
				uint32_t t = twin[gate];
				if (t == gate || pool.gone[face_of(t)]) {   // writer.cc:48-58: mesh border or neighbour already consumed
					Op bop = cb.border();
					if (t != gate) twin[gate] = gate;       // one-sided split (writer.cc:81-84)
					em.op(bop, order);
					continue;
				}
				pool.take(face_of(t));
				e0 = t;
				f = face_of(e0);
				ntri = (int)(foff[f + 1] - foff[f]) - 2;
				curtri = 0;
				e1 = nxt(e0);
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
				else { em.op(O_NM, order); em.vert(sent[v2]); em.numtri(nt); }
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
			if (seq_first) w.order_f.push_back(e0);
			++curtri;
		}
	} while (pool.left != 0);
	em.iop(I_EOM);
	w.n_conn = em.n;
}

void cut_border_walk(Mesh &m, WalkResult &w, bool eval_op_model)
{
	int udeg = 0;
	if (!m.uniform_degree(udeg)) udeg = 0;
	switch (udeg) {
	case 3: walk_impl<3>(m, w, eval_op_model); break;
	case 4: walk_impl<4>(m, w, eval_op_model); break;
	default: walk_impl<0>(m, w, eval_op_model); break;   // mixed (or unusual uniform) degrees
	}
}

}   // namespace hry
