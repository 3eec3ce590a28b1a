// PLY subset reader/writer of the product (host side).
// Behavioural contract: formats/ply/reader.cc:36-429 (property canonicalisation, list 0 = face attributes,
// list 1 = vertex attributes, `vertex_indices` -> polygons) and formats/ply/writer.cc:106-192.
// Half-edge twins are matched with the reference's sequential rule (structs/conn.h:201-214): a directed edge
// (a,b) pairs with a pending (b,a); a second pending (a,b) is dropped.  Implemented here with an open-addressing
// table instead of std::unordered_map.
#include "host.hpp"

#include <atomic>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <unordered_map>

namespace hry {

namespace {

struct Prop { std::string name; CompType type; CompType len_type; };
struct Element { std::string name; long count = 0; std::vector<Prop> props; };

CompType parse_type(const std::string &s)
{
	static const std::unordered_map<std::string, CompType> m = {
		{ "float", C_FLOAT }, { "float32", C_FLOAT }, { "double", C_DOUBLE }, { "float64", C_DOUBLE },
		{ "uint", C_UINT }, { "uint32", C_UINT }, { "int", C_INT }, { "int32", C_INT },
		{ "ushort", C_USHORT }, { "uint16", C_USHORT }, { "short", C_SHORT }, { "int16", C_SHORT },
		{ "uchar", C_UCHAR }, { "uint8", C_UCHAR }, { "char", C_CHAR }, { "int8", C_CHAR } };
	auto it = m.find(s);
	if (it == m.end()) throw Error(HRY_E_FORMAT, "Invalid data type");
	return it->second;
}

// canonical rank of well-known property names and their interpretation group (ply/reader.cc:36-68)
struct Known { const char *name; int rank; int interp; };
const Known kKnown[] = {
	{ "x", 0, 0 }, { "y", 1, 0 }, { "z", 2, 0 }, { "w", 3, 0 }, { "nx", 4, 1 }, { "ny", 5, 1 }, { "nz", 6, 1 }, { "nw", 7, 1 },
	{ "red", 8, 2 }, { "green", 9, 2 }, { "blue", 10, 2 }, { "alpha", 11, 2 },
	{ "ambient_red", 12, 3 }, { "ambient_green", 13, 3 }, { "ambient_blue", 14, 3 }, { "ambient_alpha", 15, 3 }, { "ambient_coeff", 16, 3 },
	{ "diffuse_red", 17, 4 }, { "diffuse_green", 18, 4 }, { "diffuse_blue", 19, 4 }, { "diffuse_alpha", 20, 4 }, { "diffuse_coeff", 21, 4 },
	{ "specular_red", 22, 5 }, { "specular_green", 23, 5 }, { "specular_blue", 24, 5 }, { "specular_alpha", 25, 5 }, { "specular_power", 26, 5 }, { "specular_coeff", 27, 5 },
	{ "u", 28, 16 }, { "tu", 28, 16 }, { "v", 29, 16 }, { "tv", 29, 16 }, { "tw", 30, 16 },
	{ "value", 31, 17 }, { "scale", 31, 17 }, { "confidence", 32, 18 } };
const int kKnownEnd = 33;

const Known *lookup_known(const std::string &n)
{
	for (const Known &k : kKnown) if (n == k.name) return &k;
	return nullptr;
}

}   // namespace

// Arrange scalar properties into an attribute list: well-known names first in canonical order, the rest in file
// order, each unknown name its own named interpretation.  slot_of[i] = component index of property i or -1.
void layout_attr_list(const std::vector<std::string> &names, const std::vector<CompType> &types, const std::vector<bool> &is_list,
                      AttrList &L, std::vector<int> &slot_of)
{
	size_t n = names.size();
	std::vector<int> rank(n, std::numeric_limits<int>::max()), interp(n, -1);
	int next_other = kKnownEnd, scalars = 0;
	for (size_t i = 0; i < n; ++i) {
		if (is_list[i]) continue;
		const Known *k = lookup_known(names[i]);
		rank[i] = k ? k->rank : next_other++;
		interp[i] = k ? k->interp : -1;
		++scalars;
	}
	std::vector<int> order(n);
	for (size_t i = 0; i < n; ++i) order[i] = (int)i;
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rank[a] < rank[b]; });
	slot_of.assign(n, -1);
	int others = 0;
	for (int c = 0; c < scalars; ++c) {
		int p = order[c];
		L.add_comp(types[p]);
		int id = interp[p] >= 0 ? interp[p] : kInterpOther + others++;
		L.add_interp(id, c);
		if (id >= kInterpOther) L.interp_name[id - kInterpOther] = names[p];
		slot_of[p] = c;
	}
	if (L.ncomp() > kMaxComp) throw Error(HRY_E_UNSUPPORTED, "too many components in one attribute list");
}

// ---- twin matching -------------------------------------------------------------------------------------
namespace {
struct EdgeTable {
	std::vector<uint64_t> key;
	std::vector<uint32_t> val;
	uint64_t mask;
	static constexpr uint64_t EMPTY = ~0ull, DEAD = ~0ull - 1;
	explicit EdgeTable(size_t expected)
	{
		size_t cap = 16;
		while (cap < expected * 2) cap <<= 1;
		key.assign(cap, EMPTY);
		val.assign(cap, 0);
		mask = cap - 1;
	}
	static uint64_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
	// returns slot of k or -1
	int64_t find(uint64_t k) const
	{
		for (uint64_t i = mix(k) & mask;; i = (i + 1) & mask) {
			if (key[i] == k) return (int64_t)i;
			if (key[i] == EMPTY) return -1;
		}
	}
	void insert_if_absent(uint64_t k, uint32_t v)
	{
		int64_t dead = -1;
		for (uint64_t i = mix(k) & mask;; i = (i + 1) & mask) {
			if (key[i] == k) return;
			if (key[i] == DEAD && dead < 0) dead = (int64_t)i;
			if (key[i] == EMPTY) {
				uint64_t s = dead >= 0 ? (uint64_t)dead : i;
				key[s] = k; val[s] = v;
				return;
			}
		}
	}
	void erase(int64_t slot) { key[slot] = DEAD; }
};
}   // namespace

// The reference pairs half-edges greedily in face order (structs/conn.h:201-214): a half-edge a->c takes the waiting half-edge
// c->a if there is one, else it waits itself unless an earlier a->c is still waiting.  Only half-edges of the same undirected
// edge ever interact, so the edges can be dealt out to buckets (by their smaller vertex) and every bucket paired
// independently with the very same rule, each on its own host thread -- as long as a bucket sees its half-edges in face
// order, which a stable scatter guarantees.  Result: identical twins, ~10 times sooner on 16 threads (the hash-map pass was
// 100 ms per million triangles: six times the whole encode).
static void pair_bucket(const uint64_t *pairs, size_t n, uint32_t *twin)
{
	// pairs: (directed key a<<32|c, half-edge) as two u64 per entry, in face order
	EdgeTable tab(n);
	for (size_t i = 0; i < n; ++i) {
		const uint64_t k = pairs[2 * i];
		const uint32_t h = (uint32_t)pairs[2 * i + 1];
		const uint64_t opp = (k << 32) | (k >> 32);
		int64_t s = tab.find(opp);
		if (s >= 0) {
			uint32_t o = tab.val[s];
			twin[o] = h; twin[h] = o;
			tab.erase(s);
		} else tab.insert_if_absent(k, h);
	}
}

void ensure_twins(const Mesh &cm)
{
	if (!cm.twins_pending) return;
	Mesh &m = const_cast<Mesh&>(cm);   // logically const: the twins are a function of the connectivity
	build_twins(m);
	m.twins_pending = false;
}

void match_twins_at(Mesh &m, const uint32_t *vertices, uint32_t n)
{
	std::unordered_map<uint32_t, std::vector<uint64_t>> seg;   // hub -> (larger endpoint << 32 | half-edge)
	for (uint32_t i = 0; i < n; ++i) seg[vertices[i]];
	for (uint32_t f = 0; f < m.nf; ++f) {
		const uint32_t b = m.face_off[f], e = m.face_off[f + 1];
		for (uint32_t h = b; h < e; ++h) {
			const uint32_t a = m.org[h], c = m.org[h + 1 == e ? b : h + 1];
			auto it = seg.find(std::min(a, c));
			if (it != seg.end()) it->second.push_back(((uint64_t)std::max(a, c) << 32) | h);
		}
	}
	const uint32_t NONE = 0xffffffffu;
	for (auto &kv : seg) {
		const uint32_t lo = kv.first;
		std::vector<uint64_t> &v = kv.second;
		std::sort(v.begin(), v.end());
		uint32_t cur_hi = NONE, pend_out = NONE, pend_in = NONE;
		for (uint64_t x : v) {   // structs/conn.h:201-214, run by run of equal larger endpoints, in half-edge order
			const uint32_t hi = (uint32_t)(x >> 32), h = (uint32_t)x;
			if (hi != cur_hi) { cur_hi = hi; pend_out = pend_in = NONE; }
			if (hi == lo) {
				if (pend_out != NONE) { m.twin[h] = pend_out; m.twin[pend_out] = h; pend_out = NONE; } else pend_out = h;
				continue;
			}
			const bool out = m.org[h] == lo;
			uint32_t &opposite = out ? pend_in : pend_out, &same = out ? pend_out : pend_in;
			if (opposite != NONE) { m.twin[h] = opposite; m.twin[opposite] = h; opposite = NONE; }
			else if (same == NONE) same = h;
		}
	}
}

void build_twins(Mesh &m)
{
	const uint32_t ne = m.ne(), nf = m.nf;
	m.twin.resize(ne);
	// the matching does not scale past a handful of cores, and what its threads write is read next by the caller's core: keep
	// them inside the caller's last-level cache domain (block_pool.cpp)
	unsigned n_near = 0;
	const void *near = callers_cache_cpus(&n_near);
	const unsigned nt = ne >= (1u << 18) ? (near ? std::min(host_threads(), n_near) : host_threads()) : 1u;
	if (nt < 2) {
		for (uint32_t e = 0; e < ne; ++e) m.twin[e] = e;
		EdgeTable tab(ne);
		for (uint32_t f = 0; f < nf; ++f) {
			uint32_t b = m.face_off[f], e = m.face_off[f + 1];
			for (uint32_t h = b; h < e; ++h) {
				uint32_t a = m.org[h], c = m.org[h + 1 == e ? b : h + 1];
				int64_t s = tab.find(((uint64_t)c << 32) | a);
				if (s >= 0) {
					uint32_t o = tab.val[s];
					m.twin[o] = h; m.twin[h] = o;
					tab.erase(s);
				} else tab.insert_if_absent(((uint64_t)a << 32) | c, h);
			}
		}
		return;
	}
	// buckets by the smaller endpoint: a few per thread so that uneven meshes still balance
	const uint32_t nb = nt * 8;
	const uint64_t nvp = (uint64_t)m.nv + 1;
	auto bucket_of = [&](uint32_t a, uint32_t c) { return (uint32_t)(((uint64_t)std::min(a, c) * nb) / nvp); };
	auto face_range = [&](unsigned t, uint32_t &fb, uint32_t &fe) { fb = (uint32_t)((uint64_t)nf * t / nt); fe = (uint32_t)((uint64_t)nf * (t + 1) / nt); };
	std::vector<uint64_t> count((size_t)nt * nb, 0);
	parallel_for(nt, [&](unsigned t) {
		uint32_t fb, fe; face_range(t, fb, fe);
		uint64_t *cnt = count.data() + (size_t)t * nb;
		for (uint32_t f = fb; f < fe; ++f) {
			uint32_t b = m.face_off[f], e = m.face_off[f + 1];
			for (uint32_t h = b; h < e; ++h) { m.twin[h] = h; ++cnt[bucket_of(m.org[h], m.org[h + 1 == e ? b : h + 1])]; }
		}
	}, near);
	// bucket-major, thread-minor offsets: inside a bucket the threads' (= face ranges') entries follow one another in face order
	std::vector<uint64_t> start((size_t)nt * nb), bucket_begin(nb + 1, 0);
	uint64_t run = 0;
	for (uint32_t k = 0; k < nb; ++k) {
		bucket_begin[k] = run;
		for (unsigned t = 0; t < nt; ++t) { start[(size_t)t * nb + k] = run; run += count[(size_t)t * nb + k]; }
	}
	bucket_begin[nb] = run;
	BigVec<uint64_t> pairs;
	pairs.resize(2 * (size_t)ne);
	parallel_for(nt, [&](unsigned t) {
		uint32_t fb, fe; face_range(t, fb, fe);
		uint64_t *pos = start.data() + (size_t)t * nb;
		for (uint32_t f = fb; f < fe; ++f) {
			uint32_t b = m.face_off[f], e = m.face_off[f + 1];
			for (uint32_t h = b; h < e; ++h) {
				const uint32_t a = m.org[h], c = m.org[h + 1 == e ? b : h + 1];
				const uint64_t p = pos[bucket_of(a, c)]++;
				pairs[2 * p] = ((uint64_t)a << 32) | c;
				pairs[2 * p + 1] = h;
			}
		}
	}, near);
	std::atomic<uint32_t> next{ 0 };
	parallel_for(nt, [&](unsigned) {
		for (;;) {
			const uint32_t k = next.fetch_add(1, std::memory_order_relaxed);
			if (k >= nb) break;
			pair_bucket(pairs.data() + 2 * bucket_begin[k], (size_t)(bucket_begin[k + 1] - bucket_begin[k]), m.twin.data());
		}
	}, near);
}

// ---- reader ----------------------------------------------------------------------------------------------
namespace {
struct Cur {
	const uint8_t *p, *end;
	void ws() { while (p < end && isspace(*p)) ++p; }
	std::string tok() { ws(); const uint8_t *b = p; while (p < end && !isspace(*p)) ++p; return std::string((const char*)b, (const char*)p); }
	void line() { while (p < end && *p != '\n') ++p; if (p < end) ++p; }
	void need(size_t n) const { if ((size_t)(end - p) < n) throw Error(HRY_E_FORMAT, "truncated PLY"); }
};

template <typename T> inline void put(uint8_t *d, T v) { memcpy(d, &v, sizeof(T)); }

// reads one value of type t (mode 0 ascii / 1 LE / 2 BE) into dst (host order), returns it as uint64 (list lengths, indices)
inline uint64_t read_value(Cur &c, int mode, CompType t, uint8_t *dst)
{
	if (t == C_NONE) return 1;
	if (mode == 0) {
		c.ws();
		if (c.p >= c.end) throw Error(HRY_E_FORMAT, "truncated PLY");
		char *e = nullptr;
		const char *s = (const char*)c.p;
		uint64_t ret;
		switch (t) {
		case C_CHAR: { long long v = strtoll(s, &e, 10); put<int8_t>(dst, (int8_t)v); ret = (uint64_t)(int8_t)v; break; }
		case C_UCHAR: { unsigned long long v = strtoull(s, &e, 10); put<uint8_t>(dst, (uint8_t)v); ret = (uint8_t)v; break; }
		case C_SHORT: { long long v = strtoll(s, &e, 10); put<int16_t>(dst, (int16_t)v); ret = (uint64_t)(int16_t)v; break; }
		case C_USHORT: { unsigned long long v = strtoull(s, &e, 10); put<uint16_t>(dst, (uint16_t)v); ret = (uint16_t)v; break; }
		case C_INT: { long long v = strtoll(s, &e, 10); put<int32_t>(dst, (int32_t)v); ret = (uint64_t)(int32_t)v; break; }
		case C_UINT: { unsigned long long v = strtoull(s, &e, 10); put<uint32_t>(dst, (uint32_t)v); ret = (uint32_t)v; break; }
		case C_FLOAT: { double v = strtod(s, &e); put<float>(dst, (float)v); ret = (uint64_t)(float)v; break; }
		case C_DOUBLE: { double v = strtod(s, &e); put<double>(dst, v); ret = (uint64_t)v; break; }
		default: throw Error(HRY_E_FORMAT, "Invalid data type");
		}
		if (e == s) throw Error(HRY_E_FORMAT, "malformed ASCII PLY value");
		c.p = (const uint8_t*)e;
		return ret;
	}
	int n = kTypeSize[t];
	c.need(n);
	uint8_t tmp[8];
	if (mode == 1) memcpy(tmp, c.p, n);
	else for (int i = 0; i < n; ++i) tmp[i] = c.p[n - 1 - i];
	c.p += n;
	memcpy(dst, tmp, n);
	switch (t) {
	case C_CHAR: { int8_t v; memcpy(&v, tmp, 1); return (uint64_t)v; }
	case C_UCHAR: return tmp[0];
	case C_SHORT: { int16_t v; memcpy(&v, tmp, 2); return (uint64_t)v; }
	case C_USHORT: { uint16_t v; memcpy(&v, tmp, 2); return v; }
	case C_INT: { int32_t v; memcpy(&v, tmp, 4); return (uint64_t)v; }
	case C_UINT: { uint32_t v; memcpy(&v, tmp, 4); return v; }
	case C_FLOAT: { float v; memcpy(&v, tmp, 4); return (uint64_t)v; }
	default: { double v; memcpy(&v, tmp, 8); return (uint64_t)v; }
	}
}
}   // namespace

Mesh *mesh_from_ply(const uint8_t *buf, size_t n)
{
	Cur c{ buf, buf + n };
	std::vector<Element> elems;
	int mode = -1;
	for (;;) {
		std::string id = c.tok();
		if (id.empty()) throw Error(HRY_E_FORMAT, "PLY header has no end_header");
		if (id == "end_header") break;
		if (id == "ply") continue;
		if (id == "format") {
			std::string f = c.tok();
			mode = f == "ascii" ? 0 : f == "binary_little_endian" ? 1 : f == "binary_big_endian" ? 2 : -2;
			if (mode == -2) throw Error(HRY_E_FORMAT, "Invlaid format");
			c.line();
		} else if (id == "element") {
			Element e;
			e.name = c.tok();
			e.count = atol(c.tok().c_str());
			elems.push_back(std::move(e));
		} else if (id == "property") {
			if (elems.empty()) throw Error(HRY_E_FORMAT, "Invlaid property");
			std::string ty = c.tok();
			CompType lt = C_NONE;
			if (ty == "list") { lt = parse_type(c.tok()); ty = c.tok(); }
			std::string nm = c.tok();
			elems.back().props.push_back(Prop{ nm, parse_type(ty), lt });
		} else c.line();   // comment / obj_info / unknown
	}
	c.line();
	if (mode < 0) throw Error(HRY_E_FORMAT, "PLY header has no format line");
	int fi = -1, vi = -1;
	for (size_t i = 0; i < elems.size(); ++i) { if (elems[i].name == "face") fi = (int)i; else if (elems[i].name == "vertex") vi = (int)i; }
	if (fi < 0 || vi < 0) throw Error(HRY_E_FORMAT, "PLY needs a vertex and a face element");

	std::unique_ptr<Mesh> m(new Mesh());
	if (elems[fi].count < 0 || elems[vi].count < 0 || (uint64_t)elems[fi].count > 0xfffffff0ull || (uint64_t)elems[vi].count > 0xfffffff0ull)
		throw Error(HRY_E_FORMAT, "implausible element count");
	std::vector<int> slot[2];
	const int which[2] = { fi, vi };
	for (int k = 0; k < 2; ++k) {
		const Element &el = elems[which[k]];
		std::vector<std::string> names; std::vector<CompType> types; std::vector<bool> isl;
		for (const Prop &p : el.props) { names.push_back(p.name); types.push_back(p.type); isl.push_back(p.len_type != C_NONE); }
		AttrList &L = m->lists[k];
		L.target = k;
		layout_attr_list(names, types, isl, L, slot[k]);
		L.count = (uint32_t)el.count;
		// (binary little-endian records of scalars only are copied whole below: zeros first were 60 ms per 600 MB of them)
		bool whole = mode == 1;
		for (const Prop &p : el.props) whole &= p.len_type == C_NONE;
		if (whole) L.data.resize((size_t)L.count * L.stride());
		else L.data.assign((size_t)L.count * L.stride(), 0);
	}
	m->nf = (uint32_t)elems[fi].count;
	m->nv = (uint32_t)elems[vi].count;
	int conn_prop = -1;
	for (size_t k = 0; k < elems[fi].props.size(); ++k) if (elems[fi].props[k].name == "vertex_indices") conn_prop = (int)k;
	if (conn_prop < 0 || elems[fi].props[conn_prop].len_type == C_NONE) throw Error(HRY_E_FORMAT, "PLY face element has no vertex_indices list");
	m->face_off.reserve((size_t)m->nf + 1);
	m->org.reserve((size_t)m->nf * 3);

	uint8_t scratch[8];
	bool index_range_checked = false;   // (the threaded face paths check the indices as they copy them)
	for (size_t ei = 0; ei < elems.size(); ++ei) {
		const Element &el = elems[ei];
		bool is_attr = (int)ei == fi || (int)ei == vi;
		int k = (int)ei == fi ? 0 : 1;
		AttrList *L = is_attr ? &m->lists[k] : nullptr;
		// fast path: binary little-endian records made of scalar attributes only
		bool all_scalar = true;
		for (const Prop &p : el.props) all_scalar &= p.len_type == C_NONE;
		if (is_attr && mode == 1 && all_scalar) {
			size_t rec = 0;
			std::vector<int> src_off;
			for (const Prop &p : el.props) { src_off.push_back((int)rec); rec += kTypeSize[p.type]; }
			c.need(rec * (size_t)el.count);
			bool ident = (int)rec == L->stride();
			for (size_t i = 0; ident && i < el.props.size(); ++i) ident = L->offset[slot[k][i]] == src_off[i];
			{
				const size_t n = (size_t)el.count, stride = (size_t)L->stride();
				const unsigned nt = rec * n >= ((size_t)16 << 20) ? std::max(1u, host_threads()) : 1u;
				const uint8_t *base = c.p;
				// (every slot of a record is a property of the element, rec == stride: no byte stays unwritten)
				parallel_for(nt, [&](unsigned t) {
					const size_t j0 = n * t / nt, j1 = n * (t + 1) / nt;
					if (ident) { if (j1 > j0) memcpy(L->data.data() + j0 * rec, base + j0 * rec, (j1 - j0) * rec); return; }
					for (size_t j = j0; j < j1; ++j) {
						const uint8_t *s = base + rec * j;
						uint8_t *d = L->data.data() + j * stride;
						for (size_t i = 0; i < el.props.size(); ++i) memcpy(d + L->offset[slot[k][i]], s + src_off[i], kTypeSize[el.props[i].type]);
					}
				});
			}
			c.p += rec * (size_t)el.count;
			continue;
		}
		// fast path: binary little-endian faces that are nothing but the index list, one count byte + 4-byte indices
		if ((int)ei == fi && mode == 1 && el.props.size() == 1 && kTypeSize[el.props[0].len_type] == 1 && kTypeSize[el.props[0].type] == 4 &&
		    el.props[0].type != C_FLOAT) {
			const uint32_t nf = m->nf, nv = m->nv;
			const uint8_t *p = c.p;
			const size_t avail = (size_t)(c.end - c.p);
			m->face_off.resize((size_t)nf + 1);
			m->face_off[0] = 0;
			// all triangles (the usual file): fixed 13-byte records, checked and copied by a few threads
			bool tri = avail >= (size_t)nf * 13 && nf >= (1u << 16);
			if (tri) {
				unsigned n_near = 0;
				const void *near = callers_cache_cpus(&n_near);
				const unsigned nt = std::max(1u, std::min(near ? n_near : 8u, host_threads()));
				std::atomic<bool> ok{ true }, range{ true };
				m->org.resize((size_t)nf * 3);
				uint32_t *org = m->org.data(), *foff = m->face_off.data();
				parallel_for(nt, [&](unsigned t) {
					const uint32_t fb = (uint32_t)((uint64_t)nf * t / nt), fe = (uint32_t)((uint64_t)nf * (t + 1) / nt);
					bool good = true, inside = true;
					for (uint32_t f = fb; f < fe; ++f) {
						const uint8_t *r = p + (size_t)f * 13;
						good &= r[0] == 3;
						uint32_t v[3];
						memcpy(v, r + 1, 12);
						inside &= (v[0] < nv) & (v[1] < nv) & (v[2] < nv);
						memcpy(org + (size_t)f * 3, v, 12);
						foff[f + 1] = (f + 1) * 3;
					}
					if (!good) ok.store(false, std::memory_order_relaxed);
					if (!inside) range.store(false, std::memory_order_relaxed);
				}, near);
				tri = ok.load();
				if (tri && !range.load()) throw Error(HRY_E_FORMAT, "PLY vertex index out of range");
				if (tri) {
					if (m->have_degree.size() < 4) m->have_degree.resize(4, 0);
					m->have_degree[3] = 1;
					c.p += (size_t)nf * 13;
					index_range_checked = true;
					continue;
				}
			}
			// Polygons of several degrees: where a record starts depends on every count byte before it.  One thread follows the
			// count bytes alone (a load and an add per face) and leaves a mark every 32 Ki faces -- byte offset and half-edge
			// offset -- then the blocks between the marks are checked and copied by the host threads, like the triangles above
			// (the serial loop did everything per face: 1.1 s for the 78 M faces of the configs[3] mesh; round 4).
			m->org.clear();
			const uint32_t kBlock = 1u << 15;
			const uint32_t nblk = (nf + kBlock - 1) / kBlock;
			std::vector<size_t> blk_byte(nblk + 1), blk_he(nblk + 1);
			{
				size_t at = 0, he = 0;
				for (uint32_t f = 0; f < nf; ++f) {
					if ((f & (kBlock - 1)) == 0) { blk_byte[f / kBlock] = at; blk_he[f / kBlock] = he; }
					if (at >= avail) throw Error(HRY_E_FORMAT, "truncated PLY");
					const size_t len = p[at];
					at += 1 + 4 * len;
					he += len;
				}
				if (at > avail) throw Error(HRY_E_FORMAT, "truncated PLY");
				if (he > 0xfffffff0ull) throw Error(HRY_E_UNSUPPORTED, "more than 2^32 polygon corners");
				blk_byte[nblk] = at; blk_he[nblk] = he;
			}
			m->org.resize(blk_he[nblk]);
			{
				const unsigned nt = nf >= (1u << 16) ? std::max(1u, host_threads()) : 1u;
				std::atomic<uint32_t> next{ 0 };
				std::atomic<int> bad{ 0 };   // 1: degree below 3, 2: index out of range
				std::vector<std::vector<uint8_t>> seen(nt, std::vector<uint8_t>(256, 0));
				uint32_t *org = m->org.data(), *foff = m->face_off.data();
				parallel_for(nt, [&](unsigned t) {
					std::vector<uint8_t> &have = seen[t];
					for (;;) {
						const uint32_t b = next.fetch_add(1, std::memory_order_relaxed);
						if (b >= nblk) break;
						const uint32_t f0 = b * kBlock, f1 = std::min(nf, f0 + kBlock);
						const uint8_t *q = p + blk_byte[b];
						size_t he = blk_he[b];
						bool small = false, inside = true;
						for (uint32_t f = f0; f < f1; ++f) {
							const uint32_t len = *q++;
							small |= len < 3;
							have[len] = 1;
							memcpy(org + he, q, (size_t)len * 4);
							for (uint32_t k = 0; k < len; ++k) inside &= org[he + k] < nv;
							q += (size_t)len * 4;
							he += len;
							foff[f + 1] = (uint32_t)he;
						}
						if (small) bad.store(1, std::memory_order_relaxed);
						else if (!inside && bad.load(std::memory_order_relaxed) == 0) bad.store(2, std::memory_order_relaxed);
					}
				});
				if (bad.load() == 1) throw Error(HRY_E_UNSUPPORTED, "polygon degree outside 3..255");
				if (bad.load() == 2) throw Error(HRY_E_FORMAT, "PLY vertex index out of range");
				for (unsigned t = 0; t < nt; ++t)
					for (size_t d = 0; d < 256; ++d) if (seen[t][d]) { if (d >= m->have_degree.size()) m->have_degree.resize(d + 1, 0); m->have_degree[d] = 1; }
			}
			c.p = p + blk_byte[nblk];
			index_range_checked = true;
			continue;
		}
		for (long j = 0; j < el.count; ++j) {
			for (size_t pi = 0; pi < el.props.size(); ++pi) {
				const Prop &p = el.props[pi];
				if (is_attr && slot[k][pi] >= 0) {
					read_value(c, mode, p.type, L->data.data() + (size_t)j * L->stride() + L->offset[slot[k][pi]]);
					continue;
				}
				uint64_t len = read_value(c, mode, p.len_type, scratch);
				if ((int)ei == fi && (int)pi == conn_prop) {
					if (len < 3 || len > 255) throw Error(HRY_E_UNSUPPORTED, "polygon degree outside 3..255");
					if (len >= m->have_degree.size()) m->have_degree.resize(len + 1, 0);
					m->have_degree[len] = 1;
					for (uint64_t l = 0; l < len; ++l) m->org.push_back((uint32_t)read_value(c, mode, p.type, scratch));
					m->face_off.push_back((uint32_t)m->org.size());
				} else for (uint64_t l = 0; l < len; ++l) read_value(c, mode, p.type, scratch);
			}
		}
	}
	if (m->face_off.size() != (size_t)m->nf + 1) throw Error(HRY_E_FORMAT, "PLY face count mismatch");
	if (!index_range_checked) for (uint32_t v : m->org) if (v >= m->nv) throw Error(HRY_E_FORMAT, "PLY vertex index out of range");
	m->twins_pending = true;   // matched on the device at the first upload, or by ensure_twins
	return m.release();
}

Mesh *mesh_from_arrays(uint32_t nv, const uint8_t *vrec, int v_ncomp, const uint8_t *v_types, const char *const *v_names,
                       uint32_t nf, const uint8_t *degrees, const uint32_t *indices,
                       const uint8_t *frec, int f_ncomp, const uint8_t *f_types, const char *const *f_names)
{
	std::unique_ptr<Mesh> m(new Mesh());
	m->nv = nv; m->nf = nf;
	const uint8_t *recs[2] = { frec, vrec };
	const int ncomps[2] = { f_ncomp, v_ncomp };
	const uint8_t *types[2] = { f_types, v_types };
	const char *const *names[2] = { f_names, v_names };
	const uint32_t counts[2] = { nf, nv };
	for (int k = 0; k < 2; ++k) {
		std::vector<std::string> nm; std::vector<CompType> ty; std::vector<bool> isl;
		std::vector<int> src_off;
		int rec = 0;
		for (int i = 0; i < ncomps[k]; ++i) {
			if (types[k][i] >= C_NONE) throw Error(HRY_E_ARG, "bad component type");
			nm.push_back(names[k][i]); ty.push_back((CompType)types[k][i]); isl.push_back(false);
			src_off.push_back(rec); rec += kTypeSize[types[k][i]];
		}
		AttrList &L = m->lists[k];
		L.target = k;
		std::vector<int> slot;
		layout_attr_list(nm, ty, isl, L, slot);
		L.count = counts[k];
		L.data.assign((size_t)L.count * L.stride(), 0);
		for (uint32_t j = 0; j < L.count && rec; ++j)
			for (int i = 0; i < ncomps[k]; ++i)
				memcpy(L.data.data() + (size_t)j * L.stride() + L.offset[slot[i]], recs[k] + (size_t)j * rec + src_off[i], kTypeSize[ty[i]]);
	}
	m->face_off.reserve((size_t)nf + 1);
	uint64_t tot = 0;
	for (uint32_t f = 0; f < nf; ++f) {
		int d = degrees[f];
		if (d < 3) throw Error(HRY_E_UNSUPPORTED, "polygon degree outside 3..255");
		if (d >= (int)m->have_degree.size()) m->have_degree.resize(d + 1, 0);
		m->have_degree[d] = 1;
		tot += d;
		if (tot > 0xffffffffull) throw Error(HRY_E_UNSUPPORTED, "more than 2^32-1 half-edges");
		m->face_off.push_back((uint32_t)tot);
	}
	m->org.assign(indices, indices + tot);
	for (uint32_t v : m->org) if (v >= nv) throw Error(HRY_E_ARG, "vertex index out of range");
	m->twins_pending = true;   // matched on the device at the first upload, or by ensure_twins
	return m.release();
}

// ---- writer (formats/ply/writer.cc:106-192) ------------------------------------------------------------------
namespace {
const char *type_name(CompType t)
{
	switch (t) {
	case C_FLOAT: return "float"; case C_DOUBLE: return "double"; case C_UINT: return "uint"; case C_INT: return "int";
	case C_USHORT: return "ushort"; case C_SHORT: return "short"; case C_UCHAR: return "uchar"; case C_CHAR: return "char";
	default: return "";
	}
}
std::string interp_prop_name(const AttrList &L, int interp, int k)   // writer.cc:43-66
{
	static const std::vector<std::vector<const char*>> names = {
		{ "x", "y", "z", "w" }, { "nx", "ny", "nz", "nw" }, { "red", "green", "blue" }, { "ambient_red", "ambient_green", "ambient_blue" },
		{ "diffuse_red", "diffuse_green", "diffuse_blue" }, { "specular_red", "specular_green", "specular_blue" }, { "u", "v", "tw" },
		{ "scale" }, { "confidence" } };
	static const char *group[] = { "pos", "normal", "color", "ambient", "diffuse", "specular", "tex" };
	if (interp < kInterpOther) {
		if (interp < (int)names.size()) {
			if (k < (int)names[interp].size()) return names[interp][k];
			return std::string(group[interp]) + "_" + std::to_string(k);
		}
		return "unknown_" + std::to_string(interp) + "_" + std::to_string(k);
	}
	const std::string &n = L.interp_name[interp - kInterpOther];
	return L.interp_len[interp] == 1 ? n : n + "_" + std::to_string(k);
}
}   // namespace
void print_component(std::string &o, const AttrList &L, const uint8_t *rec, int c)   // mixing.h:340-359
{
	char buf[64];
	const uint8_t *p = rec + L.offset[c];
	switch (L.stype(c)) {
	case C_CHAR: snprintf(buf, sizeof buf, "%d", (int)*(const int8_t*)p); break;
	case C_SHORT: { int16_t v; memcpy(&v, p, 2); snprintf(buf, sizeof buf, "%d", (int)v); break; }
	case C_INT: { int32_t v; memcpy(&v, p, 4); snprintf(buf, sizeof buf, "%d", v); break; }
	case C_UCHAR: snprintf(buf, sizeof buf, "%u", (unsigned)*p); break;
	case C_USHORT: { uint16_t v; memcpy(&v, p, 2); snprintf(buf, sizeof buf, "%u", (unsigned)v); break; }
	case C_UINT: { uint32_t v; memcpy(&v, p, 4); snprintf(buf, sizeof buf, "%u", v); break; }
	case C_FLOAT: { float v; memcpy(&v, p, 4); snprintf(buf, sizeof buf, "%g", (double)v); break; }
	case C_DOUBLE: { double v; memcpy(&v, p, 8); snprintf(buf, sizeof buf, "%g", v); break; }
	default: buf[0] = 0;
	}
	o += buf;
}
namespace {
inline void print_comp(std::string &o, const AttrList &L, const uint8_t *rec, int c) { print_component(o, L, rec, c); }

// General bindings (formats/ply/writer.cc:106-192): the vertex element carries the lists that EVERY vertex region binds, the face
// element those every face region binds as face lists (corner lists have no PLY form); each element writes the records its own
// region binds, in slot order.
void general_to_ply(const Mesh &m, bool ascii, ByteSink &out, bool packed)
{
	const Bindings &b = m.bind;
	auto common = [&](int nregs, auto nlists, auto list_at) {   // writer.cc:141-156: running intersection over the regions that bind anything
		std::vector<int> cur, last;
		for (int r = 0; r < nregs; ++r) {
			for (int a = 0; a < nlists(r); ++a) {
				int l = list_at(r, a);
				if (r == 0 || std::find(last.begin(), last.end(), l) != last.end()) if (std::find(cur.begin(), cur.end(), l) == cur.end()) cur.push_back(l);
			}
			if (nlists(r)) { std::swap(cur, last); cur.clear(); }
		}
		std::sort(last.begin(), last.end());
		return last;
	};
	const std::vector<int> vl = common(b.nregs_vtx(), [&](int r) { return b.nvtxlists(r); }, [&](int r, int a) { return b.vtxlist(r, a); });
	const std::vector<int> fl = common(b.nregs_face(), [&](int r) { return b.nfacelists(r); }, [&](int r, int a) { return b.facelist(r, a); });
	std::vector<char> in_v(m.lists.size(), 0), in_f(m.lists.size(), 0);
	std::string h = std::string("ply\nformat ") + (ascii ? "ascii" : "binary_little_endian") + " 1.0\ncomment decompressed using harry mesh compressor\n";
	auto props = [&](const AttrList &L) {
		for (int i = 0; i < (int)L.interp_off.size(); ++i)
			for (int k = 0; k < L.interp_len[i]; ++k)
				h += std::string("property ") + type_name(L.stype(L.interp_off[i] + k)) + " " + interp_prop_name(L, i, k) + "\n";
	};
	h += "element vertex " + std::to_string(m.nv) + "\n";
	for (int l : vl) { in_v[l] = 1; props(m.lists[l]); }
	h += "element face " + std::to_string(m.nf) + "\nproperty list uchar uint vertex_indices\n";
	for (int l : fl) { in_f[l] = 1; props(m.lists[l]); }
	h += "end_header\n";
	out.assign(h.begin(), h.end());
	std::string o;
	auto record = [&](int l, uint32_t idx, bool append) {
		const AttrList &L = m.lists[l];
		const uint8_t *rec = L.data.data() + (size_t)idx * L.stride();
		if (ascii) { for (int c = 0; c < L.ncomp(); ++c) { if (c || append) o += '\t'; print_component(o, L, rec, c); } return; }
		bool q = false;
		for (int c = 0; c < L.ncomp(); ++c) q |= L.quant[c] != 0;
		if (packed && q) for (int c = 0; c < L.ncomp(); ++c) o.append((const char*)rec + L.offset[c], (size_t)kTypeSize[L.stype(c)]);
		else o.append((const char*)rec, (size_t)L.stride());
	};
	auto flush = [&]() { out.append(o.begin(), o.end()); o.clear(); };
	for (uint32_t v = 0; v < m.nv; ++v) {
		const int r = b.vtx_reg[v];
		for (int a = 0; a < b.nvtxlists(r); ++a) if (in_v[b.vtxlist(r, a)]) record(b.vtxlist(r, a), b.vtx_attr[(size_t)v * b.nb_vtx + a], a != 0);
		if (ascii) o += '\n';
		if (o.size() > (1u << 20)) flush();
	}
	for (uint32_t f = 0; f < m.nf; ++f) {
		const int r = b.face_reg[f];
		const uint32_t s = m.face_off[f], e = m.face_off[f + 1];
		if (ascii) { o += std::to_string((int)(uint8_t)(e - s)); for (uint32_t x = s; x < e; ++x) { o += '\t'; o += std::to_string(m.org[x]); } }
		else { o += (char)(uint8_t)(e - s); o.append((const char*)&m.org[s], 4 * (size_t)(e - s)); }
		for (int a = 0; a < b.nfacelists(r); ++a) if (in_f[b.facelist(r, a)]) record(b.facelist(r, a), b.face_attr[(size_t)f * b.nb_face + a], true);
		if (ascii) o += '\n';
		if (o.size() > (1u << 20)) flush();
	}
	flush();
}
}   // namespace

void mesh_to_ply(const Mesh &m, bool ascii, ByteSink &out, bool packed)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	if (m.general) { general_to_ply(m, ascii, out, packed); return; }
	std::string h = std::string("ply\nformat ") + (ascii ? "ascii" : "binary_little_endian") + " 1.0\ncomment decompressed using harry mesh compressor\n";
	auto props = [&](const AttrList &L) {
		for (int i = 0; i < (int)L.interp_off.size(); ++i)
			for (int k = 0; k < L.interp_len[i]; ++k)
				h += std::string("property ") + type_name(L.stype(L.interp_off[i] + k)) + " " + interp_prop_name(L, i, k) + "\n";
	};
	h += "element vertex " + std::to_string(m.nv) + "\n";
	props(m.lists[1]);
	h += "element face " + std::to_string(m.nf) + "\nproperty list uchar uint vertex_indices\n";
	props(m.lists[0]);
	h += "end_header\n";
	out.assign(h.begin(), h.end());
	const AttrList &LV = m.lists[1], &LF = m.lists[0];
	// packed: every value in the width of the type the header declares for it.  The reference dumps the whole original-width
	// record even when the header announces the narrower storage type of a quantised component (writer.cc:72-75,168; SURVEY
	// App. B-12), which no PLY reader can parse; without quantised components the two forms are the same bytes.
	auto any_quant = [](const AttrList &L) { for (int c = 0; c < L.ncomp(); ++c) if (L.quant[c]) return true; return false; };
	auto put_packed = [&](const AttrList &L, const uint8_t *rec) {
		for (int c = 0; c < L.ncomp(); ++c) out.append(rec + L.offset[c], rec + L.offset[c] + kTypeSize[L.stype(c)]);
	};
	unsigned n_near = 0;
	const void *near = callers_cache_cpus(&n_near);
	const unsigned nt_fill = std::max(1u, std::min(near ? n_near : 8u, host_threads()));
	if (!ascii) {
		const bool pv = packed && any_quant(LV), pf = packed && any_quant(LF);
		if (pv) {
			// the packed records of all vertices: one block, filled by a few threads
			size_t ps = 0;
			for (int c = 0; c < LV.ncomp(); ++c) ps += kTypeSize[LV.stype(c)];
			const size_t at = out.size();
			out.resize(at + ps * m.nv);
			uint8_t *dst = out.data() + at;
			const uint32_t nv = m.nv;
			const unsigned nt = nv >= (1u << 16) ? nt_fill : 1u;
			parallel_for(nt, [&](unsigned t) {
				const uint32_t vb = (uint32_t)((uint64_t)nv * t / nt), ve = (uint32_t)((uint64_t)nv * (t + 1) / nt);
				uint8_t *w = dst + ps * vb;
				for (uint32_t v = vb; v < ve; ++v) {
					const uint8_t *rec = LV.data.data() + (size_t)v * LV.stride();
					for (int c = 0; c < LV.ncomp(); ++c) { const int k = kTypeSize[LV.stype(c)]; memcpy(w, rec + LV.offset[c], (size_t)k); w += k; }
				}
			}, near);
		} else if (LV.data.size() >= ((size_t)16 << 20)) {   // whole original-width records (writer.cc:72-75), copied by the host threads
			const size_t at = out.size(), n = LV.data.size();
			out.resize(at + n);
			const unsigned nt = std::max(1u, host_threads());
			uint8_t *dst = out.data() + at;
			parallel_for(nt, [&](unsigned t) { const size_t b = n * t / nt, e = n * (t + 1) / nt; if (e > b) memcpy(dst + b, LV.data.data() + b, e - b); });
		} else out.append(LV.data.begin(), LV.data.end());
		size_t fs = LF.stride();
		int tri = 0;
		if (!fs && m.uniform_degree(tri) && tri == 3 && m.nf >= (1u << 16)) {   // the usual file: 13-byte records, filled by a few threads
			const size_t at = out.size();
			out.resize(at + (size_t)m.nf * 13);
			const unsigned nt = nt_fill;
			uint8_t *dst = out.data() + at;
			const uint32_t *org = m.org.data();
			const uint32_t nf = m.nf;
			parallel_for(nt, [&](unsigned t) {
				const uint32_t fb = (uint32_t)((uint64_t)nf * t / nt), fe = (uint32_t)((uint64_t)nf * (t + 1) / nt);
				for (uint32_t f = fb; f < fe; ++f) { uint8_t *r = dst + (size_t)f * 13; r[0] = 3; memcpy(r + 1, org + (size_t)f * 3, 12); }
			}, near);
			return;
		}
		if (m.nf >= (1u << 16)) {
			// polygons of several degrees (and face records): face f starts f count bytes, 4 face_off[f] index bytes and f face
			// records into the element, so every thread knows where its faces go
			size_t frec = fs;
			if (fs && pf) { frec = 0; for (int c = 0; c < LF.ncomp(); ++c) frec += (size_t)kTypeSize[LF.stype(c)]; }
			const size_t at = out.size();
			const uint32_t nf = m.nf;
			out.resize(at + (size_t)nf * (1 + frec) + 4 * (size_t)m.face_off[nf]);
			uint8_t *dst = out.data() + at;
			const uint32_t *org = m.org.data(), *foff = m.face_off.data();
			const unsigned nt = std::max(1u, host_threads());
			parallel_for(nt, [&](unsigned t) {
				const uint32_t fb = (uint32_t)((uint64_t)nf * t / nt), fe = (uint32_t)((uint64_t)nf * (t + 1) / nt);
				uint8_t *w = dst + (size_t)fb * (1 + frec) + 4 * (size_t)foff[fb];
				for (uint32_t f = fb; f < fe; ++f) {
					const uint32_t b = foff[f], e = foff[f + 1];
					*w++ = (uint8_t)(e - b);
					memcpy(w, org + b, 4 * (size_t)(e - b));
					w += 4 * (size_t)(e - b);
					if (fs && pf) { const uint8_t *rec = LF.data.data() + (size_t)f * fs; for (int c = 0; c < LF.ncomp(); ++c) { const int k = kTypeSize[LF.stype(c)]; memcpy(w, rec + LF.offset[c], (size_t)k); w += k; } }
					else if (fs) { memcpy(w, LF.data.data() + (size_t)f * fs, fs); w += fs; }
				}
			});
			return;
		}
		for (uint32_t f = 0; f < m.nf; ++f) {
			uint32_t b = m.face_off[f], e = m.face_off[f + 1];
			out.push_back((uint8_t)(e - b));
			const uint8_t *p = (const uint8_t*)&m.org[b];
			out.append(p, p + 4 * (size_t)(e - b));
			if (fs && pf) put_packed(LF, LF.data.data() + (size_t)f * fs);
			else if (fs) out.append(LF.data.begin() + (size_t)f * fs, LF.data.begin() + (size_t)(f + 1) * fs);
		}
		return;
	}
	std::string o;
	for (uint32_t v = 0; v < m.nv; ++v) {
		for (int c = 0; c < LV.ncomp(); ++c) { if (c) o += '\t'; print_comp(o, LV, LV.data.data() + (size_t)v * LV.stride(), c); }
		o += '\n';
	}
	for (uint32_t f = 0; f < m.nf; ++f) {
		uint32_t b = m.face_off[f], e = m.face_off[f + 1];
		o += std::to_string((int)(uint8_t)(e - b));
		for (uint32_t x = b; x < e; ++x) { o += '\t'; o += std::to_string(m.org[x]); }
		for (int c = 0; c < LF.ncomp(); ++c) { o += '\t'; print_comp(o, LF, LF.data.data() + (size_t)f * LF.stride(), c); }
		o += '\n';
		if (o.size() > (1u << 20)) { out.append(o.begin(), o.end()); o.clear(); }
	}
	out.append(o.begin(), o.end());
}

}   // namespace hry
