// Reading the reference's single-stream format (.hry v0.1): the host side of the decoder.
//
// One adaptive arithmetic-coded stream carries everything (SURVEY.md finding 0-1), and a symbol can only be located
// after every symbol before it has been decoded with its adapted model: this part is serial by construction of the
// format, exactly like the cut-border replay it is interleaved with.  The host therefore does what the format forces
// to be serial - entropy decoding and the replay - and hands the residual byte planes to the same device
// reconstruction the chunked profile uses (unchunk.cpp).
//
// Behavioural contract: arith/coder.h:115-172 (decoder), arith/stat_adaptive.h:26-126 (adaptive tables),
// formats/hry/models.h:27-237 (context inventory, order-conditioned operation model), formats/hry/io.h:168-231,
// formats/hry/attrcode.h:443-501 (attribute symbol order), bitstream.h:14-40 (MSB-first bits, ones past the end).
#include "cbm_replay.hpp"
#include "perf_counters.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace hry {
namespace {

// MSB-first bit source; past the end of the data every bit reads as 1 (istream::get() == -1, bitstream.h:27).
struct Bits {
	const uint8_t *p, *end;
	uint64_t buf = 0;
	int have = 0;
	uint64_t take(int k)   // 0 <= k <= 56
	{
		if (k == 0) return 0;
		while (have < k) { buf = (buf << 8) | (p < end ? *p++ : 0xFFu); have += 8; }
		uint64_t v = (buf >> (have - k)) & ((k == 64) ? ~0ull : ((1ull << k) - 1));
		have -= k;
		return v;
	}
};

// arith/coder.h:115-153
struct Decoder {
	static constexpr uint64_t kHalf = 1ull << 63, kQuarter = 1ull << 62;
	uint64_t range = kHalf, value = 0, r = 0;
	Bits in;
	Decoder(const uint8_t *b, const uint8_t *e) : in{ b, e }
	{
		value = in.take(32) << 32;
		value |= in.take(32);
	}
	uint64_t target(uint64_t t)
	{
		r = range / t;
		uint64_t q = value / r;
		return q < t - 1 ? q : t - 1;
	}
	// the first half of target(): the symbol is then located by comparing cumulative counts scaled by r with the value
	// (Table::locate), which spares the second division
	void scale(uint64_t t) { r = range / t; }
	void consume(uint64_t l, uint64_t h, uint64_t t)
	{
		value -= r * l;
		range = h < t ? r * (h - l) : range - r * l;
		if (range == 0) throw Error(HRY_E_FORMAT, "corrupt stream (empty coding interval)");
		if (range <= kQuarter) {
			// renormalise until range is in (2^62, 2^63]
			int top = 63 - __builtin_clzll(range);
			int sh = (range == (1ull << top)) ? 63 - top : 62 - top;
			while (sh > 0) {
				int k = sh > 56 ? 56 : sh;
				range <<= k;
				value = (value << k) | in.take(k);
				sh -= k;
			}
		}
	}
};

// Adaptive frequency table over <= 256 symbols: counts plus sums over blocks of 16 (same cumulative frequencies as
// the reference's Fenwick tree, stat_adaptive.h:26-126; the halving above 2^62 total cannot trigger below 2^62 symbols).
struct Table {
	uint64_t cnt[256];
	uint64_t blk[16];
	uint64_t tot = 0;
	Table() { memset(cnt, 0, sizeof(cnt)); memset(blk, 0, sizeof(blk)); }
	void add(uint32_t s, uint64_t d) { cnt[s] += d; blk[s >> 4] += d; tot += d; }
	void set(uint32_t s, uint64_t f) { uint64_t d = f - cnt[s]; cnt[s] += d; blk[s >> 4] += d; tot += d; }
	void ones() { for (int i = 0; i < 256; ++i) cnt[i] = 1; for (int b = 0; b < 16; ++b) blk[b] = 16; tot = 256; }
	// find(min(value / r, tot - 1)) without the division: "target >= c" for a cumulative count c is "c r <= value and c < tot"
	// (c r <= range: no overflow).  Same symbol, same l and h as find() in every case, ties and the clamp included.
	uint32_t locate(uint64_t value, uint64_t r, uint64_t &l, uint64_t &h) const
	{
		uint64_t cum = 0;
		uint32_t b = 0;
		for (; b < 15; ++b) { const uint64_t nx = cum + blk[b]; if (nx >= tot || nx * r > value) break; cum = nx; }
		uint32_t s = b << 4;
		for (; s < 255; ++s) { const uint64_t nx = cum + cnt[s]; if (nx >= tot || nx * r > value) break; cum = nx; }
		l = cum;
		h = cum + cnt[s];
		return s;
	}
	uint32_t find(uint64_t target, uint64_t &l, uint64_t &h) const
	{
		uint64_t rem = target;
		uint32_t b = 0;
		while (b < 15 && rem >= blk[b]) rem -= blk[b++];
		uint32_t s = b << 4;
		while (s < 255 && rem >= cnt[s]) rem -= cnt[s++];
		l = target - rem;
		h = l + cnt[s];
		return s;
	}
};

enum { IOP_SYMS = 9, OP_SYMS = 7, OP_NEWVTX = 5, OP_CONNFWD = 6 };

struct Live {
	Decoder dc;
	Table t_iop, t_op, t_elem[4], t_part[2], t_vert[4], t_numtri[2];
	uint64_t c_all = 2, c_new[8], c_fwd[8];   // models.h:49-120

	Live(const uint8_t *b, const uint8_t *e, const Mesh &m) : dc(b, e)
	{
		for (int i = 0; i < IOP_SYMS; ++i) t_iop.add(i, 1);
		for (int i = 0; i < OP_SYMS; ++i) t_op.add(i, 1);
		for (int i = 0; i < 8; ++i) c_new[i] = c_fwd[i] = 1;
		for (auto &t : t_elem) t.ones();
		for (auto &t : t_part) t.ones();
		for (auto &t : t_vert) t.ones();
		for (size_t d = 0; d < m.have_degree.size(); ++d)   // models.h:209-217
			if (m.have_degree[d]) {
				uint32_t v = (uint32_t)d - 2;
				if ((v & 0xff) > 127 || (v >> 8) > 127) throw Error(HRY_E_UNSUPPORTED, "polygon degree outside the reference's model seeding range");
				t_numtri[0].add(v & 0xff, 1);
				t_numtri[1].add(v >> 8, 1);
			}
	}
	uint32_t code(Table &t)   // coder.h:154-162 + model.h:57-66
	{
		if (t.tot == 0) throw Error(HRY_E_FORMAT, "corrupt stream (symbol from an empty model)");
		uint64_t l, h;
		dc.scale(t.tot);
		if (dc.r == 0) throw Error(HRY_E_FORMAT, "corrupt stream (coding interval smaller than the model's total)");
		uint32_t s = t.locate(dc.value, dc.r, l, h);
		if (h == l) throw Error(HRY_E_FORMAT, "corrupt stream (symbol with zero frequency)");
		dc.consume(l, h, t.tot);
		return s;
	}
	uint32_t sym(Table &t) { uint32_t s = code(t); t.add(s, 1); return s; }
	uint32_t iop() { return sym(t_iop); }
	uint32_t u32(Table *t) { uint32_t v = sym(t[0]); v |= sym(t[1]) << 8; v |= sym(t[2]) << 16; v |= sym(t[3]) << 24; return v; }
	int elem() { uint32_t z = u32(t_elem); return (int)((z >> 1) ^ ((z & 1) ? 0xffffffffu : 0u)); }
	int part() { uint32_t v = sym(t_part[0]); v |= sym(t_part[1]) << 8; return (int)v; }
	uint32_t vertid() { return u32(t_vert); }
	int numtri() { uint32_t v = sym(t_numtri[0]); v |= sym(t_numtri[1]) << 8; return (int)v; }
	uint32_t op(int order)   // models.h:74-119
	{
		int i = order - 1;
		if (i > 7) i = 7;
		if (i < 0) i = 0;
		uint64_t nv = c_new[i] * c_all / (c_new[i] + c_fwd[i]);
		t_op.set(OP_NEWVTX, nv);
		t_op.set(OP_CONNFWD, c_all - nv);
		uint32_t s = code(t_op);
		if (s == OP_NEWVTX) { ++c_all; ++c_new[i]; }
		else if (s == OP_CONNFWD) { ++c_all; ++c_fwd[i]; }
		else t_op.add(s, 1);
		return s;
	}
};

// attribute symbols of one list, in stream order (attrcode.h:443-501): region (2 bytes, a single region: symbol 0 with
// l = 0 and h = t, which leaves the decoder untouched), record type, then the residual bytes of every component
void read_list(Live &lv, const AttrList &L, bool corner_list, uint32_t count, std::vector<uint8_t> &planes)
{
	int nplanes = 0;
	for (int c = 0; c < L.ncomp(); ++c) nplanes += kTypeSize[L.stype(c)];
	planes.assign((size_t)nplanes * count, 0);
	Table t_type;
	t_type.add(0, 1); t_type.add(1, 1);            // DATA, HIST (models.h:201-203)
	if (corner_list) t_type.add(2, 1);
	std::vector<Table> t_data((size_t)nplanes);
	for (auto &t : t_data) t.ones();
	for (uint32_t i = 0; i < count; ++i) {
		if (lv.sym(t_type) != 0) throw Error(HRY_E_UNSUPPORTED, "shared-attribute history records are outside the supported subset");
		for (int q = 0; q < nplanes; ++q) planes[(size_t)q * count + i] = (uint8_t)lv.sym(t_data[q]);
	}
}

// attribute symbols with general bindings (attrcode.h:443-531): per vertex its region and one record reference per list of the
// region; per face its region, its face lists, then every corner's lists.  A reference is DATA (a new record: its residual bytes
// follow), HIST (a record created earlier, by distance in creation order) or, at corners, LHIST (a record already named at this
// vertex, by distance in the vertex' own list of names).  Pure integer bookkeeping along the coding order; the residual bytes
// go to the device.  SRC says where the symbols come from: the live arithmetic decoder of a reference stream, or the decoded
// planes of a chunked container.
struct LiveSource {
	Live &lv;
	Mesh &m;
	struct PerList { Table t_type, t_ghist[4], t_lhist[2]; std::vector<Table> t_data; std::vector<int> byte_at; };
	std::vector<PerList> pl;
	Table t_regface[2], t_regvtx[2];
	int plane_list = -1;                 // this list's residual bytes also go to `planes`, plane-major with plane_count records per plane
	uint32_t plane_count = 0;
	std::vector<uint8_t> *planes = nullptr;
	LiveSource(Live &l, Mesh &mesh) : lv(l), m(mesh), pl(mesh.lists.size())
	{
		const Bindings &b = m.bind;
		for (size_t i = 0; i < m.lists.size(); ++i) {
			const AttrList &L = m.lists[i];
			PerList &P = pl[i];
			P.t_type.add(0, 1); P.t_type.add(1, 1);            // models.h:201-203
			if (L.target == 2) P.t_type.add(2, 1);
			for (auto &t : P.t_ghist) t.ones();
			for (auto &t : P.t_lhist) t.ones();
			for (int c = 0; c < L.ncomp(); ++c)
				for (int k = 0; k < kTypeSize[L.stype(c)]; ++k) P.byte_at.push_back(L.offset[c] + k);
			P.t_data.resize(P.byte_at.size());
			for (auto &t : P.t_data) t.ones();
		}
		for (int r = 0; r < b.nregs_face(); ++r) { t_regface[0].add((uint32_t)r & 0xff, 1); t_regface[1].add((uint32_t)r >> 8, 1); }   // models.h:212-217
		for (int r = 0; r < b.nregs_vtx(); ++r) { t_regvtx[0].add((uint32_t)r & 0xff, 1); t_regvtx[1].add((uint32_t)r >> 8, 1); }
	}
	uint32_t region(Table *t) { uint32_t r = lv.sym(t[0]); r |= lv.sym(t[1]) << 8; return r; }
	uint32_t region_vtx() { return region(t_regvtx); }
	uint32_t region_face() { return region(t_regface); }
	uint32_t type(int l) { return lv.sym(pl[l].t_type); }
	uint32_t ghist(int l) { return lv.u32(pl[l].t_ghist); }
	uint32_t lhist(int l) { uint32_t v = lv.sym(pl[l].t_lhist[0]); v |= lv.sym(pl[l].t_lhist[1]) << 8; return v; }
	void data(int l, uint32_t idx)
	{
		AttrList &L = m.lists[l];
		PerList &P = pl[l];
		uint8_t *rec = L.data.data() + (size_t)idx * L.stride();
		if (l == plane_list && idx < plane_count) {
			uint8_t *pp = planes->data() + idx;
			for (size_t k = 0; k < P.byte_at.size(); ++k) { const uint8_t s = (uint8_t)lv.sym(P.t_data[k]); rec[P.byte_at[k]] = s; pp[k * (size_t)plane_count] = s; }
		} else
			for (size_t k = 0; k < P.byte_at.size(); ++k) rec[P.byte_at[k]] = (uint8_t)lv.sym(P.t_data[k]);
	}
	void finish() {}
};

struct PlaneSource {
	const GenHostPlanes &hp;
	uint32_t c_regv = 0, c_regf = 0;
	std::vector<uint32_t> c_type, c_gh, c_lh, c_data;
	explicit PlaneSource(const GenHostPlanes &h) : hp(h), c_type(h.lists.size(), 0), c_gh(h.lists.size(), 0), c_lh(h.lists.size(), 0), c_data(h.lists.size(), 0) {}
	[[noreturn]] static void short_plane() { throw Error(HRY_E_FORMAT, "corrupt container (a reference plane is shorter than the mesh needs)"); }
	uint32_t region_vtx() { if (!hp.regv) return 0; if (c_regv >= hp.n_regv) short_plane(); return hp.regv[c_regv++]; }
	uint32_t region_face() { if (!hp.regf) return 0; if (c_regf >= hp.n_regf) short_plane(); return hp.regf[c_regf++]; }
	uint32_t type(int l) { const auto &P = hp.lists[l]; if (c_type[l] >= P.n_type) short_plane(); return P.type[c_type[l]++]; }
	uint32_t ghist(int l)
	{
		const auto &P = hp.lists[l];
		if (c_gh[l] >= P.n_gh) short_plane();
		const uint32_t i = c_gh[l]++;
		return (uint32_t)P.gh[0][i] | ((uint32_t)P.gh[1][i] << 8) | ((uint32_t)P.gh[2][i] << 16) | ((uint32_t)P.gh[3][i] << 24);
	}
	uint32_t lhist(int l)
	{
		const auto &P = hp.lists[l];
		if (c_lh[l] >= P.n_lh) short_plane();
		const uint32_t i = c_lh[l]++;
		return (uint32_t)P.lh[0][i] | ((uint32_t)P.lh[1][i] << 8);
	}
	void data(int l, uint32_t) { if (c_data[l] >= hp.lists[l].n_data) short_plane(); ++c_data[l]; }   // the bytes stay on the device
	void finish()
	{
		if (c_regv != hp.n_regv || c_regf != hp.n_regf) throw Error(HRY_E_FORMAT, "corrupt container (region plane longer than the mesh needs)");
		for (size_t l = 0; l < hp.lists.size(); ++l)
			if (c_type[l] != hp.lists[l].n_type || c_gh[l] != hp.lists[l].n_gh || c_lh[l] != hp.lists[l].n_lh || c_data[l] != hp.lists[l].n_data)
				throw Error(HRY_E_FORMAT, "corrupt container (a reference plane is longer than the mesh needs)");
	}
};

template <class SRC>
struct GeneralReader {
	SRC &src;
	Mesh &m;
	Bindings &b;
	std::vector<uint32_t> created;
	std::vector<GenRecordEvents> &ev;
	// per corner slot and vertex: the records named there so far, newest first
	struct Node { uint32_t idx, next; };
	std::vector<Node> pool;
	std::vector<std::vector<uint32_t>> head;
	static constexpr uint32_t NONE = 0xffffffffu;

	GeneralReader(SRC &s, Mesh &mesh, std::vector<GenRecordEvents> &events) : src(s), m(mesh), b(mesh.bind), created(mesh.lists.size(), 0), ev(events)
	{
		ev.assign(m.lists.size(), GenRecordEvents());
		head.assign(b.nb_corner, std::vector<uint32_t>());
		for (auto &h : head) h.assign(m.nv, NONE);
	}
	uint32_t new_record(int l, uint32_t he, int slot)
	{
		if (created[l] >= m.lists[l].count) throw Error(HRY_E_FORMAT, "corrupt stream (more records than the header announces)");
		const uint32_t idx = created[l]++;
		src.data(l, idx);
		ev[l].he.push_back(he); ev[l].slot.push_back((uint8_t)slot);
		return idx;
	}
	uint32_t earlier_record(int l)   // attrcode.h:463-465
	{
		const uint32_t d = src.ghist(l);
		if (d >= created[l]) throw Error(HRY_E_FORMAT, "corrupt stream (record history)");
		return created[l] - 1 - d;
	}
	void remember(int a, uint32_t v, uint32_t idx) { pool.push_back(Node{ idx, head[a][v] }); head[a][v] = (uint32_t)pool.size() - 1; }
	uint32_t named_here(int a, uint32_t v, uint32_t back)   // attrcode.h:76-79
	{
		uint32_t k = head[a][v];
		while (k != NONE && back) { k = pool[k].next; --back; }
		if (k == NONE) throw Error(HRY_E_FORMAT, "corrupt stream (per-vertex record history)");
		return pool[k].idx;
	}
	void run(const OrderVec &order_v)
	{
		if (b.nb_corner > 255 || b.nb_vtx > 255 || b.nb_face > 255) throw Error(HRY_E_UNSUPPORTED, "more than 255 lists bound to one region");
		b.corner_attr.assign((size_t)m.ne() * b.nb_corner, 0);
		for (uint32_t e : order_v) {   // attrcode.h:443-470
			const uint32_t v = m.org[e];
			const int r = (int)src.region_vtx();
			if (r >= b.nregs_vtx()) throw Error(HRY_E_FORMAT, "corrupt stream (region)");
			b.vtx_reg[v] = (uint16_t)r;
			for (int a = 0; a < b.nvtxlists(r); ++a) {
				const int l = b.vtxlist(r, a);
				const uint32_t ty = src.type(l);
				uint32_t idx;
				if (ty == 0) idx = new_record(l, e, a);
				else if (ty == 1) idx = earlier_record(l);
				else throw Error(HRY_E_FORMAT, "corrupt stream (record reference type)");
				b.vtx_attr[(size_t)v * b.nb_vtx + a] = idx;
			}
		}
		for (uint32_t f = 0; f < m.nf; ++f) {   // attrcode.h:476-531,543-548
			const int r = (int)src.region_face();
			if (r >= b.nregs_face()) throw Error(HRY_E_FORMAT, "corrupt stream (region)");
			b.face_reg[f] = (uint16_t)r;
			for (int a = 0; a < b.nfacelists(r); ++a) {
				const int l = b.facelist(r, a);
				const uint32_t ty = src.type(l);
				uint32_t idx;
				if (ty == 0) idx = new_record(l, f, a);
				else if (ty == 1) idx = earlier_record(l);
				else throw Error(HRY_E_FORMAT, "corrupt stream (record reference type)");
				b.face_attr[(size_t)f * b.nb_face + a] = idx;
			}
			for (uint32_t c = m.face_off[f]; c < m.face_off[f + 1]; ++c) {
				const uint32_t v = m.org[c];
				for (int a = 0; a < b.ncornerlists(r); ++a) {
					const int l = b.cornerlist(r, a);
					const uint32_t ty = src.type(l);
					uint32_t idx;
					if (ty == 0) { idx = new_record(l, c, a); remember(a, v, idx); }
					else if (ty == 1) { idx = earlier_record(l); remember(a, v, idx); }
					else if (ty == 2) idx = named_here(a, v, src.lhist(l));
					else throw Error(HRY_E_FORMAT, "corrupt stream (record reference type)");
					b.corner_attr[(size_t)c * b.nb_corner + a] = idx;
				}
			}
		}
		src.finish();
	}
};

}   // namespace

void read_general_stream(const uint8_t *p, size_t n, Mesh &m, OrderVec &order_v, std::vector<GenRecordEvents> &events,
                         std::vector<uint32_t> &seg_start, std::vector<uint32_t> &seg_level, int plane_list, std::vector<uint8_t> &planes)
{
	Live lv(p, p + n, m);
	cut_border_replay_with(m, lv, order_v, seg_start, seg_level);
	LiveSource src(lv, m);
	if (plane_list >= 0 && plane_list < (int)m.lists.size()) {
		src.plane_list = plane_list; src.plane_count = (uint32_t)order_v.size(); src.planes = &planes;
		planes.assign((size_t)m.lists[plane_list].coded_bytes() * order_v.size(), 0);
	}
	GeneralReader<LiveSource> gr(src, m, events);
	gr.run(order_v);
}

void read_general_planes(Mesh &m, const OrderVec &order_v, const GenHostPlanes &hp, std::vector<GenRecordEvents> &events)
{
	if (hp.lists.size() != m.lists.size()) throw Error(HRY_E_INTERNAL, "plane table does not match the lists");
	PlaneSource src(hp);
	GeneralReader<PlaneSource> gr(src, m, events);
	gr.run(order_v);
}

void read_compat_stream(const uint8_t *p, size_t n, Mesh &m, OrderVec &order_v, std::vector<uint32_t> &seg_start,
                        std::vector<uint32_t> &seg_level, std::vector<uint8_t> &vplanes, std::vector<uint8_t> &fplanes)
{
	const bool trace = getenv("HRY_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) {
		if (trace) fprintf(stderr, "[hry v0.1] %9.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
	};
	Live lv(p, p + n, m);
	cut_border_replay_with(m, lv, order_v, seg_start, seg_level);
	mark("connectivity decoded and replayed");
	PerfCounters pc;
	const bool perf = getenv("HRY_PERF") != nullptr;
	if (perf) pc.start();
	read_list(lv, m.lists[1], false, (uint32_t)order_v.size(), vplanes);
	if (perf) { pc.stop(); pc.report("v0.1 vertex records (per symbol)", (double)order_v.size() * (1 + (vplanes.size() / std::max<size_t>(1, order_v.size())))); }
	mark("vertex records' symbols decoded");
	read_list(lv, m.lists[0], false, m.nf, fplanes);
	mark("face records' symbols decoded");
}

}   // namespace hry
