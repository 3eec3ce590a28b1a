// .hry container header (formats/hry/writer.cc:104-198, reader.cc:60-177, common.h:15-16).
// All fields are little-endian raw values except the big-endian magic.  v0.1 = reference stream (compat profile);
// v0.2 = chunked profile of this implementation (the reference reader rejects it by version, reader.cc:74);
// v0.3 = sharded chunked container: the same header for the whole mesh, then one v0.2 body per shard (shard.cpp).
#include "host.hpp"

#include <algorithm>
#include <cstring>

namespace hry {
namespace {
struct Out {
	std::vector<uint8_t> &o;
	template <typename T> void v(T x) { const uint8_t *p = (const uint8_t*)&x; o.insert(o.end(), p, p + sizeof(T)); }
	void raw(const void *p, size_t n) { o.insert(o.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};
struct In {
	const uint8_t *p, *end;
	template <typename T> T v() { T x; need(sizeof(T)); memcpy(&x, p, sizeof(T)); p += sizeof(T); return x; }
	void raw(void *d, size_t n) { need(n); if (n) memcpy(d, p, n); p += n; }
	void need(size_t n) { if ((size_t)(end - p) < n) throw Error(HRY_E_FORMAT, "truncated .hry header"); }
};
}   // namespace

void write_hry_header(const Mesh &m, int ver_minor, std::vector<uint8_t> &out)
{
	Out w{ out };
	const uint8_t magic[6] = { 0xfa, 0xff, 0xaf, 0xaf, 0, (uint8_t)ver_minor };
	w.raw(magic, 6);
	const bool sh = m.shard.active();
	w.v<uint32_t>(sh ? m.shard.g_nv : m.nv); w.v<uint32_t>(sh ? m.shard.g_nf : m.nf); w.v<uint32_t>(sh ? m.shard.g_ne : m.ne());
	std::vector<char> named(m.lists.size(), 1);
	if (m.general) {   // writer.cc:124-151: the regions with the lists bound to them; a list no region names is not written
		const Bindings &b = m.bind;
		std::fill(named.begin(), named.end(), 0);
		w.v<uint16_t>((uint16_t)b.nregs_face()); w.v<uint16_t>((uint16_t)b.nregs_vtx());
		for (int r = 0; r < b.nregs_face(); ++r) {
			w.v<uint16_t>((uint16_t)b.nfacelists(r)); w.v<uint16_t>((uint16_t)b.ncornerlists(r));
			for (int a = 0; a < b.nfacelists(r); ++a) { named[b.facelist(r, a)] = 1; w.v<uint16_t>((uint16_t)b.facelist(r, a)); }
			for (int a = 0; a < b.ncornerlists(r); ++a) { named[b.cornerlist(r, a)] = 1; w.v<uint16_t>((uint16_t)b.cornerlist(r, a)); }
		}
		for (int r = 0; r < b.nregs_vtx(); ++r) {
			w.v<uint16_t>((uint16_t)b.nvtxlists(r));
			for (int a = 0; a < b.nvtxlists(r); ++a) { named[b.vtxlist(r, a)] = 1; w.v<uint16_t>((uint16_t)b.vtxlist(r, a)); }
		}
	} else {
		// region tables: one face region bound to list 0, one vertex region bound to list 1 (ply/reader.cc:399-401)
		w.v<uint16_t>(1); w.v<uint16_t>(1);
		w.v<uint16_t>(1); w.v<uint16_t>(0); w.v<uint16_t>(0);
		w.v<uint16_t>(1); w.v<uint16_t>(1);
	}
	for (size_t l = 0; l < m.lists.size(); ++l) {
		const AttrList &L = m.lists[l];
		if (!named[l]) continue;
		if (!L.have_bounds && L.ncomp() > 0) throw Error(HRY_E_INTERNAL, "attribute bounds missing");
		w.v<uint32_t>(!sh ? L.count : m.general ? m.shard.g_list_count.at(l) : l == 0 ? m.shard.g_nf : m.shard.g_nv);
		w.v<uint16_t>((uint16_t)L.ncomp());
		for (int c = 0; c < L.ncomp(); ++c) { w.v<uint8_t>(L.type[c]); w.v<uint8_t>(L.quant[c]); }
		w.v<uint16_t>((uint16_t)L.interp_off.size());
		for (size_t j = 0; j < L.interp_off.size(); ++j) {
			w.v<uint16_t>((uint16_t)L.interp_len[j]);
			if ((int)j >= kInterpOther) {
				const std::string &nm = L.interp_name[j - kInterpOther];
				w.v<uint32_t>((uint32_t)nm.size());
				w.raw(nm.data(), nm.size());
			}
		}
		w.raw(L.bmin.data(), (size_t)L.stride());
		w.raw(L.bmax.data(), (size_t)L.stride());
	}
	uint16_t cnt = 0;
	for (uint8_t d : m.have_degree) cnt += d ? 1 : 0;
	w.v<uint16_t>(cnt);
	for (size_t d = 0; d < m.have_degree.size(); ++d) if (m.have_degree[d]) w.v<uint16_t>((uint16_t)d);
}

size_t read_hry_header(const uint8_t *p, size_t n, Mesh &m, int &ver_minor, bool alloc_records)
{
	In r{ p, p + n };
	uint8_t magic[6];
	r.raw(magic, 6);
	if (magic[0] != 0xfa || magic[1] != 0xff || magic[2] != 0xaf || magic[3] != 0xaf) throw Error(HRY_E_FORMAT, "Invalid magic number");
	if (magic[4] != 0)
		throw Error(HRY_E_FORMAT, "File format version " + std::to_string(magic[4]) + "." + std::to_string(magic[5]) + " incompatible to decoder format version 0.1");
	ver_minor = magic[5];
	if (ver_minor != 1 && ver_minor != 2 && ver_minor != 3)
		throw Error(HRY_E_FORMAT, "File format version 0." + std::to_string(magic[5]) + " incompatible to decoder format version 0.1 (All 0.x-versions are incompatible to each other)");
	m.nv = r.v<uint32_t>(); m.nf = r.v<uint32_t>();
	m.declared_ne = r.v<uint32_t>();
	if ((uint64_t)m.declared_ne < 3ull * m.nf || (uint64_t)m.declared_ne > 255ull * m.nf) throw Error(HRY_E_FORMAT, "corrupt header (edge count outside 3..255 per face)");
	uint16_t nrf = r.v<uint16_t>(), nrv = r.v<uint16_t>();
	// reader.cc:86-126: region tables; a list's target is what the last region naming it binds it as
	Bindings b;
	std::vector<int> target;
	auto name_list = [&](uint16_t l, int t) { if (l >= target.size()) target.resize((size_t)l + 1, 3); target[l] = t; return l; };
	if (nrf > 128 || nrv > 128) throw Error(HRY_E_UNSUPPORTED, "more than 128 regions: the reference seeds its region models out of bounds (model.h:49-55)");
	for (int i = 0; i < nrf; ++i) {
		uint16_t nbf = r.v<uint16_t>(), nbc = r.v<uint16_t>();
		const int reg = b.add_face_region(nbf, nbc);
		b.nb_face = std::max<int>(b.nb_face, nbf); b.nb_corner = std::max<int>(b.nb_corner, nbc);
		for (int a = 0; a < nbf; ++a) b.reg_facelist[b.off_facelist[reg] + a] = name_list(r.v<uint16_t>(), 0);
		for (int a = 0; a < nbc; ++a) b.reg_cornerlist[b.off_cornerlist[reg] + a] = name_list(r.v<uint16_t>(), 2);
	}
	for (int i = 0; i < nrv; ++i) {
		uint16_t nbv = r.v<uint16_t>();
		const int reg = b.add_vtx_region(nbv);
		b.nb_vtx = std::max<int>(b.nb_vtx, nbv);
		for (int a = 0; a < nbv; ++a) b.reg_vtxlist[b.off_vtxlist[reg] + a] = name_list(r.v<uint16_t>(), 1);
	}
	if (target.size() > 4096) throw Error(HRY_E_FORMAT, "implausible number of attribute lists");
	if (b.nb_face > 255 || b.nb_vtx > 255 || b.nb_corner > 255) throw Error(HRY_E_UNSUPPORTED, "more than 255 lists bound to one region");
	const bool ply_layout = nrf == 1 && nrv == 1 && b.nfacelists(0) == 1 && b.ncornerlists(0) == 0 && b.nvtxlists(0) == 1 &&
	                        b.facelist(0, 0) == 0 && b.vtxlist(0, 0) == 1;
	m.general = !ply_layout;
	m.lists.assign(ply_layout ? 2 : target.size(), AttrList());
	for (size_t l = 0; l < m.lists.size(); ++l) {
		AttrList &L = m.lists[l];
		L.target = ply_layout ? (int)l : target[l];
		if (L.target == 3) { L.have_bounds = true; continue; }   // reader.cc:128-166: a list no region names has nothing in the file
		L.count = r.v<uint32_t>();
		uint16_t nc = r.v<uint16_t>();
		if (nc > kMaxComp) throw Error(HRY_E_UNSUPPORTED, "too many components in one attribute list");
		for (int c = 0; c < nc; ++c) {
			uint8_t t = r.v<uint8_t>(), q = r.v<uint8_t>();
			if (t >= C_NONE || q > 64) throw Error(HRY_E_FORMAT, "bad component descriptor");
			L.add_comp((CompType)t, q);
		}
		uint16_t ni = r.v<uint16_t>();
		int off = 0;
		for (int j = 0; j < ni; ++j) {
			uint16_t len = r.v<uint16_t>();
			for (int k = 0; k < len; ++k) L.add_interp(j, off + k);
			off += len;
			if (j >= kInterpOther) {
				uint32_t sl = r.v<uint32_t>();
				r.need(sl);
				std::string nm((const char*)r.p, sl);
				r.p += sl;
				if (j >= (int)L.interp_off.size()) { L.interp_off.resize(j + 1, -1); L.interp_len.resize(j + 1, 0); }
				if ((int)L.interp_name.size() < j - kInterpOther + 1) L.interp_name.resize(j - kInterpOther + 1);
				L.interp_name[j - kInterpOther] = nm;
			}
		}
		if (off > nc) throw Error(HRY_E_FORMAT, "interpretation table exceeds component count");
		if (alloc_records) L.data.assign((size_t)L.count * L.stride(), 0);
		L.bmin.resize(L.stride()); L.bmax.resize(L.stride());
		r.raw(L.bmin.data(), L.bmin.size());
		r.raw(L.bmax.data(), L.bmax.size());
		L.have_bounds = true;
	}
	if (!m.general && (m.lists[0].count != m.nf || m.lists[1].count != m.nv)) m.general = true;   // shared records: the general decoder
	if (m.general) {
		m.bind = std::move(b);
		if (alloc_records) {
			m.bind.face_reg.assign(m.nf, 0); m.bind.vtx_reg.assign(m.nv, 0);
			m.bind.face_attr.assign((size_t)m.nf * m.bind.nb_face, 0); m.bind.vtx_attr.assign((size_t)m.nv * m.bind.nb_vtx, 0);
		}
	}
	uint16_t cnt = r.v<uint16_t>();
	m.have_degree.clear();
	for (int i = 0; i < cnt; ++i) {
		uint16_t d = r.v<uint16_t>();
		if (d < 3 || d > 255) throw Error(HRY_E_UNSUPPORTED, "polygon degree outside 3..255");
		if (d >= m.have_degree.size()) m.have_degree.resize(d + 1, 0);
		m.have_degree[d] = 1;
	}
	return (size_t)(r.p - p);
}

// ---- static priors (chunked container) ----------------------------------------------------------------------------
bool plane_prior_from_hist(const uint32_t hist[256], uint64_t n, uint32_t table[256])
{
	for (int s = 0; s < 256; ++s) table[s] = 0;
	if (n < kPriorMinSyms) return false;
	for (int s = 0; s < 256; ++s)
		if (hist[s]) table[s] = (uint32_t)std::max<uint64_t>(1, ((uint64_t)hist[s] * kPriorK + n / 2) / n);
	return true;
}
// u8 mode (0 = reference initial counts, 1 = prior); prior: 32-byte bitmap of the symbols present (bit s & 7 of byte s >> 3),
// then one value per present symbol in symbol order: u8 if < 255, else 255 followed by u16
void write_prior(std::vector<uint8_t> &out, bool use, const uint32_t table[256])
{
	out.push_back(use ? 1 : 0);
	if (!use) return;
	uint8_t bm[32] = { 0 };
	for (int s = 0; s < 256; ++s) if (table[s]) bm[s >> 3] |= (uint8_t)(1u << (s & 7));
	out.insert(out.end(), bm, bm + 32);
	for (int s = 0; s < 256; ++s) {
		if (!table[s]) continue;
		if (table[s] < 255) out.push_back((uint8_t)table[s]);
		else { out.push_back(255); out.push_back((uint8_t)(table[s] & 0xff)); out.push_back((uint8_t)(table[s] >> 8)); }
	}
}
size_t read_prior(const uint8_t *p, size_t avail, bool &use, uint32_t table[256])
{
	for (int s = 0; s < 256; ++s) table[s] = 0;
	size_t k = 0;
	auto need = [&](size_t n) { if (k + n > avail) throw Error(HRY_E_FORMAT, "truncated chunked directory"); };
	need(1);
	const uint8_t mode = p[k++];
	use = mode == 1;
	if (mode == 0) return k;
	if (mode != 1) throw Error(HRY_E_FORMAT, "corrupt chunked directory (prior mode)");
	need(32);
	const uint8_t *bm = p + k;
	k += 32;
	for (int s = 0; s < 256; ++s) {
		if (!(bm[s >> 3] & (1u << (s & 7)))) continue;
		need(1);
		uint32_t v = p[k++];
		if (v == 255) { need(2); v = (uint32_t)p[k] | ((uint32_t)p[k + 1] << 8); k += 2; }
		if (v == 0) throw Error(HRY_E_FORMAT, "corrupt chunked directory (prior count)");
		table[s] = v;
	}
	return k;
}

// ---- border snapshots in the chunked container's directory (host.hpp BorderSnapshot; the oracle restates the form)
//   u32 spacing, u32 n, u32 bytes, then `bytes` bytes of LEB128 varints, then zero bytes up to a multiple of four (from the section's
//   first byte).  Per snapshot: the cursors of a restart point -- symbols consumed per plane group (5) and operation class (8), next
//   vertex, face, half-edge --, the number of counters and the (vertex, counter) pairs, the number of parts and of elements; per part
//   `size << 1 | edge_begin`; then the elements' vertices as differences from "the vertex before + 1" (before the first element: the
//   snapshot's next vertex), zigzag-folded, in runs -- `length of a run of zeros`, then the non-zero value that ends it (nothing
//   behind the last element); then the triangle counts (0 .. 9) the same way, as differences from the count before (before the
//   first: 3).  Along a border the vertices mostly follow each other and have seen three triangles: a snapshot of a thousand
//   elements is some eighty bytes, half of them its cursors.
namespace {
void put_varint(std::vector<uint8_t> &o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
int64_t unzigzag(uint64_t z) { return (int64_t)(z >> 1) ^ -(int64_t)(z & 1); }
template <typename F> void put_runs(std::vector<uint8_t> &o, size_t n, F &&diff)   // diff(i): the i-th value's difference from its prediction
{
	for (size_t i = 0; i < n;) {
		uint64_t run = 0;
		while (i < n && diff(i) == 0) { ++run; ++i; }
		put_varint(o, run);
		if (i < n) { put_varint(o, zigzag(diff(i))); ++i; }
	}
}
struct VarintReader {
	const uint8_t *p, *end;
	uint64_t get()
	{
		uint64_t v = 0;
		for (int sh = 0; sh < 64; sh += 7) {
			if (p == end) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
			const uint8_t b = *p++;
			v |= (uint64_t)(b & 0x7f) << sh;
			if (!(b & 0x80)) return v;
		}
		throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
	}
};
}   // namespace

void write_snapshot_section(uint32_t spacing, const std::vector<BorderSnapshot> &snaps, const std::vector<RestartCounters> &counters, std::vector<uint8_t> &out)
{
	const size_t sec0 = out.size();
	auto put32 = [&](uint32_t v) { const uint8_t *q = (const uint8_t*)&v; out.insert(out.end(), q, q + 4); };
	std::vector<uint8_t> body;
	for (size_t k = 0; k < snaps.size(); ++k) {
		const BorderSnapshot &S = snaps[k];
		for (int g = 0; g < G_COUNT; ++g) put_varint(body, S.n_grp[g]);
		for (int i = 0; i < 8; ++i) put_varint(body, S.n_op[i]);
		put_varint(body, S.first_vertex); put_varint(body, S.first_face); put_varint(body, S.first_halfedge);
		const RestartCounters none, &cs = k < counters.size() ? counters[k] : none;
		put_varint(body, cs.size());
		for (const auto &c : cs) { put_varint(body, c.first); put_varint(body, c.second); }
		put_varint(body, S.parts.size()); put_varint(body, S.vtx.size());
		for (uint32_t pt : S.parts) put_varint(body, pt);
		put_runs(body, S.vtx.size(), [&](size_t i) { return (int64_t)S.vtx[i] - ((int64_t)(i ? S.vtx[i - 1] : S.first_vertex - 1u) + 1); });
		put_runs(body, S.seen.size(), [&](size_t i) { return (int64_t)S.seen[i] - (int64_t)(i ? S.seen[i - 1] : 3); });
	}
	put32(spacing); put32((uint32_t)snaps.size()); put32((uint32_t)body.size());
	out.insert(out.end(), body.begin(), body.end());
	while ((out.size() - sec0) & 3) out.push_back(0);
}

size_t read_snapshot_section(const uint8_t *p, size_t avail, uint32_t nv, uint32_t &spacing, std::vector<SnapshotPoint> &out)
{
	if (avail < 12) throw Error(HRY_E_FORMAT, "truncated chunked directory");
	uint32_t n, nb;
	memcpy(&spacing, p, 4); memcpy(&n, p + 4, 4); memcpy(&nb, p + 8, 4);
	const size_t total = (12 + (size_t)nb + 3) & ~(size_t)3;
	if (nb > avail - 12 || total > avail || (uint64_t)n * 20 > nb) throw Error(HRY_E_FORMAT, "truncated chunked directory");
	for (size_t k = 12 + nb; k < total; ++k) if (p[k]) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
	VarintReader rd{ p + 12, p + 12 + nb };
	auto get32 = [&]() -> uint32_t { const uint64_t v = rd.get(); if (v > 0xffffffffull) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)"); return (uint32_t)v; };
	out.clear(); out.resize(n);
	for (uint32_t q = 0; q < n; ++q) {
		SnapshotPoint &S = out[q];
		for (int g = 0; g < G_COUNT; ++g) S.at.n_grp[g] = get32();
		for (int i = 0; i < 8; ++i) S.at.n_op[i] = get32();
		S.at.first_vertex = get32(); S.at.first_face = get32(); S.at.first_halfedge = get32(); S.at.flags = 0;
		const uint32_t nc = get32();
		if ((uint64_t)nc * 2 > nb) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
		S.counters.resize(nc);
		for (uint32_t j = 0; j < nc; ++j) { const uint32_t a = get32(), b = get32(); S.counters[j] = { a, b }; }
		const uint32_t n_parts = get32(), n_elems = get32();
		// (a run covers any number of elements in a byte or two: the bounds are the mesh's)
		if (n_parts == 0 || n_parts > n_elems || n_elems > (uint64_t)2 * nv + 4) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
		S.parts.resize(n_parts);
		uint64_t sum = 0;
		for (uint32_t i = 0; i < n_parts; ++i) {
			const uint64_t v = rd.get();
			if (v >> 1 < 1 || v >> 1 > n_elems) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
			S.parts[i] = (uint32_t)v;
			sum += v >> 1;
		}
		if (sum != n_elems) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
		S.vtx.resize(n_elems); S.seen.resize(n_elems);
		for (uint32_t i = 0; i < n_elems;) {
			const uint64_t run = rd.get();
			if (run > n_elems - i) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
			for (uint64_t r = 0; r < run; ++r, ++i) S.vtx[i] = (i ? S.vtx[i - 1] : S.at.first_vertex - 1u) + 1u;
			if (i < n_elems) { S.vtx[i] = (uint32_t)((int64_t)(i ? S.vtx[i - 1] : S.at.first_vertex - 1u) + 1 + unzigzag(rd.get())); ++i; }
		}
		for (uint32_t i = 0; i < n_elems; ++i) if (S.vtx[i] >= S.at.first_vertex || S.vtx[i] >= nv) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
		for (uint32_t i = 0; i < n_elems;) {
			const uint64_t run = rd.get();
			if (run > n_elems - i) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
			for (uint64_t r = 0; r < run; ++r, ++i) S.seen[i] = i ? S.seen[i - 1] : 3;
			if (i < n_elems) {
				const int64_t v = (int64_t)(i ? S.seen[i - 1] : 3) + unzigzag(rd.get());
				if (v < 0 || v > 9) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
				S.seen[i] = (uint8_t)v; ++i;
			}
		}
	}
	if (rd.p != rd.end) throw Error(HRY_E_FORMAT, "corrupt chunked directory (border snapshot)");
	return total;
}

}   // namespace hry
