// OBJ reader and writer of the product: the reference's formats/obj/reader.rl:27-299 and
// formats/obj/writer.cc:20-132 restated over the flat mesh model with general bindings (mesh.hpp Bindings).
//
// The reader follows the reference's scanner, not the OBJ specification -- byte parity with the reference on the same input
// is the contract.  What that means in practice (each point probed against the unmodified reference binary, tests/golden/obj):
//   * numbers: [+-] digits [. digits] | [+-] . digits, then optionally e|E, a MANDATORY sign and digits; the sign of the
//     exponent is ignored ("1e-2" is 100, "1e2" is a syntax error).  Evaluated in double as the scanner does
//     (integer part + fraction / 10^k, times sign * 10^exp), then narrowed to float              (reader.rl:32-50)
//   * "v" takes 3, 4, 6, 7 or 8 numbers (one list and one vertex region per count; beyond x y z [w] they are colours),
//     "vt" 2 or 3, "vn" 3; "o s g l p" need a blank after the keyword; unknown keywords, a stray '\r' and leading blanks
//     before a keyword are "Unable to parse this OBJ file"; a last line without line feed is ignored   (reader.rl:61-79)
//   * a corner "v/t" ALSO names normal t (the scanner runs "v/t" and "v//n"-with-empty-texture side by side): it fails with
//     "n too big" when fewer than t normals exist; "v/t/" names the texture only                     (reader.rl:54-56)
//   * "usemtl" of a material no "mtllib" file defined selects material 0; material names keep trailing blanks;
//     materials are read from <dir>/<name> with the first blank-delimited word after "newmtl"        (reader.rl:160-190)
//   * face regions are looked up by material and (texture list << 3 | normal list), 9 = none, which confuses some pairs
//     of lists exactly as the reference does                                                      (reader.rl:229-237)
// Outside the supported subset (HRY_E_UNSUPPORTED): "nan" / "inf" literals (the sign state of "inf" is undefined in the
// reference), faces with fewer than 3 corners, faces whose corners do not all carry the same kinds of index (the reference
// reads past the end of its index arrays there).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <unordered_map>

#include "host.hpp"

namespace hry {
namespace {

enum { I_POS = 0, I_NORMAL = 1, I_COLOR = 2, I_TEX = 16 };   // structs/mixing.h:20-39
enum { K_VERTEX = 0, K_TEX = 1, K_NORMAL = 2, NO_LIST = 9 };

[[noreturn]] void syntax() { throw Error(HRY_E_FORMAT, "Unable to parse this OBJ file"); }

struct Scan {
	const char *p, *end;
	static bool blank(char c) { return c == ' ' || c == '\t'; }
	bool done() const { return p == end; }
	bool is_digit() const { return p != end && (unsigned)(*p - '0') < 10u; }
	bool blanks() { const char *b = p; while (p != end && blank(*p)) ++p; return p != b; }
	float number()
	{
		if (p != end && (*p == 'n' || *p == 'N' || *p == 'i' || *p == 'I'))
			throw Error(HRY_E_UNSUPPORTED, "nan / inf literals in an OBJ file are outside the supported subset");
		double sign = 1, whole = 0, frac = 0, den = 1, ex = 0, scale = 1;
		if (p != end && (*p == '+' || *p == '-')) sign = *p++ == '-' ? -1 : 1;
		bool any = false;
		while (is_digit()) { whole = whole * 10 + (*p++ - '0'); any = true; }
		if (p != end && *p == '.') {
			++p;
			bool fd = false;
			while (is_digit()) { frac = frac * 10 + (*p++ - '0'); den *= 10; fd = true; }
			if (!any && !fd) syntax();
		} else if (!any) syntax();
		if (p != end && (*p == 'e' || *p == 'E')) {
			++p;
			if (p == end || (*p != '+' && *p != '-')) syntax();
			++p;
			if (!is_digit()) syntax();
			while (is_digit()) ex = ex * 10 + (*p++ - '0');
			scale = std::pow(10.0, ex);
		}
		double v = whole + frac / den;
		v *= sign * scale;
		return (float)v;
	}
	bool integer(int &out)
	{
		const char *q = p;
		bool neg = q != end && *q == '-';
		if (neg) ++q;
		if (q == end || (unsigned)(*q - '0') >= 10u) return false;
		unsigned v = 0;
		while (q != end && (unsigned)(*q - '0') < 10u) v = v * 10u + (unsigned)(*q++ - '0');
		p = q;
		out = neg ? -(int)v : (int)v;
		return true;
	}
};

int resolve(int n, int have)   // reader.rl:91-103
{
	if (n == 0) throw Error(HRY_E_FORMAT, "index cannot be 0");
	if (n > have) throw Error(HRY_E_FORMAT, "n too big");
	if (n < 0) {
		if (have + n < 0) throw Error(HRY_E_FORMAT, "n too small");
		return have + n;
	}
	return n - 1;
}

struct Loader {
	Mesh &m;
	std::string dir;
	int list_of[3][9];
	int vreg_of[9];
	std::vector<std::vector<int>> freg_of;   // per material: 256 lookup keys
	std::vector<std::pair<int, uint32_t>> tex_at, normal_at;   // (list, record) of every vt / vn line
	std::unordered_map<std::string, int> material;
	int cur_material = 0;
	std::vector<std::vector<uint8_t>> rec;   // growing record storage per list (moved into the lists at the end)
	std::vector<uint16_t> vreg, freg;
	std::vector<uint32_t> vattr, cattr, foff{ 0 }, org;

	Loader(Mesh &mesh, const std::string &d) : m(mesh), dir(d)
	{
		for (auto &row : list_of) for (int &x : row) x = NO_LIST;
		for (int &x : vreg_of) x = -1;
		freg_of.emplace_back(256, -1);
		m.lists.clear();
		m.general = true;
		m.bind = Bindings();
		m.bind.nb_face = 0; m.bind.nb_vtx = 1; m.bind.nb_corner = 2;   // reader.rl:257
	}
	int list_for(int kind, int n)   // reader.rl:132-148
	{
		int &l = list_of[kind][n];
		if (l != NO_LIST) return l;
		AttrList L;
		L.target = kind == K_VERTEX ? 1 : 2;
		static const int base[3] = { I_POS, I_TEX, I_NORMAL };
		for (int i = 0; i < n; ++i) {
			L.add_comp(C_FLOAT);
			L.add_interp(kind == K_VERTEX && n > 4 && i >= 3 ? I_COLOR : base[kind], i);
		}
		m.lists.push_back(std::move(L));
		rec.emplace_back();
		return l = (int)m.lists.size() - 1;
	}
	uint32_t append(int l, const float *c, int n)
	{
		const uint8_t *b = (const uint8_t*)c;
		rec[l].insert(rec[l].end(), b, b + 4 * (size_t)n);
		return m.lists[l].count++;
	}
	void vertex(const float *c, int n)   // reader.rl:191-204
	{
		const int l = list_for(K_VERTEX, n);
		const uint32_t idx = append(l, c, n);
		if (vreg_of[n] < 0) { vreg_of[n] = m.bind.add_vtx_region(1); m.bind.reg_vtxlist[m.bind.off_vtxlist[vreg_of[n]]] = (uint16_t)l; }
		vreg.push_back((uint16_t)vreg_of[n]);
		vattr.push_back(idx);
	}
	void use_material(const std::string &name) { auto it = material.find(name); cur_material = it == material.end() ? 0 : it->second; }
	void load_materials(const std::string &name)   // reader.rl:170-190
	{
		std::ifstream is(dir + "/" + name);
		if (!is) return;
		while (!is.eof()) {
			std::string word;
			is >> word;
			if (word == "newmtl") {
				std::string nm;
				is >> nm;
				material[nm] = (int)freg_of.size();
				freg_of.emplace_back(256, -1);
			} else is.ignore(std::numeric_limits<std::streamsize>::max(), '\n');
		}
	}
	void face(const std::vector<int> &vi, const std::vector<int> &ti, const std::vector<int> &ni)   // reader.rl:217-249
	{
		const size_t corners = vi.size();
		const bool ht = !ti.empty(), hn = !ni.empty();
		if (corners < 3) throw Error(HRY_E_UNSUPPORTED, "OBJ faces with fewer than 3 corners are outside the supported subset");
		if (corners > 255) throw Error(HRY_E_UNSUPPORTED, "polygon degree outside 3..255");
		if ((ht && ti.size() != corners) || (hn && ni.size() != corners))
			throw Error(HRY_E_UNSUPPORTED, "OBJ face whose corners do not all carry the same kinds of index");
		int tl = NO_LIST, nl = NO_LIST;
		if (ht) tl = tex_at[ti[0]].first;
		if (hn) nl = normal_at[ni[0]].first;
		for (size_t c = 1; c < corners; ++c) {
			if (ht && tex_at[ti[c]].first != tl) throw Error(HRY_E_FORMAT, "Inconsistent texture attribute types in face");
			if (hn && normal_at[ni[c]].first != nl) throw Error(HRY_E_FORMAT, "Inconsistent normal attribute types in face");
		}
		int &r = freg_of[cur_material][(tl << 3) | nl];
		const int ta = 0, na = ht ? 1 : 0;
		if (r < 0) {
			r = m.bind.add_face_region(0, (ht ? 1 : 0) + (hn ? 1 : 0));
			if (ht) m.bind.reg_cornerlist[m.bind.off_cornerlist[r] + ta] = (uint16_t)tl;
			if (hn) m.bind.reg_cornerlist[m.bind.off_cornerlist[r] + na] = (uint16_t)nl;
		}
		freg.push_back((uint16_t)r);
		for (size_t c = 0; c < corners; ++c) {
			org.push_back((uint32_t)vi[c]);
			uint32_t slot[2] = { 0, 0 };
			if (ht) slot[ta] = tex_at[ti[c]].second;
			if (hn) slot[na] = normal_at[ni[c]].second;
			cattr.push_back(slot[0]); cattr.push_back(slot[1]);
		}
		foff.push_back((uint32_t)org.size());
		if (corners >= m.have_degree.size()) m.have_degree.resize(corners + 1, 0);
		m.have_degree[corners] = 1;
	}
	void finish()
	{
		m.nv = (uint32_t)vreg.size(); m.nf = (uint32_t)freg.size();
		m.face_off.assign(foff.begin(), foff.end());
		m.org.assign(org.begin(), org.end());
		m.bind.vtx_reg.assign(vreg.begin(), vreg.end()); m.bind.face_reg.assign(freg.begin(), freg.end());
		m.bind.vtx_attr.assign(vattr.begin(), vattr.end()); m.bind.corner_attr.assign(cattr.begin(), cattr.end());
		for (size_t l = 0; l < m.lists.size(); ++l) m.lists[l].data.assign(rec[l].begin(), rec[l].end());
		m.twins_pending = true;
	}
};

bool starts(const Scan &s, const char *kw)   // the keyword, then a blank or the end of the line
{
	size_t k = strlen(kw);
	if ((size_t)(s.end - s.p) < k || memcmp(s.p, kw, k) != 0) return false;
	return (size_t)(s.end - s.p) == k || Scan::blank(s.p[k]);
}

}   // namespace

Mesh *mesh_from_obj(const uint8_t *buf, size_t n, const char *directory)
{
	std::unique_ptr<Mesh> m(new Mesh());
	Loader ld(*m, directory ? directory : "");
	int seen[3] = { 0, 0, 0 };
	std::vector<int> idx[3];
	float val[8];
	const char *p = (const char*)buf, *end = p + n;
	while (p != end) {
		const char *nl = (const char*)memchr(p, '\n', (size_t)(end - p));
		if (!nl) break;   // an unfinished last line never reaches the scanner's end-of-line actions
		Scan s{ p, nl };
		p = nl + 1;
		if (s.end != s.p && s.end[-1] == '\r') --s.end;
		if (memchr(s.p, '\r', (size_t)(s.end - s.p))) syntax();
		if (s.done() || *s.p == '#') continue;
		if (Scan::blank(*s.p)) { s.blanks(); if (!s.done()) syntax(); continue; }
		if (starts(s, "usemtl") || starts(s, "mtllib")) {
			const bool use = *s.p == 'u';
			s.p += 6;
			const char *after = s.p;
			if (!s.blanks()) syntax();
			std::string name(s.p, s.end);
			if (name.empty()) {   // blanks only: the first is the separator, the name is the last one
				if (s.end - after < 2) syntax();
				name.assign(1, s.end[-1]);
			}
			if (use) ld.use_material(name); else ld.load_materials(name);
			continue;
		}
		const int kind = starts(s, "vt") ? K_TEX : starts(s, "vn") ? K_NORMAL : starts(s, "v") ? K_VERTEX : -1;
		if (kind >= 0) {
			s.p += kind == K_VERTEX ? 1 : 2;
			int nc = 0;
			while (s.blanks() && !s.done()) {
				if (nc == 8) syntax();
				val[nc++] = s.number();
			}
			if (!s.done()) syntax();
			const bool ok = kind == K_VERTEX ? (nc == 3 || nc == 4 || (nc >= 6 && nc <= 8)) : kind == K_TEX ? (nc == 2 || nc == 3) : nc == 3;
			if (!ok) syntax();
			++seen[kind];
			if (kind == K_VERTEX) ld.vertex(val, nc);
			else {
				const int l = ld.list_for(kind, nc);
				(kind == K_TEX ? ld.tex_at : ld.normal_at).push_back(std::make_pair(l, ld.append(l, val, nc)));
			}
			continue;
		}
		if (starts(s, "f")) {
			s.p += 1;
			for (auto &v : idx) v.clear();
			while (s.blanks() && !s.done()) {
				int a;
				if (!s.integer(a)) syntax();
				idx[K_VERTEX].push_back(resolve(a, seen[K_VERTEX]));
				int slashes = 0, last = 0;
				for (int k = K_TEX; k <= K_NORMAL && !s.done() && *s.p == '/'; ++k) {
					++s.p; ++slashes;
					if (s.integer(a)) { idx[k].push_back(resolve(a, seen[k])); last = k; }
				}
				if (slashes == 1 && last == K_TEX) idx[K_NORMAL].push_back(resolve(a, seen[K_NORMAL]));   // "v/t" is also "v//t"
			}
			if (!s.done() || idx[K_VERTEX].size() < 2) syntax();
			ld.face(idx[K_VERTEX], idx[K_TEX], idx[K_NORMAL]);
			continue;
		}
		if (starts(s, "o") || starts(s, "s") || starts(s, "g") || starts(s, "l") || starts(s, "p")) {
			if (s.end - s.p < 2) syntax();
			continue;
		}
		syntax();
	}
	ld.finish();
	return m.release();
}

// ---- writer (formats/obj/writer.cc:20-132) ---------------------------------------------------------------------
// The writer keeps the reference's index arithmetic: normals are numbered after ALL texture coordinates (one running
// offset for both tables, writer.cc:68-73,89-93), which is what the reference's files hold -- an OBJ reader that numbers
// "vn" lines on their own sees indices that are too large by the number of "vt" lines.
void mesh_to_obj(const Mesh &m, ByteSink &out)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	const size_t nl = m.lists.size();
	auto has = [&](size_t l, int interp) { const AttrList &L = m.lists[l]; return interp < (int)L.interp_len.size() && L.interp_len[interp] != 0; };
	auto len_of = [&](size_t l, int interp) { const AttrList &L = m.lists[l]; return interp < (int)L.interp_len.size() ? L.interp_len[interp] : 0; };
	std::vector<char> is_pos(nl, 0), is_tex(nl, 0), is_normal(nl, 0);
	// region tables of the PLY layout, spelled out
	Bindings ply;
	const Bindings *b = &m.bind;
	if (!m.general) {
		ply.add_face_region(1, 0); ply.reg_facelist[0] = 0;
		ply.add_vtx_region(1); ply.reg_vtxlist[0] = 1;
		ply.nb_face = 1; ply.nb_vtx = 1;
		b = &ply;
	}
	for (int r = 0; r < b->nregs_face(); ++r)
		for (int a = 0; a < b->ncornerlists(r); ++a) {
			int l = b->cornerlist(r, a);
			if (has(l, I_TEX)) is_tex[l] = 1;
			if (has(l, I_NORMAL)) is_normal[l] = 1;
		}
	for (int r = 0; r < b->nregs_vtx(); ++r)
		for (int a = 0; a < b->nvtxlists(r); ++a) if (has(b->vtxlist(r, a), I_POS)) is_pos[b->vtxlist(r, a)] = 1;
	std::string o = "# decompressed using harry mesh compressor\n\n# vertex definitions and vertex attributes\n";
	auto flush = [&]() { out.append(o.begin(), o.end()); o.clear(); };
	auto value = [&](size_t l, uint32_t idx, int c) { const AttrList &L = m.lists[l]; print_component(o, L, L.data.data() + (size_t)idx * L.stride(), c); };
	for (size_t l = 0; l < nl; ++l) if ((size_t)m.lists[l].count * m.lists[l].stride() > m.lists[l].data.size()) throw Error(HRY_E_FORMAT, "attribute list shorter than its record count");
	for (uint32_t v = 0; v < m.nv; ++v) {
		const int r = m.general ? m.bind.vtx_reg[v] : 0;
		o += 'v';
		int cnt = 0;
		for (int a = 0; a < b->nvtxlists(r); ++a) {
			const int l = b->vtxlist(r, a);
			if (!is_pos[l]) continue;
			const uint32_t idx = m.general ? m.bind.vtx_attr[(size_t)v * m.bind.nb_vtx + a] : v;
			const AttrList &L = m.lists[l];
			// a header may announce vertices and an empty list (a decoded vertex no face reaches keeps record 0): nothing to print from
			if (idx >= L.count || (size_t)(idx + 1) * L.stride() > L.data.size()) throw Error(HRY_E_FORMAT, "vertex names a record its list does not hold");
			for (int k = 0; k < len_of(l, I_POS); ++k) { if (cnt++ >= 4) break; o += ' '; value(l, idx, L.interp_off[I_POS] + k); }
			for (int k = 0; k < len_of(l, I_COLOR); ++k) { if (cnt++ >= 8) break; o += ' '; value(l, idx, L.interp_off[I_COLOR] + k); }
			break;
		}
		o += '\n';
		if (o.size() > (1u << 20)) flush();
	}
	o += "\n# corner attributes\n# - texture coordinates\n";
	std::vector<uint32_t> tex_base(nl, 0), normal_base(nl, 0);
	uint32_t running = 0;
	for (size_t l = 0; l < nl; ++l) {
		if (!is_tex[l]) continue;
		tex_base[l] = running;
		running += m.lists[l].count;
		const int n = std::min(4, len_of(l, I_TEX));
		for (uint32_t i = 0; i < m.lists[l].count; ++i) {
			o += "vt";
			for (int k = 0; k < n; ++k) { o += ' '; value(l, i, m.lists[l].interp_off[I_TEX] + k); }
			o += '\n';
			if (o.size() > (1u << 20)) flush();
		}
	}
	o += "\n# - normals\n";
	for (size_t l = 0; l < nl; ++l) {
		if (!is_normal[l]) continue;
		normal_base[l] = running;
		running += m.lists[l].count;
		const int n = std::min(4, len_of(l, I_NORMAL));
		for (uint32_t i = 0; i < m.lists[l].count; ++i) {
			o += "vn ";
			for (int k = 0; k < n; ++k) { o += ' '; value(l, i, m.lists[l].interp_off[I_NORMAL] + k); }
			o += '\n';
			if (o.size() > (1u << 20)) flush();
		}
	}
	o += "\n# faces\n";
	for (uint32_t f = 0; f < m.nf; ++f) {
		const int r = m.general ? m.bind.face_reg[f] : 0;
		int ta = -1, na = -1;
		for (int a = 0; a < b->ncornerlists(r); ++a) {
			if (is_tex[b->cornerlist(r, a)]) ta = a;
			if (is_normal[b->cornerlist(r, a)]) na = a;
		}
		o += 'f';
		for (uint32_t e = m.face_off[f]; e < m.face_off[f + 1]; ++e) {
			o += ' '; o += std::to_string(m.org[e] + 1);
			if (ta >= 0) { o += '/'; o += std::to_string(tex_base[b->cornerlist(r, ta)] + m.bind.corner_attr[(size_t)e * m.bind.nb_corner + ta] + 1); }
			if (na >= 0) {
				if (ta < 0) o += '/';
				o += '/'; o += std::to_string(normal_base[b->cornerlist(r, na)] + m.bind.corner_attr[(size_t)e * m.bind.nb_corner + na] + 1);
			}
		}
		o += '\n';
		if (o.size() > (1u << 20)) flush();
	}
	flush();
}

}   // namespace hry
