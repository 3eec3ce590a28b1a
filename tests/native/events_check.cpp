// general_events.cpp against a plain restatement of the same bookkeeping (a push_back per symbol, a list per corner slot): every
// array equal, with and without positions.  Also a timing of both: events_check --time FILE.obj
// Built and run by tests/test_obj_cpu.py::test_fast_event_collection_equals_the_plain_one.  Usage: events_check [--time] FILE.obj...
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../../harry_amd/csrc/host/host.hpp"

using namespace hry;

static std::vector<uint8_t> slurp(const std::string &fn)
{
	std::ifstream is(fn, std::ios::binary);
	return std::vector<uint8_t>((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
}

// attrcode.h:321-393,395-416 without the values, the way one would write it down first
static void plain_events(const Mesh &m, const WalkResult &w, uint32_t pos0, Events &E)
{
	const Bindings &b = m.bind;
	static constexpr uint32_t NONE = 0xffffffffu;
	E.ls.assign(m.lists.size(), ListStream());
	for (size_t l = 0; l < m.lists.size(); ++l) { E.ls[l].nbytes = (uint32_t)m.lists[l].coded_bytes(); E.ls[l].first_at.assign(m.lists[l].count, NONE); }
	const bool code_rv = b.nregs_vtx() > 1, code_rf = b.nregs_face() > 1;
	uint32_t pos = pos0;
	auto reference = [&](int l, uint32_t idx) -> bool {
		ListStream &S = E.ls[l];
		if (S.first_at[idx] == NONE) { S.first_at[idx] = S.created++; return false; }
		S.type_sym.push_back(RefKind::hist); S.type_pos.push_back(pos++);
		S.gh_val.push_back(S.created - 1 - S.first_at[idx]); S.gh_pos.push_back(pos);
		pos += 4;
		return true;
	};
	auto data = [&](int l, uint32_t idx, uint32_t he, int slot) {
		ListStream &S = E.ls[l];
		S.type_sym.push_back(RefKind::data); S.type_pos.push_back(pos++);
		S.d_pos.push_back(pos); S.d_idx.push_back(idx); S.d_he.push_back(he); S.d_slot.push_back((RefSlot)slot);
		pos += S.nbytes;
	};
	for (uint32_t e : w.order_v) {
		const uint32_t v = m.org[e];
		const int r = b.vtx_reg[v];
		if (code_rv) { E.rv_sym.push_back((RefKind)(uint8_t)r); E.rv_pos.push_back(pos++); }
		for (int a = 0; a < b.nvtxlists(r); ++a) {
			const int l = b.vtxlist(r, a);
			const uint32_t idx = b.vtx_attr[(size_t)v * b.nb_vtx + a];
			if (!reference(l, idx)) data(l, idx, e, a);
		}
	}
	struct Node { uint32_t idx, next; };
	std::vector<Node> pool;
	std::vector<std::vector<uint32_t>> head(b.nb_corner);
	for (auto &h : head) h.assign(m.nv, NONE);
	std::vector<uint32_t> eface(m.ne());
	for (uint32_t f = 0; f < m.nf; ++f) for (uint32_t e = m.face_off[f]; e < m.face_off[f + 1]; ++e) eface[e] = f;
	for (uint32_t e0 : w.order_f) {
		const uint32_t f = eface[e0];
		const int r = b.face_reg[f];
		if (code_rf) { E.rf_sym.push_back((RefKind)(uint8_t)r); E.rf_pos.push_back(pos++); }
		for (int a = 0; a < b.nfacelists(r); ++a) {
			const int l = b.facelist(r, a);
			const uint32_t idx = b.face_attr[(size_t)f * b.nb_face + a];
			if (!reference(l, idx)) data(l, idx, f, a);
		}
		const uint32_t fb = m.face_off[f], fe = m.face_off[f + 1];
		uint32_t c = e0;
		do {
			const uint32_t v = m.org[c];
			for (int a = 0; a < b.ncornerlists(r); ++a) {
				const int l = b.cornerlist(r, a);
				const uint32_t idx = b.corner_attr[(size_t)c * b.nb_corner + a];
				uint32_t back = 0, k = head[a][v];
				while (k != NONE && pool[k].idx != idx) { k = pool[k].next; ++back; }
				if (k != NONE) {
					ListStream &S = E.ls[l];
					S.type_sym.push_back(RefKind::lhist); S.type_pos.push_back(pos++);
					S.lh_val.push_back(back); S.lh_pos.push_back(pos);
					pos += 2;
					continue;
				}
				pool.push_back(Node{ idx, head[a][v] });
				head[a][v] = (uint32_t)pool.size() - 1;
				if (!reference(l, idx)) data(l, idx, c, a);
			}
			c = c + 1 == fe ? fb : c + 1;
		} while (c != e0);
	}
	E.end_pos = pos;
}

template <typename V> static bool same(const V &a, const V &b) { return a.size() == b.size() && (a.empty() || !memcmp(a.data(), b.data(), a.size() * sizeof(typename V::value_type))); }

int main(int argc, char **argv)
{
	bool timing = false;
	int done = 0;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--time")) { timing = true; continue; }
		const std::string fn = argv[i];
		const std::vector<uint8_t> data = slurp(fn);
		const std::string dir = fn.substr(0, fn.find_last_of('/'));
		try {
			std::unique_ptr<Mesh> m(mesh_from_obj(data.data(), data.size(), dir.c_str()));
			ensure_twins(*m);
			WalkResult w;
			cut_border_walk(*m, w, false);
			for (int wp = 0; wp < 2; ++wp) {
				Events A, B;
				plain_events(*m, w, 17, A);
				collect_events(*m, w, 17, wp != 0, B);
				bool ok = A.ls.size() == B.ls.size() && same(A.rv_sym, B.rv_sym) && same(A.rf_sym, B.rf_sym) && (!wp || (A.end_pos == B.end_pos && same(A.rv_pos, B.rv_pos) && same(A.rf_pos, B.rf_pos)));
				for (size_t l = 0; ok && l < A.ls.size(); ++l) {
					const ListStream &x = A.ls[l], &y = B.ls[l];
					ok = same(x.type_sym, y.type_sym) && same(x.gh_val, y.gh_val) && same(x.lh_val, y.lh_val) && same(x.d_idx, y.d_idx) && same(x.d_he, y.d_he) && same(x.d_slot, y.d_slot) &&
					     same(x.first_at, y.first_at) && x.created == y.created && x.nbytes == y.nbytes;
					if (wp) ok = ok && same(x.type_pos, y.type_pos) && same(x.gh_pos, y.gh_pos) && same(x.lh_pos, y.lh_pos) && same(x.d_pos, y.d_pos);
				}
				if (!ok) { fprintf(stderr, "%s: the fast collection differs from the plain one (positions %d)\n", fn.c_str(), wp); return 1; }
			}
			if (timing) {
				typedef std::chrono::steady_clock Clock;
				double best[3] = { 1e9, 1e9, 1e9 };
				for (int rep = 0; rep < 7; ++rep)
					for (int k = 0; k < 3; ++k) {
						Events E;
						const auto t = Clock::now();
						if (k == 0) plain_events(*m, w, 0, E); else collect_events(*m, w, 0, k == 2, E);
						const double ms = std::chrono::duration<double, std::milli>(Clock::now() - t).count();
						if (ms < best[k]) best[k] = ms;
					}
				const double ntri = (double)m->ne() - 2.0 * m->nf;
				printf("%s: %.0f triangles: plain %.3f ms (%.1f ns per triangle), fast without positions %.3f ms (%.1f), with %.3f ms (%.1f)\n", fn.c_str(), ntri, best[0],
				       best[0] * 1e6 / ntri, best[1], best[1] * 1e6 / ntri, best[2], best[2] * 1e6 / ntri);
			}
			++done;
		} catch (const Error &e) { fprintf(stderr, "%s: %s\n", fn.c_str(), e.what()); return 1; }
	}
	printf("ok %d files\n", done);
	return 0;
}
