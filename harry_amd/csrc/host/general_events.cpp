// Which record every element names, along the coding order (attrcode.h:321-393,395-416 without the values): the host's
// bookkeeping of a mesh with general bindings -- regions, shared records, corner lists (what the OBJ reader creates).  Per list
// the stream says for every reference: a new record (DATA: its residual bytes follow; the device computes them), one created
// earlier by its distance in creation order (HIST, GlobalHistory attrcode.h:23-53), or -- at a corner -- one already named at
// this vertex by its distance in the vertex' own list of names (LHIST, LocalHistory :54-80).
//
// Round 5: written for few instructions, like the walk (cbm_walk.cpp): every output array sized once from a count of the
// references, bare cursors instead of a push_back per symbol (eight vectors grew per reference), no character-typed stores in
// the loop (a uint8_t store may alias anything and makes the compiler reload every pointer after it: the reference kinds and the
// slots are an enumeration of that width), one flat table of per-vertex name lists instead of a vector per corner slot, the face
// of a half-edge by division where every face has three or four corners, and the symbols' positions in the ONE stream only where
// a caller asks for them (the reference stream; the parallel container has none).
#include <cstring>

#include "host.hpp"

namespace hry {

void collect_events(const Mesh &m, const WalkResult &w, uint32_t pos0, bool want_positions, Events &E)
{
	const Bindings &b = m.bind;
	static constexpr uint32_t NONE = 0xffffffffu;
	const size_t nl = m.lists.size();
	E.ls.assign(nl, ListStream());
	// ---- how many references every list can get at most: sizes the arrays once
	std::vector<size_t> max_refs(nl, 0);
	{
		std::vector<size_t> per_vreg((size_t)b.nregs_vtx(), 0), per_freg((size_t)b.nregs_face(), 0), corners_freg((size_t)b.nregs_face(), 0);
		for (uint32_t e : w.order_v) ++per_vreg[b.vtx_reg[m.org[e]]];
		int ud = 0;
		const bool uniform = m.uniform_degree(ud) && ud > 0;
		for (uint32_t e0 : w.order_f) {
			const uint32_t f = uniform ? e0 / (uint32_t)ud : 0;
			if (uniform) { ++per_freg[b.face_reg[f]]; corners_freg[b.face_reg[f]] += (size_t)ud; }
		}
		if (!uniform) {   // mixed degrees: every face of a region, whatever its place in the order
			for (uint32_t f = 0; f < m.nf; ++f) { ++per_freg[b.face_reg[f]]; corners_freg[b.face_reg[f]] += m.face_off[f + 1] - m.face_off[f]; }
		}
		for (int r = 0; r < b.nregs_vtx(); ++r) for (int a = 0; a < b.nvtxlists(r); ++a) max_refs[b.vtxlist(r, a)] += per_vreg[r];
		for (int r = 0; r < b.nregs_face(); ++r) {
			for (int a = 0; a < b.nfacelists(r); ++a) max_refs[b.facelist(r, a)] += per_freg[r];
			for (int a = 0; a < b.ncornerlists(r); ++a) max_refs[b.cornerlist(r, a)] += corners_freg[r];
		}
	}
	struct Cur { RefKind *type; uint32_t *type_pos, *gh, *gh_pos, *lh, *lh_pos, *d_pos, *d_idx, *d_he; RefSlot *d_slot; uint32_t *first_at; uint32_t created, nbytes; };
	std::vector<Cur> cur(nl);
	for (size_t l = 0; l < nl; ++l) {
		ListStream &S = E.ls[l];
		S.nbytes = (uint32_t)m.lists[l].coded_bytes();
		S.first_at.resize(m.lists[l].count);
		if (!S.first_at.empty()) memset(S.first_at.data(), 0xff, S.first_at.size() * 4);
		const size_t n = max_refs[l];
		S.type_sym.resize(n); S.gh_val.resize(n); S.d_idx.resize(n); S.d_he.resize(n); S.d_slot.resize(n);
		if (m.lists[l].target == 2) S.lh_val.resize(n);
		if (want_positions) { S.type_pos.resize(n); S.gh_pos.resize(n); S.d_pos.resize(n); if (m.lists[l].target == 2) S.lh_pos.resize(n); }
		cur[l] = Cur{ S.type_sym.data(), S.type_pos.data(), S.gh_val.data(), S.gh_pos.data(), S.lh_val.data(), S.lh_pos.data(), S.d_pos.data(), S.d_idx.data(), S.d_he.data(),
		              S.d_slot.data(), S.first_at.data(), 0u, S.nbytes };
	}
	const bool code_rv = b.nregs_vtx() > 1, code_rf = b.nregs_face() > 1;
	if (code_rv) { E.rv_sym.resize(w.order_v.size()); if (want_positions) E.rv_pos.resize(w.order_v.size()); }
	if (code_rf) { E.rf_sym.resize(w.order_f.size()); if (want_positions) E.rf_pos.resize(w.order_f.size()); }
	RefKind *rv_sym = E.rv_sym.data(), *rf_sym = E.rf_sym.data();
	uint32_t *rv_pos = E.rv_pos.data(), *rf_pos = E.rf_pos.data();
	uint32_t pos = pos0;
	const bool wp = want_positions;
	// a reference to record idx of list l that is not answered by the vertex' own names: HIST if the record exists, else DATA
	auto reference = [&](Cur &C, uint32_t idx, uint32_t he, uint32_t slot) {
		uint32_t &fa = C.first_at[idx];
		if (fa == NONE) {
			fa = C.created++;
			*C.type++ = RefKind::data;
			if (wp) { *C.type_pos++ = pos++; *C.d_pos++ = pos; pos += C.nbytes; }
			*C.d_idx++ = idx; *C.d_he++ = he; *C.d_slot++ = (RefSlot)slot;
		} else {
			*C.type++ = RefKind::hist;
			*C.gh++ = C.created - 1 - fa;
			if (wp) { *C.type_pos++ = pos++; *C.gh_pos++ = pos; pos += 4; }
		}
	};
	const uint32_t *org = m.org.data();
	const uint16_t *vreg = b.vtx_reg.data(), *freg = b.face_reg.data();
	for (uint32_t e : w.order_v) {
		const uint32_t v = org[e];
		const int r = vreg[v];
		if (code_rv) { *rv_sym++ = (RefKind)(uint8_t)r; if (wp) *rv_pos++ = pos++; }
		const int na = b.nvtxlists(r);
		for (int a = 0; a < na; ++a) {
			const int l = b.vtxlist(r, a);
			const uint32_t idx = b.vtx_attr[(size_t)v * b.nb_vtx + a];
			if (idx >= m.lists[l].count) throw Error(HRY_E_ARG, "an element names a record outside its list");
			reference(cur[l], idx, e, (uint32_t)a);
		}
	}
	// per corner slot and vertex: the records named there so far, newest first (LocalHistory, attrcode.h:54-80): one pool of
	// (record, next) nodes, heads in one table [slot][vertex]
	struct Node { uint32_t idx, next; };
	size_t corner_refs = 0;
	for (size_t l = 0; l < nl; ++l) if (m.lists[l].target == 2) corner_refs += max_refs[l];
	BigVec<Node> pool(corner_refs ? corner_refs : 1);
	uint32_t pool_n = 0;
	BigVec<uint32_t> head((size_t)b.nb_corner * m.nv);
	if (!head.empty()) memset(head.data(), 0xff, head.size() * 4);
	int ud = 0;
	const bool uniform = m.uniform_degree(ud) && ud > 0;
	BigVec<uint32_t> eface;
	if (!uniform) {
		eface.resize(m.ne());
		for (uint32_t f = 0; f < m.nf; ++f) for (uint32_t e = m.face_off[f]; e < m.face_off[f + 1]; ++e) eface[e] = f;
	}
	const uint32_t nv = m.nv;
	for (uint32_t e0 : w.order_f) {
		const uint32_t f = uniform ? e0 / (uint32_t)ud : eface[e0];
		const int r = freg[f];
		if (code_rf) { *rf_sym++ = (RefKind)(uint8_t)r; if (wp) *rf_pos++ = pos++; }
		const int nfa = b.nfacelists(r);
		for (int a = 0; a < nfa; ++a) {
			const int l = b.facelist(r, a);
			const uint32_t idx = b.face_attr[(size_t)f * b.nb_face + a];
			if (idx >= m.lists[l].count) throw Error(HRY_E_ARG, "an element names a record outside its list");
			reference(cur[l], idx, f, (uint32_t)a);
		}
		const uint32_t fb = uniform ? f * (uint32_t)ud : m.face_off[f], fe = uniform ? fb + (uint32_t)ud : m.face_off[f + 1];
		const int nca = b.ncornerlists(r);
		if (!nca) continue;
		uint32_t c = e0;
		do {
			const uint32_t v = org[c];
			for (int a = 0; a < nca; ++a) {
				const int l = b.cornerlist(r, a);
				const uint32_t idx = b.corner_attr[(size_t)c * b.nb_corner + a];
				uint32_t &hd = head[(size_t)a * nv + v];
				uint32_t back = 0, k = hd;
				while (k != NONE && pool[k].idx != idx) { k = pool[k].next; ++back; }
				Cur &C = cur[l];
				if (k != NONE) {
					if (back > 0xffffu) throw Error(HRY_E_UNSUPPORTED, "more than 65536 different records of one list at one vertex (io.h:104 codes 16 bits)");
					*C.type++ = RefKind::lhist;
					*C.lh++ = back;
					if (wp) { *C.type_pos++ = pos++; *C.lh_pos++ = pos; pos += 2; }
					continue;
				}
				pool[pool_n] = Node{ idx, hd };
				hd = pool_n++;
				if (idx >= m.lists[l].count) throw Error(HRY_E_ARG, "an element names a record outside its list");
				reference(C, idx, c, (uint32_t)a);
			}
			c = c + 1 == fe ? fb : c + 1;
		} while (c != e0);
	}
	// ---- the arrays at their sizes
	for (size_t l = 0; l < nl; ++l) {
		ListStream &S = E.ls[l];
		const Cur &C = cur[l];
		const size_t nt = (size_t)(C.type - S.type_sym.data()), ng = (size_t)(C.gh - S.gh_val.data()), nd = (size_t)(C.d_idx - S.d_idx.data());
		const size_t nlh = S.lh_val.empty() ? 0 : (size_t)(C.lh - S.lh_val.data());
		S.type_sym.resize(nt); S.gh_val.resize(ng); S.lh_val.resize(nlh); S.d_idx.resize(nd); S.d_he.resize(nd); S.d_slot.resize(nd);
		if (wp) { S.type_pos.resize(nt); S.gh_pos.resize(ng); S.lh_pos.resize(nlh); S.d_pos.resize(nd); }
		S.created = C.created;
	}
	E.end_pos = pos;
}

}   // namespace hry
