// The reference stream (.hry v0.1) of a mesh with GENERAL bindings: several regions, records
// shared between elements, corner attributes -- what the OBJ reader creates (formats/obj/reader.rl:108-277) and what
// formats/hry/attrcode.h:23-80,135-154,321-393,443-531 code.  The PLY layout keeps its own faster pipeline (codec.cpp).
//
// Division of labour, as everywhere in the compat profile:
//   host    the cut-border walk; which record every vertex / face / corner names and whether the stream says so with a new
//           record (DATA), a distance in creation order (HIST) or a distance in the vertex' own list of names (LHIST) --
//           integer bookkeeping along the coding order; the position of every symbol in the single stream
//   device  prediction + residuals of every record coded as DATA (general.hip), the adaptive models of every byte plane by
//           counting, the range coder (kernels.hip: the same kernels as the PLY layout; planes carry explicit positions)
// and for the decoder: host = serial entropy decode + replay + the same bookkeeping (compat_read.cpp), device = un-prediction.
// (The parallel container has no single sequence: there the bookkeeping is the device's too -- events.hip, general_planes_encode.)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "codec_math.hpp"
#include "context.hpp"
#include "kernels.hpp"

#include <exception>
#include <thread>

namespace hry {

using namespace dev;
typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

namespace dev {
void launch_face_rank(hipStream_t st, const ConnView &cv, const uint32_t *order_f, uint32_t n, uint32_t *frank);
void launch_gen_vtx_resid(hipStream_t st, const ConnView &cv, const GenView &gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                          const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes);
void launch_gen_face_resid(hipStream_t st, const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes);
void launch_gen_corner_resid(hipStream_t st, const ConnView &cv, const GenView &gv, const uint32_t *frank, const uint32_t *ev_he, const uint8_t *ev_slot,
                             const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes);
constexpr int kSrcCap = 24;   // general.hip
struct GenChainJob {
	int32_t kind, comp;
	uint32_t n, pad2;
	uint8_t *rec;
	const uint32_t *src, *ev_he;
	const uint8_t *nsrc, *ev_slot;
	ListDesc ld;
};
void launch_gen_sources(hipStream_t st, int kind, const ConnView &cv, const GenView &gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                        uint32_t n, uint32_t *src, uint8_t *nsrc);
void launch_gen_chain(hipStream_t st, int kind, int stype, const ConnView &cv, const GenView &gv, const uint32_t *rank, const GenChainJob *jobs, uint32_t njobs);
void launch_faces_unfold(hipStream_t st, uint32_t n, const ListDesc &ld, uint8_t *rec);
}
bool reconstruct_vertex_list_fast(Context &cx, Mesh &m, int l, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                                  const std::vector<uint32_t> &seg_level, const std::vector<uint8_t> &vplanes, const uint8_t *d_vplanes = nullptr);   // unchunk.cpp
bool vertex_list_fast_applicable(const Mesh &m, int l, size_t n_order);                                                                             // unchunk.cpp
void reconstruct_vertex_list_detached(Context &cx, Mesh &t, const Mesh &conn, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                                      const std::vector<uint32_t> &seg_level, const uint8_t *d_vplanes, long long trace_origin);                    // unchunk.cpp
long long trace_origin_ns();                                                                                                                          // unchunk.cpp

void check_general(const Mesh &m)
{
	const Bindings &b = m.bind;
	if (m.lists.size() > (size_t)kMaxLists) throw Error(HRY_E_UNSUPPORTED, "more than 16 attribute lists");
	if (b.nregs_face() > 128 || b.nregs_vtx() > 128) throw Error(HRY_E_UNSUPPORTED, "more than 128 regions: the reference seeds its region models out of bounds (model.h:49-55)");
	if (b.nb_face > 255 || b.nb_vtx > 255 || b.nb_corner > 255) throw Error(HRY_E_UNSUPPORTED, "more than 255 lists bound to one region");
	if (b.face_reg.size() != m.nf || b.vtx_reg.size() != m.nv || b.face_attr.size() != (size_t)m.nf * b.nb_face ||
	    b.vtx_attr.size() != (size_t)m.nv * b.nb_vtx || b.corner_attr.size() != (size_t)m.ne() * b.nb_corner)
		throw Error(HRY_E_ARG, "binding tables do not match the element counts");
	auto bound = [&](int l, int want) {
		if (l < 0 || l >= (int)m.lists.size()) throw Error(HRY_E_ARG, "a region binds a list that does not exist");
		if (m.lists[l].target != want) throw Error(HRY_E_ARG, "a region binds a list of another kind");
	};
	for (int r = 0; r < b.nregs_face(); ++r) {
		for (int a = 0; a < b.nfacelists(r); ++a) bound(b.facelist(r, a), 0);
		for (int a = 0; a < b.ncornerlists(r); ++a) bound(b.cornerlist(r, a), 2);
	}
	for (int r = 0; r < b.nregs_vtx(); ++r) for (int a = 0; a < b.nvtxlists(r); ++a) bound(b.vtxlist(r, a), 1);
	for (uint32_t f = 0; f < m.nf; ++f) if (b.face_reg[f] >= b.nregs_face()) throw Error(HRY_E_ARG, "face region out of range");
	for (uint32_t v = 0; v < m.nv; ++v) if (b.vtx_reg[v] >= b.nregs_vtx()) throw Error(HRY_E_ARG, "vertex region out of range");
	// every record an element names exists (a decoded header may announce elements and an empty list)
	auto holds = [&](int l, uint32_t rec) { if (rec >= m.lists[l].count || (size_t)(rec + 1) * m.lists[l].stride() > m.lists[l].data.size()) throw Error(HRY_E_ARG, "an element names a record its list does not hold"); };
	for (uint32_t f = 0; f < m.nf; ++f) {
		const int r = b.face_reg[f];
		for (int a = 0; a < b.nfacelists(r); ++a) holds(b.facelist(r, a), b.face_attr[(size_t)f * b.nb_face + a]);
		for (uint32_t h = m.face_off[f]; h < m.face_off[f + 1]; ++h)
			for (int a = 0; a < b.ncornerlists(r); ++a) holds(b.cornerlist(r, a), b.corner_attr[(size_t)h * b.nb_corner + a]);
	}
	for (uint32_t v = 0; v < m.nv; ++v) { const int r = b.vtx_reg[v]; for (int a = 0; a < b.nvtxlists(r); ++a) holds(b.vtxlist(r, a), b.vtx_attr[(size_t)v * b.nb_vtx + a]); }
}

// connectivity + every list + the binding tables -> HBM
// (a mesh that hry_mesh_upload has made resident stays so, tables included, like a mesh in the PLY layout)
void upload_general(Context &cx, Mesh &m)
{
	const bool mesh_there = m.device_token != 0 && m.device_token == cx.resident_token && !m.twins_pending;
	if (mesh_there && cx.gen_token == m.device_token) return;
	if (!mesh_there) cx.upload_mesh(m);
	const Bindings &b = m.bind;
	auto put = [&](DevBuf &d, const void *src, size_t bytes) {
		d.ensure(std::max<size_t>(bytes, 16));
		if (bytes) HIP_OK(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, cx.stream));
	};
	put(cx.d_vreg, b.vtx_reg.data(), b.vtx_reg.size() * 2);
	put(cx.d_freg, b.face_reg.data(), b.face_reg.size() * 2);
	put(cx.d_vattr, b.vtx_attr.data(), b.vtx_attr.size() * 4);
	put(cx.d_cattr, b.corner_attr.data(), b.corner_attr.size() * 4);
	put(cx.d_fattr, b.face_attr.data(), b.face_attr.size() * 4);
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.gen_token = m.device_token;
}
static GenView gen_view(const Context &cx, const Mesh &m)
{
	GenView gv;
	gv.vtx_reg = cx.d_vreg.as<uint16_t>(); gv.face_reg = cx.d_freg.as<uint16_t>();
	gv.vtx_attr = cx.d_vattr.as<uint32_t>(); gv.corner_attr = cx.d_cattr.as<uint32_t>();
	gv.nb_vtx = m.bind.nb_vtx; gv.nb_corner = m.bind.nb_corner;
	return gv;
}

namespace {

// a device arena filled from host vectors in one go
struct Arena {
	struct Piece { const void *p; size_t bytes, at; };
	std::vector<Piece> pieces;
	size_t bytes = 0;
	size_t add(const void *p, size_t n)
	{
		const size_t at = (bytes + 15) & ~(size_t)15;
		pieces.push_back(Piece{ p, n, at });
		bytes = at + n;
		return at;
	}
	template <typename V> size_t add(const V &v) { return add(v.data(), v.size() * sizeof(typename V::value_type)); }
	// the pieces -> the context's pinned block -> HBM, behind everything on the stream; no wait: the block is the context's, and the
	// context's next call finds the stream drained.  (Until round 5 a pageable vector that grew piece by piece, its copy, and a wait
	// for it because the vector went out of scope: 4 of the 11 ms of a 180 000-triangle scene's encode.)
	void send(Context &cx, void *dst)
	{
		if (!bytes) return;
		cx.h_gen.ensure(bytes);
		uint8_t *h = cx.h_gen.as<uint8_t>();
		for (const Piece &q : pieces) if (q.bytes) memcpy(h + q.at, q.p, q.bytes);
		HIP_OK(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, cx.stream));
	}
};

}   // namespace

void finish_stream(Context &cx, uint32_t ns, std::vector<uint8_t> &payload);

void encode_general(Context &cx, Mesh &m, std::vector<uint8_t> &out)
{
	HIP_OK(hipSetDevice(cx.device));
	auto t_all = Clock::now();
	cx.timing = hry_timing{};
	check_codable(m);
	check_general(m);
	for (auto &L : m.lists) if (!L.have_bounds && L.ncomp()) { device_bounds(cx, m); break; }
	for (auto &L : m.lists) if (!L.have_bounds) { L.bmin.assign(L.stride(), 0); L.bmax.assign(L.stride(), 0); L.have_bounds = true; }
	upload_general(cx, m);

	out.clear();
	write_hry_header(m, 1, out);
	auto t_walk = Clock::now();
	WalkResult w;
	cut_border_walk(m, w, false);   // the operation model is evaluated on the device (k_opmodel_*), the groups' places in the ONE symbol sequence come out of the walk -- also from its threads (cbm_walk.cpp: the components' pieces are put in coding order)
	Events E;
	collect_events(m, w, w.n_conn, true, E);
	cx.timing.host_walk_ms = ms_since(t_walk);
	if ((uint64_t)E.end_pos + test_extra("HRY_TEST_EXTRA_SYMBOLS") >= (1ull << 31)) throw Error(HRY_E_UNSUPPORTED, "more than 2^31 symbols in one compat stream");
	const uint32_t ns = E.end_pos;
	const uint32_t vc = (uint32_t)w.order_v.size(), fc = (uint32_t)w.order_f.size();

	// ---- H2D: orders, repaired twins, connectivity groups (as codec.cpp), then everything collect_events produced
	auto t_h2d = Clock::now();
	HIP_OK(hipEventRecord(cx.ev[0], cx.stream));
	cx.d_order_v.ensure(std::max<size_t>((size_t)vc * 4, 16));
	cx.d_order_f.ensure(std::max<size_t>((size_t)fc * 4, 16));
	cx.d_rank.ensure(std::max<size_t>((size_t)m.nv * 4 + (size_t)m.nf * 4, 16));
	uint32_t *d_rank = cx.d_rank.as<uint32_t>(), *d_frank = d_rank + m.nv;
	if (vc) HIP_OK(hipMemcpyAsync(cx.d_order_v.p, w.order_v.data(), (size_t)vc * 4, hipMemcpyHostToDevice, cx.stream));
	if (fc) HIP_OK(hipMemcpyAsync(cx.d_order_f.p, w.order_f.data(), (size_t)fc * 4, hipMemcpyHostToDevice, cx.stream));
	upload_repaired_twins(cx, m, w);
	size_t ngrp = 0;
	for (int g = 0; g < G_COUNT; ++g) ngrp += w.grp_val[g].size();
	const size_t nop = w.op_sc.size();
	cx.d_grp_val.ensure(std::max<size_t>(ngrp * 4, 16));
	cx.d_grp_pos.ensure(std::max<size_t>(ngrp * 4, 16));
	// operations as the walk wrote them (symbol | order class << 3) + where the connectivity groups sit between them: the
	// device evaluates the operation model and places the records (k_opmodel_*)
	std::vector<uint32_t> op_thr, op_cum;
	op_position_table(w, op_thr, op_cum);
	const size_t op_bytes = (nop + 15) & ~(size_t)15, ngr = op_thr.size();
	cx.d_op.ensure(std::max<size_t>(op_bytes + ngr * 8 + dev::op_model_scratch_bytes((uint32_t)nop) + 64, 64));
	size_t goff[G_COUNT + 1] = { 0 };
	for (int g = 0; g < G_COUNT; ++g) {
		size_t n = w.grp_val[g].size();
		goff[g + 1] = goff[g] + n;
		if (!n) continue;
		HIP_OK(hipMemcpyAsync(cx.d_grp_val.as<uint32_t>() + goff[g], w.grp_val[g].data(), n * 4, hipMemcpyHostToDevice, cx.stream));
		HIP_OK(hipMemcpyAsync(cx.d_grp_pos.as<uint32_t>() + goff[g], w.grp_pos[g].data(), n * 4, hipMemcpyHostToDevice, cx.stream));
	}
	uint8_t *d_opsc = cx.d_op.as<uint8_t>();
	uint32_t *d_opthr = (uint32_t*)(d_opsc + op_bytes), *d_opcum = d_opthr + ngr;
	void *d_opscratch = d_opcum + ngr;
	if (nop) HIP_OK(hipMemcpyAsync(d_opsc, w.op_sc.data(), nop, hipMemcpyHostToDevice, cx.stream));
	if (ngr) {
		HIP_OK(hipMemcpyAsync(d_opthr, op_thr.data(), ngr * 4, hipMemcpyHostToDevice, cx.stream));
		HIP_OK(hipMemcpyAsync(d_opcum, op_cum.data(), ngr * 4, hipMemcpyHostToDevice, cx.stream));
	}

	// initial count tables (models.h:197-218)
	std::vector<uint32_t> inits;
	auto add_init = [&](const std::vector<uint32_t> &t) { uint32_t id = (uint32_t)(inits.size() / 256); inits.insert(inits.end(), t.begin(), t.end()); return id; };
	std::vector<uint32_t> ones(256, 1), iop_init(256, 0), nt0(256, 0), nt1(256, 0), ty2(256, 0), ty3(256, 0), rv(256, 0), rf(256, 0);
	for (int i = 0; i < 9; ++i) iop_init[i] = 1;
	for (size_t d = 3; d < m.have_degree.size(); ++d) if (m.have_degree[d]) { ++nt0[(d - 2) & 0xff]; ++nt1[(d - 2) >> 8]; }
	ty2[0] = ty2[1] = 1; ty3[0] = ty3[1] = ty3[2] = 1;
	for (int r = 0; r < m.bind.nregs_vtx(); ++r) ++rv[r];
	for (int r = 0; r < m.bind.nregs_face(); ++r) ++rf[r];
	const uint32_t id_ones = add_init(ones), id_iop = add_init(iop_init), id_nt0 = add_init(nt0), id_nt1 = add_init(nt1),
	               id_ty2 = add_init(ty2), id_ty3 = add_init(ty3), id_rv = add_init(rv), id_rf = add_init(rf);
	auto total_of = [&](uint32_t id) { uint32_t s = 0; for (int i = 0; i < 256; ++i) s += inits[(size_t)id * 256 + i]; return s; };
	cx.d_init.ensure(inits.size() * 4);
	HIP_OK(hipMemcpyAsync(cx.d_init.p, inits.data(), inits.size() * 4, hipMemcpyHostToDevice, cx.stream));

	// the arena: symbol planes the host made, position tables, event tables; then room for the residual planes the kernels make
	Arena A;
	struct ListAt { size_t type_sym, type_pos, gh_vals, gh_planes, gh_pos, lh_vals, lh_planes, lh_pos, d_pos, d_idx, d_he, d_slot, planes; };
	std::vector<ListAt> at(m.lists.size());
	for (size_t l = 0; l < m.lists.size(); ++l) {
		const ListStream &S = E.ls[l];
		ListAt &T = at[l];
		T.type_sym = A.add(S.type_sym); T.type_pos = A.add(S.type_pos);
		// (the distances go up as 32-bit values and are split into their byte planes on the device, like the connectivity groups:
		// the host's byte loops were a millisecond of an 80 000-triangle scene)
		T.gh_vals = A.add(S.gh_val); T.gh_pos = A.add(S.gh_pos);
		T.lh_vals = A.add(S.lh_val); T.lh_pos = A.add(S.lh_pos);
		T.d_pos = A.add(S.d_pos); T.d_idx = A.add(S.d_idx); T.d_he = A.add(S.d_he); T.d_slot = A.add(S.d_slot);
	}
	const size_t rv_sym = A.add(E.rv_sym), rv_pos = A.add(E.rv_pos), rf_sym = A.add(E.rf_sym), rf_pos = A.add(E.rf_pos);
	size_t arena_bytes = (A.bytes + 15) & ~(size_t)15;
	for (size_t l = 0; l < m.lists.size(); ++l) {
		at[l].planes = arena_bytes; arena_bytes += ((size_t)E.ls[l].d_pos.size() * E.ls[l].nbytes + 15) & ~(size_t)15;
		at[l].gh_planes = arena_bytes; arena_bytes += (E.ls[l].gh_val.size() * 4 + 15) & ~(size_t)15;
		at[l].lh_planes = arena_bytes; arena_bytes += (E.ls[l].lh_val.size() * 2 + 15) & ~(size_t)15;
	}
	cx.d_gen.ensure(std::max<size_t>(arena_bytes, 16));
	A.send(cx, cx.d_gen.p);
	uint8_t *arena = cx.d_gen.as<uint8_t>();
	for (size_t l = 0; l < m.lists.size(); ++l) {
		launch_split_bytes(cx.stream, (const uint32_t*)(arena + at[l].gh_vals), (uint32_t)E.ls[l].gh_val.size(), 4, arena + at[l].gh_planes);
		launch_split_bytes(cx.stream, (const uint32_t*)(arena + at[l].lh_vals), (uint32_t)E.ls[l].lh_val.size(), 2, arena + at[l].lh_planes);
	}

	size_t conn_plane_bytes = 0;
	for (int g = 0; g < G_COUNT; ++g) conn_plane_bytes += w.grp_val[g].size() * kGroupBytes[g];
	cx.d_connplanes.ensure(std::max<size_t>(conn_plane_bytes, 16));

	// ---- model jobs
	std::vector<PlaneJob> jobs;
	std::vector<ChunkRef> chunks;
	uint32_t max_total = 2;
	auto add_job = [&](const uint8_t *sym, uint32_t n, uint32_t init_id, const uint32_t *pos_tab, uint32_t pos_add) {
		if (!n) return;
		PlaneJob j{};
		j.sym = sym; j.n = n; j.init = cx.d_init.as<uint32_t>() + (size_t)init_id * 256; j.t0 = total_of(init_id);
		j.pos_tab = pos_tab; j.pos_add = pos_add; j.pos_base = 0; j.pos_stride = 0;
		j.chunk0 = (uint32_t)chunks.size();
		for (uint32_t f = 0; f < n; f += kChunk) chunks.push_back(ChunkRef{ (uint32_t)jobs.size(), f });
		jobs.push_back(j);
		max_total = std::max(max_total, j.t0 + n);
	};
	{
		size_t poff = 0;
		for (int g = 0; g < G_COUNT; ++g) {
			uint32_t n = (uint32_t)w.grp_val[g].size();
			for (int k = 0; k < kGroupBytes[g]; ++k) {
				uint32_t init_id = g == G_IOP ? id_iop : g == G_NUMTRI ? (k == 0 ? id_nt0 : id_nt1) : id_ones;
				add_job(cx.d_connplanes.as<uint8_t>() + poff + (size_t)k * n, n, init_id, cx.d_grp_pos.as<uint32_t>() + goff[g], (uint32_t)k);
			}
			poff += (size_t)n * kGroupBytes[g];
		}
	}
	add_job(arena + rv_sym, (uint32_t)E.rv_sym.size(), id_rv, (const uint32_t*)(arena + rv_pos), 0);
	add_job(arena + rf_sym, (uint32_t)E.rf_sym.size(), id_rf, (const uint32_t*)(arena + rf_pos), 0);
	for (size_t l = 0; l < m.lists.size(); ++l) {
		const ListStream &S = E.ls[l];
		const ListAt &T = at[l];
		add_job(arena + T.type_sym, (uint32_t)S.type_sym.size(), m.lists[l].target == 2 ? id_ty3 : id_ty2, (const uint32_t*)(arena + T.type_pos), 0);
		for (int k = 0; k < 4; ++k) add_job(arena + T.gh_planes + (size_t)k * S.gh_val.size(), (uint32_t)S.gh_val.size(), id_ones, (const uint32_t*)(arena + T.gh_pos), (uint32_t)k);
		for (int k = 0; k < 2; ++k) add_job(arena + T.lh_planes + (size_t)k * S.lh_val.size(), (uint32_t)S.lh_val.size(), id_ones, (const uint32_t*)(arena + T.lh_pos), (uint32_t)k);
		const uint32_t nd = (uint32_t)S.d_pos.size();
		for (uint32_t k = 0; k < S.nbytes; ++k) add_job(arena + T.planes + (size_t)k * nd, nd, id_ones, (const uint32_t*)(arena + T.d_pos), k);
	}
	max_total = std::max<uint32_t>(max_total, (uint32_t)nop + 8);   // the operation model's total: 7 + the operations so far
	cx.d_jobs.ensure(std::max<size_t>(jobs.size() * sizeof(PlaneJob), 16));
	cx.d_chunks.ensure(std::max<size_t>(chunks.size() * sizeof(ChunkRef), 16));
	cx.d_hist.ensure(std::max<size_t>(chunks.size() * 256 * 4, 16));
	if (!jobs.empty()) HIP_OK(hipMemcpyAsync(cx.d_jobs.p, jobs.data(), jobs.size() * sizeof(PlaneJob), hipMemcpyHostToDevice, cx.stream));
	if (!chunks.empty()) HIP_OK(hipMemcpyAsync(cx.d_chunks.p, chunks.data(), chunks.size() * sizeof(ChunkRef), hipMemcpyHostToDevice, cx.stream));
	cx.ensure_magic(max_total + 1);
	cx.d_rec_sym.ensure(std::max<size_t>((size_t)ns * sizeof(SymRec), 16));
	cx.d_sym_l.ensure(std::max<size_t>((size_t)ns * 4, 16));
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.timing.h2d_ms = ms_since(t_h2d);

	// ---- device: residuals of the records coded as data
	ConnView cv = cx.conn_view();
	const GenView gv = gen_view(cx, m);
	HIP_OK(hipEventRecord(cx.ev[1], cx.stream));
	HIP_OK(hipMemsetAsync(d_rank, 0xff, (size_t)m.nv * 4, cx.stream));
	launch_rank(cx.stream, cx.d_order_v.as<uint32_t>(), vc, cv.org, d_rank);
	launch_face_rank(cx.stream, cv, cx.d_order_f.as<uint32_t>(), fc, d_frank);
	for (size_t l = 0; l < m.lists.size(); ++l) {
		const ListStream &S = E.ls[l];
		const ListAt &T = at[l];
		const uint32_t nd = (uint32_t)S.d_pos.size();
		if (!nd || !S.nbytes) continue;
		const ListDesc ld = make_list_desc(m.lists[l]);
		const uint32_t *he = (const uint32_t*)(arena + T.d_he), *idx = (const uint32_t*)(arena + T.d_idx);
		const uint8_t *slot = arena + T.d_slot;
		const uint8_t *rec = cx.d_rec[l].as<uint8_t>();
		if (m.lists[l].target == 1) launch_gen_vtx_resid(cx.stream, cv, gv, d_rank, he, slot, idx, nd, rec, ld, arena + T.planes);
		else if (m.lists[l].target == 2) launch_gen_corner_resid(cx.stream, cv, gv, d_frank, he, slot, idx, nd, rec, ld, arena + T.planes);
		else launch_gen_face_resid(cx.stream, idx, nd, rec, ld, arena + T.planes);
	}
	HIP_OK(hipEventRecord(cx.ev[2], cx.stream));
	// ---- device: models -> per-symbol records in stream order
	{
		size_t poff = 0;
		for (int g = 0; g < G_COUNT; ++g) {
			uint32_t n = (uint32_t)w.grp_val[g].size();
			launch_split_bytes(cx.stream, cx.d_grp_val.as<uint32_t>() + goff[g], n, kGroupBytes[g], cx.d_connplanes.as<uint8_t>() + poff);
			poff += (size_t)n * kGroupBytes[g];
		}
	}
	const MagicEnt *magic = cx.d_magic.as<MagicEnt>();
	SymRec *rec = cx.d_rec_sym.as<SymRec>();
	uint32_t *sym_l = cx.d_sym_l.as<uint32_t>();
	launch_op_model(cx.stream, d_opsc, (uint32_t)nop, d_opthr, d_opcum, (uint32_t)ngr, d_opscratch, magic, rec, sym_l);
	launch_model(cx.stream, cx.d_jobs.as<PlaneJob>(), (uint32_t)jobs.size(), cx.d_chunks.as<ChunkRef>(), (uint32_t)chunks.size(), cx.d_hist.as<uint32_t>(), magic, rec, sym_l);

	if (cx.keep_stages) {
		cx.stage_put_host("order_v", w.order_v.data(), (size_t)vc * 4);
		cx.stage_put_host("order_f", w.order_f.data(), (size_t)fc * 4);
		cx.stage_put("rec", cx.d_rec_sym.p, (size_t)ns * sizeof(SymRec));
		cx.stage_put("sym_l", cx.d_sym_l.p, (size_t)ns * 4);
	}
	std::vector<uint8_t> payload;
	finish_stream(cx, ns, payload);
	out.insert(out.end(), payload.begin(), payload.end());

	cx.timing.k_predict_ms = cx.elapsed(1, 2);
	cx.timing.k_model_ms = cx.elapsed(2, 3);
	cx.timing.k_rchain_ms = cx.elapsed(3, 4);
	cx.timing.device_ms = cx.elapsed(1, 5);
	cx.timing.n_symbols = ns;
	cx.timing.payload_bytes = payload.size();
	cx.timing.total_ms = ms_since(t_all);
}

// Device part of a decode with general bindings: connectivity, bindings and the lists (holding residual codes in record layout,
// or -- chunked container -- zeros, the codes being scattered from the decoded planes by `fill`) go up; every list with records
// to reconstruct gets its source table and its chains.  ev[l] empty: the list is final already (vertex fast path) or has no records.
static void reconstruct_general(Context &cx, Mesh &m, const OrderVec &order_v, const std::vector<GenRecordEvents> &ev,
                                const std::function<void()> &fill)
{
	auto t_h2d = Clock::now();
	upload_general(cx, m);   // connectivity, bindings, and the residual codes in record layout
	if (fill) fill();
	const uint32_t vc = (uint32_t)order_v.size();
	cx.d_order_v.ensure(std::max<size_t>((size_t)vc * 4, 16));
	cx.d_rank.ensure(std::max<size_t>((size_t)m.nv * 4, 16));
	if (vc) HIP_OK(hipMemcpyAsync(cx.d_order_v.p, order_v.data(), (size_t)vc * 4, hipMemcpyHostToDevice, cx.stream));
	Arena A;
	const size_t nl = m.lists.size();
	std::vector<size_t> he_at(nl), slot_at(nl), src_at(nl), nsrc_at(nl);
	for (size_t l = 0; l < nl; ++l) { he_at[l] = A.add(ev[l].he); slot_at[l] = A.add(ev[l].slot); }
	size_t arena_bytes = (A.bytes + 15) & ~(size_t)15;
	for (size_t l = 0; l < nl; ++l) {
		if (m.lists[l].target != 1 && m.lists[l].target != 2) continue;
		const size_t nd = ev[l].he.size();
		src_at[l] = arena_bytes; arena_bytes += (nd * kSrcCap * 4 + 15) & ~(size_t)15;
		nsrc_at[l] = arena_bytes; arena_bytes += (nd + 15) & ~(size_t)15;
	}
	const size_t jobs_at = arena_bytes;
	size_t max_jobs = 0;
	for (const AttrList &L : m.lists) max_jobs += (size_t)L.ncomp();
	arena_bytes += (max_jobs + 1) * sizeof(GenChainJob);
	cx.d_gen.ensure(std::max<size_t>(arena_bytes, 16));
	A.send(cx, cx.d_gen.p);
	uint8_t *arena = cx.d_gen.as<uint8_t>();
	std::vector<GenChainJob> jobs, lead;   // lead: one per list (the source table is per record)
	for (size_t l = 0; l < nl; ++l) {
		const AttrList &L = m.lists[l];
		const uint32_t nd = (uint32_t)ev[l].he.size();
		if (!nd || !L.ncomp() || (L.target != 1 && L.target != 2)) continue;
		GenChainJob j{};
		j.kind = L.target == 1 ? 0 : 1; j.n = nd; j.rec = cx.d_rec[l].as<uint8_t>();
		j.src = (const uint32_t*)(arena + src_at[l]); j.nsrc = arena + nsrc_at[l];
		j.ev_he = (const uint32_t*)(arena + he_at[l]); j.ev_slot = arena + slot_at[l];
		j.ld = make_list_desc(L);
		lead.push_back(j);
		for (int c = 0; c < L.ncomp(); ++c) { j.comp = c; jobs.push_back(j); }   // the components of a record are predicted independently
	}
	// one launch per kind and storage type (the kernel is instantiated for each); inside a launch every job has its own wavefront
	std::stable_sort(jobs.begin(), jobs.end(), [](const GenChainJob &a, const GenChainJob &b) {
		return std::make_pair(a.kind, (int)a.ld.stype[a.comp]) < std::make_pair(b.kind, (int)b.ld.stype[b.comp]); });
	if (!jobs.empty()) HIP_OK(hipMemcpyAsync(arena + jobs_at, jobs.data(), jobs.size() * sizeof(GenChainJob), hipMemcpyHostToDevice, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.timing.h2d_ms = ms_since(t_h2d);

	ConnView cv = cx.conn_view();
	const GenView gv = gen_view(cx, m);
	HIP_OK(hipEventRecord(cx.ev[3], cx.stream));
	HIP_OK(hipMemsetAsync(cx.d_rank.p, 0xff, (size_t)m.nv * 4, cx.stream));
	launch_rank(cx.stream, cx.d_order_v.as<uint32_t>(), vc, cv.org, cx.d_rank.as<uint32_t>());
	for (const GenChainJob &j : lead)
		launch_gen_sources(cx.stream, j.kind, cv, gv, cx.d_rank.as<uint32_t>(), j.ev_he, j.ev_slot, j.n, const_cast<uint32_t*>(j.src), const_cast<uint8_t*>(j.nsrc));
	for (size_t l = 0; l < nl; ++l) {   // face lists: no prediction (attrcode.h:245-270)
		const AttrList &L = m.lists[l];
		if (L.target == 0 && L.ncomp() && !ev[l].he.empty()) launch_faces_unfold(cx.stream, (uint32_t)ev[l].he.size(), make_list_desc(L), cx.d_rec[l].as<uint8_t>());
	}
	HIP_OK(hipEventRecord(cx.ev[7], cx.stream));
	for (size_t a = 0; a < jobs.size();) {
		size_t e = a + 1;
		while (e < jobs.size() && jobs[e].kind == jobs[a].kind && jobs[e].ld.stype[jobs[e].comp] == jobs[a].ld.stype[jobs[a].comp]) ++e;
		launch_gen_chain(cx.stream, jobs[a].kind, jobs[a].ld.stype[jobs[a].comp], cv, gv, cx.d_rank.as<uint32_t>(), (const GenChainJob*)(arena + jobs_at) + a, (uint32_t)(e - a));
		a = e;
	}
	HIP_OK(hipEventRecord(cx.ev[4], cx.stream));
	for (size_t l = 0; l < nl; ++l)
		if (!m.lists[l].data.empty()) HIP_OK(hipMemcpyAsync(m.lists[l].data.data(), cx.d_rec[l].p, m.lists[l].data.size(), hipMemcpyDeviceToHost, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.timing.k_chain_ms = cx.elapsed(7, 4);
}

// ---------------------------------------------------------------------------------------------------------
Mesh *decode_general(Context &cx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> m)
{
	HIP_OK(hipSetDevice(cx.device));
	auto t_all = Clock::now();
	cx.timing = hry_timing{};
	if (m->lists.size() > (size_t)kMaxLists) throw Error(HRY_E_UNSUPPORTED, "more than 16 attribute lists");
	for (const AttrList &L : m->lists)
		for (int c = 0; c < L.ncomp(); ++c) {
			if (L.stype(c) == C_DOUBLE) throw Error(HRY_E_UNSUPPORTED, "lossless double components are outside the supported subset");
			if (kTypeSize[L.stype(c)] == 8) throw Error(HRY_E_UNSUPPORTED, "8-byte storage types (more than 32 quantisation bits, lossless 64-bit integers) are outside the supported subset");   // (before the stream is read: the record chains have no such form)
		}
	auto t_walk = Clock::now();
	OrderVec order_v;
	std::vector<uint32_t> seg_start, seg_level;
	std::vector<GenRecordEvents> ev;
	std::vector<uint8_t> vplanes;
	// one vertex region with one list (every OBJ whose "v" lines have the same number of values): candidate for the vertex chains
	int fast_l = -1;
	if (m->bind.nregs_vtx() == 1 && m->bind.nvtxlists(0) == 1 && !getenv("HRY_GENERIC_VERTEX")) fast_l = m->bind.vtxlist(0, 0);
	read_general_stream(p + hdr, n - hdr, *m, order_v, ev, seg_start, seg_level, fast_l, vplanes);
	cx.timing.host_walk_ms = ms_since(t_walk);
	double fast_ms = 0;
	if (fast_l >= 0 && ev[fast_l].he.size() == order_v.size() && !order_v.empty()) {   // ... and every vertex created its own record
		if (reconstruct_vertex_list_fast(cx, *m, fast_l, order_v, seg_start, seg_level, vplanes)) { ev[fast_l].he.clear(); ev[fast_l].slot.clear(); fast_ms = cx.elapsed(3, 4); }
	}

	reconstruct_general(cx, *m, order_v, ev, nullptr);
	cx.timing.k_predict_ms = cx.elapsed(3, 4) + fast_ms;
	cx.timing.device_ms = cx.timing.k_predict_ms;
	cx.timing.payload_bytes = n - hdr;
	cx.timing.total_ms = ms_since(t_all);
	m->device_token = 0;
	return m.release();
}


// ---------------------------------------------------------------------------------------------------------
// chunked container (.hry v0.2) with general bindings
// ---------------------------------------------------------------------------------------------------------
std::vector<GenPlane> general_plane_layout(const Mesh &m)
{
	std::vector<GenPlane> p;
	if (m.bind.nregs_vtx() > 1) p.push_back(GenPlane{ GP_REGV, -1, 0, INIT_REGV });
	if (m.bind.nregs_face() > 1) p.push_back(GenPlane{ GP_REGF, -1, 0, INIT_REGF });
	for (size_t l = 0; l < m.lists.size(); ++l) {
		const AttrList &L = m.lists[l];
		if (L.target == 3) continue;
		p.push_back(GenPlane{ GP_TYPE, (int)l, 0, L.target == 2 ? INIT_TYPE3 : INIT_TYPE2 });
		for (int k = 0; k < 4; ++k) p.push_back(GenPlane{ GP_GHIST, (int)l, k, INIT_ONES });
		if (L.target == 2) for (int k = 0; k < 2; ++k) p.push_back(GenPlane{ GP_LHIST, (int)l, k, INIT_ONES });
		for (int k = 0; k < L.coded_bytes(); ++k) p.push_back(GenPlane{ GP_DATA, (int)l, k, INIT_ONES });
	}
	return p;
}

// The planes of a mesh with general bindings in the parallel container: which record every element names -- on the device
// (events.hip; HRY_HOST_EVENTS: the host's loop and an arena of its arrays, as until round 5) -- and the residuals of the records
// coded as data (general.hip).  Per list: kinds (one byte a reference), four planes of creation-order distances, two of per-vertex
// distances at corner lists, the data planes.
void general_planes_encode(Context &cx, Mesh &m, const WalkResult &w, std::vector<PlaneRef> &planes)
{
	const uint32_t vc = (uint32_t)w.order_v.size(), fc = (uint32_t)w.order_f.size();
	const size_t nl = m.lists.size();
	struct Lst { uint32_t nt = 0, ng = 0, nlh = 0, nd = 0, nbytes = 0; const uint8_t *type_sym = nullptr, *d_slot = nullptr; const uint32_t *gh_vals = nullptr, *lh_vals = nullptr, *d_idx = nullptr, *d_he = nullptr;
	             uint8_t *gh = nullptr, *lh = nullptr, *data = nullptr; };
	std::vector<Lst> ls(nl);
	const uint8_t *rv = nullptr, *rf = nullptr;
	uint32_t n_rv = 0, n_rf = 0;
	ConnView cv = cx.conn_view();
	const GenView gv = gen_view(cx, m);
	const Bindings &b = m.bind;
	static const bool host_events_env = getenv("HRY_HOST_EVENTS") != nullptr;
	bool host_events = host_events_env;
	Events E;   // (the host's arrays live until the arena has been copied into pinned memory: Arena::send)
again:
	if (host_events) {
		const auto t_events = Clock::now();
		collect_events(m, w, 0, false, E);
		cx.timing.host_walk_ms += ms_since(t_events);   // (host bookkeeping along the coding order, like the walk: it was missing from the record)
		if (getenv("HRY_TRACE")) fprintf(stderr, "[hry enc] %8.3f ms  which record every element names (host)\n", ms_since(t_events));
		Arena A;
		struct At { size_t type_sym, gh_vals, gh, lh_vals, lh, d_idx, d_he, d_slot, planes; };
		std::vector<At> at(nl);
		for (size_t l = 0; l < nl; ++l) {
			const ListStream &S = E.ls[l];
			At &T = at[l];
			T.type_sym = A.add(S.type_sym);
			T.gh_vals = A.add(S.gh_val); T.lh_vals = A.add(S.lh_val);   // (32-bit values: split into byte planes on the device)
			T.d_idx = A.add(S.d_idx); T.d_he = A.add(S.d_he); T.d_slot = A.add(S.d_slot);
		}
		const size_t rv_at = A.add(E.rv_sym), rf_at = A.add(E.rf_sym);
		size_t arena_bytes = (A.bytes + 15) & ~(size_t)15;
		for (size_t l = 0; l < nl; ++l) {
			at[l].planes = arena_bytes; arena_bytes += ((size_t)E.ls[l].d_idx.size() * E.ls[l].nbytes + 15) & ~(size_t)15;
			at[l].gh = arena_bytes; arena_bytes += (E.ls[l].gh_val.size() * 4 + 15) & ~(size_t)15;
			at[l].lh = arena_bytes; arena_bytes += (E.ls[l].lh_val.size() * 2 + 15) & ~(size_t)15;
		}
		cx.d_gen.ensure(std::max<size_t>(arena_bytes, 16));
		A.send(cx, cx.d_gen.p);
		uint8_t *arena = cx.d_gen.as<uint8_t>();
		for (size_t l = 0; l < nl; ++l) {
			const ListStream &S = E.ls[l];
			Lst &L = ls[l];
			L.nt = (uint32_t)S.type_sym.size(); L.ng = (uint32_t)S.gh_val.size(); L.nlh = (uint32_t)S.lh_val.size(); L.nd = (uint32_t)S.d_idx.size(); L.nbytes = S.nbytes;
			L.type_sym = arena + at[l].type_sym; L.gh_vals = (const uint32_t*)(arena + at[l].gh_vals); L.lh_vals = (const uint32_t*)(arena + at[l].lh_vals);
			L.d_idx = (const uint32_t*)(arena + at[l].d_idx); L.d_he = (const uint32_t*)(arena + at[l].d_he); L.d_slot = arena + at[l].d_slot;
			L.gh = arena + at[l].gh; L.lh = arena + at[l].lh; L.data = arena + at[l].planes;
		}
		rv = arena + rv_at; rf = arena + rf_at; n_rv = (uint32_t)E.rv_sym.size(); n_rf = (uint32_t)E.rf_sym.size();
	} else {
		// ---- on the device.  Most references a list can get, without a pass over the orders: every coded element at every slot
		// of the region with the most slots for it (corner lists: every half-edge)
		std::vector<uint32_t> max_refs(nl, 0);
		for (size_t l = 0; l < nl; ++l) {
			const int tg = m.lists[l].target;
			if (tg == 3) continue;
			const int nreg = tg == 1 ? b.nregs_vtx() : b.nregs_face();
			uint32_t most = 0;
			for (int r = 0; r < nreg; ++r) {
				uint32_t c = 0;
				const int na = tg == 1 ? b.nvtxlists(r) : tg == 0 ? b.nfacelists(r) : b.ncornerlists(r);
				for (int a = 0; a < na; ++a) c += (uint32_t)(tg == 1 ? b.vtxlist(r, a) : tg == 0 ? b.facelist(r, a) : b.cornerlist(r, a)) == (uint32_t)l;
				most = std::max(most, c);
			}
			const uint64_t elems = tg == 1 ? vc : tg == 0 ? fc : m.ne();
			if (elems * most >= (1ull << 31)) throw Error(HRY_E_UNSUPPORTED, "more than 2^31 references to one list");
			max_refs[l] = (uint32_t)(elems * most);
		}
		// the regions' tables (a few words) through the context's pinned block
		Arena A;
		std::vector<int32_t> off_f(b.off_facelist.begin(), b.off_facelist.end()), off_v(b.off_vtxlist.begin(), b.off_vtxlist.end()), off_c(b.off_cornerlist.begin(), b.off_cornerlist.end());
		const size_t a_off_f = A.add(off_f), a_off_v = A.add(off_v), a_off_c = A.add(off_c);
		const size_t a_lf = A.add(b.reg_facelist), a_lv = A.add(b.reg_vtxlist), a_lc = A.add(b.reg_cornerlist);
		size_t bytes = (A.bytes + 15) & ~(size_t)15;
		auto take = [&](size_t n) { const size_t at = bytes; bytes += (n + 15) & ~(size_t)15; return at; };
		struct At { size_t type_sym, gh_vals, lh_vals, d_idx, d_he, d_slot, gh, lh, planes, ws; };
		std::vector<At> at(nl);
		// (the vertices' names at the corner slots are shared by the corner lists: at most every half-edge at every corner slot)
		uint32_t most_corner_slots = 0;
		bool any_corner = false;
		for (int r = 0; r < b.nregs_face(); ++r) most_corner_slots = std::max(most_corner_slots, (uint32_t)b.ncornerlists(r));
		for (size_t l = 0; l < nl; ++l) any_corner |= m.lists[l].target == 2 && max_refs[l] != 0;
		if ((uint64_t)m.ne() * most_corner_slots >= (1ull << 31) || (uint64_t)b.nb_corner * m.nv >= (1ull << 31)) throw Error(HRY_E_UNSUPPORTED, "more than 2^31 corner references");
		const uint32_t corner_refs_max = any_corner ? m.ne() * most_corner_slots : 0u, head_words = any_corner ? (uint32_t)b.nb_corner * m.nv : 0u;
		for (size_t l = 0; l < nl; ++l) {
			const size_t r = max_refs[l];
			const uint32_t n_order = m.lists[l].target == 1 ? vc : fc;
			ls[l].nbytes = (uint32_t)m.lists[l].coded_bytes();
			at[l].type_sym = take(r); at[l].gh_vals = take(r * 4); at[l].lh_vals = take(m.lists[l].target == 2 ? r * 4 : 0);
			at[l].d_idx = take(r * 4); at[l].d_he = take(r * 4); at[l].d_slot = take(r);
			at[l].gh = take(r * 4); at[l].lh = take(m.lists[l].target == 2 ? r * 2 : 0); at[l].planes = take(r * ls[l].nbytes);
			at[l].ws = take(r ? dev::events_list_workspace_bytes(n_order, (uint32_t)r, m.lists[l].count) : 0);
		}
		const size_t a_counts = take((nl * 4 + 4) * 4), a_rv = take(vc), a_rf = take(fc), a_names = take(dev::events_names_workspace_bytes(fc, corner_refs_max, head_words));
		// (the arena is sized from upper bounds -- every half-edge times the most slots a region binds to the list -- where the
		// host's loop needs the actual counts: a scene whose bounds do not fit the device's memory takes the host's loop)
		bool no_room = false;
		try { cx.d_gen.ensure(std::max<size_t>(bytes, 16)); } catch (const Error &) { (void)hipGetLastError(); no_room = true; }
		if (no_room) { host_events = true; goto again; }
		A.send(cx, cx.d_gen.p);
		uint8_t *arena = cx.d_gen.as<uint8_t>();
		EvRegions rg{ (const int32_t*)(arena + a_off_f), (const int32_t*)(arena + a_off_v), (const int32_t*)(arena + a_off_c),
		              (const uint16_t*)(arena + a_lf), (const uint16_t*)(arena + a_lv), (const uint16_t*)(arena + a_lc), cx.d_fattr.as<uint32_t>(), b.nb_face };
		uint32_t *d_counts = (uint32_t*)(arena + a_counts), *d_err = d_counts + nl * 4;
		HIP_OK(hipMemsetAsync(d_err, 0, 16, cx.stream));
		if (any_corner) dev::launch_corner_places(cx.stream, cv, gv, rg, cx.d_order_f.as<uint32_t>(), fc, corner_refs_max, head_words, m.nv, arena + a_names);
		for (int phase = 0; phase < 2; ++phase)
			for (size_t l = 0; l < nl; ++l) {
				const int tg = m.lists[l].target;
				if (tg == 3) { if (!phase) HIP_OK(hipMemsetAsync(d_counts + 4 * l, 0, 16, cx.stream)); continue; }
				const uint32_t *order = tg == 1 ? cx.d_order_v.as<uint32_t>() : cx.d_order_f.as<uint32_t>();
				const uint32_t n_order = tg == 1 ? vc : fc;
				if (!phase) dev::launch_list_refs(cx.stream, tg, (uint32_t)l, m.lists[l].count, cv, gv, rg, order, n_order, max_refs[l], m.nv, fc, corner_refs_max, head_words, arena + a_names,
				                                  arena + at[l].ws, d_counts + 4 * l, d_err);
				else dev::launch_list_kinds(cx.stream, tg, m.lists[l].count, cv, n_order, max_refs[l], m.nv, fc, corner_refs_max, head_words, arena + a_names, arena + at[l].ws, arena + at[l].type_sym,
				                            (uint32_t*)(arena + at[l].gh_vals), (uint32_t*)(arena + at[l].lh_vals), (uint32_t*)(arena + at[l].d_idx), (uint32_t*)(arena + at[l].d_he),
				                            arena + at[l].d_slot, d_counts + 4 * l, d_err);
			}
		if (b.nregs_vtx() > 1) { dev::launch_region_symbols(cx.stream, 1, cv, gv, cx.d_order_v.as<uint32_t>(), vc, arena + a_rv); rv = arena + a_rv; n_rv = vc; }
		if (b.nregs_face() > 1) { dev::launch_region_symbols(cx.stream, 0, cv, gv, cx.d_order_f.as<uint32_t>(), fc, arena + a_rf); rf = arena + a_rf; n_rf = fc; }
		// how many of every kind: the one thing the host needs (the planes' lengths)
		cx.h_small.ensure(std::max<size_t>((nl * 4 + 4) * 4, 4096));
		uint32_t *h_counts = cx.h_small.as<uint32_t>();
		HIP_OK(hipMemcpyAsync(h_counts, d_counts, (nl * 4 + 4) * 4, hipMemcpyDeviceToHost, cx.stream));
		HIP_OK(hipStreamSynchronize(cx.stream));
		const uint32_t err = h_counts[nl * 4];
		if (err & 4u) { host_events = true; goto again; }   // a hub: a vertex with more names at a slot than a device thread walks (events.hip: kMaxNames) -- the host's loop
		if (err & 1u) throw Error(HRY_E_ARG, "an element names a record outside its list");
		if (err & 2u) throw Error(HRY_E_UNSUPPORTED, "more than 65536 different records of one list at one vertex (io.h:104 codes 16 bits)");
		for (size_t l = 0; l < nl; ++l) {
			Lst &L = ls[l];
			L.nt = h_counts[4 * l]; L.ng = h_counts[4 * l + 1]; L.nlh = h_counts[4 * l + 2]; L.nd = h_counts[4 * l + 3];
			if (L.nt > max_refs[l] || (uint64_t)L.ng + L.nlh + L.nd != L.nt) throw Error(HRY_E_INTERNAL, "references on the device: the kinds do not add up");
			L.type_sym = arena + at[l].type_sym; L.gh_vals = (const uint32_t*)(arena + at[l].gh_vals); L.lh_vals = (const uint32_t*)(arena + at[l].lh_vals);
			L.d_idx = (const uint32_t*)(arena + at[l].d_idx); L.d_he = (const uint32_t*)(arena + at[l].d_he); L.d_slot = arena + at[l].d_slot;
			L.gh = arena + at[l].gh; L.lh = arena + at[l].lh; L.data = arena + at[l].planes;
		}
	}
	for (size_t l = 0; l < nl; ++l) {
		launch_split_bytes(cx.stream, ls[l].gh_vals, ls[l].ng, 4, ls[l].gh);
		launch_split_bytes(cx.stream, ls[l].lh_vals, ls[l].nlh, 2, ls[l].lh);
	}
	cx.d_rank.ensure(std::max<size_t>((size_t)m.nv * 4 + (size_t)m.nf * 4, 16));
	uint32_t *d_rank = cx.d_rank.as<uint32_t>(), *d_frank = d_rank + m.nv;
	HIP_OK(hipMemsetAsync(d_rank, 0xff, (size_t)m.nv * 4, cx.stream));
	launch_rank(cx.stream, cx.d_order_v.as<uint32_t>(), vc, cv.org, d_rank);
	launch_face_rank(cx.stream, cv, cx.d_order_f.as<uint32_t>(), fc, d_frank);
	for (size_t l = 0; l < nl; ++l) {
		const Lst &L = ls[l];
		if (!L.nd || !L.nbytes) continue;
		const ListDesc ld = make_list_desc(m.lists[l]);
		const uint8_t *rec = cx.d_rec[l].as<uint8_t>();
		if (m.lists[l].target == 1) launch_gen_vtx_resid(cx.stream, cv, gv, d_rank, L.d_he, L.d_slot, L.d_idx, L.nd, rec, ld, L.data);
		else if (m.lists[l].target == 2) launch_gen_corner_resid(cx.stream, cv, gv, d_frank, L.d_he, L.d_slot, L.d_idx, L.nd, rec, ld, L.data);
		else launch_gen_face_resid(cx.stream, L.d_idx, L.nd, rec, ld, L.data);
	}
	for (const GenPlane &g : general_plane_layout(m)) {
		const Lst *L = g.list >= 0 ? &ls[g.list] : nullptr;
		switch (g.what) {
		case GP_REGV: planes.push_back(PlaneRef{ rv, n_rv, g.init }); break;
		case GP_REGF: planes.push_back(PlaneRef{ rf, n_rf, g.init }); break;
		case GP_TYPE: planes.push_back(PlaneRef{ L->type_sym, L->nt, g.init }); break;
		case GP_GHIST: planes.push_back(PlaneRef{ L->gh + (size_t)g.byte * L->ng, L->ng, g.init }); break;
		case GP_LHIST: planes.push_back(PlaneRef{ L->lh + (size_t)g.byte * L->nlh, L->nlh, g.init }); break;
		default: planes.push_back(PlaneRef{ L->data + (size_t)g.byte * L->nd, L->nd, g.init }); break;
		}
	}
}

namespace dev { void launch_residuals_to_rec(hipStream_t st, const uint8_t *planes, uint32_t n, const ListDesc &ld, uint8_t *rec); }

void general_planes_decode(Context &cx, Mesh &m, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                           const std::vector<uint32_t> &seg_level, const uint8_t *d_syms, const std::vector<uint64_t> &plane_off,
                           const std::vector<uint32_t> &nsym, uint32_t first)
{
	const std::vector<GenPlane> layout = general_plane_layout(m);
	// the planes that say which record every element names come down; the residual bytes stay where they are
	std::vector<std::vector<uint8_t>> host(layout.size());
	for (size_t q = 0; q < layout.size(); ++q) {
		if (layout[q].what == GP_DATA) continue;
		host[q].resize(nsym[first + q]);
		if (!host[q].empty()) HIP_OK(hipMemcpyAsync(host[q].data(), d_syms + plane_off[first + q], host[q].size(), hipMemcpyDeviceToHost, cx.stream));
	}
	HIP_OK(hipStreamSynchronize(cx.stream));
	GenHostPlanes hp;
	hp.lists.resize(m.lists.size());
	std::vector<size_t> data_plane0(m.lists.size(), 0);
	for (size_t q = 0; q < layout.size(); ++q) {
		const GenPlane &g = layout[q];
		const uint32_t n = nsym[first + q];
		switch (g.what) {
		case GP_REGV: hp.regv = host[q].data(); hp.n_regv = n; if (!hp.regv) hp.regv = (const uint8_t*)""; break;
		case GP_REGF: hp.regf = host[q].data(); hp.n_regf = n; if (!hp.regf) hp.regf = (const uint8_t*)""; break;
		case GP_TYPE: hp.lists[g.list].type = host[q].data(); hp.lists[g.list].n_type = n; break;
		case GP_GHIST:
			if (g.byte && n != hp.lists[g.list].n_gh) throw Error(HRY_E_FORMAT, "history planes of different length");
			hp.lists[g.list].gh[g.byte] = host[q].data(); hp.lists[g.list].n_gh = n; break;
		case GP_LHIST:
			if (g.byte && n != hp.lists[g.list].n_lh) throw Error(HRY_E_FORMAT, "history planes of different length");
			hp.lists[g.list].lh[g.byte] = host[q].data(); hp.lists[g.list].n_lh = n; break;
		default:
			if (g.byte == 0) { data_plane0[g.list] = first + q; hp.lists[g.list].n_data = n; }
			else if (n != hp.lists[g.list].n_data) throw Error(HRY_E_FORMAT, "residual planes of different length");
			break;
		}
	}
	std::vector<GenRecordEvents> ev;
	// A vertex list in which every vertex owns a record takes the PLY layout's chains, straight from the decoded planes -- and
	// BESIDE the host's bookkeeping of the other lists (which record every corner / face names: 6 ms per 180 000 triangles, as
	// long as the vertex chain): a helper thread drives the chain on copies of the connectivity and the list's records while this
	// thread reads the planes.  Whether the list qualifies is known from the planes' lengths (one residual symbol in the list's first
	// data plane per record coded as data).
	int fast_l = -1;
	if (m.bind.nregs_vtx() == 1 && m.bind.nvtxlists(0) == 1 && !getenv("HRY_GENERIC_VERTEX") && !order_v.empty()) {
		const int l = m.bind.vtxlist(0, 0);
		if (hp.lists[l].n_data == order_v.size() && m.lists[l].coded_bytes() > 0 && vertex_list_fast_applicable(m, l, order_v.size())) fast_l = l;
		// ... and from its reference kinds: every vertex must create its record (kind DATA).  A damaged stream whose data plane merely
		// has the right length would otherwise get a whole device reconstruction before the reader refuses it.
		if (fast_l >= 0) {
			const GenHostPlanes::L &P = hp.lists[l];
			bool all_data = P.n_type == order_v.size() && P.type != nullptr;
			for (uint32_t i = 0; all_data && i < P.n_type; ++i) all_data = P.type[i] == 0;
			if (!all_data) fast_l = -1;
		}
	}
	if (fast_l < 0) read_general_planes(m, order_v, hp, ev);
	else {
		Mesh t;
		t.nv = m.nv; t.nf = m.nf; t.declared_ne = m.declared_ne; t.have_degree = m.have_degree;   // (the connectivity itself is lent: both threads only read it)
		{
			BigVec<uint8_t> held;
			held.swap(m.lists[fast_l].data);
			t.lists[1] = m.lists[fast_l];      // the description; the records move
			t.lists[1].data.swap(held);
		}
		std::exception_ptr chain_error, reader_error;
		const long long origin = trace_origin_ns();
		std::thread chain([&] {
			try { reconstruct_vertex_list_detached(cx, t, m, order_v, seg_start, seg_level, d_syms + plane_off[data_plane0[fast_l]], origin); }
			catch (...) { chain_error = std::current_exception(); }
		});
		try { read_general_planes(m, order_v, hp, ev); } catch (...) { reader_error = std::current_exception(); }
		chain.join();
		t.lists[1].data.swap(m.lists[fast_l].data);
		if (reader_error) std::rethrow_exception(reader_error);
		if (chain_error) std::rethrow_exception(chain_error);
		ev[fast_l].he.clear(); ev[fast_l].slot.clear();
	}
	reconstruct_general(cx, m, order_v, ev, [&] {
		for (size_t l = 0; l < m.lists.size(); ++l) {
			const uint32_t nd = (uint32_t)ev[l].he.size();
			if (!nd || !m.lists[l].coded_bytes()) continue;
			launch_residuals_to_rec(cx.stream, d_syms + plane_off[data_plane0[l]], nd, make_list_desc(m.lists[l]), cx.d_rec[l].as<uint8_t>());
		}
	});
}

}   // namespace hry
