// C ABI of libharry_amd.so (include/harry_amd.h).  Thin: argument checks, exception -> status translation.
#include <cstdlib>
#include <cstring>
#include <new>

#include "device/context.hpp"

using namespace hry;

struct hry_ctx { Context cx; explicit hry_ctx(int d) : cx(d) {} };
struct hry_mesh { Mesh m; };
struct hry_plan { ShardPlan p; };
struct hry_walk {
	WalkResult w; uint32_t info[2]; std::vector<uint8_t> vplanes, fplanes; std::vector<uint32_t> seg_start, seg_level;
	mutable std::vector<uint8_t> op_sym, op_class;   // unpacked from w.op_sc on first request
	mutable std::vector<uint32_t> op_thr, op_cum;    // op_position_table, on first request
	mutable bool have_table = false;
	mutable std::vector<uint8_t> snap_section;       // the border snapshots as the chunked container's directory holds them, on first request
	void table() const { if (!have_table) { op_position_table(w, op_thr, op_cum); have_table = true; } }
	void snapshots() const
	{
		if (!snap_section.empty() || w.snapshots.empty()) return;
		std::vector<RestartCounters> rc, sc;
		(void)select_restart_points(w.marks, w.named, rc, &w.snapshots, &sc);
		write_snapshot_section(w.snapshot_faces, w.snapshots, sc, snap_section);
	}
	void unpack_ops() const
	{
		if (op_sym.size() == w.op_sc.size()) return;
		op_sym.resize(w.op_sc.size()); op_class.resize(w.op_sc.size());
		for (size_t i = 0; i < w.op_sc.size(); ++i) { op_sym[i] = op_u8(w.op_sc[i]) & 7; op_class[i] = op_u8(w.op_sc[i]) >> 3; }
	}
};

static thread_local std::string g_last_error;

template <typename F> static int guarded(F &&f)
{
	try { f(); return HRY_OK; }
	catch (const Error &e) { g_last_error = e.what(); return e.code; }
	catch (const std::bad_alloc &) { g_last_error = "out of memory"; return HRY_E_NOMEM; }
	catch (const std::exception &e) { g_last_error = e.what(); return HRY_E_INTERNAL; }
}
static uint8_t *dup_bytes(const std::vector<uint8_t> &v)
{
	uint8_t *p = (uint8_t*)malloc(v.size() ? v.size() : 1);
	if (!p) throw std::bad_alloc();
	if (!v.empty()) memcpy(p, v.data(), v.size());
	return p;
}

extern "C" {

const char *hry_last_error(void) { return g_last_error.c_str(); }
int hry_abi_version(void) { return HRY_ABI_VERSION; }

int hry_device_count(void)
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}
int hry_ctx_create(int device, hry_ctx **out)
{
	if (!out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] { *out = new hry_ctx(device); });
}
void hry_ctx_destroy(hry_ctx *ctx) { delete ctx; }
int hry_ctx_timing(const hry_ctx *ctx, hry_timing *out)
{
	if (!ctx || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = ctx->cx.timing;
	return HRY_OK;
}
void *hry_ctx_stream(const hry_ctx *ctx) { return ctx ? (void*)ctx->cx.stream : nullptr; }

int hry_mesh_from_ply(const uint8_t *ply, size_t n, hry_mesh **out)
{
	if (!ply || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<Mesh> m(mesh_from_ply(ply, n));
		*out = new hry_mesh{ std::move(*m) };
	});
}
int hry_mesh_from_arrays(uint32_t nv, const uint8_t *vrec, int v_ncomp, const uint8_t *v_types, const char *const *v_names,
                         uint32_t nf, const uint8_t *degrees, const uint32_t *indices,
                         const uint8_t *frec, int f_ncomp, const uint8_t *f_types, const char *const *f_names, hry_mesh **out)
{
	if (!out || (nf && (!degrees || !indices)) || (v_ncomp && (!v_types || !v_names || (nv && !vrec))) || (f_ncomp && (!f_types || !f_names || (nf && !frec)))) {
		g_last_error = "null argument";
		return HRY_E_ARG;
	}
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<Mesh> m(mesh_from_arrays(nv, vrec, v_ncomp, v_types, v_names, nf, degrees, indices, frec, f_ncomp, f_types, f_names));
		*out = new hry_mesh{ std::move(*m) };
	});
}
int hry_mesh_to_ply(const hry_mesh *m, int ascii, uint8_t **out, size_t *out_len)
{
	if (!m || !out || !out_len) { g_last_error = "null argument"; return HRY_E_ARG; }
	return guarded([&] {
		ByteSink v;
		mesh_to_ply(m->m, (ascii & HRY_PLY_ASCII) != 0, v, (ascii & HRY_PLY_PACKED) != 0);
		*out = v.release(out_len);
	});
}
int hry_mesh_from_obj(const uint8_t *obj, size_t n, const char *dir, hry_mesh **out)
{
	if (!obj || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<Mesh> m(mesh_from_obj(obj, n, dir));
		*out = new hry_mesh{ std::move(*m) };
	});
}
int hry_mesh_to_obj(const hry_mesh *m, int, uint8_t **out, size_t *out_len)
{
	if (!m || !out || !out_len) { g_last_error = "null argument"; return HRY_E_ARG; }
	return guarded([&] {
		ByteSink v;
		mesh_to_obj(m->m, v);
		*out = v.release(out_len);
	});
}
int hry_mesh_general(const hry_mesh *m) { return m && m->m.general ? 1 : 0; }
int hry_list_target(const hry_mesh *m, int l) { return m && l >= 0 && (size_t)l < m->m.lists.size() ? m->m.lists[l].target : 3; }
int hry_mesh_nregions(const hry_mesh *m, int which)
{
	if (!m) return 0;
	if (!m->m.general) return 1;
	return which == 0 ? m->m.bind.nregs_face() : m->m.bind.nregs_vtx();
}
int hry_mesh_region_lists(const hry_mesh *m, int kind, int r, uint16_t *out, int cap)
{
	if (!m || r < 0) return 0;
	if (!m->m.general) { if (kind == 2 || r != 0) return 0; if (out && cap > 0) out[0] = kind == 0 ? 0 : 1; return 1; }
	const Bindings &b = m->m.bind;
	if (r >= (kind == 1 ? b.nregs_vtx() : b.nregs_face())) return 0;
	const int n = kind == 0 ? b.nfacelists(r) : kind == 1 ? b.nvtxlists(r) : b.ncornerlists(r);
	for (int a = 0; a < n && a < cap && out; ++a) out[a] = (uint16_t)(kind == 0 ? b.facelist(r, a) : kind == 1 ? b.vtxlist(r, a) : b.cornerlist(r, a));
	return n;
}
size_t hry_mesh_regions_of(const hry_mesh *m, int which, const uint16_t **out)
{
	if (!m || !out || !m->m.general) return 0;
	const BigVec<uint16_t> &v = which == 0 ? m->m.bind.face_reg : m->m.bind.vtx_reg;
	*out = v.data();
	return v.size();
}
size_t hry_mesh_bindings(const hry_mesh *m, int kind, const uint32_t **out, int *slots)
{
	if (!m || !out || !slots || !m->m.general) return 0;
	const Bindings &b = m->m.bind;
	const BigVec<uint32_t> &v = kind == 0 ? b.face_attr : kind == 1 ? b.vtx_attr : b.corner_attr;
	*slots = kind == 0 ? b.nb_face : kind == 1 ? b.nb_vtx : b.nb_corner;
	*out = v.data();
	return kind == 0 ? m->m.nf : kind == 1 ? m->m.nv : m->m.ne();
}
void hry_mesh_free(hry_mesh *m) { delete m; }
hry_mesh *hry_mesh_clone(const hry_mesh *m)
{
	if (!m) return nullptr;
	try { hry_mesh *c = new hry_mesh{ m->m }; c->m.device_token = 0; return c; }
	catch (...) { g_last_error = "out of memory"; return nullptr; }
}

uint32_t hry_mesh_nv(const hry_mesh *m) { return m->m.nv; }
uint32_t hry_mesh_nf(const hry_mesh *m) { return m->m.nf; }
uint32_t hry_mesh_ne(const hry_mesh *m) { return m->m.ne(); }
uint64_t hry_mesh_ntri(const hry_mesh *m) { return m->m.ntri(); }
const uint32_t *hry_mesh_face_offsets(const hry_mesh *m) { return m->m.face_off.data(); }
const uint32_t *hry_mesh_org(const hry_mesh *m) { return m->m.org.data(); }
const uint32_t *hry_mesh_twin(const hry_mesh *m)
{
	try { ensure_twins(m->m); } catch (...) { return nullptr; }
	return m->m.twin.data();
}
int hry_mesh_nlists(const hry_mesh *m) { return (int)m->m.lists.size(); }
int hry_list_ncomp(const hry_mesh *m, int l) { return m->m.lists[l].ncomp(); }
uint32_t hry_list_count(const hry_mesh *m, int l) { return m->m.lists[l].count; }
int hry_list_stride(const hry_mesh *m, int l) { return m->m.lists[l].stride(); }
int hry_list_type(const hry_mesh *m, int l, int c) { return m->m.lists[l].type[c]; }
int hry_list_quant(const hry_mesh *m, int l, int c) { return m->m.lists[l].quant[c]; }
int hry_list_offset(const hry_mesh *m, int l, int c) { return m->m.lists[l].offset[c]; }
const uint8_t *hry_list_data(const hry_mesh *m, int l) { return m->m.lists[l].data.data(); }
const uint8_t *hry_list_min(const hry_mesh *m, int l) { return m->m.lists[l].have_bounds ? m->m.lists[l].bmin.data() : nullptr; }
const uint8_t *hry_list_max(const hry_mesh *m, int l) { return m->m.lists[l].have_bounds ? m->m.lists[l].bmax.data() : nullptr; }

int hry_bounds(hry_ctx *ctx, hry_mesh *m)
{
	if (!ctx || !m) { g_last_error = "null argument"; return HRY_E_ARG; }
	return guarded([&] { device_bounds(ctx->cx, m->m); });
}
int hry_requant(hry_ctx *ctx, hry_mesh *m, const hry_quant *q, size_t nq, int clear)
{
	if (!ctx || !m || (nq && !q)) { g_last_error = "null argument"; return HRY_E_ARG; }
	return guarded([&] { device_requant(ctx->cx, m->m, q, nq, clear != 0); });
}
int hry_mesh_upload(hry_ctx *ctx, hry_mesh *m)
{
	if (!ctx || !m) { g_last_error = "null argument"; return HRY_E_ARG; }
	return guarded([&] { if (m->m.general) upload_general(ctx->cx, m->m); else ctx->cx.upload_mesh(m->m); });
}
int hry_encode(hry_ctx *ctx, hry_mesh *m, const hry_opts *opts, uint8_t **out, size_t *out_len)
{
	if (!ctx || !m || !out || !out_len) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr; *out_len = 0;
	return guarded([&] {
		hry_opts o = opts ? *opts : hry_opts{};
		ctx->cx.keep_stages = o.keep_stages != 0;
		ctx->cx.device_recurrence = (o.flags & HRY_FLAG_DEVICE_RECURRENCE) != 0;
		ctx->cx.stages.clear();
		if (o.profile == HRY_PROFILE_CHUNKED) {
			// straight into the buffer the caller gets: a vector first cost a zero fill, a second set of fresh pages and a copy --
			// 10 ms of a 75 ms encode of the 12.6 M-triangle share of configs[3] (39 MB of container)
			ByteSink sink;
			encode_chunked(ctx->cx, m->m, o.chunk_syms, sink);
			*out = sink.release(out_len);
			return;
		}
		std::vector<uint8_t> v;
		if (o.profile == HRY_PROFILE_COMPAT) encode_compat(ctx->cx, m->m, v);
		else throw Error(HRY_E_ARG, "unknown profile");
		*out = dup_bytes(v);
		*out_len = v.size();
	});
}
int hry_decode(hry_ctx *ctx, const uint8_t *hry, size_t n, const hry_opts *opts, hry_mesh **out)
{
	if (!ctx || !hry || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		hry_opts o = opts ? *opts : hry_opts{};
		ctx->cx.keep_stages = o.keep_stages != 0;
		ctx->cx.stages.clear();
		std::unique_ptr<Mesh> m(decode_any(ctx->cx, hry, n, o.shard_index, o.shard_count, (o.flags & HRY_FLAG_PARTIAL) != 0));
		*out = new hry_mesh{ std::move(*m) };
	});
}
int hry_encode_sharded(hry_ctx *const *ctx, int n_ctx, hry_mesh *m, const hry_quant *quant, size_t n_quant, int clear,
                       const hry_opts *opts, uint8_t **out, size_t *out_len, hry_shard_timing *timing)
{
	if (!ctx || n_ctx <= 0 || !m || !out || !out_len || (n_quant && !quant)) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr; *out_len = 0;
	return guarded([&] {
		hry_opts o = opts ? *opts : hry_opts{};
		o.profile = opts ? o.profile : HRY_PROFILE_CHUNKED;
		if (o.profile != HRY_PROFILE_CHUNKED) throw Error(HRY_E_UNSUPPORTED, "the reference's single stream (compat) does not shard: one recurrence over the whole file");
		std::vector<Context*> cxs;
		for (int i = 0; i < n_ctx; ++i) { if (!ctx[i]) throw Error(HRY_E_ARG, "null context"); ctx[i]->cx.keep_stages = false; ctx[i]->cx.stages.clear(); cxs.push_back(&ctx[i]->cx); }
		ByteSink v;
		hry_shard_timing st{};
		encode_sharded(cxs.data(), n_ctx, m->m, quant, n_quant, clear != 0, o.shard_count, o.chunk_syms, v, st, (o.flags & HRY_FLAG_KEEP_MESH) == 0);
		if (timing) *timing = st;
		*out = v.release(out_len);
	});
}
int hry_decode_sharded(hry_ctx *const *ctx, int n_ctx, const uint8_t *hry, size_t n, const hry_opts *opts, hry_mesh **out, hry_shard_timing *timing)
{
	if (!ctx || n_ctx <= 0 || !hry || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		hry_opts o = opts ? *opts : hry_opts{};
		std::vector<Context*> cxs;
		for (int i = 0; i < n_ctx; ++i) { if (!ctx[i]) throw Error(HRY_E_ARG, "null context"); ctx[i]->cx.keep_stages = false; ctx[i]->cx.stages.clear(); cxs.push_back(&ctx[i]->cx); }
		std::unique_ptr<Mesh> g(new Mesh());
		int minor = 0;
		const bool sharded = n >= 6 && hry[4] == 0 && hry[5] == 3;
		if (!sharded) {   // an unsharded file: the first context decodes it
			std::unique_ptr<Mesh> m(decode_any(*cxs[0], hry, n, 0, 0, false));
			if (timing) { *timing = hry_shard_timing{}; timing->n_contexts = 1; timing->n_segments = 1; timing->total_ms = cxs[0]->timing.total_ms; }
			*out = new hry_mesh{ std::move(*m) };
			return;
		}
		const size_t hdr = read_hry_header(hry, n, *g, minor, false);
		std::unique_ptr<Mesh> m(decode_sharded(cxs.data(), n_ctx, hry, n, hdr, std::move(g), o.shard_index, o.shard_count,
		                                       (o.flags & HRY_FLAG_PARTIAL) != 0 || o.shard_count > 1, timing));
		*out = new hry_mesh{ std::move(*m) };
	});
}
int hry_container_check(const uint8_t *hry, size_t n, int *complete)
{
	if (!hry) { g_last_error = "null argument"; return HRY_E_ARG; }
	if (complete) *complete = 0;
	return guarded([&] {
		Mesh m;
		int minor = 0;
		const size_t hdr = read_hry_header(hry, n, m, minor, false);
		if (minor != 3) { if (complete) *complete = 1; return; }
		ShardedDirectory dir;
		std::vector<uint32_t> counts;
		for (const AttrList &L : m.lists) counts.push_back(L.count);
		parse_sharded_directory(hry, n, hdr, m.nv, m.nf, m.declared_ne, dir, true, m.general ? &counts : nullptr);
		if (complete) *complete = dir.complete ? 1 : 0;
	});
}
int hry_mesh_partial(const hry_mesh *m) { return m && m->m.partial ? 1 : 0; }
void hry_free(void *p) { BlockPool::give(p); }   // (a block of the recycling pool goes back to it, anything else to the C library)
int hry_container_info(const uint8_t *hry, size_t n, uint32_t info[8])
{
	if (!hry || !info) { g_last_error = "null argument"; return HRY_E_ARG; }
	return guarded([&] {
		Mesh m;
		int minor = 0;
		const size_t hdr = read_hry_header(hry, n, m, minor, false);
		for (int i = 0; i < 8; ++i) info[i] = 0;
		info[0] = (uint32_t)minor; info[1] = (uint32_t)hdr; info[2] = m.nv; info[3] = m.nf; info[4] = m.declared_ne; info[7] = 1;
		size_t body = hdr;
		if (minor == 3) {
			ShardedDirectory dir;
			std::vector<uint32_t> counts;
			for (const AttrList &L : m.lists) counts.push_back(L.count);
			parse_sharded_directory(hry, n, hdr, m.nv, m.nf, m.declared_ne, dir, true, m.general ? &counts : nullptr);
			info[7] = (uint32_t)dir.segments.size();
			if (dir.segments.empty()) return;
			body = dir.segments[0].offset + dir.segments[0].body_at;   // (general bindings: runs carry their record ranges)
		}
		if (minor >= 2) {
			if (n < body + 8) throw Error(HRY_E_FORMAT, "truncated chunked directory");
			memcpy(&info[5], hry + body, 4); memcpy(&info[6], hry + body + 4, 4);
		}
	});
}

int hry_shard_plan(const hry_mesh *m, int n_shards, hry_plan **out)
{
	if (!m || !out || n_shards <= 0) { g_last_error = "invalid argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<hry_plan> p(new hry_plan());
		shard_plan(m->m, (uint32_t)n_shards, p->p);
		*out = p.release();
	});
}
void hry_plan_free(hry_plan *p) { delete p; }
int hry_walk_run_shard(hry_mesh *m, const hry_plan *p, int shard, hry_walk **out)
{
	if (!m || !p || !out || shard < 0) { g_last_error = "invalid argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<hry_walk> w(new hry_walk());
		check_codable(m->m);
		ensure_twins(m->m);
		if (p->p.g_nv != m->m.nv || p->p.g_nf != m->m.nf || p->p.g_ne != m->m.ne()) throw Error(HRY_E_ARG, "the plan belongs to another mesh");
		ComponentAnalysis part;
		ShardInfo info;
		shard_components(p->p, (uint32_t)shard, part, info);
		int ud = 0;
		const bool uniform = m->m.uniform_degree(ud) && (ud == 3 || ud == 4);
		BigVec<uint32_t> eface;
		if (!uniform && p->p.A.eface.size() != m->m.ne()) {
			eface.resize(m->m.ne());
			for (uint32_t f = 0; f < m->m.nf; ++f) for (uint32_t h = m->m.face_off[f]; h < m->m.face_off[f + 1]; ++h) eface[h] = f;
		}
		WalkState marks(m->m.nv, m->m.nf);
		{ uint64_t nfs = 0; for (uint32_t k = 0; k < part.ncomp; ++k) nfs += part.n_faces[k]; w->w.snapshot_faces = snapshot_spacing((uint32_t)nfs); }   // (what an encode of the shard asks for)
		cut_border_walk_in_place(m->m, part, uniform ? nullptr : eface.empty() ? p->p.A.eface.data() : eface.data(), marks, w->w);
		w->info[0] = w->w.n_conn; w->info[1] = w->w.numtri_coded ? 1 : 0;
		*out = w.release();
	});
}
int hry_analysis_check(hry_ctx *ctx, hry_mesh *m)
{
	if (!ctx || !m) { g_last_error = "invalid argument"; return HRY_E_ARG; }
	return guarded([&] {
		check_codable(m->m);
		if (m->m.general || !m->m.shard.seeds.empty()) throw Error(HRY_E_ARG, "analysis check: a mesh in the PLY layout that is not a shard");
		if (m->m.device_token == 0 || m->m.device_token != ctx->cx.resident_token) ctx->cx.upload_mesh(m->m);
		ComponentAnalysis D, H;
		device_component_analysis(ctx->cx, m->m, D);
		analyse_components(m->m, H);
		if (D.ncomp != H.ncomp) throw Error(HRY_E_INTERNAL, "analysis: " + std::to_string(D.ncomp) + " components on the device, " + std::to_string(H.ncomp) + " on the host");
		if (D.ncomp < 2) return;
		auto same = [&](const char *what, const std::vector<uint32_t> &d, const std::vector<uint32_t> &h) {
			if (d.size() != h.size()) throw Error(HRY_E_INTERNAL, std::string("analysis: size of ") + what);
			for (size_t i = 0; i < d.size(); ++i)
				if (d[i] != h[i]) throw Error(HRY_E_INTERNAL, std::string("analysis: ") + what + "[" + std::to_string(i) + "] = " + std::to_string(d[i]) + " on the device, " + std::to_string(h[i]) + " on the host");
		};
		same("seed", D.seed, H.seed); same("n_faces", D.n_faces, H.n_faces); same("n_halfedges", D.n_halfedges, H.n_halfedges);
		same("fresh", D.fresh, H.fresh); same("group", D.group, H.group);
		same("face_lo", D.face_lo, H.face_lo); same("face_hi", D.face_hi, H.face_hi); same("vtx_lo", D.vtx_lo, H.vtx_lo); same("vtx_hi", D.vtx_hi, H.vtx_hi);
		// (component NUMBERS are the roots' order in both: by_rank / rank_of agree too)
		same("by_rank", D.by_rank, H.by_rank);
	});
}
uint32_t hry_plan_ncomponents(const hry_plan *p) { return p ? p->p.A.ncomp : 0; }
uint32_t hry_plan_ngroups(const hry_plan *p)
{
	uint32_t n = 0;
	if (p) for (uint32_t k = 0; k < p->p.A.ncomp; ++k) n += p->p.A.group[k] == k;
	return n;
}
uint64_t hry_plan_triangles(const hry_plan *p, int shard) { return p && shard >= 0 && (uint32_t)shard < p->p.n_shards ? p->p.shard_triangles[shard] : 0; }
int hry_shard_extract(const hry_mesh *m, const hry_plan *p, int shard, hry_mesh **out)
{
	if (!m || !p || !out || shard < 0) { g_last_error = "invalid argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<Mesh> s(shard_extract(m->m, p->p, (uint32_t)shard));
		*out = new hry_mesh{ std::move(*s) };
	});
}
int hry_merge(const uint8_t *const *parts, const size_t *sizes, size_t n, uint8_t **out, size_t *out_len)
{
	if (!parts || !sizes || !out || !out_len) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr; *out_len = 0;
	return guarded([&] {
		ByteSink v;
		merge_containers(parts, sizes, n, v);
		*out = v.release(out_len);
	});
}
size_t hry_mesh_runs(const hry_mesh *m, const uint32_t **runs)
{
	if (!m || !runs) return 0;
	const std::vector<ShardRun> &r = m->m.shard.active() ? m->m.shard.runs : m->m.covered;
	*runs = (const uint32_t*)r.data();
	return r.size();
}
size_t hry_shard_elements(const hry_mesh *m, int which, const uint32_t **idx)
{
	if (!m || !idx) return 0;
	if (which >= 16) {   // general bindings: input index in the whole mesh of every record of list which - 16
		if ((size_t)(which - 16) >= m->m.shard.record_of.size()) return 0;
		*idx = m->m.shard.record_of[which - 16].data();
		return m->m.shard.record_of[which - 16].size();
	}
	const std::vector<uint32_t> &v = which == 2 ? m->m.shard.seeds : which ? m->m.shard.vertex_of : m->m.shard.face_of;
	*idx = v.data();
	return v.size();
}
int hry_list_set_bounds(hry_mesh *m, int l, const uint8_t *min_rec, const uint8_t *max_rec)
{
	if (!m || l < 0 || (size_t)l >= m->m.lists.size() || !min_rec || !max_rec) { g_last_error = "invalid argument"; return HRY_E_ARG; }
	AttrList &L = m->m.lists[l];
	for (int c = 0; c < L.ncomp(); ++c) if (L.quant[c]) { g_last_error = "bounds of an already quantised list come from its header"; return HRY_E_ARG; }
	L.bmin.assign(min_rec, min_rec + L.stride());
	L.bmax.assign(max_rec, max_rec + L.stride());
	L.bmin_at.clear(); L.bmax_at.clear();
	L.have_bounds = true;
	return HRY_OK;
}
uint32_t hry_list_min_at(const hry_mesh *m, int l, int c) { return m && l >= 0 && (size_t)l < m->m.lists.size() && c >= 0 && (size_t)c < m->m.lists[l].bmin_at.size() ? m->m.lists[l].bmin_at[c] : 0; }
uint32_t hry_list_max_at(const hry_mesh *m, int l, int c) { return m && l >= 0 && (size_t)l < m->m.lists.size() && c >= 0 && (size_t)c < m->m.lists[l].bmax_at.size() ? m->m.lists[l].bmax_at[c] : 0; }

int hry_stage_get(hry_ctx *ctx, const char *name, void **host_copy, size_t *bytes)
{
	if (!ctx || !name || !host_copy || !bytes) { g_last_error = "null argument"; return HRY_E_ARG; }
	auto it = ctx->cx.stages.find(name);
	if (it == ctx->cx.stages.end()) { g_last_error = std::string("no such stage: ") + name; return HRY_E_ARG; }
	return guarded([&] { *host_copy = dup_bytes(it->second); *bytes = it->second.size(); });
}
int hry_walk_run(hry_mesh *m, hry_walk **out)
{
	if (!m || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<hry_walk> w(new hry_walk());
		check_codable(m->m);
		cut_border_walk(m->m, w->w);
		w->info[0] = w->w.n_conn; w->info[1] = w->w.numtri_coded ? 1 : 0;
		*out = w.release();
	});
}
int hry_walk_run_plain(hry_mesh *m, hry_walk **out)
{
	if (!m || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<hry_walk> w(new hry_walk());
		check_codable(m->m);
		w->w.snapshot_faces = snapshot_spacing(m->m.nf);   // (what a chunked encode asks for)
		cut_border_walk(m->m, w->w, false);
		w->info[0] = w->w.n_conn; w->info[1] = w->w.numtri_coded ? 1 : 0;
		*out = w.release();
	});
}
size_t hry_walk_get(const hry_walk *w, const char *name, const void **ptr)
{
	if (!w || !name || !ptr) return 0;
	std::string n(name);
	const WalkResult &r = w->w;
	auto ret = [&](const auto &v) { *ptr = v.data(); return v.size(); };
	if (n == "order_v") return ret(r.order_v);
	if (n == "order_f") return ret(r.order_f);
	if (n == "op_sym") { w->unpack_ops(); return ret(w->op_sym); }
	if (n == "op_class") { w->unpack_ops(); return ret(w->op_class); }
	if (n == "op_l") return ret(r.op_l);
	if (n == "op_h") return ret(r.op_h);
	if (n == "op_t") return ret(r.op_t);
	if (n == "op_pos") return ret(r.op_pos);
	if (n == "op_thr") { w->table(); return ret(w->op_thr); }   // where the connectivity groups sit between the operations (what the
	if (n == "op_cum") { w->table(); return ret(w->op_cum); }   // device's operation model places its records with)
	if (n == "info") { *ptr = w->info; return 2; }
	if (n == "snap_section") { w->snapshots(); return ret(w->snap_section); }
	if (n == "marks") { *ptr = r.marks.data(); return r.marks.size() * (sizeof(ComponentMark) / 4); }
	if (n == "seg_start") return ret(w->seg_start);
	if (n == "seg_level") return ret(w->seg_level);
	if (n == "vplanes") return ret(w->vplanes);
	if (n == "fplanes") return ret(w->fplanes);
	if (n.size() == 8 && n.compare(0, 3, "grp") == 0 && n[3] >= '0' && n[3] < '0' + G_COUNT) {
		int g = n[3] - '0';
		if (n.compare(4, 4, "_val") == 0) return ret(r.grp_val[g]);
		if (n.compare(4, 4, "_pos") == 0) return ret(r.grp_pos[g]);
	}
	*ptr = nullptr;
	return 0;
}
void hry_walk_free(hry_walk *w) { delete w; }

int hry_stream_read_host(const void *hry, size_t bytes, hry_mesh **mesh, hry_walk **out)
{
	if (!hry || !mesh || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*mesh = nullptr; *out = nullptr;
	return guarded([&] {
		std::unique_ptr<hry_mesh> m(new hry_mesh());
		std::unique_ptr<hry_walk> w(new hry_walk());
		int minor = 0;
		size_t hdr = read_hry_header((const uint8_t*)hry, bytes, m->m, minor);
		if (minor != 1) throw Error(HRY_E_ARG, "not a single-stream (v0.1) file");
		if (m->m.general) throw Error(HRY_E_UNSUPPORTED, "hry_stream_read_host returns the planes of the PLY layout only");
		std::vector<uint32_t> seg_start, seg_level;
		OrderVec order_v;
		read_compat_stream((const uint8_t*)hry + hdr, bytes - hdr, m->m, order_v, seg_start, seg_level, w->vplanes, w->fplanes);
		w->w.order_v.assign(order_v.begin(), order_v.end());
		w->info[0] = w->info[1] = 0;
		*mesh = m.release();
		*out = w.release();
	});
}

int hry_walk_replay(const hry_mesh *src, const hry_walk *walk, int use_restart_points, hry_mesh **mesh, hry_walk **out)
{
	if (!src || !walk || !mesh || !out) { g_last_error = "null argument"; return HRY_E_ARG; }
	*mesh = nullptr; *out = nullptr;
	return guarded([&] {
		const WalkResult &r = walk->w;
		// the 21 connectivity planes of the chunked container, built on the host: groups split into little-endian bytes,
		// operations split by order class
		std::vector<uint8_t> planes[21];
		static const int first_plane[G_COUNT] = { 0, 1, 5, 7, 11 };
		for (int g = 0; g < G_COUNT; ++g)
			for (int b = 0; b < kGroupBytes[g]; ++b) {
				std::vector<uint8_t> &pl = planes[first_plane[g] + b];
				pl.resize(r.grp_val[g].size());
				for (size_t i = 0; i < pl.size(); ++i) pl[i] = (uint8_t)(r.grp_val[g][i] >> (8 * b));
			}
		for (size_t i = 0; i < r.op_sc.size(); ++i) planes[13 + (op_u8(r.op_sc[i]) >> 3)].push_back(op_u8(r.op_sc[i]) & 7);
		std::unique_ptr<hry_mesh> m(new hry_mesh());
		std::unique_ptr<hry_walk> w(new hry_walk());
		m->m.nv = src->m.nv; m->m.nf = src->m.nf; m->m.declared_ne = src->m.ne(); m->m.have_degree = src->m.have_degree;
		// use_restart_points: 1 the restart points at component starts, 3 also the border snapshots inside components -- through the
		// directory's form and back, as a decoder receives them (the counters of a span depend on which points exist: never the snapshots alone)
		std::vector<RestartPoint> restarts;
		std::vector<RestartCounters> rcounters, scounters;
		std::vector<SnapshotPoint> snaps;
		const bool with_snaps = (use_restart_points & 2) != 0 && !r.snapshots.empty();
		if (use_restart_points & 3) restarts = select_restart_points(r.marks, r.named, rcounters, with_snaps ? &r.snapshots : nullptr, with_snaps ? &scounters : nullptr);
		if (use_restart_points == 2) throw Error(HRY_E_ARG, "replay: border snapshots go with the restart points");
		if (with_snaps) {
			std::vector<uint8_t> sec;
			write_snapshot_section(r.snapshot_faces, r.snapshots, scounters, sec);
			uint32_t spacing = 0;
			if (read_snapshot_section(sec.data(), sec.size(), m->m.nv, spacing, snaps) != sec.size() || spacing != r.snapshot_faces) throw Error(HRY_E_INTERNAL, "border snapshots: the section does not read back");
		}
		OrderVec order_v;
		PlaneView views[21];
		for (int k = 0; k < 21; ++k) views[k] = PlaneView(planes[k]);
		cut_border_replay(m->m, views, restarts, rcounters, order_v, w->seg_start, w->seg_level, nullptr, with_snaps ? &snaps : nullptr);
		w->w.order_v.assign(order_v.begin(), order_v.end());
		w->info[0] = (uint32_t)restarts.size(); w->info[1] = (uint32_t)snaps.size();
		*mesh = m.release();
		*out = w.release();
	});
}

int hry_range_encode_lht(hry_ctx *ctx, const uint64_t *lht, size_t n, uint8_t **out, size_t *out_len)
{
	if (!ctx || (n && !lht) || !out || !out_len) { g_last_error = "null argument"; return HRY_E_ARG; }
	return guarded([&] {
		std::vector<uint8_t> v;
		range_encode_lht(ctx->cx, lht, n, v);
		*out = dup_bytes(v);
		*out_len = v.size();
	});
}

}   // extern "C"
