// Sharding of one mesh over several GPUs (north_star: "meshes shard by independent connected
// component across the 8 GPUs of one node").
//
// What the reference fixes and a split has to honour:
//   * the numbering of the decoded mesh runs across ALL components in coding order: vertices, faces and half-edges are
//     numbered as the cut-border machine creates them (cbm/encoder.h:61-68,215; cbm/decoder.h:48,75,145,162);
//   * the coding order of the components is the start-face sequence (formats/hry/writer.cc:28-46): face 0 first, then the
//     iteration order of a std::unordered_set that holds the faces of the WHOLE mesh;
//   * a component that touches a vertex introduced by an earlier one names it by its index (TRIxxx start operations,
//     NM operations: cbm/encoder.h:79-113,187), and its operation classes depend on that vertex's triangle count.
// So the unit of distribution is a GROUP: components tied together by shared vertices.  A plan (host analysis of the
// connectivity only, no walk) labels the components, orders them, ties them, and takes the exclusive scans of the vertices /
// faces / half-edges each one introduces.  A shard is the sub-mesh of some groups in its own compact numbering, plus the seed
// face of each of its components (in coding order) and the place of every run of consecutive components in the full
// numbering.  Shards are coded independently (one v0.2 body each, no data-path collective) and merged into one .hry v0.3:
//
//     header of the whole mesh (minor version 3) | u32 n_segments | n_segments x u64 bytes | segments
//     segment: u32 n_runs | n_runs x { u32 first_vertex, first_face, first_halfedge, n_vertices, n_faces, n_halfedges } | v0.2 body
//
// A segment decodes on its own (any GPU); its runs say where its vertices, faces and half-edges go in the whole mesh.
#include "host.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <numeric>

namespace hry {
namespace {
constexpr uint32_t NONE32 = 0xffffffffu;

struct Ranges {
	unsigned nt;
	explicit Ranges(uint64_t n) : nt(n >= (1u << 16) ? host_threads() : 1u) {}
	void of(uint32_t n, unsigned t, uint32_t &b, uint32_t &e) const { b = (uint32_t)((uint64_t)n * t / nt); e = (uint32_t)((uint64_t)n * (t + 1) / nt); }
};
}   // namespace

void shard_plan(const Mesh &m, uint32_t n_shards, ShardPlan &plan, bool light)
{
	if (n_shards == 0) throw Error(HRY_E_ARG, "need at least one shard");
	if (m.nf == 0) throw Error(HRY_E_UNSUPPORTED, "mesh without faces");
	if (m.shard.active()) throw Error(HRY_E_ARG, "a shard cannot be sharded again");
	plan = ShardPlan();
	plan.n_shards = n_shards;
	plan.g_nv = m.nv; plan.g_nf = m.nf; plan.g_ne = m.ne();
	plan.have_degree = m.have_degree;
	ComponentAnalysis &A = plan.A;
	plan.light = light && !m.general;
	A.want_vertex_owner = !plan.light;
	const bool trace = getenv("HRY_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry plan] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	analyse_components(m, A);
	mark("components analysed");
	shard_plan_finish(m, n_shards, plan);
}

// the plan of a mesh whose components somebody else has analysed (device/analysis.cpp: the tables per coding rank, the index
// intervals; no per-face labels): a light plan, for shards that are coded where they lie
void shard_plan_from_analysis(const Mesh &m, uint32_t n_shards, ComponentAnalysis &&A, ShardPlan &plan)
{
	if (n_shards == 0) throw Error(HRY_E_ARG, "need at least one shard");
	if (m.nf == 0) throw Error(HRY_E_UNSUPPORTED, "mesh without faces");
	if (m.shard.active()) throw Error(HRY_E_ARG, "a shard cannot be sharded again");
	if (m.general) throw Error(HRY_E_INTERNAL, "shard plan: general bindings need the per-face labels");
	const uint32_t nc = A.ncomp;
	if (nc == 0 || A.seed.size() != nc || A.n_faces.size() != nc || A.n_halfedges.size() != nc || A.fresh.size() != nc || A.group.size() != nc || A.face_lo.size() != nc ||
	    A.face_hi.size() != nc || A.vtx_lo.size() != nc || A.vtx_hi.size() != nc)
		throw Error(HRY_E_INTERNAL, "shard plan: incomplete analysis");
	plan = ShardPlan();
	plan.n_shards = n_shards;
	plan.g_nv = m.nv; plan.g_nf = m.nf; plan.g_ne = m.ne();
	plan.have_degree = m.have_degree;
	plan.light = true;
	plan.A = std::move(A);
	shard_plan_finish(m, n_shards, plan);
}

// scans, groups onto shards, and (unless the plan is light) where every element goes: plan.A is complete
void shard_plan_finish(const Mesh &m, uint32_t n_shards, ShardPlan &plan)
{
	ComponentAnalysis &A = plan.A;
	const bool trace = getenv("HRY_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[hry plan] %8.2f ms  (finish) %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); };
	const uint32_t nc = A.ncomp;
	// numbering of the decoded mesh: exclusive scans in coding order
	plan.base_v.resize(nc + 1); plan.base_f.resize(nc + 1); plan.base_he.resize(nc + 1);
	plan.base_v[0] = plan.base_f[0] = plan.base_he[0] = 0;
	for (uint32_t k = 0; k < nc; ++k) {
		plan.base_v[k + 1] = plan.base_v[k] + A.fresh[k];
		plan.base_f[k + 1] = plan.base_f[k] + A.n_faces[k];
		plan.base_he[k + 1] = plan.base_he[k] + A.n_halfedges[k];
	}
	// ---- general bindings (regions, shared records, corner lists; structs/attr.h:101-189).  A record of a list belongs to the
	// component that names it first in coding order -- the decoder numbers the records of a list in the order they are first
	// coded (attrcode.h:443-531), so that component "introduces" it like a vertex -- and components that name a common record (two
	// parts of a scene using the same "vn" line) are tied into one group: the later one refers to it by its distance in the
	// creation order (GlobalHistory, attrcode.h:23-53), which only exists inside one stream.  OBJ-sized inputs: plain loops.
	if (m.general) {
		const Bindings &b = m.bind;
		const size_t nl = m.lists.size();
		if (b.face_reg.size() != m.nf || b.vtx_reg.size() != m.nv || b.face_attr.size() != (size_t)m.nf * b.nb_face ||
		    b.vtx_attr.size() != (size_t)m.nv * b.nb_vtx || b.corner_attr.size() != (size_t)m.ne() * b.nb_corner)
			throw Error(HRY_E_ARG, "binding tables do not match the element counts");
		plan.record_owner.resize(nl);
		for (size_t l = 0; l < nl; ++l) plan.record_owner[l].assign(m.lists[l].count, NONE32);
		std::vector<uint32_t> parent(A.group);   // union-find over coding ranks; a root is the smallest rank of its set (as in the analysis)
		auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
		auto unite = [&](uint32_t x, uint32_t y) { x = find(x); y = find(y); if (x == y) return; if (x > y) std::swap(x, y); parent[y] = x; };
		auto each_reference = [&](auto &&fn) {   // fn(list, record, coding rank of the naming component)
			for (uint32_t f = 0; f < m.nf; ++f) {
				const uint32_t k = A.rank_of[A.comp[f]];
				const int r = b.face_reg[f];
				if (r >= b.nregs_face()) throw Error(HRY_E_ARG, "face region out of range");
				for (int a = 0; a < b.nfacelists(r); ++a) fn(b.facelist(r, a), b.face_attr[(size_t)f * b.nb_face + a], k);
				for (uint32_t h = m.face_off[f]; h < m.face_off[f + 1]; ++h)
					for (int a = 0; a < b.ncornerlists(r); ++a) fn(b.cornerlist(r, a), b.corner_attr[(size_t)h * b.nb_corner + a], k);
			}
			for (uint32_t v = 0; v < m.nv; ++v) {
				const uint32_t k = A.vertex_owner[v];
				if (k == NONE32) continue;
				const int r = b.vtx_reg[v];
				if (r >= b.nregs_vtx()) throw Error(HRY_E_ARG, "vertex region out of range");
				for (int a = 0; a < b.nvtxlists(r); ++a) fn(b.vtxlist(r, a), b.vtx_attr[(size_t)v * b.nb_vtx + a], k);
			}
		};
		each_reference([&](int l, uint32_t rec, uint32_t k) {
			if ((size_t)l >= nl || rec >= m.lists[l].count) throw Error(HRY_E_ARG, "an element names a record its list does not hold");
			uint32_t &o = plan.record_owner[l][rec];
			o = std::min(o, k);
		});
		each_reference([&](int l, uint32_t rec, uint32_t k) { const uint32_t o = plan.record_owner[l][rec]; if (o != k) unite(o, k); });
		for (uint32_t k = 0; k < nc; ++k) A.group[k] = find(k);
		plan.fresh_rec.assign(nl, std::vector<uint32_t>(nc, 0)); plan.base_rec.assign(nl, std::vector<uint32_t>(nc + 1, 0));
		for (size_t l = 0; l < nl; ++l) {
			for (uint32_t o : plan.record_owner[l]) if (o != NONE32) ++plan.fresh_rec[l][o];
			for (uint32_t k = 0; k < nc; ++k) plan.base_rec[l][k + 1] = plan.base_rec[l][k] + plan.fresh_rec[l][k];
		}
	}
	// groups in the order of their first component, with their triangle counts
	std::vector<uint32_t> groups;          // representative ranks, ascending
	std::vector<uint64_t> gtri(nc, 0);     // indexed by representative
	uint64_t total = 0;
	for (uint32_t k = 0; k < nc; ++k) {
		const uint64_t tri = (uint64_t)A.n_halfedges[k] - 2ull * A.n_faces[k];
		if (A.group[k] == k) groups.push_back(k);
		gtri[A.group[k]] += tri;
		total += tri;
	}
	std::vector<uint32_t> shard_of_group(nc, 0);
	plan.shard_triangles.assign(n_shards, 0);
	if (groups.size() >= 8ull * n_shards) {
		// many groups: cut their sequence where the running triangle count passes k / n of the total (midpoint rule), which
		// keeps long runs of consecutive components together
		uint64_t run = 0;
		for (uint32_t g : groups) {
			const uint64_t mid = run + gtri[g] / 2;
			uint32_t s = total ? (uint32_t)std::min<uint64_t>(n_shards - 1, (unsigned __int128)mid * n_shards / total) : 0;
			shard_of_group[g] = s;
			run += gtri[g];
		}
	} else {
		// few groups: largest first onto the least loaded shard (ties: lower index) -- the same greedy rule as
		// harry_amd.sharding.assign_components
		std::vector<uint32_t> by_size(groups);
		std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) { return gtri[a] > gtri[b]; });
		std::vector<uint64_t> load(n_shards, 0);
		for (uint32_t g : by_size) {
			uint32_t s = 0;
			for (uint32_t j = 1; j < n_shards; ++j) if (load[j] < load[s]) s = j;
			shard_of_group[g] = s;
			load[s] += gtri[g];
		}
	}
	plan.shard_of.resize(nc);
	for (uint32_t k = 0; k < nc; ++k) {
		plan.shard_of[k] = shard_of_group[A.group[k]];
		plan.shard_triangles[plan.shard_of[k]] += (uint64_t)A.n_halfedges[k] - 2ull * A.n_faces[k];
	}
	mark("groups onto shards");
	if (!m.uniform_degree(plan.udeg)) plan.udeg = 0;
	if (plan.light) return;   // shards coded where they lie (shard_components): nobody asks where an element goes
	// ---- where every face / half-edge / vertex goes: compact numbering per shard, ascending input index.  One pass over the mesh
	// for ALL shards (thread ranges count per shard, a prefix over the ranges places them).
	if (!m.uniform_degree(plan.udeg)) plan.udeg = 0;
	const uint32_t nf = m.nf, nv = m.nv;
	const uint32_t *foff = m.face_off.data();
	const Ranges R(m.ne());
	const unsigned nt = R.nt;
	const size_t S = n_shards;
	// vertices that no face references are never coded (the reference's walk does not reach them) but its bounds scan reads every
	// record (structs/quant.h:30-44): shard 0 carries them, so that the shards' bounds combine to the whole mesh's
	auto shard_of_face = [&](uint32_t f) { return plan.shard_of[A.rank_of[A.comp[f]]]; };
	auto shard_of_vertex = [&](uint32_t v) { const uint32_t o = A.vertex_owner[v]; return o != NONE32 ? plan.shard_of[o] : 0u; };
	std::vector<uint32_t> cf((size_t)(nt + 1) * S, 0), ch((size_t)(nt + 1) * S, 0), cv((size_t)(nt + 1) * S, 0);
	parallel_for(nt, [&](unsigned t) {
		uint32_t b, e;
		// counted in the thread's own memory: the threads' rows of the tables are 4 S bytes apart -- with two shards eight threads
		// shared a cache line, and this pass took 115 - 157 ms for 25 M triangles (the rest of the plan: 55)
		std::vector<uint32_t> f_(S, 0), h_(S, 0), v_(S, 0);
		R.of(nf, t, b, e);
		for (uint32_t f = b; f < e; ++f) { const uint32_t s = shard_of_face(f); ++f_[s]; h_[s] += foff[f + 1] - foff[f]; }
		R.of(nv, t, b, e);
		for (uint32_t v = b; v < e; ++v) ++v_[shard_of_vertex(v)];
		std::copy(f_.begin(), f_.end(), cf.begin() + (size_t)(t + 1) * S);
		std::copy(h_.begin(), h_.end(), ch.begin() + (size_t)(t + 1) * S);
		std::copy(v_.begin(), v_.end(), cv.begin() + (size_t)(t + 1) * S);
	});
	for (unsigned t = 0; t < nt; ++t)
		for (size_t s = 0; s < S; ++s) { cf[(t + 1) * S + s] += cf[t * S + s]; ch[(t + 1) * S + s] += ch[t * S + s]; cv[(t + 1) * S + s] += cv[t * S + s]; }
	plan.shard_faces.resize(S); plan.shard_vertices.resize(S); plan.shard_ne.resize(S);
	for (size_t s = 0; s < S; ++s) { plan.shard_faces[s].resize(cf[nt * S + s]); plan.shard_vertices[s].resize(cv[nt * S + s]); plan.shard_ne[s] = ch[nt * S + s]; }
	plan.local_face.resize(nf); plan.local_he.resize(nf); plan.local_vertex.resize(nv);
	mark("counted");
	parallel_for(nt, [&](unsigned t) {
		uint32_t b, e;
		std::vector<uint32_t> lf(cf.begin() + (size_t)t * S, cf.begin() + (size_t)(t + 1) * S), lh(ch.begin() + (size_t)t * S, ch.begin() + (size_t)(t + 1) * S),
		                      lv(cv.begin() + (size_t)t * S, cv.begin() + (size_t)(t + 1) * S);
		R.of(nf, t, b, e);
		for (uint32_t f = b; f < e; ++f) {
			const uint32_t s = shard_of_face(f);
			plan.local_face[f] = lf[s]; plan.local_he[f] = lh[s];
			plan.shard_faces[s][lf[s]++] = f;
			lh[s] += foff[f + 1] - foff[f];
		}
		R.of(nv, t, b, e);
		for (uint32_t v = b; v < e; ++v) {
			const uint32_t s = shard_of_vertex(v);
			plan.local_vertex[v] = lv[s];
			plan.shard_vertices[s][lv[s]++] = v;
		}
	});
	mark("shard index");
	if (m.general) {   // the records of every list, shard by shard (a record no element names goes with shard 0, like an unreferenced vertex)
		const size_t nl = m.lists.size();
		plan.local_record.resize(nl); plan.shard_records.assign(nl, std::vector<BigVec<uint32_t>>(S));
		for (size_t l = 0; l < nl; ++l) {
			const uint32_t cnt = m.lists[l].count;
			plan.local_record[l].resize(cnt);
			for (uint32_t r = 0; r < cnt; ++r) {
				const uint32_t o = plan.record_owner[l][r], s = o != NONE32 ? plan.shard_of[o] : 0u;
				plan.local_record[l][r] = (uint32_t)plan.shard_records[l][s].size();
				plan.shard_records[l][s].push_back(r);
			}
		}
	}
}

// The components of one shard for a walk of the whole mesh where it lies (cut_border_walk_in_place): seeds as faces of the whole
// mesh, groups as ranks inside the list; and what the shard's segment says about itself (runs in the whole mesh's numbering).
void shard_components(const ShardPlan &plan, uint32_t shard, ComponentAnalysis &part, ShardInfo &info)
{
	if (shard >= plan.n_shards) throw Error(HRY_E_ARG, "shard index out of range");
	const ComponentAnalysis &A = plan.A;
	const uint32_t nc = A.ncomp;
	part = ComponentAnalysis();
	info = ShardInfo();
	info.g_nv = plan.g_nv; info.g_nf = plan.g_nf; info.g_ne = plan.g_ne;
	std::vector<uint32_t> local_rank(nc, NONE32);
	bool open = false;
	for (uint32_t k = 0; k < nc; ++k) {
		if (plan.shard_of[k] != shard) { open = false; continue; }
		local_rank[k] = part.ncomp++;
		part.seed.push_back(A.seed[k]); part.n_faces.push_back(A.n_faces[k]); part.n_halfedges.push_back(A.n_halfedges[k]); part.fresh.push_back(A.fresh[k]);
		part.group.push_back(local_rank[A.group[k]]);   // (a group lies in one shard: its root is here too, and comes first)
		if (!open) { info.runs.push_back(ShardRun{ plan.base_v[k], plan.base_f[k], plan.base_he[k], 0, 0, 0 }); open = true; }
		ShardRun &r = info.runs.back();
		r.n_vertices += A.fresh[k]; r.n_faces += A.n_faces[k]; r.n_halfedges += A.n_halfedges[k];
	}
	part.by_rank.resize(part.ncomp); part.rank_of.resize(part.ncomp);
	for (uint32_t k = 0; k < part.ncomp; ++k) part.by_rank[k] = part.rank_of[k] = k;
}

// The index intervals of the whole mesh a shard's faces and vertices lie in, merged where they are less than `gap` elements apart
// (a little of the neighbours travels along: harmless, and a few long copies beat thousands of short ones).
void shard_intervals(const ShardPlan &plan, uint32_t shard, uint32_t gap, std::vector<std::pair<uint32_t, uint32_t>> &faces, std::vector<std::pair<uint32_t, uint32_t>> &vertices)
{
	const ComponentAnalysis &A = plan.A;
	auto collect = [&](const std::vector<uint32_t> &lo, const std::vector<uint32_t> &hi, std::vector<std::pair<uint32_t, uint32_t>> &out) {
		out.clear();
		for (uint32_t k = 0; k < A.ncomp; ++k) if (plan.shard_of[k] == shard && lo[k] < hi[k]) out.push_back({ lo[k], hi[k] });
		std::sort(out.begin(), out.end());
		size_t n = 0;
		for (const auto &iv : out) {
			if (n && iv.first <= out[n - 1].second + gap) out[n - 1].second = std::max(out[n - 1].second, iv.second);
			else out[n++] = iv;
		}
		out.resize(n);
	};
	if (A.face_lo.size() != A.ncomp || A.vtx_lo.size() != A.ncomp) throw Error(HRY_E_INTERNAL, "shard: plan without its intervals");
	collect(A.face_lo, A.face_hi, faces);
	collect(A.vtx_lo, A.vtx_hi, vertices);
}

Mesh *shard_extract(const Mesh &m, const ShardPlan &plan, uint32_t shard)
{
	ensure_twins(m);
	if (plan.light) throw Error(HRY_E_ARG, "the plan was made for coding the shards in place: it has no element index");
	if (shard >= plan.n_shards) throw Error(HRY_E_ARG, "shard index out of range");
	if (plan.g_nv != m.nv || plan.g_nf != m.nf || plan.g_ne != m.ne() || plan.A.comp.size() != m.nf || plan.A.vertex_owner.size() != m.nv || plan.local_face.size() != m.nf)
		throw Error(HRY_E_ARG, "the plan belongs to another mesh");
	const ComponentAnalysis &A = plan.A;
	const uint32_t nf = m.nf, nv = m.nv, nc = A.ncomp;
	const uint32_t *foff = m.face_off.data();
	const BigVec<uint32_t> &faces = plan.shard_faces[shard], &verts = plan.shard_vertices[shard];
	const uint32_t lnf = (uint32_t)faces.size(), lnv = (uint32_t)verts.size(), lne = plan.shard_ne[shard];
	const Ranges R(lne);
	const unsigned nt = R.nt;
	std::unique_ptr<Mesh> out(new Mesh());
	Mesh &s = *out;
	s.nv = lnv; s.nf = lnf;
	s.have_degree = m.have_degree;   // the tables of the whole mesh: numtri model and its presence follow the container header
	s.face_off.resize((size_t)lnf + 1); s.face_off[0] = 0;
	s.org.resize(lne); s.twin.resize(lne);
	s.shard.g_nv = plan.g_nv; s.shard.g_nf = plan.g_nf; s.shard.g_ne = plan.g_ne;
	s.shard.vertex_of.assign(verts.begin(), verts.end()); s.shard.face_of.assign(faces.begin(), faces.end());
	// face of a half-edge: uniform degree by division, else the table the analysis built
	const int udeg = plan.udeg;
	if (!udeg && A.eface.size() != m.ne()) throw Error(HRY_E_INTERNAL, "shard: plan without its half-edge table");
	const uint32_t *eface = udeg ? nullptr : A.eface.data();
	auto face_of = [&](uint32_t h) -> uint32_t { return udeg ? h / (uint32_t)udeg : eface[h]; };
	std::atomic<bool> bad{ false };
	parallel_for(nt, [&](unsigned t) {
		uint32_t b, e;
		R.of(lnf, t, b, e);
		for (uint32_t lf = b; lf < e; ++lf) {
			const uint32_t f = faces[lf];
			const uint32_t lo = plan.local_he[f], deg = foff[f + 1] - foff[f];
			s.face_off[(size_t)lf + 1] = lo + deg;
			for (uint32_t k = 0; k < deg; ++k) {
				const uint32_t h = foff[f] + k, lh = lo + k;
				const uint32_t v = m.org[h];
				if (v >= nv || (A.vertex_owner[v] == NONE32 ? 0u : plan.shard_of[A.vertex_owner[v]]) != shard) { bad.store(true, std::memory_order_relaxed); s.org[lh] = 0; }
				else s.org[lh] = plan.local_vertex[v];
				const uint32_t o = m.twin[h];
				if (o == h) { s.twin[lh] = lh; continue; }
				const uint32_t fo = face_of(o);
				if (fo >= nf || plan.shard_of[A.rank_of[A.comp[fo]]] != shard) { bad.store(true, std::memory_order_relaxed); s.twin[lh] = lh; continue; }
				s.twin[lh] = plan.local_he[fo] + (o - foff[fo]);
			}
		}
	});
	if (bad.load()) throw Error(HRY_E_INTERNAL, "shard: a selected face reaches outside its group");
	if (m.general) {
		// regions and list formats of the whole mesh; every list with the records of this shard; every element's slots renumbered
		const Bindings &b = m.bind;
		const size_t nl = m.lists.size();
		if (plan.local_record.size() != nl) throw Error(HRY_E_ARG, "the plan belongs to another mesh");
		s.general = true;
		s.lists.assign(nl, AttrList());
		s.shard.g_list_count.resize(nl); s.shard.record_of.resize(nl);
		for (size_t l = 0; l < nl; ++l) {
			const AttrList &L = m.lists[l];
			AttrList &D = s.lists[l];
			D.target = L.target; D.type = L.type; D.quant = L.quant; D.offset = L.offset;
			D.interp_off = L.interp_off; D.interp_len = L.interp_len; D.interp_name = L.interp_name;
			D.bmin = L.bmin; D.bmax = L.bmax; D.have_bounds = L.have_bounds;
			const BigVec<uint32_t> &src = plan.shard_records[l][shard];
			D.count = (uint32_t)src.size();
			const size_t st = (size_t)L.stride();
			D.data.resize(src.size() * st);
			for (size_t i = 0; i < src.size() && st; ++i) memcpy(D.data.data() + i * st, L.data.data() + (size_t)src[i] * st, st);
			s.shard.g_list_count[l] = L.count;
			s.shard.record_of[l].assign(src.begin(), src.end());
		}
		Bindings &d = s.bind;
		d.reg_facelist = b.reg_facelist; d.reg_vtxlist = b.reg_vtxlist; d.reg_cornerlist = b.reg_cornerlist;
		d.off_facelist = b.off_facelist; d.off_vtxlist = b.off_vtxlist; d.off_cornerlist = b.off_cornerlist;
		d.nb_face = b.nb_face; d.nb_vtx = b.nb_vtx; d.nb_corner = b.nb_corner;
		d.face_reg.resize(lnf); d.vtx_reg.resize(lnv);
		d.face_attr.assign((size_t)lnf * b.nb_face, 0); d.vtx_attr.assign((size_t)lnv * b.nb_vtx, 0); d.corner_attr.assign((size_t)lne * b.nb_corner, 0);
		for (uint32_t lf = 0; lf < lnf; ++lf) {
			const uint32_t f = faces[lf];
			const int r = b.face_reg[f];
			d.face_reg[lf] = (uint16_t)r;
			for (int a = 0; a < b.nfacelists(r); ++a) d.face_attr[(size_t)lf * b.nb_face + a] = plan.local_record[b.facelist(r, a)][b.face_attr[(size_t)f * b.nb_face + a]];
			for (uint32_t k = 0; k < foff[f + 1] - foff[f]; ++k)
				for (int a = 0; a < b.ncornerlists(r); ++a)
					d.corner_attr[(size_t)(plan.local_he[f] + k) * b.nb_corner + a] = plan.local_record[b.cornerlist(r, a)][b.corner_attr[(size_t)(foff[f] + k) * b.nb_corner + a]];
		}
		for (uint32_t lv = 0; lv < lnv; ++lv) {
			const uint32_t v = verts[lv];
			const int r = b.vtx_reg[v];
			d.vtx_reg[lv] = (uint16_t)r;
			for (int a = 0; a < b.nvtxlists(r); ++a) d.vtx_attr[(size_t)lv * b.nb_vtx + a] = plan.local_record[b.vtxlist(r, a)][b.vtx_attr[(size_t)v * b.nb_vtx + a]];
		}
	}
	// attribute records and list formats
	for (int l = 0; l < 2 && !m.general; ++l) {
		const AttrList &L = m.lists[l];
		AttrList &D = s.lists[l];
		D.target = L.target; D.type = L.type; D.quant = L.quant; D.offset = L.offset;
		D.interp_off = L.interp_off; D.interp_len = L.interp_len; D.interp_name = L.interp_name;
		D.bmin = L.bmin; D.bmax = L.bmax; D.have_bounds = L.have_bounds;
		const BigVec<uint32_t> &src = l == 0 ? faces : verts;
		D.count = (uint32_t)src.size();
		const size_t st = (size_t)L.stride();
		D.data.resize(src.size() * st);
		if (st && L.count != (l == 0 ? nf : nv)) throw Error(HRY_E_UNSUPPORTED, "attribute lists must have one record per element");
		if (st) parallel_for(nt, [&](unsigned t) {
			uint32_t b, e;
			R.of((uint32_t)src.size(), t, b, e);
			if (st == 12) for (uint32_t i = b; i < e; ++i) memcpy(D.data.data() + (size_t)i * 12, L.data.data() + (size_t)src[i] * 12, 12);
			else for (uint32_t i = b; i < e; ++i) memcpy(D.data.data() + (size_t)i * st, L.data.data() + (size_t)src[i] * st, st);
		});
	}
	// components of the shard in coding order: seeds and runs, and what the walk would otherwise find out again (labels, sizes, ties)
	std::vector<uint32_t> local_rank(nc, NONE32);
	{
		uint32_t n = 0;
		for (uint32_t k = 0; k < nc; ++k) if (plan.shard_of[k] == shard) local_rank[k] = n++;
		s.shard.comp_faces.reserve(n); s.shard.comp_halfedges.reserve(n); s.shard.comp_fresh.reserve(n); s.shard.comp_group.reserve(n);
		for (uint32_t k = 0; k < nc; ++k) {
			if (plan.shard_of[k] != shard) continue;
			s.shard.comp_faces.push_back(A.n_faces[k]); s.shard.comp_halfedges.push_back(A.n_halfedges[k]); s.shard.comp_fresh.push_back(A.fresh[k]);
			s.shard.comp_group.push_back(local_rank[A.group[k]]);   // (a group lies in one shard: its root is here too, and comes first)
		}
	}
	bool open = false;
	for (uint32_t k = 0; k < nc; ++k) {
		if (plan.shard_of[k] != shard) { open = false; continue; }
		s.shard.seeds.push_back(plan.local_face[A.seed[k]]);
		if (!open) { s.shard.runs.push_back(ShardRun{ plan.base_v[k], plan.base_f[k], plan.base_he[k], 0, 0, 0 }); open = true; }
		ShardRun &r = s.shard.runs.back();
		r.n_vertices += A.fresh[k]; r.n_faces += A.n_faces[k]; r.n_halfedges += A.n_halfedges[k];
		if (m.general) {
			const size_t nl = m.lists.size();
			if (s.shard.run_records.size() < s.shard.runs.size() * 2 * nl) {
				for (size_t l = 0; l < nl; ++l) { s.shard.run_records.push_back(plan.base_rec[l][k]); s.shard.run_records.push_back(0); }
			}
			uint32_t *rr = s.shard.run_records.data() + (s.shard.runs.size() - 1) * 2 * nl;
			for (size_t l = 0; l < nl; ++l) rr[2 * l + 1] += plan.fresh_rec[l][k];
		}
	}
	return out.release();
}

// ---- merge ---------------------------------------------------------------------------------------------------------
namespace {
struct PartView { size_t hdr; uint32_t nseg; const uint8_t *lens; const uint8_t *segs; size_t seg_bytes; };
PartView parse_part(const uint8_t *p, size_t n)
{
	Mesh tmp;
	int minor = 0;
	PartView v{};
	v.hdr = read_hry_header(p, n, tmp, minor, false);
	if (minor != 3) throw Error(HRY_E_ARG, "merge: not a sharded (.hry v0.3) container");
	ShardedDirectory dir;   // every check a reader makes: a damaged part must not end up inside a merged container
	std::vector<uint32_t> counts;
	for (const AttrList &L : tmp.lists) counts.push_back(L.count);
	parse_sharded_directory(p, n, v.hdr, tmp.nv, tmp.nf, tmp.declared_ne, dir, true, tmp.general ? &counts : nullptr);
	v.nseg = (uint32_t)dir.segments.size();
	v.lens = p + v.hdr + 4;
	v.segs = v.lens + 8ull * v.nseg;
	uint64_t tot = 0;
	for (const auto &sg : dir.segments) tot += sg.bytes;
	v.seg_bytes = (size_t)tot;
	return v;
}
}   // namespace

void merge_containers(const uint8_t *const *parts, const size_t *sizes, size_t n, ByteSink &out)
{
	if (n == 0) throw Error(HRY_E_ARG, "merge: nothing to merge");
	std::vector<PartView> pv(n);
	uint64_t nseg = 0, bytes = 0;
	for (size_t i = 0; i < n; ++i) {
		if (!parts[i]) throw Error(HRY_E_ARG, "null argument");
		pv[i] = parse_part(parts[i], sizes[i]);
		if (pv[i].hdr != pv[0].hdr || memcmp(parts[i], parts[0], pv[0].hdr) != 0) throw Error(HRY_E_ARG, "merge: the parts describe different meshes (headers differ)");
		nseg += pv[i].nseg; bytes += pv[i].seg_bytes;
	}
	if (nseg > 0xffffffffull) throw Error(HRY_E_UNSUPPORTED, "too many segments");
	const size_t hdr = pv[0].hdr;
	out.clear();
	out.resize(hdr + 4 + 8 * (size_t)nseg + (size_t)bytes);   // (not zero-filled: every byte is written below)
	uint8_t *o = out.data();
	memcpy(o, parts[0], hdr);
	const uint32_t ns32 = (uint32_t)nseg;
	memcpy(o + hdr, &ns32, 4);
	std::vector<size_t> len_at(n), seg_at(n);
	size_t la = hdr + 4, sa = hdr + 4 + 8 * (size_t)nseg;
	for (size_t i = 0; i < n; ++i) { len_at[i] = la; seg_at[i] = sa; la += 8ull * pv[i].nseg; sa += pv[i].seg_bytes; }
	// the segments are most of a container: copied by a few threads, each its parts (fresh pages of the output included)
	const unsigned nt = bytes >= (16u << 20) ? (unsigned)std::min<size_t>(n, host_threads()) : 1u;
	parallel_for(nt, [&](unsigned t) {
		for (size_t i = t; i < n; i += nt) {
			if (pv[i].nseg) memcpy(o + len_at[i], pv[i].lens, 8ull * pv[i].nseg);
			if (pv[i].seg_bytes) memcpy(o + seg_at[i], pv[i].segs, pv[i].seg_bytes);
		}
	});
}

// ---- directory of a sharded container -------------------------------------------------------------------------------
// Everything a reader needs before it touches a segment body, checked against the header's sizes: segment extents inside the
// buffer, run tables inside their segments, runs inside the mesh, no two runs (of any segments) overlapping, and -- unless
// allow_gaps -- every face and every half-edge covered (vertices need not be: the reference never codes a vertex no face
// references).  Host-only, so the sanitizer build of tests/native reaches it.
void parse_sharded_directory(const uint8_t *p, size_t n, size_t hdr, uint32_t gnv, uint32_t gnf, uint32_t gne, ShardedDirectory &dir, bool allow_gaps,
                             const std::vector<uint32_t> *list_counts)
{
	dir = ShardedDirectory();
	if (n < hdr + 4) throw Error(HRY_E_FORMAT, "truncated sharded container");
	uint32_t nseg;
	memcpy(&nseg, p + hdr, 4);
	if ((uint64_t)nseg * 8 > n - hdr - 4) throw Error(HRY_E_FORMAT, "truncated sharded container");
	size_t off = hdr + 4 + 8ull * nseg;
	dir.segments.resize(nseg);
	for (uint32_t si = 0; si < nseg; ++si) {
		ShardedDirectory::Segment &sg = dir.segments[si];
		uint64_t len;
		memcpy(&len, p + hdr + 4 + 8ull * si, 8);
		if (len > n - off) throw Error(HRY_E_FORMAT, "truncated sharded container");
		sg.offset = off; sg.bytes = (size_t)len;
		off += (size_t)len;
		if (sg.bytes < 4) throw Error(HRY_E_FORMAT, "truncated segment");
		uint32_t nr;
		memcpy(&nr, p + sg.offset, 4);
		const size_t nl2 = list_counts ? 2 * list_counts->size() : 0;
		const size_t run_bytes = sizeof(ShardRun) + 4 * nl2;
		if ((uint64_t)nr * run_bytes > sg.bytes - 4) throw Error(HRY_E_FORMAT, "truncated segment (runs)");
		sg.runs.resize(nr);
		sg.run_records.resize((size_t)nr * nl2);
		sg.nrec.assign(nl2 / 2, 0);
		for (uint32_t j = 0; j < nr; ++j) {
			const uint8_t *rp = p + sg.offset + 4 + (size_t)j * run_bytes;
			memcpy(&sg.runs[j], rp, sizeof(ShardRun));
			if (nl2) memcpy(sg.run_records.data() + (size_t)j * nl2, rp + sizeof(ShardRun), 4 * nl2);
			for (size_t l = 0; l < nl2 / 2; ++l) {
				const uint64_t first = sg.run_records[(size_t)j * nl2 + 2 * l], cnt = sg.run_records[(size_t)j * nl2 + 2 * l + 1];
				if (first + cnt > (*list_counts)[l] || (uint64_t)sg.nrec[l] + cnt > (*list_counts)[l]) throw Error(HRY_E_FORMAT, "corrupt sharded container (records outside their list)");
				sg.nrec[l] += (uint32_t)cnt;
			}
		}
		sg.body_at = 4 + run_bytes * (size_t)nr;
		uint64_t lnv = 0, lnf = 0, lne = 0;
		for (const ShardRun &r : sg.runs) {
			if ((uint64_t)r.first_vertex + r.n_vertices > gnv || (uint64_t)r.first_face + r.n_faces > gnf || (uint64_t)r.first_halfedge + r.n_halfedges > gne)
				throw Error(HRY_E_FORMAT, "corrupt sharded container (run outside the mesh)");
			if (r.n_halfedges < r.n_faces) throw Error(HRY_E_FORMAT, "corrupt sharded container (run with fewer half-edges than faces)");
			lnv += r.n_vertices; lnf += r.n_faces; lne += r.n_halfedges;
		}
		if (lnv > gnv || lnf > gnf || lne > gne) throw Error(HRY_E_FORMAT, "corrupt sharded container (runs exceed the mesh)");
		sg.nv = (uint32_t)lnv; sg.nf = (uint32_t)lnf; sg.ne = (uint32_t)lne;
	}
	if (off != n) throw Error(HRY_E_FORMAT, "sharded container: segment sizes do not add up");
	// overlap and coverage over the runs of ALL segments
	struct Iv { uint64_t b, e; };
	std::vector<Iv> fv, hv, vv;
	for (const auto &sg : dir.segments)
		for (const ShardRun &r : sg.runs) {
			if (r.n_faces) fv.push_back(Iv{ r.first_face, (uint64_t)r.first_face + r.n_faces });
			if (r.n_halfedges) hv.push_back(Iv{ r.first_halfedge, (uint64_t)r.first_halfedge + r.n_halfedges });
			if (r.n_vertices) vv.push_back(Iv{ r.first_vertex, (uint64_t)r.first_vertex + r.n_vertices });
		}
	auto disjoint = [](std::vector<Iv> &v, uint64_t &covered) {
		std::sort(v.begin(), v.end(), [](const Iv &a, const Iv &b) { return a.b < b.b; });
		covered = 0;
		for (size_t i = 0; i < v.size(); ++i) {
			if (i && v[i - 1].e > v[i].b) return false;
			covered += v[i].e - v[i].b;
		}
		return true;
	};
	uint64_t cf = 0, ch = 0, cvv = 0;
	if (!disjoint(fv, cf) || !disjoint(hv, ch) || !disjoint(vv, cvv)) throw Error(HRY_E_FORMAT, "corrupt sharded container (overlapping runs)");
	// faces and half-edges are numbered by the same order of the components (exclusive scans, cbm/decoder.h:48,75,145,162): a run
	// that comes later in the faces comes later in the half-edges -- what keeps the merged face offsets monotone
	{
		std::vector<const ShardRun*> byf;
		for (const auto &sg : dir.segments)
			for (const ShardRun &r : sg.runs) {
				if (r.n_faces) byf.push_back(&r);
				else if (r.n_halfedges || r.n_vertices) throw Error(HRY_E_FORMAT, "corrupt sharded container (run without faces)");
			}
		std::sort(byf.begin(), byf.end(), [](const ShardRun *a, const ShardRun *b) { return a->first_face < b->first_face; });
		for (size_t i = 1; i < byf.size(); ++i) {
			const uint64_t he_end = (uint64_t)byf[i - 1]->first_halfedge + byf[i - 1]->n_halfedges;
			if (he_end > byf[i]->first_halfedge)
				throw Error(HRY_E_FORMAT, "corrupt sharded container (faces and half-edges of the runs are ordered differently)");
			// runs that follow each other without a face between them leave no half-edge between them either: a half-edge no
			// face owns would end up inside the polygon in front of it when a partial mesh is filled up
			if ((uint64_t)byf[i - 1]->first_face + byf[i - 1]->n_faces == byf[i]->first_face && he_end != byf[i]->first_halfedge)
				throw Error(HRY_E_FORMAT, "corrupt sharded container (half-edges between two runs that no face owns)");
		}
		if (!byf.empty()) {   // the same at both ends of the mesh
			if (byf.front()->first_face == 0 && byf.front()->first_halfedge != 0) throw Error(HRY_E_FORMAT, "corrupt sharded container (half-edges in front of the first face)");
			if ((uint64_t)byf.back()->first_face + byf.back()->n_faces == gnf && (uint64_t)byf.back()->first_halfedge + byf.back()->n_halfedges != gne)
				throw Error(HRY_E_FORMAT, "corrupt sharded container (half-edges behind the last face)");
		}
	}
	if (list_counts)   // the record ranges of a list must not overlap either
		for (size_t l = 0; l < list_counts->size(); ++l) {
			std::vector<Iv> rv;
			for (const auto &sg : dir.segments)
				for (size_t j = 0; j < sg.runs.size(); ++j) {
					const uint64_t first = sg.run_records[j * 2 * list_counts->size() + 2 * l], cnt = sg.run_records[j * 2 * list_counts->size() + 2 * l + 1];
					if (cnt) rv.push_back(Iv{ first, first + cnt });
				}
			uint64_t c = 0;
			if (!disjoint(rv, c)) throw Error(HRY_E_FORMAT, "corrupt sharded container (overlapping record ranges)");
		}
	dir.complete = cf == gnf && ch == gne;
	if (!dir.complete && !allow_gaps) throw Error(HRY_E_FORMAT, "sharded container does not cover the mesh (missing segments)");
}

// ---- bounds of the whole mesh from the bounds of its shards ---------------------------------------------------------
// ONE sequential scan over the whole mesh (structs/quant.h:30-44: strict comparisons, the first element wins a tie, the
// maximum starts at numeric_limits<T>::min()) restated as a combination of per-shard scans: every shard reports, per
// component, its extreme value and 1 + the index IN THE WHOLE MESH of the first element that holds it (0: the scan's initial
// value); the smaller key wins between equal values (+-0.0 compare equal and differ in their bits).
namespace {
template <typename F> void with_comp_type(CompType t, F &&f)
{
	switch (t) {
	case C_FLOAT: f(float()); break; case C_DOUBLE: f(double()); break; case C_ULONG: f(uint64_t()); break; case C_LONG: f(int64_t()); break;
	case C_UINT: f(uint32_t()); break; case C_INT: f(int32_t()); break; case C_USHORT: f(uint16_t()); break; case C_SHORT: f(int16_t()); break;
	case C_UCHAR: f(uint8_t()); break; case C_CHAR: f(int8_t()); break; default: break;
	}
}
}   // namespace

void combine_shard_bounds(const std::vector<const Mesh*> &shards, int l, std::vector<uint8_t> &bmin, std::vector<uint8_t> &bmax)
{
	if (shards.empty()) throw Error(HRY_E_ARG, "no shards");
	const AttrList &F = shards[0]->lists[l];
	bmin.assign(F.stride(), 0); bmax.assign(F.stride(), 0);
	for (int c = 0; c < F.ncomp(); ++c)
		with_comp_type(F.type[c], [&](auto tag) {
			typedef decltype(tag) T;
			bool have = false;
			T best_mn = T(), best_mx = T();
			uint64_t key_mn = 0, key_mx = 0;
			for (const Mesh *s : shards) {
				const AttrList &L = s->lists[l];
				if (!L.have_bounds || L.bmin_at.size() != (size_t)F.ncomp() || L.type != F.type) throw Error(HRY_E_ARG, "shard without device-computed bounds");
				static const std::vector<uint32_t> none;
				const std::vector<uint32_t> &whole = s->general ? ((size_t)l < s->shard.record_of.size() ? s->shard.record_of[l] : none) : l == 0 ? s->shard.face_of : s->shard.vertex_of;
				auto key = [&](uint32_t at) -> uint64_t { return at == 0 ? 0 : (at - 1 < whole.size() ? (uint64_t)whole[at - 1] + 1 : at); };
				T mn, mx;
				memcpy(&mn, L.bmin.data() + L.offset[c], sizeof(T)); memcpy(&mx, L.bmax.data() + L.offset[c], sizeof(T));
				const uint64_t kmn = key(L.bmin_at[c]), kmx = key(L.bmax_at[c]);
				if (!have || mn < best_mn || (!(best_mn < mn) && kmn < key_mn)) { best_mn = mn; key_mn = kmn; }
				if (!have || mx > best_mx || (!(best_mx > mx) && kmx < key_mx)) { best_mx = mx; key_mx = kmx; }
				have = true;
			}
			memcpy(bmin.data() + F.offset[c], &best_mn, sizeof(T)); memcpy(bmax.data() + F.offset[c], &best_mx, sizeof(T));
		});
}

}   // namespace hry
