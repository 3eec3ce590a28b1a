// The component analysis of a mesh whose connectivity is resident in HBM (twins.hip: k_cc_*): what host/cbm_walk.cpp's
// analyse_components finds with 1.9 CPU-seconds of passes over the half-edges and the corners of the 100 M-triangle configs[3]
// mesh -- the connected components, their coding order (the reference's start-face sequence, writer.cc:40-46), how many faces,
// half-edges and new vertices each brings and which of them share a vertex (cbm/encoder.h:79-113,187) -- so that the walk can run
// on the host threads from its first component on, every component writing where it belongs.  Same tables as the host's, entry
// for entry (tests/test_gpu_chunked.py compares them); only the per-face labels stay on the device.
#include <algorithm>
#include <chrono>
#include <numeric>

#include "context.hpp"
#include "kernels.hpp"

namespace hry {

// A: every table of the analysis per coding rank (ComponentAnalysis: ncomp, by_rank, rank_of, seed, n_faces, n_halfedges, fresh, group,
// face_lo / face_hi, vtx_lo / vtx_hi); A.comp and A.vertex_owner stay empty.  The mesh must be resident on cx with its twins.
// A mesh of one component returns after the labelling with A.ncomp = 1 and no table.
void device_component_analysis(Context &cx, const Mesh &m, ComponentAnalysis &A)
{
	HIP_OK(hipSetDevice(cx.device));
	const bool trace = getenv("HRY_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) {
		if (trace) fprintf(stderr, "[hry walk] %8.2f ms  (device) %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
	};
	if (!m.shard.seeds.empty()) throw Error(HRY_E_INTERNAL, "device analysis: a shard brings its coding order along");
	if (m.device_token == 0 || m.device_token != cx.resident_token) throw Error(HRY_E_INTERNAL, "device analysis: the mesh is not resident");
	const dev::ConnView cv = cx.conn_view();
	const uint32_t nf = m.nf, nv = m.nv;
	if (cv.nf != nf || cv.ne != m.ne()) throw Error(HRY_E_INTERNAL, "device analysis: resident connectivity of another mesh");
	hipStream_t st = cx.stream;
	cx.d_cscratch.ensure(dev::components_workspace_bytes(nv, nf));
	const dev::ComponentsWorkspace ws = dev::components_workspace(cx.d_cscratch.p, nv, nf);
	uint32_t *d_label = ws.label, *d_num = ws.num;
	dev::launch_components_label(st, cv, ws);
	uint32_t ncomp = 0;
	HIP_OK(hipMemcpyAsync(&ncomp, d_num + nf, 4, hipMemcpyDeviceToHost, st));
	std::vector<uint32_t> spans;
	start_face_spans(nf, spans);   // (host, a handful of spans: beside the kernels)
	HIP_OK(hipStreamSynchronize(st));
	mark("components labelled");
	A = ComponentAnalysis();
	A.ncomp = ncomp;
	if (ncomp < 2) return;   // (one component: the caller walks it in the sequential loop and needs no table)
	// per-component tables behind the workspace's labels: 5 x u32 + 1 x u64 by component number, 5 x u32 by rank, the spans
	const size_t words = (size_t)12 * ncomp + spans.size() + 8;
	cx.d_small.ensure(words * 4 + 64);
	uint32_t *d_nfaces = cx.d_small.as<uint32_t>(), *d_nhe = d_nfaces + ncomp, *d_flo = d_nhe + ncomp, *d_fhi = d_flo + ncomp;
	uint64_t *d_key = (uint64_t*)(d_fhi + ncomp + ((4 * (size_t)ncomp) & 1));   // 8-byte aligned
	uint32_t *d_rank = (uint32_t*)(d_key + ncomp), *d_tie = d_rank + ncomp, *d_fresh = d_tie + ncomp, *d_vlo = d_fresh + ncomp, *d_vhi = d_vlo + ncomp;
	uint32_t *d_spans = d_vhi + ncomp;
	HIP_OK(hipMemsetAsync(d_nfaces, 0, (size_t)2 * ncomp * 4, st));       // faces, half-edges
	HIP_OK(hipMemsetAsync(d_flo, 0xff, (size_t)ncomp * 4, st));
	HIP_OK(hipMemsetAsync(d_fhi, 0, (size_t)ncomp * 4, st));
	HIP_OK(hipMemsetAsync(d_key, 0xff, (size_t)ncomp * 8, st));
	HIP_OK(hipMemcpyAsync(d_spans, spans.data(), spans.size() * 4, hipMemcpyHostToDevice, st));
	dev::launch_components_faces(st, cv, d_label, d_num, d_spans, (uint32_t)(spans.size() / 4), d_nfaces, d_nhe, d_flo, d_fhi, d_key);
	std::vector<uint64_t> key(ncomp);
	std::vector<uint32_t> by_num((size_t)4 * ncomp);   // faces, half-edges, lowest face, highest face + 1, by component number
	HIP_OK(hipMemcpyAsync(key.data(), d_key, (size_t)ncomp * 8, hipMemcpyDeviceToHost, st));
	HIP_OK(hipMemcpyAsync(by_num.data(), d_nfaces, (size_t)4 * ncomp * 4, hipMemcpyDeviceToHost, st));
	// (the vertex words go to their neutral value meanwhile)
	HIP_OK(hipMemsetAsync(ws.vfirst, 0xff, (size_t)nv * 4, st));
	HIP_OK(hipMemsetAsync(d_fresh, 0, (size_t)ncomp * 4, st));
	HIP_OK(hipMemsetAsync(d_vlo, 0xff, (size_t)ncomp * 4, st));
	HIP_OK(hipMemsetAsync(d_vhi, 0, (size_t)ncomp * 4, st));
	HIP_OK(hipStreamSynchronize(st));
	mark("faces counted, keys on the host");
	// coding order: components by the smallest key of the start-face sequence among their faces
	std::vector<uint32_t> &by_rank = A.by_rank, &rank_of = A.rank_of;
	by_rank.resize(ncomp); rank_of.resize(ncomp);
	std::iota(by_rank.begin(), by_rank.end(), 0u);
	std::sort(by_rank.begin(), by_rank.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });
	for (uint32_t k = 0; k < ncomp; ++k) rank_of[by_rank[k]] = k;
	mark("  keys sorted (host)");
	HIP_OK(hipMemcpyAsync(d_rank, rank_of.data(), (size_t)ncomp * 4, hipMemcpyHostToDevice, st));
	dev::launch_components_vertices(st, cv, nv, ncomp, ws, d_rank, d_tie, d_fresh, d_vlo, d_vhi);
	std::vector<uint32_t> by_rank_tab((size_t)4 * ncomp);   // group, new vertices, lowest vertex, highest + 1, by rank
	HIP_OK(hipMemcpyAsync(by_rank_tab.data(), d_tie, (size_t)4 * ncomp * 4, hipMemcpyDeviceToHost, st));
	A.seed.resize(ncomp); A.n_faces.resize(ncomp); A.n_halfedges.resize(ncomp); A.face_lo.resize(ncomp); A.face_hi.resize(ncomp);
	for (uint32_t k = 0; k < ncomp; ++k) {
		const uint32_t c = by_rank[k];
		A.seed[k] = (uint32_t)key[c];
		A.n_faces[k] = by_num[c]; A.n_halfedges[k] = by_num[(size_t)ncomp + c];
		A.face_lo[k] = by_num[(size_t)2 * ncomp + c]; A.face_hi[k] = by_num[(size_t)3 * ncomp + c];
	}
	HIP_OK(hipStreamSynchronize(st));
	A.group.assign(by_rank_tab.begin(), by_rank_tab.begin() + ncomp);
	A.fresh.assign(by_rank_tab.begin() + ncomp, by_rank_tab.begin() + (size_t)2 * ncomp);
	A.vtx_lo.assign(by_rank_tab.begin() + (size_t)2 * ncomp, by_rank_tab.begin() + (size_t)3 * ncomp);
	A.vtx_hi.assign(by_rank_tab.begin() + (size_t)3 * ncomp, by_rank_tab.end());
	mark("vertex bases and groups");
}

}   // namespace hry
