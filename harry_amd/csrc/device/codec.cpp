// Device pipeline of the compat profile (reference-identical .hry v0.1 stream) + bounds / requantisation.
//
//   host walk (cbm_walk.cpp)  ->  H2D: order, repaired twins, connectivity symbol planes
//   k_rank, k_predict_vtx, k_face_planes              prediction + residuals + byte planes
//   k_split_bytes, k_op_records, k_type_records,
//   k_model_hist / _scan / _lht                       exact adaptive models by counting  -> (magic(t), x, l) per symbol
//   k_rchain                                          serial range recurrence           -> r_k, bit position S_k
//   k_low_accumulate, k_carry_*                       low register as one big number    -> payload bytes
//
// Reference: formats/hry/writer.cc:200-214 (compress), attrcode.h:395-416 (encode), arith/coder.h:58-112.
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <system_error>
#include <vector>

#include "codec_math.hpp"
#include "context.hpp"
#include "kernels.hpp"

namespace hry {

using namespace dev;
typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

// ---------------------------------------------------------------------------------------------------------
Context::Context(int dev) : device(dev)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(HRY_E_NODEVICE, "no HIP device available: the .hry path has no CPU fallback");
	if (dev < 0 || dev >= n) throw Error(HRY_E_ARG, "invalid device index");
	HIP_OK(hipSetDevice(dev));
	// the codec's main stream outranks the helper streams: its launches (connectivity streams of a decode, the reconstruction
	// chain) must not queue behind the tens of thousands of attribute-stream workgroups on stream3
	int prio_lo = 0, prio_hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
	HIP_OK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_hi));
	for (auto &e : ev) HIP_OK(hipEventCreate(&e));
}
Context::~Context()
{
	(void)hipSetDevice(device);
	if (stream) (void)hipStreamSynchronize(stream);
	for (auto &e : ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : slice_ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : chain_ev) if (e) (void)hipEventDestroy(e);
	if (pipe_stream) { (void)hipStreamSynchronize(pipe_stream); (void)hipStreamDestroy(pipe_stream); }
	if (pipe_ev) (void)hipEventDestroy(pipe_ev);
	for (auto &e : pipe_slot_ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : stage_ev) if (e) (void)hipEventDestroy(e);
	if (stream) (void)hipStreamDestroy(stream);
	if (stream2) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
	if (stream3) { (void)hipStreamSynchronize(stream3); (void)hipStreamDestroy(stream3); }
	for (auto &u : up_stream) if (u) { (void)hipStreamSynchronize(u); (void)hipStreamDestroy(u); }
	for (auto &e : up_ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : ev_x) if (e) (void)hipEventDestroy(e);
	if (ev_payload) (void)hipEventDestroy(ev_payload);
	for (auto &e : attr_ev) if (e) (void)hipEventDestroy(e);
	for (int g = 1; g < kAttrGroups; ++g) if (attr_stream[g]) { (void)hipStreamSynchronize(attr_stream[g]); (void)hipStreamDestroy(attr_stream[g]); }   // [0] is stream3
	if (h_stage) (void)hipHostFree(h_stage);
	if (h_down) (void)hipHostFree(h_down);
}
void Context::stage_put(const char *name, const void *dptr, size_t bytes)
{
	if (!keep_stages) return;
	std::vector<uint8_t> h(bytes);
	if (bytes) {
		HIP_OK(hipStreamSynchronize(stream));
		HIP_OK(hipMemcpy(h.data(), dptr, bytes, hipMemcpyDeviceToHost));
	}
	stages[name] = std::move(h);
}
void Context::stage_put_host(const char *name, const void *hptr, size_t bytes)
{
	if (!keep_stages) return;
	stages[name] = std::vector<uint8_t>((const uint8_t*)hptr, (const uint8_t*)hptr + bytes);
}
float Context::elapsed(int a, int b)
{
	float ms = 0;
	(void)hipEventElapsedTime(&ms, ev[a], ev[b]);
	return ms;
}
void Context::ensure_magic(uint32_t n)
{
	if (n <= magic_n) return;
	uint32_t want = n + n / 4 + 1024;
	DevBuf nb;
	nb.ensure((size_t)want * sizeof(MagicEnt));
	// the table is a pure function of t: recompute rather than copy
	launch_magic_table(stream, nb.as<MagicEnt>(), 0, want);
	HIP_OK(hipStreamSynchronize(stream));
	std::swap(d_magic.p, nb.p);
	std::swap(d_magic.cap, nb.cap);
	magic_n = want;
}
void Context::upload_mesh(Mesh &m, bool with_records)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	HIP_OK(hipSetDevice(device));
	if (m.lists.size() > (size_t)kMaxLists) throw Error(HRY_E_UNSUPPORTED, "more than 16 attribute lists");
	for (size_t l = 0; l < m.lists.size() && with_records; ++l) {
		d_rec[l].ensure(std::max<size_t>(m.lists[l].data.size(), 16));
		if (!m.lists[l].data.empty()) HIP_OK(hipMemcpyAsync(d_rec[l].p, m.lists[l].data.data(), m.lists[l].data.size(), hipMemcpyHostToDevice, stream));
	}
	uint32_t ne = m.ne();
	d_org.ensure(std::max<size_t>((size_t)ne * 4, 16));
	d_twin.ensure(std::max<size_t>((size_t)ne * 4, 16));
	d_foff.ensure(((size_t)m.nf + 1) * 4);
	HIP_OK(hipMemcpyAsync(d_org.p, m.org.data(), (size_t)ne * 4, hipMemcpyHostToDevice, stream));
	if (!m.twins_pending) HIP_OK(hipMemcpyAsync(d_twin.p, m.twin.data(), (size_t)ne * 4, hipMemcpyHostToDevice, stream));
	HIP_OK(hipMemcpyAsync(d_foff.p, m.face_off.data(), ((size_t)m.nf + 1) * 4, hipMemcpyHostToDevice, stream));
	int ud = 0;
	res_has_eface = !m.uniform_degree(ud);
	res_udeg = (uint32_t)ud;
	if (res_has_eface) {   // mixed polygon degrees: face of every half-edge, derived on the device from the offsets
		d_eface.ensure(std::max<size_t>((size_t)ne * 4, 16));
		dev::launch_edge_faces(stream, d_foff.as<uint32_t>(), m.nf, d_eface.as<uint32_t>());
	}
	res_nv = m.nv; res_nf = m.nf; res_ne = ne;
	if (m.twins_pending) {   // a freshly read mesh: half-edge twin matching on the device (twins.hip), and down for the host's walk
		if (ne && m.nv) {
			for (uint32_t v : m.org) if (v >= m.nv) throw Error(HRY_E_ARG, "vertex index out of range");
			d_cscratch.ensure(dev::twin_workspace_bytes(m.nv, ne));
			const uint32_t *d_over = nullptr;
			dev::launch_twins(stream, conn_view(), m.nv, d_twin.as<uint32_t>(), d_cscratch.p, &d_over);
			m.twin.resize(ne);
			std::vector<uint32_t> over(1 + dev::twin_overflow_capacity(), 0);
			HIP_OK(hipMemcpyAsync(m.twin.data(), d_twin.p, (size_t)ne * 4, hipMemcpyDeviceToHost, stream));
			HIP_OK(hipMemcpyAsync(over.data(), d_over, over.size() * 4, hipMemcpyDeviceToHost, stream));
			HIP_OK(hipStreamSynchronize(stream));
			if (over[0] > dev::twin_overflow_capacity()) {   // a mesh of hubs: the host's matcher does all of it
				m.twins_pending = true;
				ensure_twins(m);
				HIP_OK(hipMemcpyAsync(d_twin.p, m.twin.data(), (size_t)ne * 4, hipMemcpyHostToDevice, stream));
			} else if (over[0]) {
				match_twins_at(m, over.data() + 1, over[0]);   // the few hubs, with the reference's rule, on the host
				HIP_OK(hipMemcpyAsync(d_twin.p, m.twin.data(), (size_t)ne * 4, hipMemcpyHostToDevice, stream));
			}
		} else m.twin.resize(ne);
		HIP_OK(hipStreamSynchronize(stream));
		m.twins_pending = false;
	}
	HIP_OK(hipStreamSynchronize(stream));
	if (!with_records) { resident_token = 0; return; }   // connectivity only (twin matching for a caller that shards the mesh): nothing stays resident
	m.device_token = next_token++;
	resident_token = m.device_token;
}
void Context::ensure_second_stream()
{
	HIP_OK(hipSetDevice(device));
	if (!stream2) HIP_OK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
}
// what upload_mesh does besides the three copies
void Context::adopt_conn(Mesh &m)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	HIP_OK(hipSetDevice(device));
	const uint32_t ne = m.ne();
	int ud = 0;
	res_has_eface = !m.uniform_degree(ud);
	res_udeg = (uint32_t)ud;
	if (res_has_eface) {
		d_eface.ensure(std::max<size_t>((size_t)ne * 4, 16));
		dev::launch_edge_faces(stream, d_foff.as<uint32_t>(), m.nf, d_eface.as<uint32_t>());
	}
	res_nv = m.nv; res_nf = m.nf; res_ne = ne;
	m.twins_pending = false;
	HIP_OK(hipStreamSynchronize(stream));
	resident_token = 0;
}
ConnView Context::conn_view() const
{
	ConnView cv;
	cv.org = d_org.as<uint32_t>(); cv.twin = d_twin.as<uint32_t>(); cv.foff = d_foff.as<uint32_t>();
	cv.eface = res_has_eface ? d_eface.as<uint32_t>() : nullptr;
	cv.udeg = res_udeg ? res_udeg : 3; cv.nf = res_nf; cv.ne = res_ne;
	return cv;
}

ListDesc make_list_desc(const AttrList &L)
{
	ListDesc d{};
	d.ncomp = L.ncomp(); d.stride = L.stride();
	int p = 0;
	for (int c = 0; c < L.ncomp(); ++c) {
		d.stype[c] = L.stype(c); d.otype[c] = L.type[c]; d.quant[c] = L.quant[c];
		d.off[c] = (uint16_t)L.offset[c]; d.plane[c] = (uint16_t)p;
		p += kTypeSize[L.stype(c)];
	}
	d.nplanes = p;
	return d;
}

void check_codable(const Mesh &m)
{
	if (m.partial) throw Error(HRY_E_ARG, "partially decoded mesh (a share of a sharded container): only its runs are real");
	for (size_t l = 0; l < m.lists.size(); ++l)
		for (int c = 0; c < m.lists[l].ncomp(); ++c) {
			CompType st = m.lists[l].stype(c);
			if (st == C_DOUBLE) throw Error(HRY_E_UNSUPPORTED, "lossless double components: the reference's residual code reads out of bounds for 8-byte floats (prediction.h:33-44); quantise them with -q");
			if (m.general && kTypeSize[st] == 8) throw Error(HRY_E_UNSUPPORTED, "8-byte storage types (more than 32 quantisation bits, lossless 64-bit integers) are outside the supported subset");
		}
	if (!m.general && (m.lists[0].count != m.nf || m.lists[1].count != m.nv)) throw Error(HRY_E_UNSUPPORTED, "attribute lists must have one record per element");
	int dmax = (int)m.have_degree.size() - 1;
	if (dmax - 2 >= 128) throw Error(HRY_E_UNSUPPORTED, "polygons with more than 129 edges: the reference seeds its numtri model out of bounds (model.h:49-55)");
}

// ---------------------------------------------------------------------------------------------------------
// bounds (a1) and requantisation (a2 on the host, a3 on the device)
// ---------------------------------------------------------------------------------------------------------
// A large result into PAGEABLE host memory (the container the caller receives): the runtime's own path stages it through pinned
// memory and copies it out on the calling thread, 20 - 22 GB/s -- the rate of one core's memcpy, not of the link (292 MB of the
// 100 M-triangle mesh's container: 13 - 14 ms).  Here the DMA writes a ring of pinned slots and helper threads copy finished slots
// out while the next ones are in flight; this thread only issues copies and waits for their events (the helpers make no runtime
// call).  Small results, or HRY_NO_STAGED_FETCH: the runtime's path.
void fetch_to_host(Context &cx, void *dst, const void *d_src, size_t bytes)
{
	static const bool off = getenv("HRY_NO_STAGED_FETCH") != nullptr;
	static const size_t min_bytes = [] { const char *e = getenv("HRY_STAGED_FETCH_MIN"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)48 << 20; }();
	constexpr int kSlots = 8;
	static const size_t slot_bytes = [] { const char *e = getenv("HRY_STAGED_FETCH_SLOT"); const size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)4 << 20; return v < 4096 ? (size_t)4096 : v; }();
	static const unsigned max_helpers = [] { const char *e = getenv("HRY_STAGED_FETCH_THREADS"); const int v = e ? atoi(e) : 3; return (unsigned)(v < 1 ? 1 : v > 16 ? 16 : v); }();
	const unsigned n_helpers = std::min(max_helpers, host_threads() > 1 ? host_threads() - 1 : 0u);
	if (off || bytes < min_bytes || n_helpers == 0) {
		if (bytes) HIP_OK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, cx.stream));
		HIP_OK(hipStreamSynchronize(cx.stream));
		return;
	}
	cx.h_fetch.ensure(slot_bytes * kSlots);
	for (auto &e : cx.stage_ev) if (!e) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	const size_t n_chunks = (bytes + slot_bytes - 1) / slot_bytes;
	std::mutex mu;
	std::condition_variable cv;
	size_t landed = 0, taken = 0, copied[kSlots];   // chunks whose DMA has finished / that a helper has taken; per slot: chunks copied out of it
	for (auto &c : copied) c = 0;
	bool stop = false;
	std::vector<std::thread> helpers;
	const void *node = callers_node_cpus();
	helpers.reserve(n_helpers);   // (no growth -- and so no destruction of a joinable thread -- between two creations)
	auto finish = [&] { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_all(); for (auto &h : helpers) h.join(); };
	try {
	for (unsigned t = 0; t < n_helpers; ++t) helpers.emplace_back([&, node] {
		stay_on_node(node);
		for (;;) {
			size_t c;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return stop || taken < landed; });
				if (taken >= landed) return;
				c = taken++;
			}
			const size_t off_b = c * slot_bytes, n = std::min(slot_bytes, bytes - off_b);
			memcpy((uint8_t*)dst + off_b, cx.h_fetch.as<uint8_t>() + (c % kSlots) * slot_bytes, n);
			{ std::lock_guard<std::mutex> g(mu); ++copied[c % kSlots]; }
			cv.notify_all();
		}
	});
	} catch (const std::system_error &) {   // no thread to be had: the helpers that did start go home, the runtime's own path brings the bytes down
		finish();
		HIP_OK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, cx.stream));
		HIP_OK(hipStreamSynchronize(cx.stream));
		return;
	}
	try {
		size_t issued = 0, waited = 0;
		while (waited < n_chunks) {
			// keep the ring full: chunk c may go into its slot once chunk c - kSlots has been copied out of it
			while (issued < n_chunks && issued < waited + kSlots) {
				const size_t c = issued, slot = c % kSlots;
				if (c >= (size_t)kSlots) {
					std::unique_lock<std::mutex> lk(mu);
					if (copied[slot] < c / kSlots) break;   // its slot is still being read: wait for a landing first, then look again
				}
				const size_t off_b = c * slot_bytes, n = std::min(slot_bytes, bytes - off_b);
				HIP_OK(hipMemcpyAsync(cx.h_fetch.as<uint8_t>() + slot * slot_bytes, (const uint8_t*)d_src + off_b, n, hipMemcpyDeviceToHost, cx.stream));
				HIP_OK(hipEventRecord(cx.stage_ev[slot], cx.stream));
				++issued;
			}
			if (waited < issued) {
				HIP_OK(hipEventSynchronize(cx.stage_ev[waited % kSlots]));
				++waited;
				{ std::lock_guard<std::mutex> g(mu); landed = waited; }
				cv.notify_all();
			} else {   // nothing in flight and the next slot is busy: until a helper has emptied it
				const size_t slot = issued % kSlots;
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return copied[slot] >= issued / kSlots; });
			}
		}
		{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { size_t done = 0; for (auto c : copied) done += c; return done == n_chunks; }); }
	} catch (...) { finish(); throw; }
	finish();
}

void device_bounds(Context &cx, Mesh &m, const Mesh *records)
{
	HIP_OK(hipSetDevice(cx.device));
	if (records && records->device_token != 0 && records->device_token == cx.resident_token) {
		if (records->lists.size() != m.lists.size()) throw Error(HRY_E_INTERNAL, "bounds: list count");   // (the records are resident: nothing to bring up)
	} else if (records) {   // the scan reads nothing but the records: the caller has no use for the connectivity on this device
		if (m.lists.size() > (size_t)kMaxLists || records->lists.size() != m.lists.size()) throw Error(HRY_E_UNSUPPORTED, "more than 16 attribute lists");
		for (size_t l = 0; l < m.lists.size(); ++l) {
			const BigVec<uint8_t> &src = records->lists[l].data;
			cx.d_rec[l].ensure(std::max<size_t>(src.size(), 16));
			if (!src.empty()) HIP_OK(hipMemcpyAsync(cx.d_rec[l].p, src.data(), src.size(), hipMemcpyHostToDevice, cx.stream));
		}
		cx.resident_token = 0;
	} else if (m.device_token == 0 || m.device_token != cx.resident_token) cx.upload_mesh(m);
	const int nparts = 256;   // one block per compute unit
	const size_t out_bytes = 24 * (size_t)dev::kMaxComp;   // per list: (min, max, where) of every component
	cx.d_small.ensure((size_t)nparts * dev::kMaxComp * (8 + 8 + 8) + out_bytes * kMaxLists + 64);
	uint8_t *pmin = cx.d_small.as<uint8_t>(), *pmax = pmin + (size_t)nparts * dev::kMaxComp * 8;
	uint32_t *pidx = (uint32_t*)(pmax + (size_t)nparts * dev::kMaxComp * 8);
	uint8_t *outs = (uint8_t*)(pidx + 2 * (size_t)nparts * dev::kMaxComp);
	// every list's scan one behind the other on the stream, the results into pinned memory, ONE wait (a wait and a pageable copy per
	// list were 0.8 ms of a 5 ms encode of an OBJ scene with three float lists)
	if (m.lists.size() > (size_t)kMaxLists) throw Error(HRY_E_UNSUPPORTED, "more than 16 attribute lists");
	cx.h_small.ensure(std::max<size_t>(out_bytes * kMaxLists, 4096));
	uint8_t *res_all = cx.h_small.as<uint8_t>();
	bool scanned[kMaxLists] = {}, any = false;
	for (size_t l = 0; l < m.lists.size(); ++l) {
		AttrList &L = m.lists[l];
		bool quantised = false;
		for (int c = 0; c < L.ncomp(); ++c) quantised |= L.quant[c] != 0;
		if (quantised && L.have_bounds) continue;   // they came with the quantisation (file header or an earlier hry_requant)
		if (quantised) throw Error(HRY_E_UNSUPPORTED, "bounds of an already quantised list come from its header");
		L.bmin.assign(L.stride(), 0); L.bmax.assign(L.stride(), 0);
		L.bmin_at.assign(L.ncomp(), 0); L.bmax_at.assign(L.ncomp(), 0);
		scanned[l] = true;
		if (!L.ncomp()) continue;
		BoundsPlan plan{};
		plan.n = L.ncomp(); plan.stride = L.stride();
		for (int c = 0; c < L.ncomp(); ++c) {
			plan.off[c] = (uint16_t)L.offset[c]; plan.type[c] = (uint8_t)L.type[c];
		}
		launch_bounds(cx.stream, cx.d_rec[l].as<uint8_t>(), L.count, plan, pmin, pmax, pidx, nparts, outs + l * out_bytes);
		HIP_OK(hipMemcpyAsync(res_all + l * out_bytes, outs + l * out_bytes, (size_t)L.ncomp() * 24, hipMemcpyDeviceToHost, cx.stream));
		any = true;
	}
	if (any) HIP_OK(hipStreamSynchronize(cx.stream));
	for (size_t l = 0; l < m.lists.size(); ++l) {
		if (!scanned[l]) continue;
		AttrList &L = m.lists[l];
		const uint8_t *res = res_all + l * out_bytes;
		for (int c = 0; c < L.ncomp(); ++c) {
			memcpy(L.bmin.data() + L.offset[c], res + (size_t)c * 24, kTypeSize[L.type[c]]);
			memcpy(L.bmax.data() + L.offset[c], res + (size_t)c * 24 + 8, kTypeSize[L.type[c]]);
			memcpy(&L.bmin_at[c], res + (size_t)c * 24 + 16, 4);
			memcpy(&L.bmax_at[c], res + (size_t)c * 24 + 20, 4);
		}
		L.have_bounds = true;
	}
}

namespace {
template <typename T> T rd(const uint8_t *p) { T v; memcpy(&v, p, sizeof(T)); return v; }
template <typename T> void wr(uint8_t *p, T v) { memcpy(p, &v, sizeof(T)); }
template <typename F> void host_with_type(CompType t, F &&f)
{
	switch (t) {
	case C_FLOAT: f(float()); break; case C_DOUBLE: f(double()); break; case C_ULONG: f(uint64_t()); break; case C_LONG: f(int64_t()); break;
	case C_UINT: f(uint32_t()); break; case C_INT: f(int32_t()); break; case C_USHORT: f(uint16_t()); break; case C_SHORT: f(int16_t()); break;
	case C_UCHAR: f(uint8_t()); break; case C_CHAR: f(int8_t()); break; default: break;
	}
}
// per-component extent shared inside an interpretation group (structs/quant.h:46-96)
std::vector<uint8_t> shared_extent(const AttrList &L)
{
	int n = L.ncomp();
	std::vector<uint8_t> ext(L.stride(), 0), s(L.stride(), 0);
	for (int c = 0; c < n; ++c)
		host_with_type(L.type[c], [&](auto tag) {
			typedef decltype(tag) T;
			wr<T>(ext.data() + L.offset[c], (T)(rd<T>(L.bmax.data() + L.offset[c]) - rd<T>(L.bmin.data() + L.offset[c])));
			wr<T>(s.data() + L.offset[c], std::numeric_limits<T>::min());
		});
	std::vector<int> lead(n, 0);   // components outside every interpretation share group 0, as in the reference
	for (size_t i = 0; i < L.interp_off.size(); ++i)
		for (int j = 0; j < L.interp_len[i]; ++j) lead[L.interp_off[i] + j] = L.interp_off[i];
	auto get_as = [&](const std::vector<uint8_t> &r, int j, auto tag) {
		typedef decltype(tag) T;
		T out = T();
		host_with_type(L.type[j], [&](auto st) { typedef decltype(st) S; out = (T)rd<S>(r.data() + L.offset[j]); });
		return out;
	};
	for (int j = 0; j < n; ++j) {
		int k = lead[j];
		host_with_type(L.type[k], [&](auto tag) {
			typedef decltype(tag) T;
			T cur = rd<T>(s.data() + L.offset[k]), v = get_as(ext, j, T());
			wr<T>(s.data() + L.offset[k], std::max(cur, v));
		});
	}
	for (int j = 0; j < n; ++j) {
		int k = lead[j];
		host_with_type(L.type[j], [&](auto tag) { typedef decltype(tag) T; wr<T>(s.data() + L.offset[j], get_as(s, k, T())); });
	}
	return s;
}
}   // namespace

// validation and expansion of a quantisation request as the reference CLI does it (main.cc:74-91): the quantisation of every
// component afterwards
std::vector<std::vector<uint8_t>> requant_targets(const Mesh &m, const hry_quant *q, size_t nq, bool clear)
{
	const int nl = (int)m.lists.size();
	std::vector<std::vector<uint8_t>> nquant(nl);
	for (int l = 0; l < nl; ++l) nquant[l] = m.lists[l].quant;
	struct One { int l, c, q; };
	std::vector<One> reqs;
	for (size_t i = 0; i < nq; ++i) {
		if (q[i].bits < 0) throw Error(HRY_E_ARG, "Invalid quantization bits");
		if (q[i].list < 0 || q[i].list >= nl) throw Error(HRY_E_ARG, "Invalid list index");
		const AttrList &L = m.lists[q[i].list];
		if (q[i].comp == -1) {
			for (int c = 0; c < L.ncomp(); ++c) {
				if (q[i].bits > kTypeSize[L.type[c]] * 8) throw Error(HRY_E_ARG, "Invalid quantization bits");
				reqs.push_back(One{ q[i].list, c, q[i].bits });
			}
		} else {
			if (q[i].comp < 0 || q[i].comp >= L.ncomp()) throw Error(HRY_E_ARG, "Invalid attribute index");
			if (q[i].bits > kTypeSize[L.type[q[i].comp]] * 8) throw Error(HRY_E_ARG, "Invalid quantization bits");
			reqs.push_back(One{ q[i].list, q[i].comp, q[i].bits });
		}
	}
	if (clear) for (int l = 0; l < nl; ++l) std::fill(nquant[l].begin(), nquant[l].end(), 0);
	for (const One &r : reqs) nquant[r.l][r.c] = (uint8_t)r.q;
	return nquant;
}
// what k_requant does to the records of list L (bounds known) to take its components to the quantisation `to`
dev::RequantPlan requant_plan(const AttrList &L, const std::vector<uint8_t> &to)
{
	std::vector<uint8_t> scale = shared_extent(L);
	RequantPlan plan{};
	for (int c = 0; c < L.ncomp(); ++c) {
		int sq = L.quant[c], dq = to[c];
		if (sq == dq) continue;
		if (dq > 30 || sq > 30) throw Error(HRY_E_UNSUPPORTED, "more than 30 quantisation bits: the reference evaluates 1 << bits in int (quant.h:135)");
		RequantComp &rc = plan.c[plan.n++];
		rc.off = L.offset[c];
		rc.src_type = sq ? storage_type(L.type[c], sq) : L.type[c];
		rc.src_bits = sq; rc.dst_bits = dq; rc.dst_type = L.type[c]; rc.pad = 0;
		rc.mn = 0; rc.scale = 0;
		memcpy(&rc.mn, L.bmin.data() + L.offset[c], kTypeSize[L.type[c]]);
		memcpy(&rc.scale, scale.data() + L.offset[c], kTypeSize[L.type[c]]);
		if (L.type[c] != C_FLOAT && L.type[c] != C_DOUBLE && rc.scale == 0) throw Error(HRY_E_UNSUPPORTED, "constant integer component: the reference divides by a zero extent (quant.h:106)");
	}
	return plan;
}

uint64_t test_extra(const char *name) { const char *e = getenv(name); return e ? (uint64_t)strtoull(e, nullptr, 10) : 0ull; }

namespace dev { void launch_scatter_u32(hipStream_t st, const uint32_t *pairs, uint32_t n, uint32_t *dst); }
// A walk repairs a handful of twins (non-manifold edges, neighbours consumed from the other side) -- the whole array went up for
// them: 1.2 GB for the configs[3] mesh.  Now the entries the walk names go up as (half-edge, twin) pairs and are scattered.
// patches_only: `host` is a mesh other threads are walking other parts of (a shard coded in place) -- never the whole array,
// whose other entries are theirs to change and, where contexts share a device's arrays, theirs to bring up
void upload_repaired_twins(Context &cx, const Mesh &host, const WalkResult &w, bool patches_only)
{
	if (!w.twins_changed) return;
	const size_t ne = host.ne(), np = w.twin_patches.size();
	if (!patches_only && (np == 0 || np > ne / 16)) {
		HIP_OK(hipMemcpyAsync(cx.d_twin.p, host.twin.data(), ne * 4, hipMemcpyHostToDevice, cx.stream));
		return;
	}
	if (np == 0) return;
	std::vector<uint32_t> &pairs = cx.h_twin_patch;
	pairs.resize(2 * np);
	for (size_t i = 0; i < np; ++i) {
		const uint32_t h = w.twin_patches[i];
		if (h >= ne) throw Error(HRY_E_INTERNAL, "walk: repaired twin out of range");
		pairs[2 * i] = h; pairs[2 * i + 1] = host.twin[h];
	}
	cx.d_patch.ensure(pairs.size() * 4);
	HIP_OK(hipMemcpyAsync(cx.d_patch.p, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice, cx.stream));
	dev::launch_scatter_u32(cx.stream, cx.d_patch.as<uint32_t>(), (uint32_t)np, cx.d_twin.as<uint32_t>());
}

void device_requant(Context &cx, Mesh &m, const hry_quant *q, size_t nq, bool clear)
{
	HIP_OK(hipSetDevice(cx.device));
	const int nl = (int)m.lists.size();
	const std::vector<std::vector<uint8_t>> nquant = requant_targets(m, q, nq, clear);
	bool any = false;
	for (int l = 0; l < nl; ++l) any |= nquant[l] != m.lists[l].quant;
	if (!any) return;
	for (int l = 0; l < nl; ++l) if (!m.lists[l].have_bounds && m.lists[l].ncomp()) { device_bounds(cx, m); break; }
	if (m.device_token == 0 || m.device_token != cx.resident_token) cx.upload_mesh(m);

	for (int l = 0; l < nl; ++l) {
		AttrList &L = m.lists[l];
		if (nquant[l] == L.quant) continue;
		const RequantPlan plan = requant_plan(L, nquant[l]);
		launch_requant(cx.stream, cx.d_rec[l].as<uint8_t>(), L.count, L.stride(), plan);
		if (!L.data.empty()) HIP_OK(hipMemcpyAsync(L.data.data(), cx.d_rec[l].p, L.data.size(), hipMemcpyDeviceToHost, cx.stream));
		HIP_OK(hipStreamSynchronize(cx.stream));
		L.quant = nquant[l];
	}
}

// ---------------------------------------------------------------------------------------------------------
// range coder back end shared by encode_compat and hry_range_encode_lht:
// records (device) -> payload bytes (host)
// ---------------------------------------------------------------------------------------------------------
void finish_stream(Context &cx, uint32_t ns, std::vector<uint8_t> &payload)
{
	cx.d_r.ensure(std::max<size_t>((size_t)ns * 8, 16));
	cx.d_s.ensure(std::max<size_t>((size_t)ns * 4, 16));
	cx.d_state.ensure(64);
	uint64_t st[2] = { 1ull << 63, 0 };   // R = HALF, no shifts yet (coder.h:47)
	HIP_OK(hipMemcpyAsync(cx.d_state.p, st, 16, hipMemcpyHostToDevice, cx.stream));
	HIP_OK(hipEventRecord(cx.ev[3], cx.stream));
	if (!cx.device_recurrence && ns) {
		// The serial recurrence on a host core ("host or one-lane"; same arithmetic as k_rchain, 8 times
		// faster than a lone wavefront's scalar unit).  It runs BEHIND the device: the records come down slice by slice into
		// pinned memory while the core works on the slices before, and (r, S) of a finished slice go back up at once.
		// A ring of kRing slice buffers and events, reused: slot i % kRing receives slice i once (r, S) of slice i - kRing have
		// gone up from it -- the stream runs its copies in order, so "download slice i" is simply enqueued behind "upload slice
		// i - kRing" and the host waits for the download's event before it touches the slot.  Pinned memory stays at
		// kRing x 7 MB whatever the stream's length (28 M triangles used to pin 4 GB for the life of the context).
		const uint32_t SL = 1u << 18, kRing = 4;
		const uint32_t nsl = (ns + SL - 1) / SL;
		const size_t slot = std::min<size_t>(ns, SL);
		cx.h_rec.ensure(slot * kRing * sizeof(SymRec)); cx.h_r.ensure(slot * kRing * 8); cx.h_s.ensure(slot * kRing * 4);
		while (cx.slice_ev.size() < kRing) { hipEvent_t e; HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); cx.slice_ev.push_back(e); }
		SymRec *rec = cx.h_rec.as<SymRec>();
		uint64_t *rr = cx.h_r.as<uint64_t>();
		uint32_t *ss = cx.h_s.as<uint32_t>();
		auto download = [&](uint32_t i) {
			const uint32_t b0 = i * SL, n0 = std::min(SL, ns - b0);
			HIP_OK(hipMemcpyAsync(rec + (size_t)(i % kRing) * slot, cx.d_rec_sym.as<SymRec>() + b0, (size_t)n0 * sizeof(SymRec), hipMemcpyDeviceToHost, cx.stream));
			HIP_OK(hipEventRecord(cx.slice_ev[i % kRing], cx.stream));
		};
		for (uint32_t i = 0; i < nsl && i < kRing; ++i) download(i);
		uint64_t R = st[0], S = st[1];
		for (uint32_t i = 0; i < nsl; ++i) {
			const uint32_t b0 = i * SL, n0 = std::min(SL, ns - b0);
			const size_t at = (size_t)(i % kRing) * slot;
			HIP_OK(hipEventSynchronize(cx.slice_ev[i % kRing]));
			for (uint32_t k = 0; k < n0; ++k) {
				const SymRec &q = rec[at + k];
				uint64_t r = cm::div_by_magic(R, q.magic, q.meta & 63u);
				uint64_t prod = r * q.x;
				uint64_t Rn = (q.meta & kMetaSub) ? R - prod : prod;
				uint64_t y = Rn - 1;
				uint32_t sh = (y ? (uint32_t)__builtin_clzll(y) : 64u) - 1u;
				rr[at + k] = r; ss[at + k] = (uint32_t)S;
				R = Rn << sh;
				S += sh;
			}
			HIP_OK(hipMemcpyAsync(cx.d_r.as<uint64_t>() + b0, rr + at, (size_t)n0 * 8, hipMemcpyHostToDevice, cx.stream));
			HIP_OK(hipMemcpyAsync(cx.d_s.as<uint32_t>() + b0, ss + at, (size_t)n0 * 4, hipMemcpyHostToDevice, cx.stream));
			if (i + kRing < nsl) download(i + kRing);
		}
		st[0] = R; st[1] = S;
		HIP_OK(hipEventRecord(cx.ev[4], cx.stream));
	} else {
		if (ns) launch_rchain(cx.stream, cx.d_rec_sym.as<SymRec>(), ns, cx.d_r.as<uint64_t>(), cx.d_s.as<uint32_t>(), cx.d_state.as<uint64_t>());
		HIP_OK(hipEventRecord(cx.ev[4], cx.stream));
		HIP_OK(hipMemcpyAsync(st, cx.d_state.p, 16, hipMemcpyDeviceToHost, cx.stream));
		HIP_OK(hipStreamSynchronize(cx.stream));
	}
	uint64_t total_shift = st[1];
	// (HRY_TEST_EXTRA_BITS / HRY_TEST_EXTRA_SYMBOLS: counted on top of the stream's own -- the tests pin the refusals at their boundaries without a mesh of 180 M triangles)
	if (total_shift + 64 + test_extra("HRY_TEST_EXTRA_BITS") >= (1ull << 32)) throw Error(HRY_E_UNSUPPORTED, "compat stream longer than 2^32 bits: use the chunked profile");
	uint64_t nbits = total_shift + 64;   // flush: the 64 bits of the low register (coder.h:58-67)
	size_t nbytes = (size_t)((nbits + 7) / 8);
	uint32_t nw = (uint32_t)((nbits + 31) / 32) + 2;
	cx.d_acc.ensure((size_t)nw * 8);
	// (no folded copy of the accumulators: the carry kernels fold where they read, kernels.hip)
	cx.d_summary.ensure(((size_t)nw / 1024 + 2) * 4);
	cx.d_bytes.ensure((size_t)nw * 4);
	HIP_OK(hipMemsetAsync(cx.d_acc.p, 0, (size_t)nw * 8, cx.stream));
	launch_low_accumulate(cx.stream, cx.d_r.as<uint64_t>(), cx.d_s.as<uint32_t>(), cx.d_sym_l.as<uint32_t>(), ns, cx.d_acc.as<uint64_t>());
	launch_carry(cx.stream, cx.d_acc.as<uint64_t>(), nw, cx.d_v.as<uint64_t>(), cx.d_summary.as<uint32_t>(), cx.d_bytes.as<uint8_t>());
	HIP_OK(hipEventRecord(cx.ev[5], cx.stream));
	payload.resize(nbytes);
	HIP_OK(hipMemcpyAsync(payload.data(), cx.d_bytes.p, nbytes, hipMemcpyDeviceToHost, cx.stream));
	HIP_OK(hipStreamSynchronize(cx.stream));
	cx.stage_put("r", cx.d_r.p, (size_t)ns * 8);
	cx.stage_put("S", cx.d_s.p, (size_t)ns * 4);
}

void range_encode_lht(Context &cx, const uint64_t *lht, size_t n, std::vector<uint8_t> &out)
{
	HIP_OK(hipSetDevice(cx.device));
	if (n >= (1ull << 31)) throw Error(HRY_E_ARG, "too many symbols");
	std::vector<SymRec> rec(n);
	std::vector<uint32_t> ls(n);
	for (size_t i = 0; i < n; ++i) {
		uint64_t l = lht[3 * i], h = lht[3 * i + 1], t = lht[3 * i + 2];
		if (!(l < h && h <= t) || t >= (1ull << 32) || t < 1) throw Error(HRY_E_ARG, "need l < h <= t < 2^32");
		bool sub = h == t, noop = sub && l == 0;
		SymRec r{ 0, 0, 0 };
		uint32_t shift = 0;
		if (t >= 2) cm::make_magic((uint32_t)t, r.magic, shift);
		r.x = (uint32_t)(sub ? l : h - l);
		r.meta = shift | (sub ? kMetaSub : 0u) | (noop ? kMetaNoop : 0u);
		rec[i] = r;
		ls[i] = (uint32_t)l;
	}
	cx.d_rec_sym.ensure(std::max<size_t>(n * sizeof(SymRec), 16));
	cx.d_sym_l.ensure(std::max<size_t>(n * 4, 16));
	if (n) {
		HIP_OK(hipMemcpyAsync(cx.d_rec_sym.p, rec.data(), n * sizeof(SymRec), hipMemcpyHostToDevice, cx.stream));
		HIP_OK(hipMemcpyAsync(cx.d_sym_l.p, ls.data(), n * 4, hipMemcpyHostToDevice, cx.stream));
	}
	finish_stream(cx, (uint32_t)n, out);
}

// ---------------------------------------------------------------------------------------------------------
// compat encode
// ---------------------------------------------------------------------------------------------------------
void encode_compat(Context &cx, Mesh &m, std::vector<uint8_t> &out)
{
	HIP_OK(hipSetDevice(cx.device));
	auto t_all = Clock::now();
	if (m.general) { encode_general(cx, m, out); return; }
	cx.timing = hry_timing{};
	check_codable(m);
	// the reference's single stream carries one coder state and one numbering across the whole file: it does not shard
	if (m.shard.active()) throw Error(HRY_E_UNSUPPORTED, "a shard of a larger mesh codes into the sharded chunked container only (HRY_PROFILE_CHUNKED)");
	for (int l = 0; l < 2; ++l) if (!m.lists[l].have_bounds && m.lists[l].ncomp()) { device_bounds(cx, m); break; }
	for (int l = 0; l < 2; ++l) if (!m.lists[l].have_bounds) { m.lists[l].bmin.assign(m.lists[l].stride(), 0); m.lists[l].bmax.assign(m.lists[l].stride(), 0); m.lists[l].have_bounds = true; }
	if (m.device_token == 0 || m.device_token != cx.resident_token) cx.upload_mesh(m);

	// ---- host: header + cut-border walk
	out.clear();
	write_hry_header(m, 1, out);
	auto t_walk = Clock::now();
	WalkResult w;
	cut_border_walk(m, w, false);   // the operation model is evaluated on the device (k_opmodel_*), the groups' places in the ONE symbol sequence come out of the walk -- also from its threads (cbm_walk.cpp: the components' pieces are put in coding order)
	cx.timing.host_walk_ms = ms_since(t_walk);

	const uint32_t vc = (uint32_t)w.order_v.size(), fc = (uint32_t)w.order_f.size();
	const ListDesc ldv = make_list_desc(m.lists[1]), ldf = make_list_desc(m.lists[0]);
	const uint32_t sv = 1 + (uint32_t)ldv.nplanes, sf = 1 + (uint32_t)ldf.nplanes;   // symbols per vertex / face (reg_* symbols are exact no-ops)
	const uint64_t ns64 = (uint64_t)w.n_conn + (uint64_t)vc * sv + (uint64_t)fc * sf;
	if (ns64 + test_extra("HRY_TEST_EXTRA_SYMBOLS") >= (1ull << 31)) throw Error(HRY_E_UNSUPPORTED, "more than 2^31 symbols in one compat stream: use the chunked profile");
	const uint32_t ns = (uint32_t)ns64;
	const uint32_t base_v = w.n_conn, base_f = w.n_conn + vc * sv;

	// ---- H2D
	auto t_h2d = Clock::now();
	HIP_OK(hipEventRecord(cx.ev[0], cx.stream));
	cx.d_order_v.ensure(std::max<size_t>((size_t)vc * 4, 16));
	cx.d_order_f.ensure(std::max<size_t>((size_t)fc * 4, 16));
	cx.d_rank.ensure(std::max<size_t>((size_t)m.nv * 4, 16));
	if (vc) HIP_OK(hipMemcpyAsync(cx.d_order_v.p, w.order_v.data(), (size_t)vc * 4, hipMemcpyHostToDevice, cx.stream));
	if (fc) HIP_OK(hipMemcpyAsync(cx.d_order_f.p, w.order_f.data(), (size_t)fc * 4, hipMemcpyHostToDevice, cx.stream));
	// the resident copy of the twins is current unless the walk repaired some (non-manifold edges, consumed neighbours)
	upload_repaired_twins(cx, m, w);   // (encoder.h:150,193-198)
	// connectivity groups: values + positions, packed back to back
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

	// ---- model jobs: every byte plane with its initial counts (models.h:197-218, model.h:38-55)
	std::vector<uint32_t> inits;   // 256-entry tables
	auto add_init = [&](const std::vector<uint32_t> &t) { uint32_t id = (uint32_t)(inits.size() / 256); inits.insert(inits.end(), t.begin(), t.end()); return id; };
	std::vector<uint32_t> ones(256, 1), iop_init(256, 0), nt0(256, 0), nt1(256, 0);
	for (int i = 0; i < 9; ++i) iop_init[i] = 1;
	for (size_t d = 3; d < m.have_degree.size(); ++d) if (m.have_degree[d]) { ++nt0[(d - 2) & 0xff]; ++nt1[(d - 2) >> 8]; }
	const uint32_t id_ones = add_init(ones), id_iop = add_init(iop_init), id_nt0 = add_init(nt0), id_nt1 = add_init(nt1);
	auto total_of = [&](uint32_t id) { uint32_t s = 0; for (int i = 0; i < 256; ++i) s += inits[(size_t)id * 256 + i]; return s; };

	size_t conn_plane_bytes = 0;
	for (int g = 0; g < G_COUNT; ++g) conn_plane_bytes += w.grp_val[g].size() * kGroupBytes[g];
	cx.d_connplanes.ensure(std::max<size_t>(conn_plane_bytes, 16));
	cx.d_vplanes.ensure(std::max<size_t>((size_t)vc * ldv.nplanes, 16));
	cx.d_fplanes.ensure(std::max<size_t>((size_t)fc * ldf.nplanes, 16));
	cx.d_init.ensure(inits.size() * 4);
	HIP_OK(hipMemcpyAsync(cx.d_init.p, inits.data(), inits.size() * 4, hipMemcpyHostToDevice, cx.stream));

	struct JobH { PlaneJob j; uint32_t init_id; };
	std::vector<PlaneJob> jobs;
	std::vector<ChunkRef> chunks;
	uint32_t max_total = 2 + std::max(vc, fc);
	auto add_job = [&](const uint8_t *sym, uint32_t n, uint32_t init_id, const uint32_t *pos_tab, uint32_t pos_add, uint32_t pos_base, uint32_t pos_stride) {
		if (!n) return;
		PlaneJob j{};
		j.sym = sym; j.n = n; j.init = cx.d_init.as<uint32_t>() + (size_t)init_id * 256; j.t0 = total_of(init_id);
		j.pos_tab = pos_tab; j.pos_add = pos_add; j.pos_base = pos_base; j.pos_stride = pos_stride;
		j.chunk0 = (uint32_t)chunks.size();
		for (uint32_t f = 0; f < n; f += kChunk) chunks.push_back(ChunkRef{ (uint32_t)jobs.size(), f });
		jobs.push_back(j);
		max_total = std::max(max_total, j.t0 + n);
	};
	{
		size_t poff = 0;
		for (int g = 0; g < G_COUNT; ++g) {
			uint32_t n = (uint32_t)w.grp_val[g].size();
			for (int b = 0; b < kGroupBytes[g]; ++b) {
				uint32_t init_id = g == G_IOP ? id_iop : g == G_NUMTRI ? (b == 0 ? id_nt0 : id_nt1) : id_ones;
				add_job(cx.d_connplanes.as<uint8_t>() + poff + (size_t)b * n, n, init_id, cx.d_grp_pos.as<uint32_t>() + goff[g], (uint32_t)b, 0, 0);
			}
			poff += (size_t)n * kGroupBytes[g];
		}
	}
	for (int p = 0; p < ldv.nplanes; ++p) add_job(cx.d_vplanes.as<uint8_t>() + (size_t)p * vc, vc, id_ones, nullptr, 0, base_v + 1 + p, sv);
	for (int p = 0; p < ldf.nplanes; ++p) add_job(cx.d_fplanes.as<uint8_t>() + (size_t)p * fc, fc, id_ones, nullptr, 0, base_f + 1 + p, sf);
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

	// ---- device: prediction + residuals + planes
	ConnView cv = cx.conn_view();
	HIP_OK(hipEventRecord(cx.ev[1], cx.stream));
	HIP_OK(hipMemsetAsync(cx.d_rank.p, 0xff, (size_t)m.nv * 4, cx.stream));
	launch_rank(cx.stream, cx.d_order_v.as<uint32_t>(), vc, cv.org, cx.d_rank.as<uint32_t>());
	launch_predict_vtx(cx.stream, cv, cx.d_order_v.as<uint32_t>(), vc, cx.d_rank.as<uint32_t>(), cx.d_rec[1].as<uint8_t>(), ldv, cx.d_vplanes.as<uint8_t>());
	launch_face_planes(cx.stream, cv, cx.d_order_f.as<uint32_t>(), fc, cx.d_rec[0].as<uint8_t>(), ldf, cx.d_fplanes.as<uint8_t>());
	HIP_OK(hipEventRecord(cx.ev[2], cx.stream));
	// ---- device: models -> per-symbol records in global stream order
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
	launch_type_records(cx.stream, vc, base_v, sv, magic, rec, sym_l);
	launch_type_records(cx.stream, fc, base_f, sf, magic, rec, sym_l);
	launch_model(cx.stream, cx.d_jobs.as<PlaneJob>(), (uint32_t)jobs.size(), cx.d_chunks.as<ChunkRef>(), (uint32_t)chunks.size(), cx.d_hist.as<uint32_t>(), magic, rec, sym_l);

	if (cx.keep_stages) {
		cx.stage_put_host("order_v", w.order_v.data(), (size_t)vc * 4);
		cx.stage_put_host("order_f", w.order_f.data(), (size_t)fc * 4);
		cx.stage_put_host("twin", m.twin.data(), (size_t)m.ne() * 4);
		cx.stage_put("rank", cx.d_rank.p, (size_t)m.nv * 4);
		cx.stage_put("vplanes", cx.d_vplanes.p, (size_t)vc * ldv.nplanes);
		cx.stage_put("fplanes", cx.d_fplanes.p, (size_t)fc * ldf.nplanes);
		cx.stage_put("rec", cx.d_rec_sym.p, (size_t)ns * sizeof(SymRec));
		cx.stage_put("sym_l", cx.d_sym_l.p, (size_t)ns * 4);
		uint32_t lay[8] = { w.n_conn, vc, fc, sv, sf, ns, (uint32_t)w.numtri_coded, 0 };
		cx.stage_put_host("layout", lay, sizeof lay);
	}

	// ---- device: serial recurrence, big-number low register, carries; D2H
	std::vector<uint8_t> payload;
	auto t_fin = Clock::now();
	finish_stream(cx, ns, payload);
	(void)t_fin;
	out.insert(out.end(), payload.begin(), payload.end());
	cx.stage_put_host("payload", payload.data(), payload.size());

	cx.timing.k_predict_ms = cx.elapsed(1, 2);
	cx.timing.k_model_ms = cx.elapsed(2, 3);
	cx.timing.k_rchain_ms = cx.elapsed(3, 4);
	cx.timing.device_ms = cx.elapsed(1, 5);
	cx.timing.n_symbols = ns;
	cx.timing.payload_bytes = payload.size();
	cx.timing.total_ms = ms_since(t_all);
}

}   // namespace hry
