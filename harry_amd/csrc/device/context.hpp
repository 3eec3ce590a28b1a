// Device context of the codec: HIP device, one stream, grow-only workspace buffers, resident mesh.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../host/host.hpp"
#include "dev_types.hpp"

namespace hry {

inline void hip_check(hipError_t e, const char *what)
{
	if (e != hipSuccess) throw Error(HRY_E_NODEVICE, std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define HIP_OK(x) ::hry::hip_check((x), #x)

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf &operator=(const DevBuf&) = delete;
	~DevBuf() { if (p) (void)hipFree(p); }
	void ensure(size_t n)
	{
		if (n <= cap) return;
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = n + n / 8 + 256;
		HIP_OK(hipMalloc(&p, want));
		cap = want;
	}
	template <typename T> T *as() const { return (T*)p; }
};

// pinned host memory, grow-only (persistent across calls: fresh pinned or pageable blocks cost a page fault per 4 KiB)
struct PinBuf {
	void *p = nullptr;
	size_t cap = 0;
	PinBuf() = default;
	PinBuf(const PinBuf&) = delete;
	PinBuf &operator=(const PinBuf&) = delete;
	~PinBuf() { if (p) (void)hipHostFree(p); }
	void ensure(size_t n)
	{
		if (n <= cap) return;
		if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
		size_t want = n + n / 8 + 4096;
		HIP_OK(hipHostMalloc(&p, want, hipHostMallocDefault));
		cap = want;
	}
	template <typename T> T *as() const { return (T*)p; }
};

constexpr int kMaxLists = 16;   // attribute lists of one mesh kept in HBM (the OBJ reader creates at most 8)

struct Context {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr;   // uploads and connectivity-only kernels of the pipelined decode (created on first use)
	hipStream_t stream3 = nullptr;   // attribute streams' entropy decode, next to the connectivity streams' (created on first use)
	static constexpr int kUploadStreams = 3;
	hipStream_t up_stream[kUploadStreams] = {};   // further uploaders of finished spans beside stream2 (unchunk.cpp: SpanUploader; created on first use)
	hipEvent_t up_ev[kUploadStreams] = {};
	hipEvent_t ev_x[3] = {};         // cross-stream ordering events (created with stream3)
	hipEvent_t ev_payload = nullptr; // chunked decode: the attribute streams' part of a large payload is on the device (created with stream3)
	// chunked decode: the attribute streams are launched in groups by how far into their plane they end (unchunk.cpp); group g
	// runs on attr_stream[g] and raises attr_ev[g]
	static constexpr int kAttrGroups = 3;   // (+ the codec's three streams: more streams than hardware queues serialise)
	hipStream_t attr_stream[kAttrGroups] = {};
	hipEvent_t attr_ev[kAttrGroups] = {};
	void *h_down = nullptr;          // pinned landing buffer for the vertex records of the pipelined decode (device -> host per slice)
	size_t h_down_cap = 0;
	PinBuf h_mirror;                 // pipelined decode with border snapshots: pinned copies of the helper threads' stretches (face offsets, origins, twins, decode order), made by the helpers themselves
	void *h_stage = nullptr;         // pinned staging memory for uploads that run next to a busy host thread (copies from
	size_t h_stage_cap = 0;          // pageable memory make the runtime pin and unpin pages: TLB shootdowns for every thread)
	hipStream_t pipe_stream = nullptr;   // chunked encode: what finished groups of a walk on several threads have coded goes to the planes beside the walk (chunked.cpp: EncodePipeline; created on first use)
	hipEvent_t pipe_ev = nullptr;
	static constexpr int kPipeSlots = 3;
	hipEvent_t pipe_slot_ev[kPipeSlots] = {};   // a slot of h_pipe / d_pipe is free again
	hipEvent_t ev[8] = {};
	// the float chains of a large mesh run in batches (unchunk.cpp: ChainBatches): a pair of timing events around every batch's
	// launches, so that hry_timing.k_chain_ms is the sum over the batches of a decode (created on first use)
	static constexpr int kChainBatchEvents = 16;
	hipEvent_t chain_ev[2 * kChainBatchEvents] = {};
	hry_timing timing{};

	// resident mesh (hry_mesh_upload): attribute records and connectivity stay in HBM across encodes
	uint64_t resident_token = 0;
	uint64_t gen_token = 0;          // the mesh whose binding tables (d_vreg .. d_cattr) are in HBM (general.cpp: upload_general)
	uint64_t next_token = 1;
	DevBuf d_rec[kMaxLists], d_org, d_twin, d_foff, d_eface;
	DevBuf d_vreg, d_freg, d_vattr, d_cattr, d_fattr, d_gen;   // general bindings (general.cpp): region and record tables, event arena
	uint32_t res_nv = 0, res_nf = 0, res_ne = 0, res_udeg = 0;
	bool res_has_eface = false;

	// reciprocal table, valid for totals < magic_n
	DevBuf d_magic;
	uint32_t magic_n = 0;

	// workspace
	DevBuf d_order_v, d_order_f, d_rank, d_vplanes, d_fplanes, d_connplanes, d_grp_val, d_grp_pos, d_op, d_jobs, d_chunks, d_hist, d_init,
	       d_rec_sym, d_sym_l, d_r, d_s, d_state, d_acc, d_v, d_summary, d_bytes, d_small;
	// chunked profile
	DevBuf d_cjobs, d_cscratch, d_csizes, d_coffs, d_cout, d_csyms, d_patch;
	DevBuf d_split;   // chunked encode in two kernels: per stream the place of its records and the streams' order, longest first (the records: d_rec_sym)
	DevBuf d_pipe, d_nt_val, d_nt_planes;   // EncodePipeline: run tables and twin pairs of the batches; the polygons' triangle counts and their two byte planes
	std::vector<uint32_t> h_twin_patch;   // (half-edge, twin) pairs on their way to d_patch (upload_repaired_twins)
	std::vector<uint32_t> inplace_twin_patches;   // a shard coded in place: the half-edges whose twins its walks repaired in the WHOLE mesh's host array (sharded.cpp brings them to the resident copy on another device)

	bool keep_stages = false;
	bool device_recurrence = false; // HRY_FLAG_DEVICE_RECURRENCE: k_rchain instead of the host core
	PinBuf h_pipe;                  // chunked encode: the pipeline's staging slots (run tables + the runs' entries, gathered)
	PinBuf h_fetch;                 // large results on their way down: a ring of pinned slots (fetch_to_host, codec.cpp)
	hipEvent_t stage_ev[8] = {};    // ... and an event per slot
	PinBuf h_gen;                   // general bindings: the events' arena on its way up (general.cpp)
	PinBuf h_small;                 // a few words that come down asynchronously (a copy into pageable memory keeps its caller until it has happened)
	PinBuf h_conn;                  // chunked decode: the connectivity planes, down for the host's replay
	PinBuf h_rec, h_r, h_s;         // compat: symbol records down, (r, S) up, slice by slice (codec.cpp finish_stream)
	std::vector<hipEvent_t> slice_ev;
	std::map<std::string, std::vector<uint8_t>> stages;

	explicit Context(int dev);
	~Context();
	void stage_put(const char *name, const void *dptr, size_t bytes);
	void stage_put_host(const char *name, const void *hptr, size_t bytes);
	void ensure_magic(uint32_t n);
	void upload_mesh(Mesh &m, bool with_records = true);
	void adopt_conn(Mesh &m);        // the connectivity is in d_foff / d_org / d_twin already (unchunk.cpp: SpanUploader): the rest of upload_mesh
	void ensure_second_stream();
	dev::ConnView conn_view() const;
	float elapsed(int a, int b);
};

// ---- planes of the chunked container (chunked.cpp, unchunk.cpp, general.cpp)
// initial counts of a plane that carries no static prior (the reference's initial model of that context, models.h:197-218)
enum { INIT_ONES = 0, INIT_IOP = 1, INIT_NT0 = 2, INIT_NT1 = 3, INIT_OP = 4, INIT_REGV = 5, INIT_REGF = 6, INIT_TYPE2 = 7, INIT_TYPE3 = 8, INIT_KINDS = 9 };
void build_init_tables(const Mesh &m, std::vector<uint32_t> &tabs);   // INIT_KINDS x 256
struct PlaneRef { const uint8_t *dptr; uint32_t n; int init; };
// Chunks of an attribute plane grow with their position: 1 Ki symbols each up to symbol 32 Ki, 2 Ki up to 64 Ki, 4 Ki up to 128 Ki
// ... (length = position / 16 rounded down to a power of two, at least 1 Ki, at most the container's chunk size).  A stream is one
// serial wavefront, 0.26 us per symbol: the decoder's reconstruction chain walks the vertices in order at 16 ns each and finds
// every chunk decoded when it gets there, instead of waiting for the first full-size chunk (2.1 ms at 8 Ki symbols).
// Connectivity planes keep one size (they are needed whole, first).  The oracle restates the same rule.
inline uint32_t attr_chunk_len(uint64_t pos, uint32_t chunk_syms)
{
	uint32_t len = 1024;
	while (len < chunk_syms && (uint64_t)len * 2 <= pos / 16) len *= 2;
	return len < chunk_syms ? len : chunk_syms;
}
// general bindings: the attribute planes that follow the 21 connectivity planes, in container order (the oracle restates it):
// the region of every vertex / face (low byte; only with more than one region), then per list a region binds, in list order:
// the kind of every reference, the creation-order distances (4 planes), at corner lists the per-vertex distances (2 planes),
// the residual bytes of the records coded as data
enum { GP_REGV = 0, GP_REGF, GP_TYPE, GP_GHIST, GP_LHIST, GP_DATA };
struct GenPlane { int what, list, byte, init; };
std::vector<GenPlane> general_plane_layout(const Mesh &m);
// encode: collects the references on the host, computes the residuals on the device, returns the planes (device pointers into
// cx.d_gen) in layout order; order_v / order_f / repaired twins must be resident (d_order_v, d_order_f, d_twin)
void general_planes_encode(Context &cx, Mesh &m, const WalkResult &w, std::vector<PlaneRef> &planes);
// decode: the decoded planes (device, plane k at d_syms + plane_off[k], nsym[k] symbols; first = index of the first attribute
// plane) -> bindings + records of m
void general_planes_decode(Context &cx, Mesh &m, const OrderVec &order_v, const std::vector<uint32_t> &seg_start,
                           const std::vector<uint32_t> &seg_level, const uint8_t *d_syms, const std::vector<uint64_t> &plane_off,
                           const std::vector<uint32_t> &nsym, uint32_t first);

void check_general(const Mesh &m);             // general.cpp
// device -> pageable host memory, behind everything on cx.stream; returns when the bytes are there (codec.cpp)
void fetch_to_host(Context &cx, void *dst, const void *d_src, size_t bytes);
void upload_general(Context &cx, Mesh &m);     // connectivity + every list + the binding tables -> HBM

// codec entry points (codec.cpp / chunked.cpp)
// records: m holds the lists' formats and counts only, their records are those of *records and nothing else travels to the device
void device_bounds(Context &cx, Mesh &m, const Mesh *records = nullptr);
// analysis.cpp: host/cbm_walk.cpp's analyse_components on the device, for a mesh resident on cx (every table but the per-face labels)
void device_component_analysis(Context &cx, const Mesh &m, ComponentAnalysis &A);
void device_requant(Context &cx, Mesh &m, const hry_quant *q, size_t nq, bool clear);
std::vector<std::vector<uint8_t>> requant_targets(const Mesh &m, const hry_quant *q, size_t nq, bool clear);   // validated request -> quantisation of every component
dev::RequantPlan requant_plan(const AttrList &L, const std::vector<uint8_t> &to);
// the twins the walk repaired (cbm/encoder.h:150,193-198) into the resident copy: the few entries it names, else the whole array
void upload_repaired_twins(Context &cx, const Mesh &host, const WalkResult &w, bool patches_only = false);
uint64_t test_extra(const char *name);   // HRY_TEST_EXTRA_SYMBOLS / _BITS: counted on top of a reference stream's own (tests of the format's limits)
void encode_compat(Context &cx, Mesh &m, std::vector<uint8_t> &out);
// A shard coded where it lies in the whole mesh (sharded.cpp: the in-process executor): the mesh handed to encode_chunked is a
// SKELETON -- the shard's sizes, the lists' formats and bounds, its runs, no arrays; the context's connectivity and record arrays
// are those of the whole mesh in the whole mesh's numbering, filled over the shard's index intervals; the walk goes over the
// whole mesh's host arrays with the shard's components.
struct InPlaceShard {
	Mesh *whole;
	const ComponentAnalysis *part;                                   // the shard's components (shard_components)
	const uint32_t *eface;                                           // face of every half-edge of the whole mesh (mixed degrees), else nullptr
	WalkState *marks;                                                // of the whole mesh; shared by the workers
	const std::vector<std::pair<uint32_t, uint32_t>> *face_intervals;   // the shard's faces, as uploaded (repaired twins go up over the same intervals)
	std::function<void()> arrays_ready;                              // called after the walk, before anything touches the device: returns when the
	                                                                 // shard's intervals are in HBM (and quantised, if the caller quantises)
	std::function<void()> before_walk, after_walk;                   // around the walk on the host threads (the executor takes the workers' walks in turn)
};
void encode_chunked(Context &cx, Mesh &m, int chunk_syms, ByteSink &out, const InPlaceShard *in_place = nullptr);   // (bytes that are not zero-filled first and go to the caller as they are)
void encode_general(Context &cx, Mesh &m, std::vector<uint8_t> &out);   // general.cpp: regions, shared records, corner lists (reference stream only)
Mesh *decode_general(Context &cx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> m);
void finish_stream(Context &cx, uint32_t ns, std::vector<uint8_t> &payload);
Mesh *decode_any(Context &cx, const uint8_t *p, size_t n, int shard_index = 0, int shard_count = 0, bool allow_partial = false);
// sharded.cpp: one mesh over several contexts (devices) from one process
void encode_sharded(Context *const *cxs, int n_ctx, Mesh &m, const hry_quant *q, size_t nq, bool clear, int n_shards, int chunk_syms,
                    ByteSink &out, hry_shard_timing &st, bool store_bounds = true);
Mesh *decode_sharded(Context *const *cxs, int n_ctx, const uint8_t *p, size_t n, size_t hdr, std::unique_ptr<Mesh> g, int shard_index, int shard_count,
                     bool allow_partial, hry_shard_timing *st);
void range_encode_lht(Context &cx, const uint64_t *lht, size_t n, std::vector<uint8_t> &out);

dev::ListDesc make_list_desc(const AttrList &L);
void check_codable(const Mesh &m);

}   // namespace hry
