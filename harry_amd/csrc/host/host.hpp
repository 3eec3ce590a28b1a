// Host-side declarations of the product (PLY I/O, cut-border walk, .hry header).
#pragma once
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/harry_amd.h"
#include "mesh.hpp"

namespace hry {

// ---- ply_io.cpp
Mesh *mesh_from_ply(const uint8_t *buf, size_t n);
Mesh *mesh_from_arrays(uint32_t nv, const uint8_t *vrec, int v_ncomp, const uint8_t *v_types, const char *const *v_names,
                       uint32_t nf, const uint8_t *degrees, const uint32_t *indices,
                       const uint8_t *frec, int f_ncomp, const uint8_t *f_types, const char *const *f_names);
void mesh_to_ply(const Mesh &m, bool ascii, ByteSink &out, bool packed = false);
void build_twins(Mesh &m);
// the readers leave the twin matching pending (Mesh::twins_pending); a context does it on the device when the mesh is first
// uploaded (device/twins.hip), host-only entry points do it here
void ensure_twins(const Mesh &m);
// the half-edges whose smaller endpoint is one of the given vertices, matched on the host (the device leaves hubs with more
// than a few dozen such half-edges to it); every other entry of m.twin is final already
void match_twins_at(Mesh &m, const uint32_t *vertices, uint32_t n);
void print_component(std::string &o, const AttrList &L, const uint8_t *rec, int c);   // one value as the reference prints it (mixing.h:340-359)

// ---- obj_io.cpp (formats/obj/reader.rl:108-299, writer.cc:20-132): meshes with general bindings (mesh.hpp Bindings)
Mesh *mesh_from_obj(const uint8_t *buf, size_t n, const char *directory);   // directory: where "mtllib" files are looked up
void mesh_to_obj(const Mesh &m, ByteSink &out);

// ---- context numbering of a .hry stream (formats/hry/models.h:183-237), shared with the device code
enum {
	CTX_IOP = 0, CTX_OP = 1, CTX_ELEM = 2, CTX_PART = 6, CTX_VERT = 8, CTX_NUMTRI = 12, CTX_REGFACE = 14, CTX_REGVTX = 16, CTX_ATTR0 = 18,
	ATTR_TYPE = 0, ATTR_GHIST = 1, ATTR_LHIST = 5, ATTR_DATA = 7
};
// connectivity symbol groups the walk emits as byte planes
enum ConnGroup { G_IOP = 0, G_ELEM, G_PART, G_VERT, G_NUMTRI, G_COUNT };
static constexpr int kGroupBytes[G_COUNT] = { 1, 4, 2, 4, 2 };
static constexpr int kGroupCtx[G_COUNT] = { CTX_IOP, CTX_ELEM, CTX_PART, CTX_VERT, CTX_NUMTRI };

// Output of the host-side cut-border walk (the inputs the device path needs)
// State of the connectivity coding at the start of a connected component: how many symbols of every connectivity plane group
// and of every operation class precede it, and the first vertex index / face / half-edge it will create.
struct ComponentMark {
	uint32_t n_grp[G_COUNT];
	uint32_t n_op[8];
	uint32_t first_vertex, first_face, first_halfedge;
	uint32_t min_ref;   // smallest vertex index the component names explicitly (TRIxxx start, NM operation); 0xffffffff = none
};
// A restart point of the chunked container (v0.2 directory): a ComponentMark at which a decoder may start replaying
// independently.  flags bit 0: some component between this point and the next names a vertex created before this point.
struct RestartPoint {
	uint32_t n_grp[G_COUNT];
	uint32_t n_op[8];
	uint32_t first_vertex, first_face, first_halfedge;
	uint32_t flags;
};
// An explicit naming of a vertex (TRIxxx start, NM operation) with the number of triangles seen at it so far (the "order"
// the operation planes are split by, models.h:69-72) and the component (index into marks) that names it.
// snap: how many border snapshots (below) its component had taken when the vertex was named -- 0 = before the first
struct NamedVertex { uint32_t mark, id, count, snap; };
// counters a restart point carries: (vertex, order counter at the start of the span) for every older vertex its span names
typedef std::vector<std::pair<uint32_t, uint32_t>> RestartCounters;
constexpr uint32_t kRestartFaces = 8192;   // a restart point at the first component start >= this many faces after the previous one
constexpr uint32_t kRestartWords = G_COUNT + 8 + 4;
// A restart point INSIDE a connected component (round 6; chunked container, directory extension): the cut-border at the first
// moment between two operations -- the polygon in hand complete -- at which the component has coded j * snapshot_faces faces,
// j = 1, 2, ...  What a decoder needs to start replaying there (cbm/decoder.h:27-211, cbm/cutborder.h:49-333): the plane cursors and
// the next vertex / face / half-edge like any restart point; the parts of the border with their edge_begin flags; of every
// element the vertex (the decoder's numbering) and how many triangles have been seen at it (it selects the plane the next
// operation is read from, models.h:101-105; counts from 9 on are one class).  The elements' half-edges are NOT stored: they
// only ever become twins of edges the span creates, so a decoder replays with placeholders and joins the spans afterwards.
struct BorderSnapshot {
	uint32_t mark = 0;                        // the component it lies in (index into WalkResult::marks)
	uint32_t n_grp[G_COUNT] = { 0, 0, 0, 0, 0 }, n_op[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // symbols consumed: since the component's mark while the
	uint32_t first_vertex = 0, first_face = 0, first_halfedge = 0;                      // walk runs, absolute once finish_snapshots() has run
	std::vector<uint32_t> parts;              // bottom of the stack first: size << 1 | edge_begin
	std::vector<uint32_t> vtx;                // per element, part by part, head -> tail
	std::vector<uint8_t> seen;                // min(triangles seen at the vertex, 9)
	std::vector<uint32_t> orig;               // (walk on two cores only: the elements' vertices in the mesh's numbering, until the counts are filled in)
};
constexpr uint32_t kSnapshotMinFaces = 1u << 17;
// spacing of the snapshots for a container that describes nf faces: at least 2^17 faces (0.9 ms of replay on a core of the test
// boxes: eight stretches for the million triangles of BASELINE configs[1]), and at most some sixty snapshots a mesh (a power of two;
// 0 = none: HRY_NO_SNAPSHOTS).  HRY_SNAPSHOT_FACES overrides (tests: small meshes)
uint32_t snapshot_spacing(uint32_t nf);
// canonical selection, shared by every writer of the container (the oracle restates it)
// snaps (finished: absolute cursors) / snap_counters: the border snapshots of the walk and, out, per snapshot the older vertices
// its span names that are not on its border, with their counters
std::vector<RestartPoint> select_restart_points(const std::vector<ComponentMark> &marks, const std::vector<NamedVertex> &named,
                                                std::vector<RestartCounters> &counters, const std::vector<BorderSnapshot> *snaps = nullptr,
                                                std::vector<RestartCounters> *snap_counters = nullptr);

// One byte per cut-border operation: symbol | order class << 3.  A distinct type on purpose: stores through a character
// type may alias anything, so a plain uint8_t stream makes the compiler reload every pointer of the walk's hot loop after each
// operation it writes; an enumeration with the same representation does not (read it back through uint8_t freely).
enum class OpByte : uint8_t {};
inline uint8_t op_u8(OpByte b) { return (uint8_t)b; }

// Progress of a walk on several threads (walk_components_parallel): what a finished GROUP of tied components has coded is final
// from then on -- runs of the coded vertices / faces (positions in order_v / order_f; the polygons' triangle counts lie like
// order_f) and the twins it repaired.  The encoder's device side takes them while the other groups are still being walked
// (device/chunked.cpp: EncodePipeline).  Called by the walking threads, several at a time.
struct WalkProgress {
	virtual ~WalkProgress() {}
	// before the first walk: the arrays the runs refer to (they do not move until the walk returns); numtri = nullptr when the
	// mesh has one polygon degree
	virtual void begin(const uint32_t *order_v, size_t n_v, const uint32_t *order_f, size_t n_f, const uint32_t *numtri) = 0;
	// runs: (first, length) pairs; twin_pairs: (half-edge, its final twin) pairs
	virtual void group_done(const uint32_t *v_runs, uint32_t n_v_runs, const uint32_t *f_runs, uint32_t n_f_runs, const uint32_t *twin_pairs, uint32_t n_pairs) = 0;
};

struct WalkResult {
	WalkProgress *progress = nullptr;   // in: told about every finished group of a walk on several threads (nullptr: nobody listens)
	BigVec<uint32_t> order_v;   // one half-edge per coded vertex, in coding order (attrcode.h:297,310-314)
	BigVec<uint32_t> order_f;   // one half-edge per face, in coding order (attrcode.h:298,315-319)
	// connectivity symbols: values (low byte first, one entry per symbol) + position in the global symbol sequence
	BigVec<uint32_t> grp_val[G_COUNT];
	BigVec<uint32_t> grp_pos[G_COUNT];
	// cut-border operations: raw symbol | order class << 3 (one byte each), and -- for the reference stream -- the
	// order-conditioned model already evaluated (models.h:91-119) as cumulative-frequency triples
	BigVec<OpByte> op_sc;
	uint32_t n_op_class[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // operations per order class (sizes of the chunked container's operation planes)
	BigVec<uint32_t> op_l, op_h, op_t, op_pos;
	std::vector<ComponentMark> marks; // one per connected component, in coding order
	std::vector<NamedVertex> named;   // every explicit naming of a vertex, in coding order
	uint32_t n_conn = 0;             // symbols in the connectivity part of the global sequence
	bool numtri_coded = false;       // false when a single polygon degree makes every numtri symbol an exact no-op
	bool numtri_positions = true;    // in: fill grp_pos[G_NUMTRI] (one entry per face; only a single symbol sequence needs it)
	bool twins_changed = false;      // the walk repaired at least one twin (cbm/encoder.h:150,193-198): the device copy is stale
	std::vector<uint32_t> twin_patches;   // ... at these half-edges (with repetitions); their twins are final when the walk returns
	uint32_t snapshot_faces = 0;     // in: a border snapshot every so many faces of a component (0: none)
	std::vector<BorderSnapshot> snapshots;   // in stream order
};

// per-face / per-vertex marks of the walk.  Not character types (see OpByte above): a byte store in the hot loop would force every
// cached pointer and counter back through memory
enum class Gone : uint8_t { no = 0, yes = 1 };
typedef uint16_t OnCount;
// Shared state of a walk: per-vertex and per-face marks.  Connected components touch disjoint faces, and disjoint
// vertices unless they share a (non-manifold) vertex, so several components can be walked at the same time on
// these arrays as long as components that share a vertex are walked in coding order by one thread -- also the shards of one
// mesh by the workers of the in-process executor (device/sharded.cpp), which is why the type is public.
struct WalkState {
	BigVec<Gone> gone;         // face consumed
	BigVec<OnCount> on;        // how many border elements reference a vertex (cutborder.h:69)
	BigVec<uint32_t> sent;     // original vertex -> transmitted index (encoder.h:28-52)
	BigVec<uint16_t> seen;     // triangles seen per vertex (selects the op model class)
	WalkState(uint32_t nv, uint32_t nf) : gone(nf, Gone::no), on(nv, 0), sent(nv, 0xffffffffu), seen(nv, 0) {}
	WalkState(uint32_t nv, uint32_t nf, unsigned n_threads);   // the same, filled by several threads
};

// A walk of the components on several threads rests on knowing the components before it starts -- from the half-edge twins as the
// matching left them.  The walk REPAIRS twins where more than two faces meet in an edge (cbm/encoder.h:150,193-198), and a repair
// can cut a component in two (the faces behind the old pairing are reached later, as a component of their own): the walk of that
// component then consumes fewer faces than the analysis promised.  Thrown at that point, with the mesh's twins partly repaired:
// cut_border_walk catches it, matches the twins afresh (the matching is a function of the connectivity) and walks on one thread, as
// the reference does; callers that walk shards in place (cut_border_walk_in_place) see it as the error it is.
struct WalkMismatch : Error {
	WalkMismatch() : Error(HRY_E_INTERNAL, "walk on several threads: a repaired half-edge twin split a connected component (cbm/encoder.h:150,193-198); such a mesh is walked on one thread (HRY_HOST_THREADS=1)") {}
};
// ---- cbm_walk.cpp: cbm::encode restated over flat arrays (cbm/encoder.h:54-217, cutborder.h:49-333)
// eval_op_model: the order-conditioned operation model of the reference stream evaluated per operation (op_l / op_h / op_t / op_pos);
// the product evaluates it on the device (k_opmodel_*) and asks for one_sequence instead: positions of the connectivity groups in
// ONE symbol sequence, i.e. the walk on one thread
void cut_border_walk(Mesh &m, WalkResult &out, bool eval_op_model = true, bool one_sequence = false);
// the connectivity groups of a walk in stream order, as the operations see them: operation i (0-based among the operations) sits at
// position i + cum[j] of the symbol sequence, j = the last group with thr[j] <= i (none: i)
void op_position_table(const WalkResult &w, std::vector<uint32_t> &thr, std::vector<uint32_t> &cum);

// The connected components of a mesh as the walk will code them, without walking: which faces form a component, the coding
// order (start-face sequence of the reference, writer.cc:40-46, or the seed list of a shard), how many vertices / faces /
// half-edges each one introduces (their exclusive scans are the numbering of the decoded mesh, cbm/decoder.h:48,75,145,162),
// and which components share a vertex (cbm/encoder.h:79-113,187: they name each other's vertices and must stay together).
struct ComponentAnalysis {
	uint32_t ncomp = 0;
	BigVec<uint32_t> comp;                       // per face: component number (arbitrary, dense)
	std::vector<uint32_t> by_rank, rank_of;      // coding rank <-> component number
	std::vector<uint32_t> seed, n_faces, n_halfedges, fresh, group;   // per coding rank; group = smallest rank tied to it
	std::vector<uint32_t> face_lo, face_hi, vtx_lo, vtx_hi;   // per coding rank: the index intervals [lo, hi) its faces / the vertices it introduces lie in
	bool want_vertex_owner = false;
	BigVec<uint32_t> vertex_owner;               // per vertex: coding rank of the component that introduces it (0xffffffff: unused)
	BigVec<uint32_t> eface;                      // mixed polygon degrees only: face of every half-edge (kept for the shard planner)
};
void analyse_components(const Mesh &m, ComponentAnalysis &A);
void start_face_spans(uint32_t nf, std::vector<uint32_t> &spans);   // (lowest face, highest face, position of the first, ascending) per span, sorted by the lowest face
// Some components of `m`, walked where they lie (no sub-mesh): `part` lists them in coding order (seed faces of m, sizes, the
// vertices each introduces, groups as ranks inside the list); vertex indices start at 0 with the list's first component -- what a
// shard of m codes (shard.cpp: shard_components).  `st` may be shared with other calls walking OTHER groups of the same mesh at
// the same time; eface: the face of every half-edge for mixed polygon degrees (ComponentAnalysis::eface), else nullptr.
void cut_border_walk_in_place(Mesh &m, const ComponentAnalysis &part, const uint32_t *eface, WalkState &st, WalkResult &out);

// ---- general_events.cpp: which record every element of a mesh with general bindings names, along the coding order
// (attrcode.h:321-393).  Kinds and slots are enumerations of a byte's width, not character types (see OpByte)
enum class RefKind : uint8_t { data = 0, hist = 1, lhist = 2 };   // io.h:95-98 (also: a region number where regions are coded)
enum class RefSlot : uint8_t {};
// everything the stream says about one list, as symbols with (for the reference stream) the position of every symbol in it
struct ListStream {   // (BigVec: sized once for the most a list can get, not filled first)
	BigVec<RefKind> type_sym;
	BigVec<uint32_t> type_pos;
	BigVec<uint32_t> gh_val, gh_pos;   // distance in creation order (4 bytes each, io.h:99-103)
	BigVec<uint32_t> lh_val, lh_pos;   // distance in the vertex' own names (2 bytes each, io.h:104-108)
	BigVec<uint32_t> d_pos, d_idx, d_he;   // records coded as data: position of the first residual byte, the record, where
	BigVec<RefSlot> d_slot;
	uint32_t nbytes = 0;                    // residual bytes per record
	BigVec<uint32_t> first_at;              // record -> its rank among the records created so far (GlobalHistory::tidxlist)
	uint32_t created = 0;
};
struct Events {
	std::vector<ListStream> ls;
	BigVec<RefKind> rv_sym, rf_sym;    // region of every vertex / face (low byte; the high byte never carries information)
	BigVec<uint32_t> rv_pos, rf_pos;
	uint32_t end_pos = 0;
};
// pos0: the position of the first symbol (behind the connectivity); want_positions: fill the *_pos arrays (the ONE symbol sequence
// of the reference stream; the parallel container asks for none)
void collect_events(const Mesh &m, const WalkResult &w, uint32_t pos0, bool want_positions, Events &E);

// ---- shard.cpp: a mesh shards by groups of connected components
struct ShardPlan {
	uint32_t n_shards = 0, g_nv = 0, g_nf = 0, g_ne = 0;
	ComponentAnalysis A;
	std::vector<uint32_t> shard_of;               // per coding rank
	std::vector<uint32_t> base_v, base_f, base_he; // per coding rank (+ end): position in the numbering of the decoded mesh
	std::vector<uint64_t> shard_triangles;        // per shard
	std::vector<uint8_t> have_degree;
	// where every element goes, built once for all shards (one pass over the mesh instead of one per shard)
	BigVec<uint32_t> local_face, local_he;        // per face: its index in its shard, the first half-edge of it there
	BigVec<uint32_t> local_vertex;                // per vertex: its index in the shard that owns it (unreferenced vertices: shard 0)
	std::vector<BigVec<uint32_t>> shard_faces, shard_vertices;   // per shard: its faces / vertices, ascending input index
	std::vector<uint32_t> shard_ne;               // per shard: half-edges
	int udeg = 0;                                 // the one polygon degree of the mesh, 0 = mixed (then A.eface holds the face of every half-edge)
	bool light = false;                           // made without the per-element index (local_* / shard_* below are empty)
	// general bindings: every list's records belong to the component that first names them (coding order); components that name
	// a common record are tied into one group like components that share a vertex
	std::vector<BigVec<uint32_t>> record_owner, local_record;       // per list, per record: coding rank of its component (0xffffffff: unnamed), index in its shard
	std::vector<std::vector<BigVec<uint32_t>>> shard_records;       // per list, per shard: its records, ascending input index
	std::vector<std::vector<uint32_t>> base_rec, fresh_rec;         // per list, per coding rank (+ end for the bases): place in the decoder's record numbering
};
// light: for shards that are coded where they lie in the whole mesh (shard_components + cut_border_walk_in_place: the in-process
// executor) -- no per-element index, shard_extract refuses such a plan
void shard_plan(const Mesh &m, uint32_t n_shards, ShardPlan &plan, bool light = false);
void shard_plan_from_analysis(const Mesh &m, uint32_t n_shards, ComponentAnalysis &&A, ShardPlan &plan);   // light; A: the tables per coding rank + intervals
void shard_plan_finish(const Mesh &m, uint32_t n_shards, ShardPlan &plan);   // (internal: plan.A is complete)
void shard_components(const ShardPlan &plan, uint32_t shard, ComponentAnalysis &part, ShardInfo &info);
void shard_intervals(const ShardPlan &plan, uint32_t shard, uint32_t gap, std::vector<std::pair<uint32_t, uint32_t>> &faces, std::vector<std::pair<uint32_t, uint32_t>> &vertices);
Mesh *shard_extract(const Mesh &m, const ShardPlan &plan, uint32_t shard);
// several single- or multi-segment sharded containers (.hry v0.3) of the same mesh -> one
void merge_containers(const uint8_t *const *parts, const size_t *sizes, size_t n, ByteSink &out);
// the directory of a sharded container, validated against the header's sizes (throws HRY_E_FORMAT on damage)
struct ShardedDirectory {
	struct Segment {
		size_t offset = 0, bytes = 0, body_at = 0;   // body_at: v0.2 body inside the segment
		std::vector<ShardRun> runs;
		std::vector<uint32_t> run_records;           // general bindings: 2 x lists words per run (first record, records), else empty
		std::vector<uint32_t> nrec;                  // ... records of every list in the segment
		uint32_t nv = 0, nf = 0, ne = 0;
	};
	std::vector<Segment> segments;
	bool complete = false;   // every face and half-edge of the mesh lies in some run
};
// list_counts: the record counts of the header's lists when the mesh has general bindings (then every run carries its record
// ranges), nullptr for the PLY layout
void parse_sharded_directory(const uint8_t *p, size_t n, size_t hdr, uint32_t gnv, uint32_t gnf, uint32_t gne, ShardedDirectory &dir, bool allow_gaps = false,
                             const std::vector<uint32_t> *list_counts = nullptr);
// bounds of list l of the whole mesh from device-computed bounds of its shards (the scan's own tie rule, see shard.cpp)
void combine_shard_bounds(const std::vector<const Mesh*> &shards, int l, std::vector<uint8_t> &bmin, std::vector<uint8_t> &bmax);
// a worker thread of the in-process multi-GPU executor limits the helper threads of the host phases it starts (0: no limit)
void set_thread_budget(unsigned n);
// the CPUs of memory node `node` (nullptr: unknown / single node); block_pool.cpp
const void *node_cpus(int node);

// a byte plane of decoded connectivity symbols (the replay reads them where they landed: pinned memory of the context, or a vector)
struct PlaneView {
	const uint8_t *p = nullptr;
	size_t n = 0;
	PlaneView() = default;
	PlaneView(const uint8_t *q, size_t c) : p(q), n(c) {}
	PlaneView(const std::vector<uint8_t> &v) : p(v.data()), n(v.size()) {}
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	const uint8_t *data() const { return p; }
	const uint8_t *begin() const { return p; }
	const uint8_t *end() const { return p + n; }
	uint8_t operator[](size_t i) const { return p[i]; }
};
// ---- cbm_unwalk.cpp: cbm::decode restated over flat arrays (cbm/decoder.h:27-211)
// seg_start: first decode rank of every connected component (+ end sentinel); seg_level[k]: 0 = the component touches no vertex
// coded before it, else 1 + the level of the latest component it reads from (shared non-manifold vertices)
// on_span (optional): called by the thread that finished a span of the parallel replay -- faces [f0, f1), half-edges [h0, h1) and
// vertices [v0, v1) are final in m.face_off (entries f0 + 1 .. f1) / m.org / m.twin / order_v from then on (a span links half-edges
// of its own components only).  Not called at all when the replay runs as one sequence.
// index / n_spans: which span of how many; comp_first[0 .. n_comp): the first vertex of every component the span holds (valid until
// cut_border_replay returns)
struct SpanDone {
	// ends_inside: the span stops inside a component (at a border snapshot): the component's last listed here goes on in the next span
	virtual void span(uint32_t index, uint32_t n_spans, uint32_t f0, uint32_t f1, uint32_t h0, uint32_t h1, uint32_t v0, uint32_t v1, const uint32_t *comp_first, uint32_t n_comp, bool ends_inside) = 0;
	virtual ~SpanDone() {}
};
struct SnapshotPoint;
// snaps: the border snapshots of the directory (restart points inside components; nullptr / HRY_NO_SNAPSHOT_REPLAY: not used)
void cut_border_replay(Mesh &m, const PlaneView *conn_planes, const std::vector<RestartPoint> &restarts,
                       const std::vector<RestartCounters> &counters,
                       OrderVec &order_v, std::vector<uint32_t> &seg_start, std::vector<uint32_t> &seg_level, SpanDone *on_span = nullptr,
                       const std::vector<SnapshotPoint> *snaps = nullptr);
unsigned cpu_allowance();           // CPUs this process may keep busy: affinity mask and control-group quota (HRY_CPUS overrides)
unsigned host_threads();            // HRY_HOST_THREADS, default min(16, cpu_allowance()) (large hosts: see cbm_walk.cpp)
uint32_t parallel_min_faces();      // HRY_PARALLEL_MIN_FACES, default 65536
// helper threads run on the CPUs of the memory node their creator is on (block_pool.cpp; HRY_NO_NUMA_BIND switches it off)
const void *callers_node_cpus();       // nullptr: one node, or unknown
void stay_on_node(const void *cpus);   // confines the calling thread
const void *callers_neighbour_cpus();   // the caller's cache domain (else memory node) without the caller's own core (nullptr: unknown)
const void *callers_cache_cpus(unsigned *n_cpus);   // CPUs sharing the caller's last-level cache (nullptr: unknown)
void run_on_helpers(unsigned n, void (*fn)(void*, unsigned), void *arg, const void *cpus);   // thread_pool.cpp: helper threads kept between calls
// CONTRACT: the indices may run ONE AFTER THE OTHER on the calling thread (a helper thread that cannot be created leaves its index
// to the caller, after index 0).  A body must therefore never wait for another index (no barriers, no hand-overs between indices);
// every body here takes work off a shared counter or owns a range computed from (index, n_threads), which stays what was asked for.
template <typename F> inline void parallel_for(unsigned n_threads, F &&body, const void *cpus = nullptr)   // body(thread index), returns when every index has
{
	if (n_threads <= 1) { body(0); return; }
	typedef typename std::remove_reference<F>::type Body;
	struct Job { Body *body; std::exception_ptr err; std::mutex mu; } job{ &body, nullptr, {} };
	run_on_helpers(n_threads, [](void *a, unsigned t) {
		Job &j = *(Job*)a;
		try { (*j.body)(t); } catch (...) { std::lock_guard<std::mutex> g(j.mu); if (!j.err) j.err = std::current_exception(); }
	}, &job, cpus);
	if (job.err) std::rethrow_exception(job.err);
}

// ---- static priors of the chunked container's planes (header.cpp; the oracle restates the rule and the directory form)
// Every chunk of a plane starts its adaptive table from the plane's histogram scaled to about kPriorK counts (symbols that
// do not occur get 0) instead of the reference's flat initial counts; planes shorter than kPriorMinSyms keep those.
constexpr uint32_t kPriorK = 1024, kPriorMinSyms = 1024;
bool plane_prior_from_hist(const uint32_t hist[256], uint64_t n, uint32_t table[256]);
void write_prior(std::vector<uint8_t> &out, bool use, const uint32_t table[256]);
size_t read_prior(const uint8_t *p, size_t avail, bool &use, uint32_t table[256]);   // returns bytes consumed; throws on damage

// ---- border snapshots in the chunked container's directory (header.cpp)
// a snapshot as a decoder holds it: where it lies (the cursors of a restart point), the counters of the older vertices its span
// names that are not on the border, the border itself
struct SnapshotPoint {
	RestartPoint at;
	RestartCounters counters;
	std::vector<uint32_t> parts, vtx;   // parts: size << 1 | edge_begin, bottom of the stack first; vtx per element, head -> tail
	std::vector<uint8_t> seen;
};
void write_snapshot_section(uint32_t spacing, const std::vector<BorderSnapshot> &snaps, const std::vector<RestartCounters> &counters, std::vector<uint8_t> &out);
size_t read_snapshot_section(const uint8_t *p, size_t avail, uint32_t nv, uint32_t &spacing, std::vector<SnapshotPoint> &out);   // returns bytes consumed; throws on damage

// ---- header.cpp (formats/hry/writer.cc:104-198 / reader.cc:60-177)
// reference single-stream format (compat_read.cpp): serial entropy decode + replay on the host; residual byte planes
// (plane-major, one plane per coded byte) are returned for the device reconstruction
void read_compat_stream(const uint8_t *p, size_t n, Mesh &m, OrderVec &order_v, std::vector<uint32_t> &seg_start,
                        std::vector<uint32_t> &seg_level, std::vector<uint8_t> &vplanes, std::vector<uint8_t> &fplanes);

// the same for a stream whose header announces general bindings (regions, shared records, corner lists; mesh.hpp Bindings):
// fills m.bind, leaves in every list the RESIDUAL codes of its records (record layout, in creation order) and says where each
// record was created: the half-edge of the vertex / the corner (the face index for face lists) and the slot of the list there
struct GenRecordEvents { std::vector<uint32_t> he; std::vector<uint8_t> slot; };
// plane_list >= 0: the residual bytes of that list additionally plane-major (one plane per coded byte, order_v.size() records each)
void read_general_stream(const uint8_t *p, size_t n, Mesh &m, OrderVec &order_v, std::vector<GenRecordEvents> &events,
                         std::vector<uint32_t> &seg_start, std::vector<uint32_t> &seg_level, int plane_list, std::vector<uint8_t> &planes);

// the same bookkeeping over the decoded planes of a chunked container (host copies): region planes (absent with one region),
// per list the reference kinds, the creation-order distances (4 byte planes), the per-vertex distances (2 byte planes) and the
// number of records coded as data (their residual bytes stay on the device)
struct GenHostPlanes {
	const uint8_t *regv = nullptr, *regf = nullptr;
	uint32_t n_regv = 0, n_regf = 0;
	struct L { const uint8_t *type = nullptr, *gh[4] = { nullptr, nullptr, nullptr, nullptr }, *lh[2] = { nullptr, nullptr }; uint32_t n_type = 0, n_gh = 0, n_lh = 0, n_data = 0; };
	std::vector<L> lists;
};
void read_general_planes(Mesh &m, const OrderVec &order_v, const GenHostPlanes &hp, std::vector<GenRecordEvents> &events);

// a shard writes the sizes of the full mesh (m.shard.g_*): the header of a sharded container describes the whole
void write_hry_header(const Mesh &m, int ver_minor, std::vector<uint8_t> &out);
// parses the header into a mesh skeleton (lists allocated unless alloc_records is false, no connectivity); returns bytes consumed
size_t read_hry_header(const uint8_t *p, size_t n, Mesh &m, int &ver_minor, bool alloc_records = true);

}   // namespace hry
