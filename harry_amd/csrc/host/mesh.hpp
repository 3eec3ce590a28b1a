// Host-side mesh model of the product: flat arrays only (what the kernels consume).
// Reference counterparts: structs/mesh.h:19-40, structs/conn.h:72-170, structs/attr.h:24-189, structs/mixing.h:41-200.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace hry {

// ---- big host arrays -------------------------------------------------------------------------------------------------
// The connectivity, the attribute records and the walk's outputs are tens of megabytes per million triangles and are
// allocated by every call.  From the C library such blocks are mmap'ed fresh each time: one page fault per 4 KiB on first
// touch (4.5 ms of a 20 ms decode went there).  They come from a small recycling pool instead -- a freed block keeps its
// pages and serves the next request of a similar size -- and vectors of them do not value-initialise on resize (every
// element is written before it is read; callers that need zeros say assign(n, 0)).
struct BlockPool {
	static void *take(size_t bytes);            // 64-byte aligned, capacity >= bytes
	static void give(void *p) noexcept;         // back to the pool (or to the C library when the pool is full)
	static constexpr size_t kMinBytes = 256u << 10;   // smaller requests bypass the pool
};
template <typename T> struct PoolAlloc {
	typedef T value_type;
	PoolAlloc() noexcept {}
	template <typename U> PoolAlloc(const PoolAlloc<U>&) noexcept {}
	T *allocate(size_t n)
	{
		const size_t bytes = n * sizeof(T);
		void *p = bytes >= BlockPool::kMinBytes ? BlockPool::take(bytes) : ::operator new(bytes);
		return (T*)p;
	}
	void deallocate(T *p, size_t n) noexcept
	{
		if (n * sizeof(T) >= BlockPool::kMinBytes) BlockPool::give(p); else ::operator delete(p);
	}
	template <typename U> void construct(U *p) noexcept { ::new ((void*)p) U; }   // default-init: no fill on resize()
	template <typename U, typename A0, typename... As> void construct(U *p, A0 &&a0, As &&... as) { ::new ((void*)p) U(std::forward<A0>(a0), std::forward<As>(as)...); }
	template <typename U> bool operator==(const PoolAlloc<U>&) const noexcept { return true; }
	template <typename U> bool operator!=(const PoolAlloc<U>&) const noexcept { return false; }
};
template <typename T> using BigVec = std::vector<T, PoolAlloc<T>>;
typedef BigVec<uint32_t> OrderVec;   // the decode order (one half-edge per coded vertex): 4 bytes per vertex, per call

struct Error : std::runtime_error {
	int code;
	Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

enum CompType : uint8_t { C_FLOAT, C_DOUBLE, C_ULONG, C_LONG, C_UINT, C_INT, C_USHORT, C_SHORT, C_UCHAR, C_CHAR, C_NONE };
static constexpr int kTypeSize[11] = { 4, 8, 8, 8, 4, 4, 2, 2, 1, 1, 0 };
enum { kInterpOther = 19 };   // mixing.h:20-39: ids >= OTHER carry a name
enum { kMaxComp = 32 };

inline CompType storage_type(CompType t, int q)   // mixing.h:101-108
{
	if (q == 0) return t;
	return q <= 8 ? C_UCHAR : q <= 16 ? C_USHORT : q <= 32 ? C_UINT : C_ULONG;
}

// One attribute list = AoS records; slot offsets follow the ORIGINAL component types (mixing.h:60),
// a quantised value occupies the low bytes of its slot.
struct AttrList {
	int target = 0;                        // 0 face, 1 vertex, 2 corner, 3 none (structs/attr.h:22)
	std::vector<CompType> type;
	std::vector<uint8_t> quant;
	std::vector<int> offset{0};            // ncomp + 1
	std::vector<int> interp_off, interp_len;   // indexed by interpretation id
	std::vector<std::string> interp_name;      // id - kInterpOther
	uint32_t count = 0;
	BigVec<uint8_t> data;
	std::vector<uint8_t> bmin, bmax;       // records in original types; empty until computed
	std::vector<uint32_t> bmin_at, bmax_at; // 1 + index of the first element that holds the bound, 0 = the initial value (device_bounds)
	bool have_bounds = false;

	int ncomp() const { return (int)type.size(); }
	int stride() const { return offset.back(); }
	CompType stype(int c) const { return storage_type(type[c], quant[c]); }
	int coded_bytes() const { int n = 0; for (int c = 0; c < ncomp(); ++c) n += kTypeSize[stype(c)]; return n; }
	void add_comp(CompType t, int q = 0)
	{
		type.push_back(t);
		quant.push_back((uint8_t)q);
		offset.push_back(offset.back() + kTypeSize[t]);
	}
	void add_interp(int id, int comp_index)   // mixing.h:156-166
	{
		if (id >= (int)interp_off.size()) {
			interp_off.resize(id + 1, -1);
			interp_len.resize(id + 1, 0);
			if (id >= kInterpOther) interp_name.resize(id - kInterpOther + 1);
		}
		if (interp_off[id] == -1) interp_off[id] = comp_index;
		++interp_len[id];
	}
};

// A shard of a larger mesh (host/shard.cpp): whole groups of connected components that share no vertex
// with the rest, in the numbering of the shard itself.  The vertex / face / half-edge numbering of a decoded mesh follows the
// coding order across ALL components of the full mesh (cbm/encoder.h:61-68,215; decoder.h:48,75,145,162), so every run of
// components that are consecutive in that order keeps its place in the full numbering.
struct ShardRun { uint32_t first_vertex, first_face, first_halfedge, n_vertices, n_faces, n_halfedges; };
struct ShardInfo {
	uint32_t g_nv = 0, g_nf = 0, g_ne = 0;   // sizes of the full mesh (the merged container's header); g_nf == 0: not a shard
	std::vector<uint32_t> seeds;             // start face of every component of the shard, in coding order (replaces the
	                                         // reference's start-face rule, writer.cc:40-46, which depends on the full face count)
	std::vector<ShardRun> runs;              // in coding order; the shard's own numbering lays them out back to back
	std::vector<uint32_t> vertex_of, face_of; // input index in the full mesh of every vertex / face of the shard (bounds ties, tests)
	// what the planner already knows about the shard's components, in their coding order (= the order of `seeds`): sizes, the
	// vertices each one introduces, the ties -- the walk on several threads takes them instead of labelling the components again
	std::vector<uint32_t> comp_faces, comp_halfedges, comp_fresh, comp_group;   // per component; group = smallest rank tied to it
	// general bindings (regions, shared records, corner lists): the records of every list are numbered by the decoder in the order
	// they are first coded (attrcode.h:443-531: cur_idx), across all components -- so every run also has its place in that
	// numbering, per list: run_records[run * 2 * nlists + 2 * l] = first record, [... + 1] = records the run creates.
	std::vector<uint32_t> g_list_count;      // records of every list of the full mesh (the merged container's header)
	std::vector<uint32_t> run_records;
	std::vector<std::vector<uint32_t>> record_of;   // per list: input index in the full mesh of every record of the shard (bounds ties)
	bool active() const { return g_nf != 0; }
};

// General attribute bindings (structs/attr.h:101-189): what the OBJ reader creates and what any .hry header may announce.
// Faces and vertices belong to regions; a region names the lists its elements carry records of (face regions also the lists of
// their corners); an element has one slot per list of its region, holding the index of ITS record in that list -- records are
// shared (several corners / vertices may name the same one).  The PLY layout (one face region -> list 0, one vertex region ->
// list 1, element i owns record i) keeps all of this implicit: Mesh::general == false.
struct Bindings {
	std::vector<uint16_t> reg_facelist, reg_vtxlist, reg_cornerlist;   // region x slot -> list
	std::vector<int> off_facelist{0}, off_vtxlist{0}, off_cornerlist{0};
	int nb_face = 0, nb_vtx = 0, nb_corner = 0;                        // slots per face / vertex / corner (max over regions)
	BigVec<uint16_t> face_reg, vtx_reg;                                // element -> region
	BigVec<uint32_t> face_attr, vtx_attr, corner_attr;                 // element x slot -> record of the bound list
	int nregs_face() const { return (int)off_facelist.size() - 1; }
	int nregs_vtx() const { return (int)off_vtxlist.size() - 1; }
	int nfacelists(int r) const { return off_facelist[r + 1] - off_facelist[r]; }
	int nvtxlists(int r) const { return off_vtxlist[r + 1] - off_vtxlist[r]; }
	int ncornerlists(int r) const { return off_cornerlist[r + 1] - off_cornerlist[r]; }
	int facelist(int r, int a) const { return reg_facelist[off_facelist[r] + a]; }
	int vtxlist(int r, int a) const { return reg_vtxlist[off_vtxlist[r] + a]; }
	int cornerlist(int r, int a) const { return reg_cornerlist[off_cornerlist[r] + a]; }
	int add_face_region(int nface, int ncorner)   // structs/mesh.h:106-113
	{
		off_facelist.push_back(off_facelist.back() + nface); off_cornerlist.push_back(off_cornerlist.back() + ncorner);
		reg_facelist.resize(off_facelist.back(), 0); reg_cornerlist.resize(off_cornerlist.back(), 0);
		return nregs_face() - 1;
	}
	int add_vtx_region(int n)   // structs/mesh.h:114-119
	{
		off_vtxlist.push_back(off_vtxlist.back() + n); reg_vtxlist.resize(off_vtxlist.back(), 0);
		return nregs_vtx() - 1;
	}
};

// Output bytes of the writers (PLY, OBJ, containers): grows like a vector, but new bytes are not zero-filled (a 19 MB file is 4 600
// fresh pages: filling them once is enough) and the buffer can be handed to the C boundary as it is.  Large buffers come from the
// recycling pool like every other big array (a 290 MB container from the C library is 71 000 fresh pages per encode, faulted in
// under the device's copy, and as many returned to the kernel by free: 80 ms of a 490 ms sharded encode of the configs[3] mesh);
// BlockPool::give takes both kinds back, and so does hry_free.
class ByteSink {
	uint8_t *p_ = nullptr;
	size_t n_ = 0, cap_ = 0;
	static uint8_t *get(size_t c)
	{
		uint8_t *q = (uint8_t*)(c >= BlockPool::kMinBytes ? BlockPool::take(c) : malloc(c));
		if (!q) throw std::bad_alloc();
		return q;
	}
	void grow(size_t want)
	{
		size_t c = cap_ ? cap_ : 4096;
		while (c < want) c += c / 2 + 4096;
		uint8_t *q = get(c);
		if (n_) memcpy(q, p_, n_);
		BlockPool::give(p_);
		p_ = q; cap_ = c;
	}
public:
	ByteSink() = default;
	ByteSink(const ByteSink&) = delete;
	ByteSink &operator=(const ByteSink&) = delete;
	~ByteSink() { BlockPool::give(p_); }
	size_t size() const { return n_; }
	bool empty() const { return n_ == 0; }
	uint8_t *data() { return p_; }
	const uint8_t *data() const { return p_; }
	const uint8_t *begin() const { return p_; }
	const uint8_t *end() const { return p_ + n_; }
	uint8_t &operator[](size_t i) { return p_[i]; }
	void clear() { n_ = 0; }
	void reserve(size_t c) { if (c > cap_) grow(c); }
	void resize(size_t n) { if (n > cap_) grow(n); n_ = n; }   // new bytes are NOT initialised
	void push_back(uint8_t b) { if (n_ == cap_) grow(n_ + 1); p_[n_++] = b; }
	template <typename It> void append(It b, It e) { const size_t k = (size_t)(e - b); if (n_ + k > cap_) grow(n_ + k); if (k) memcpy(p_ + n_, &*b, k); n_ += k; }
	template <typename It> void assign(It b, It e) { n_ = 0; append(b, e); }
	uint8_t *release(size_t *n) { uint8_t *q = p_ ? p_ : (uint8_t*)malloc(1); if (n) *n = n_; p_ = nullptr; n_ = cap_ = 0; return q; }   // the caller frees with BlockPool::give (hry_free)
};

struct Mesh {
	uint32_t nv = 0, nf = 0;
	BigVec<uint32_t> face_off{0};        // nf + 1
	BigVec<uint32_t> org;                // per half-edge
	BigVec<uint32_t> twin;               // per half-edge, flat id; self = border
	bool twins_pending = false;          // a reader left the matching for later: on the device at the first upload (twins.hip), else
	                                     // on the host when first asked for (ensure_twins)
	std::vector<uint8_t> have_degree;    // have_degree[d] != 0 iff a polygon with d edges exists (faces.h:44-56)
	std::vector<AttrList> lists = std::vector<AttrList>(2);   // PLY layout: [0] face attributes, [1] vertex attributes (formats/ply/reader.cc:388-400)
	bool general = false;                // true: `bind` holds regions and element -> record maps, any number of lists
	Bindings bind;
	uint64_t device_token = 0;           // identity of the HBM-resident copy, 0 = none
	uint32_t declared_ne = 0;            // half-edge count announced by a .hry header (the connectivity follows later)
	ShardInfo shard;                     // set by shard_extract: this mesh is a shard of a larger one
	std::vector<ShardRun> covered;       // set by the decoder of a sharded container: the runs of the whole numbering that were decoded
	std::vector<uint32_t> covered_records;   // ... general bindings: their record ranges (2 x lists words per run, as ShardInfo::run_records)
	bool partial = false;                // a share of a sharded container's segments (shard_count > 1): everything outside `covered` is
	                                     // filler (empty faces, zero records) -- readable through the accessors, refused by every consumer

	uint32_t ne() const { return face_off.back(); }
	uint64_t ntri() const { return (uint64_t)ne() - 2ull * nf; }
	bool uniform_degree(int &d) const
	{
		d = 0;
		for (size_t i = 0; i < have_degree.size(); ++i)
			if (have_degree[i]) { if (d) return false; d = (int)i; }
		return d != 0;
	}
};

}   // namespace hry
