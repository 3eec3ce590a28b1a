/*
 * hry_oracle.cc -- CPU ORACLE: single-threaded restatement of the reference .hry v0.1 codec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (harry_amd/) links, loads or calls this file.
 * It exists so that every HIP kernel and the host-side walk of the product can be checked, bit for bit,
 * against the behaviour of maxvonbuelow/harry (reference mounted at /root/reference in the build
 * container).  Pinning: tests/test_oracle_golden.py compares this restatement with .hry files and decoded
 * meshes produced by the unmodified reference binary (tests/golden/, generator tests/golden/make_golden.py)
 * and, where /root/reference exists, with oracle/_ref/harry_ref run live.
 *
 * Every block cites the reference file:line whose behaviour it restates.  The code is written from the
 * behavioural description, not transcribed: flat half-edge ids instead of (face,
 * edge) pairs, explicit typed dispatch instead of the visitor templates, one translation unit.
 */
#include "hry_oracle.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <limits>
#include <list>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace ho {

// ------------------------------------------------------------------------------------------------
// Component types (structs/mixing.h:18-19)
// ------------------------------------------------------------------------------------------------
enum Type : uint8_t { T_FLOAT, T_DOUBLE, T_ULONG, T_LONG, T_UINT, T_INT, T_USHORT, T_SHORT, T_UCHAR, T_CHAR, T_NONE };
static const int TSIZE[11] = { 4, 8, 8, 8, 4, 4, 2, 2, 1, 1, 0 };
enum { I_POS = 0, I_NORMAL = 1, I_COLOR = 2, I_COLOR_AMBIENT = 3, I_COLOR_DIFFUSE = 4, I_COLOR_SPECULAR = 5,
       I_TEX = 16, I_SCALE = 17, I_CONFIDENCE = 18, I_OTHER = 19 };
enum Target { TG_FACE, TG_VTX, TG_CORNER, TG_NONE };

static Type quant_type(int q)   // structs/mixing.h:101-108
{
	if (q <= 8) return T_UCHAR;
	if (q <= 16) return T_USHORT;
	if (q <= 32) return T_UINT;
	if (q <= 64) return T_ULONG;
	return T_NONE;
}

// record format: component types, quantisation bits, storage types, slot offsets (structs/mixing.h:41-99).
// Slot offsets depend on the ORIGINAL type only (mixing.h:60), so a quantised value lives in the low bytes.
struct Fmt {
	std::vector<Type> type, stype;
	std::vector<int> quant, off{0};
	void add(Type t, int q = 0)
	{
		type.push_back(t);
		quant.push_back(q);
		stype.push_back(q == 0 ? t : quant_type(q));
		off.push_back(off.back() + TSIZE[t]);
	}
	void setquant(int i, int q) { quant[i] = q; stype[i] = q == 0 ? type[i] : quant_type(q); }
	int size() const { return (int)type.size(); }
	int bytes() const { return off.back(); }
};

struct Interps {   // structs/mixing.h:139-200
	std::vector<int> off, len;
	std::vector<std::string> names;
	void append(int interp, int o)
	{
		if (interp >= (int)off.size()) {
			off.resize(interp + 1, -1);
			len.resize(interp + 1, 0);
			if (interp >= I_OTHER) names.resize(interp - I_OTHER + 1, "");
		}
		if (off[interp] == -1) off[interp] = o;
		++len[interp];
	}
	void describe(int interp, const std::string &name)
	{
		int id = interp - I_OTHER;
		if (id < 0) return;
		if (id >= (int)names.size()) names.resize(id + 1, "");
		names[id] = name;
	}
	int size() const { return (int)off.size(); }
};

struct List {   // structs/attr.h:24-99
	Fmt fmt;
	Interps interps;
	int target = TG_NONE;
	uint32_t count = 0;
	std::vector<uint8_t> data;          // count * fmt.bytes()
	std::vector<uint8_t> bmin, bmax;    // records in the dequantised (original) types
	uint8_t *rec(uint32_t i) { return data.data() + (size_t)i * fmt.bytes(); }
	const uint8_t *rec(uint32_t i) const { return data.data() + (size_t)i * fmt.bytes(); }
};

typedef uint32_t he_t;   // flat half-edge id = face_offset[f] + local edge

struct Mesh {
	// connectivity (structs/conn.h:72-170, structs/faces.h:18-98)
	std::vector<uint32_t> foff{0};
	std::vector<uint32_t> org;
	std::vector<he_t> twin;
	std::vector<uint32_t> eface;
	std::vector<char> have_deg;
	uint32_t conn_nv = 0;
	// attributes + bindings (structs/attr.h:101-189); the oracle supports what the PLY reader creates:
	// one face region bound to list 0, one vertex region bound to list 1, identity element->attribute maps
	std::vector<List> lists;
	uint32_t nv = 0, nf = 0;
	// a shard of a larger mesh (sharded container .hry v0.3 of this implementation, see "sharded container" below): sizes of
	// the whole mesh for the header, the start face of every component in coding order, and the place of every run of
	// consecutive components in the numbering of the whole decoded mesh
	uint32_t g_nv = 0, g_nf = 0, g_ne = 0;
	std::vector<uint32_t> seeds;
	std::vector<std::array<uint32_t, 6>> runs;   // first_vertex, first_face, first_halfedge, n_vertices, n_faces, n_halfedges
	bool is_shard() const { return g_nf != 0; }
	// general bindings (structs/attr.h:101-189): what the OBJ reader creates and what any .hry may announce.  Off = the PLY
	// layout above (kept implicit: the 28 M-triangle checks would not fit otherwise).
	struct Bind {
		bool on = false;
		std::vector<uint16_t> face_reg, vtx_reg;                            // element -> region
		int nb_face = 0, nb_vtx = 0, nb_corner = 0;                         // slots per element (max over regions)
		std::vector<uint32_t> face_attr, vtx_attr, corner_attr;             // element x slot -> record of the bound list
		std::vector<uint16_t> reg_facelist, reg_vtxlist, reg_cornerlist;    // region x slot -> list
		std::vector<int> off_facelist{0}, off_vtxlist{0}, off_cornerlist{0};
		int nregs_face() const { return (int)off_facelist.size() - 1; }
		int nregs_vtx() const { return (int)off_vtxlist.size() - 1; }
		int nfacelists(int r) const { return off_facelist[r + 1] - off_facelist[r]; }
		int nvtxlists(int r) const { return off_vtxlist[r + 1] - off_vtxlist[r]; }
		int ncornerlists(int r) const { return off_cornerlist[r + 1] - off_cornerlist[r]; }
		int facelist(int r, int a) const { return reg_facelist[off_facelist[r] + a]; }
		int vtxlist(int r, int a) const { return reg_vtxlist[off_vtxlist[r] + a]; }
		int cornerlist(int r, int a) const { return reg_cornerlist[off_cornerlist[r] + a]; }
		int add_face_region(int nface, int ncorner)   // mesh.h:106-113
		{
			off_facelist.push_back(off_facelist.back() + nface); off_cornerlist.push_back(off_cornerlist.back() + ncorner);
			reg_facelist.resize(off_facelist.back(), 0); reg_cornerlist.resize(off_cornerlist.back(), 0);
			return nregs_face() - 1;
		}
		int add_vtx_region(int n)   // mesh.h:114-119
		{
			off_vtxlist.push_back(off_vtxlist.back() + n); reg_vtxlist.resize(off_vtxlist.back(), 0);
			return nregs_vtx() - 1;
		}
	} bind;
	void make_general()   // the PLY layout spelled out: one face region -> list 0, one vertex region -> list 1, private records
	{
		if (bind.on) return;
		bind.on = true;
		bind.add_face_region(1, 0); bind.reg_facelist[0] = 0;
		bind.add_vtx_region(1); bind.reg_vtxlist[0] = 1;
		bind.nb_face = 1; bind.nb_vtx = 1; bind.nb_corner = 0;
		bind.face_reg.assign(nf, 0); bind.vtx_reg.assign(nv, 0);
		bind.face_attr.resize(nf); bind.vtx_attr.resize(nv);
		for (uint32_t i = 0; i < nf; ++i) bind.face_attr[i] = i;
		for (uint32_t i = 0; i < nv; ++i) bind.vtx_attr[i] = i;
	}

	uint32_t add_face(int ne)   // conn.h:83-93
	{
		uint32_t f = (uint32_t)foff.size() - 1;
		uint32_t o = foff.back();
		foff.push_back(o + ne);
		if (ne >= (int)have_deg.size()) have_deg.resize(ne + 1, 0);
		have_deg[ne] = 1;
		org.resize(o + ne, 0);
		twin.resize(o + ne);
		eface.resize(o + ne, f);
		for (int i = 0; i < ne; ++i) twin[o + i] = o + i;
		return f;
	}
	void set_org(he_t e, uint32_t v) { org[e] = v; conn_nv = std::max(conn_nv, v + 1); }
	int deg(uint32_t f) const { return (int)(foff[f + 1] - foff[f]); }
	he_t next(he_t e) const { uint32_t f = eface[e]; return e + 1 == foff[f + 1] ? foff[f] : e + 1; }
	he_t prev(he_t e) const { uint32_t f = eface[e]; return e == foff[f] ? foff[f + 1] - 1 : e - 1; }
	uint32_t dest(he_t e) const { return org[next(e)]; }
	void merge(he_t a, he_t b) { twin[a] = b; twin[b] = a; }   // conn.h:161-165 (a==b makes a border)
	uint32_t num_face() const { return (uint32_t)foff.size() - 1; }
	uint32_t num_edge() const { return foff.back(); }
	uint64_t num_tri() const { return (uint64_t)foff.back() - 2ull * num_face(); }
};

static std::string g_err;

// ------------------------------------------------------------------------------------------------
// typed scalar helpers: float<->ordered int, residual folding, predictor
// (formats/hry/transform.h:19-48, formats/hry/prediction.h:21-147)
// ------------------------------------------------------------------------------------------------
template <int S> struct ints;
template <> struct ints<1> { typedef int8_t s; typedef uint8_t u; };
template <> struct ints<2> { typedef int16_t s; typedef uint16_t u; };
template <> struct ints<4> { typedef int32_t s; typedef uint32_t u; };
template <> struct ints<8> { typedef int64_t s; typedef uint64_t u; };

template <typename T, typename U> static inline U bitcast(T v) { U r; static_assert(sizeof(T) == sizeof(U), ""); memcpy(&r, &v, sizeof(U)); return r; }

// transform.h:19-23: i ^ ((-(unsigned(i) >> msb)) >> 1): negative floats get their low bits inverted
template <int S> static inline typename ints<S>::s flip_float_bits(typename ints<S>::s i)
{
	typedef typename ints<S>::u U;
	U sign = (U)((U)i >> (S * 8 - 1));
	U m = (U)((U)(0 - sign) >> 1);
	return (typename ints<S>::s)((U)i ^ m);
}

// prediction.h:33-44: the sign-flip table is indexed with sizeof(T) instead of sizeof(T)-1, so the mask that is
// applied to 4-byte types is ZERO.  8-byte types read past the table: unsupported here.
template <typename T> static inline uint64_t signmask_as_indexed()
{
	static const uint64_t tbl[8] = { 0x80ull, 0x8000ull, 0, 0x80000000ull, 0, 0, 0, 0x8000000000000000ull };
	if (sizeof(T) >= 8) throw std::runtime_error("oracle: lossless residuals of 8-byte floating types are unspecified in the reference (prediction.h:33-44)");
	return tbl[sizeof(T)];
}

template <typename T> static inline int bits_of(int q) { return q == 0 ? (int)sizeof(T) * 8 : q; }   // prediction.h:21-25
template <typename T> static inline T low_mask(int bits)   // prediction.h:27-31
{
	return bits == (int)(sizeof(T) << 3) ? T(-1) : T((1 << bits) - 1);
}

// prediction.h:81-99 (integral form).  The expressions keep the reference's C++ types so that integer
// promotion/truncation behaves identically for 1-, 2-, 4- and 8-byte T.
template <typename T> static T fold_residual_int(const T raw, const T pred, int bits)
{
	const T max_pos = low_mask<T>(bits) - pred;
	if (pred == T(0)) return raw;
	const T balanced_max = std::min(T(pred), max_pos);
	if (raw < pred) {
		const T dlt = pred - raw;
		if (dlt > balanced_max) return dlt + balanced_max;
		return T(dlt << 1) - 1;
	} else {
		const T dlt = raw - pred;
		if (dlt > balanced_max) return dlt + balanced_max;
		return T(dlt << 1);
	}
}
// prediction.h:46-64
template <typename T> static T unfold_residual_int(const T delta, const T pred, int bits)
{
	const T sel[2] = { T(0), T(~T(0)) };
	const T max_pos = low_mask<T>(bits) - pred;
	if (pred == T(0)) return delta;
	const T balanced_max = std::min(T(pred - T(1)), max_pos);
	if ((delta >> 1) > balanced_max) {
		if (max_pos >= pred) return pred + delta - balanced_max - T(1);
		else return pred - delta + balanced_max;
	}
	return pred + (T(delta >> 1) ^ sel[delta & 1]);
}
template <typename T> static T fold_residual(const T raw, const T pred, int q, std::false_type) { return fold_residual_int<T>(raw, pred, bits_of<T>(q)); }
template <typename T> static T unfold_residual(const T d, const T pred, int q, std::false_type) { return unfold_residual_int<T>(d, pred, bits_of<T>(q)); }
// prediction.h:101-113 / 65-73: floats are mapped to order-preserving ints first
template <typename T> static T fold_residual(const T raw, const T pred, int q, std::true_type)
{
	typedef typename ints<sizeof(T)>::s S;
	typedef typename ints<sizeof(T)>::u U;
	U m = (U)signmask_as_indexed<T>();
	U p = (U)flip_float_bits<sizeof(T)>(bitcast<T, S>(pred)) ^ m;
	U r = (U)flip_float_bits<sizeof(T)>(bitcast<T, S>(raw)) ^ m;
	return bitcast<U, T>(fold_residual_int<U>(r, p, bits_of<T>(q)));
}
template <typename T> static T unfold_residual(const T delta, const T pred, int q, std::true_type)
{
	typedef typename ints<sizeof(T)>::s S;
	typedef typename ints<sizeof(T)>::u U;
	U m = (U)signmask_as_indexed<T>();
	U p = (U)flip_float_bits<sizeof(T)>(bitcast<T, S>(pred)) ^ m;
	U v = unfold_residual_int<U>(bitcast<T, U>(delta), p, bits_of<T>(q));
	S s = (S)v ^ (S)m;
	return bitcast<S, T>(flip_float_bits<sizeof(T)>(s));
}
// prediction.h:121-147 parallelogram predictor: saturating for integers, plain fp for floats
template <typename T> static T paral_predict(const T v0, const T v1, const T v2, int q, std::false_type)
{
	const T max = low_mask<T>(bits_of<T>(q));
	if (v1 < v2) {
		const T tmp = v2 - v1;
		if (tmp > v0) return T(0);
		else return v0 - tmp;
	} else {
		const T tmp = v1 - v2;
		const T v = v0 + tmp;
		if ((v > max) || (v < v0)) return max;
		return v;
	}
}
template <typename T> static T paral_predict(const T v0, const T v1, const T v2, int, std::true_type) { return v0 + (v1 - v2); }

// big accumulator type per storage type (structs/mixing.h:110-127)
template <typename T> struct big_of { typedef int64_t type; };
template <> struct big_of<float> { typedef double type; };
template <> struct big_of<double> { typedef double type; };
template <> struct big_of<uint64_t> { typedef uint64_t type; };
// transform.h:90-93
static inline double div_round(double n, double d) { return n / d; }
static inline int64_t div_round(int64_t n, int64_t d) { return (n + (d >> 1)) / d; }
static inline uint64_t div_round(uint64_t n, uint64_t d) { return (n + (d >> 1)) / d; }

// invoke f(T()) with the C++ type of a storage type (mixing.h:272-289 dispatch)
template <typename F> static void with_type(Type t, F &&f)
{
	switch (t) {
	case T_FLOAT: f(float()); break;
	case T_DOUBLE: f(double()); break;
	case T_ULONG: f(uint64_t()); break;
	case T_LONG: f(int64_t()); break;
	case T_UINT: f(uint32_t()); break;
	case T_INT: f(int32_t()); break;
	case T_USHORT: f(uint16_t()); break;
	case T_SHORT: f(int16_t()); break;
	case T_UCHAR: f(uint8_t()); break;
	case T_CHAR: f(int8_t()); break;
	default: throw std::runtime_error("oracle: bad component type");
	}
}
template <typename T> static inline T ld(const uint8_t *p) { T v; memcpy(&v, p, sizeof(T)); return v; }
template <typename T> static inline void st(uint8_t *p, T v) { memcpy(p, &v, sizeof(T)); }

// ------------------------------------------------------------------------------------------------
// bit I/O (arith/bitstream.h:16-62): MSB first, last byte zero padded, reads past the end give 0xFF bytes
// ------------------------------------------------------------------------------------------------
struct BitSink {
	std::vector<uint8_t> &out;
	int fill = 0;
	uint8_t cur = 0;
	explicit BitSink(std::vector<uint8_t> &o) : out(o) {}
	void put(unsigned bit)
	{
		cur |= (uint8_t)(bit << (7 - fill));
		if (++fill == 8) pad();
	}
	void pad()
	{
		if (fill == 0) return;
		out.push_back(cur);
		cur = 0; fill = 0;
	}
};
struct BitSource {
	const uint8_t *p, *end;
	int used = 8;
	uint8_t cur = 0;
	BitSource(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
	unsigned get()
	{
		if (used == 8) {
			cur = p < end ? *p++ : 0xFF;   // istream::get() == -1 past EOF (bitstream.h:27)
			used = 0;
		}
		return (cur >> (7 - used++)) & 1;
	}
};

// ------------------------------------------------------------------------------------------------
// Moffat-Neal-Witten range coder (arith/coder.h:27-172), templated on the register type like the reference's
// arith::Encoder<TF> / arith::Decoder<TF>: 64-bit registers for the reference stream (the default instantiation,
// which formats/hry uses), 32-bit registers for the streams of the chunked container.
// ------------------------------------------------------------------------------------------------
template <typename TF> struct RangeEncoderT {
	static constexpr int B = (int)sizeof(TF) * 8;
	static constexpr TF HALF = TF(1) << (B - 1), QUARTER = TF(1) << (B - 2);
	TF low = 0, range = HALF;
	uint64_t pending = 0;
	BitSink sink;
	bool done = false;
	explicit RangeEncoderT(std::vector<uint8_t> &o) : sink(o) {}
	void emit(unsigned b)   // coder.h:105-112 bit-plus-follow
	{
		sink.put(b);
		for (; pending > 0; --pending) sink.put(!b);
	}
	void encode(TF l, TF h, TF t)   // coder.h:69-91
	{
		TF r = range / t;
		low = (TF)(low + r * l);
		range = h < t ? (TF)(r * (h - l)) : (TF)(range - r * l);
		while (range <= QUARTER) {
			if (low <= HALF && (TF)(low + range) <= HALF) emit(0);
			else if (low >= HALF) { emit(1); low -= HALF; }
			else { ++pending; low -= QUARTER; }
			low = (TF)(low << 1);
			range = (TF)(range << 1);
		}
	}
	void finish()   // coder.h:58-67: the B bits of low, then pad
	{
		if (done) return;
		done = true;
		for (int i = B - 1; i >= 0; --i) emit((unsigned)((low >> i) & 1));
		sink.pad();
	}
};
template <typename TF> struct RangeDecoderT {
	static constexpr int B = (int)sizeof(TF) * 8;
	static constexpr TF HALF = TF(1) << (B - 1), QUARTER = TF(1) << (B - 2);
	TF range = HALF, value = 0, r = 0;
	BitSource src;
	RangeDecoderT(const uint8_t *b, const uint8_t *e) : src(b, e)
	{
		for (int i = 0; i < B; ++i) value = (TF)(2 * value + src.get());   // coder.h:124-129
	}
	TF target(TF t)   // coder.h:134-138
	{
		r = range / t;
		return std::min<TF>(t - 1, value / r);
	}
	void consume(TF l, TF h, TF t)   // coder.h:140-153
	{
		value = (TF)(value - r * l);
		range = h < t ? (TF)(r * (h - l)) : (TF)(range - r * l);
		while (range <= QUARTER) {
			range = (TF)(range << 1);
			value = (TF)(2 * value + src.get());
		}
	}
};
typedef RangeEncoderT<uint64_t> RangeEncoder;
typedef RangeDecoderT<uint64_t> RangeDecoder;
typedef RangeEncoderT<uint32_t> ChunkEncoder;   // arith::Encoder<uint32_t>
typedef RangeDecoderT<uint32_t> ChunkDecoder;   // arith::Decoder<uint32_t>

// ------------------------------------------------------------------------------------------------
// adaptive frequency table: raw counts + Fenwick tree (arith/stat_adaptive.h:26-126)
// ------------------------------------------------------------------------------------------------
struct FreqTable {
	std::vector<uint64_t> tree, cnt;
	uint32_t n, top;
	explicit FreqTable(uint32_t n_ = 256) : tree(n_, 0), cnt(n_, 0), n(n_)
	{
		top = 1;
		while ((top << 1) <= n) top <<= 1;   // highest power of two <= n (msb.h:6-13)
	}
	uint64_t prefix(uint32_t k) const   // sum of the first k counts
	{
		uint64_t s = 0;
		for (uint32_t i = k; i != 0; i &= i - 1) s += tree[i - 1];
		return s;
	}
	uint64_t total() const { return prefix(n); }
	void bump(uint32_t s, uint64_t d)
	{
		for (uint32_t i = s + 1; i <= n; i += i & (0 - i)) tree[i - 1] += d;
		cnt[s] += d;
	}
	void inc(uint32_t s, uint64_t d = 1)   // stat_adaptive.h:77-82 incl. the (practically dead) halving
	{
		bump(s, d);
		if (total() > (1ull << 62))
			for (uint32_t i = 0; i < n; ++i) bump(i, 0 - (cnt[i] >> 1));
	}
	void set(uint32_t s, uint64_t f) { bump(s, f - cnt[s]); }   // stat_adaptive.h:87-90 (no halving test)
	void range_of(uint32_t s, uint64_t &l, uint64_t &h) const { h = prefix(s + 1); l = h - cnt[s]; }
	uint32_t find(uint64_t target, uint64_t &l, uint64_t &h) const   // stat_adaptive.h:55-72
	{
		uint32_t s = 0;
		uint64_t rem = target;
		for (uint32_t step = top; step > 0; step >>= 1) {
			if (s + step <= n && tree[s + step - 1] <= rem) {
				rem -= tree[s + step - 1];
				s += step;
			}
		}
		l = target - rem;
		h = l + cnt[s];
		return s;
	}
};

// ------------------------------------------------------------------------------------------------
// context inventory of a .hry (formats/hry/models.h:183-237) with a flat numbering for traces
// ------------------------------------------------------------------------------------------------
enum { CTX_IOP = 0, CTX_OP = 1, CTX_ELEM = 2, CTX_PART = 6, CTX_VERT = 8, CTX_NUMTRI = 12, CTX_REGFACE = 14, CTX_REGVTX = 16, CTX_ATTR0 = 18 };
enum { ATTR_TYPE = 0, ATTR_GHIST = 1, ATTR_LHIST = 5, ATTR_DATA = 7 };
enum InitOp { IOP_INIT, IOP_TRI100, IOP_TRI010, IOP_TRI001, IOP_TRI110, IOP_TRI101, IOP_TRI011, IOP_TRI111, IOP_EOM };   // cbm/base.h:16-22
enum Op { OP_BORDER, OP_CONNBWD, OP_SPLIT, OP_UNION, OP_NM, OP_NEWVTX, OP_CONNFWD, OP_CLOSE };                            // cbm/base.h:23
enum { A_DATA = 0, A_HIST = 1, A_LHIST = 2 };

struct Models {
	std::vector<FreqTable> tab;       // indexed by ctx id
	std::vector<int> attr_base;       // per list
	// order-conditioned op model state (models.h:49-120)
	uint64_t c_all = 2, c_new[8], c_fwd[8];
	int order = 0;

	explicit Models(const Mesh &m)
	{
		tab.reserve(64);
		tab.emplace_back(IOP_EOM + 1);                       // CTX_IOP: 9 symbols, all 1 (models.h:27-32)
		for (uint32_t i = 0; i <= IOP_EOM; ++i) tab[CTX_IOP].inc(i);
		tab.emplace_back(OP_CONNFWD + 1);                    // CTX_OP: 7 symbols, all 1 (models.h:56-60)
		for (uint32_t i = 0; i <= OP_CONNFWD; ++i) tab[CTX_OP].inc(i);
		for (int i = 0; i < 8; ++i) c_new[i] = c_fwd[i] = 1;
		for (int i = 0; i < 4 + 2 + 4; ++i) { tab.emplace_back(256); ones(tab.back()); }   // elem, part, vert (ModelMult init=true)
		for (int i = 0; i < 2 + 2 + 2; ++i) tab.emplace_back(256);                           // numtri, regface, regvtx (init=false)
		// models.h:209-217; ModelMult::init(T) walks the bytes of the value as (signed) char (model.h:49-55)
		for (size_t d = 0; d < m.have_deg.size(); ++d)
			if (m.have_deg[d]) seed16(CTX_NUMTRI, (uint16_t)(d - 2));
		if (m.bind.on) {          // models.h:212-217
			for (int r = 0; r < m.bind.nregs_face(); ++r) seed16(CTX_REGFACE, (uint16_t)r);
			for (int r = 0; r < m.bind.nregs_vtx(); ++r) seed16(CTX_REGVTX, (uint16_t)r);
		} else {
			seed16(CTX_REGFACE, 0);   // one face region, one vertex region
			seed16(CTX_REGVTX, 0);
		}
		for (size_t l = 0; l < m.lists.size(); ++l) {
			attr_base.push_back((int)tab.size());
			tab.emplace_back(256);                           // attr_type: DATA, HIST (+LHIST for corner lists) (models.h:201-203)
			tab.back().inc(A_DATA); tab.back().inc(A_HIST);
			if (m.lists[l].target == TG_CORNER) tab.back().inc(A_LHIST);
			for (int i = 0; i < 4 + 2; ++i) { tab.emplace_back(256); ones(tab.back()); }   // ghist, lhist
			const Fmt &f = m.lists[l].fmt;
			for (int c = 0; c < f.size(); ++c)
				for (int b = 0; b < TSIZE[f.stype[c]]; ++b) { tab.emplace_back(256); ones(tab.back()); }   // models.h:126-144
		}
	}
	static void ones(FreqTable &t) { for (uint32_t j = 0; j < 256; ++j) t.inc(j); }
	void seed16(int ctx, uint16_t v)
	{
		signed char b0 = (signed char)(v & 0xff), b1 = (signed char)(v >> 8);
		if (b0 < 0 || b1 < 0) throw std::runtime_error("oracle: model seed byte >= 128 is out of bounds in the reference (model.h:49-55)");
		tab[ctx].inc((uint32_t)b0);
		tab[ctx + 1].inc((uint32_t)b1);
	}
	int data_ctx(int l, const Fmt &f, int comp) const
	{
		int c = attr_base[l] + ATTR_DATA;
		for (int i = 0; i < comp; ++i) c += TSIZE[f.stype[i]];
		return c;
	}
	// models.h:91-119
	int order_class() const { int o = order - 1; return o < 8 ? o : 7; }
	void op_prepare()
	{
		int i = order_class();
		uint64_t nv = c_new[i] * c_all / (c_new[i] + c_fwd[i]);
		tab[CTX_OP].set(OP_NEWVTX, nv);
		tab[CTX_OP].set(OP_CONNFWD, c_all - nv);
	}
	void op_update(uint32_t s)
	{
		int i = order_class();
		if (s == OP_NEWVTX) { ++c_all; ++c_new[i]; }
		else if (s == OP_CONNFWD) { ++c_all; ++c_fwd[i]; }
		else tab[CTX_OP].inc(s);
	}
};

// ------------------------------------------------------------------------------------------------
// symbol layer (formats/hry/io.h:19-231)
// ------------------------------------------------------------------------------------------------
enum { REC_OP0 = 8192, REC_SLOTS = 8200 };   // recording slots: context id, or REC_OP0 + order class for operations (above every context id)
struct SymWriter {
	Models &md;
	RangeEncoder &rc;
	std::vector<ho_sym> *trace;
	std::vector<std::vector<uint8_t>> *record = nullptr;   // chunked profile: collect symbols per context instead of coding
	// chunked profile: state at the start of every connected component (restart points of the container directory)
	struct Mark { uint32_t n_grp[5], n_op[8], first_vertex, first_face, min_ref; };
	std::vector<Mark> marks;
	uint32_t min_ref = 0xffffffffu;
	void mark_component(uint32_t next_id, uint32_t nfaces)
	{
		snaps_in_component = 0;
		if (!record) return;
		if (!marks.empty()) marks.back().min_ref = min_ref;
		Mark k;
		k.n_grp[0] = (uint32_t)(*record)[CTX_IOP].size(); k.n_grp[1] = (uint32_t)(*record)[CTX_ELEM].size();
		k.n_grp[2] = (uint32_t)(*record)[CTX_PART].size(); k.n_grp[3] = (uint32_t)(*record)[CTX_VERT].size();
		k.n_grp[4] = (uint32_t)(*record)[CTX_NUMTRI].size();
		for (int i = 0; i < 8; ++i) k.n_op[i] = (uint32_t)(*record)[REC_OP0 + i].size();
		k.first_vertex = next_id; k.first_face = nfaces; k.min_ref = 0xffffffffu;
		marks.push_back(k);
		min_ref = 0xffffffffu;
	}
	void finish_marks() { if (!marks.empty()) marks.back().min_ref = min_ref; }
	SymWriter(Models &m, RangeEncoder &r, std::vector<ho_sym> *t) : md(m), rc(r), trace(t) {}
	void code(int ctx, uint32_t s)
	{
		FreqTable &f = md.tab[ctx];
		uint64_t l, h, t = f.total();
		f.range_of(s, l, h);
		if (trace) trace->push_back(ho_sym{ (uint32_t)ctx, s, l, h, t });
		rc.encode(l, h, t);
	}
	void sym(int ctx, uint32_t s)      // model.h:57-66
	{
		if (record) { (*record)[ctx].push_back((uint8_t)s); return; }
		code(ctx, s); md.tab[ctx].inc(s);
	}
	void bytes(int ctx, const uint8_t *p, int n) { for (int i = 0; i < n; ++i) sym(ctx + i, p[i]); }
	void iop(uint32_t s) { sym(CTX_IOP, s); }
	void op(uint32_t s)   // models.h:74-80
	{
		if (record) { (*record)[REC_OP0 + md.order_class()].push_back((uint8_t)s); return; }
		md.op_prepare(); code(CTX_OP, s); md.op_update(s);
	}
	void elem(int i)   // io.h:150-153 + transform.h:25-30 zigzag
	{
		uint32_t c = (uint32_t)i, z = (c << 1) ^ ((c >> 31) ? 0xffffffffu : 0u);
		bytes(CTX_ELEM, (const uint8_t*)&z, 4);
	}
	void part(int p) { uint16_t v = (uint16_t)p; bytes(CTX_PART, (const uint8_t*)&v, 2); }
	void vertid(uint32_t v) { if (v < min_ref) min_ref = v; bytes(CTX_VERT, (const uint8_t*)&v, 4); }
	// chunked profile: every explicit naming of a vertex with the number of triangles seen at it so far (the "order" the
	// operation model conditions on, models.h:69-72), per component -- restart points carry the counters of the older
	// vertices their span names, so that a decoder can start there without the components before it
	struct Named { uint32_t mark, id, count, snap; };   // snap: border snapshots its component had taken before the naming
	std::vector<Named> named;
	void name(uint32_t v, uint32_t count) { if (record && !marks.empty()) named.push_back(Named{ (uint32_t)marks.size() - 1, v, count, snaps_in_component }); vertid(v); }
	// chunked profile, round 6: restart points INSIDE a component -- the cut-border at the first moment between two operations, the
	// polygon in hand complete, at which the component has coded j * snapshot_faces faces (j = 1, 2, ...): the cursors of a restart
	// point, the parts with their edge_begin flags, per element the vertex (the decoder's number) and min(triangles seen at it, 9).
	// The elements' half-edges are not part of it (cbm/decoder.h:179-197: a border edge only ever becomes the twin of a new edge).
	struct Snap { uint32_t mark, n_grp[5], n_op[8], first_vertex, first_face; std::vector<uint32_t> parts, vtx; std::vector<uint8_t> seen; };
	std::vector<Snap> snaps;
	uint32_t snapshot_faces = 0, snaps_in_component = 0;
	void numtri(int n) { if (n != 0) { uint16_t v = (uint16_t)n; bytes(CTX_NUMTRI, (const uint8_t*)&v, 2); } }   // io.h:162-165
	void reg_face(uint16_t r) { bytes(CTX_REGFACE, (const uint8_t*)&r, 2); }
	void reg_vtx(uint16_t r) { bytes(CTX_REGVTX, (const uint8_t*)&r, 2); }
	void attr_type(int l, uint8_t t) { sym(md.attr_base[l] + ATTR_TYPE, t); }
	void attr_ghist(int l, uint32_t d) { attr_type(l, A_HIST); bytes(md.attr_base[l] + ATTR_GHIST, (const uint8_t*)&d, 4); }    // io.h:99-103
	void attr_lhist(int l, uint16_t d) { attr_type(l, A_LHIST); bytes(md.attr_base[l] + ATTR_LHIST, (const uint8_t*)&d, 2); }  // io.h:104-108
};
struct SymReader {
	Models &md;
	RangeDecoder &rc;
	// chunked profile: symbols were entropy-decoded per context beforehand and are consumed from these planes
	std::vector<std::vector<uint8_t>> *planes = nullptr;
	std::vector<size_t> cursor;
	int fixed_numtri = -1;   // >= 0: numtri is not transmitted (single polygon degree)
	// chunked container, round 6: the border snapshots of the directory, checked by cbm_decode against its own state
	struct Snap { uint32_t n_grp[5], n_op[8], first_vertex, first_face, first_halfedge; std::vector<std::pair<uint32_t, uint32_t>> counters; std::vector<uint32_t> parts, vtx; std::vector<uint8_t> seen; };
	const std::vector<Snap> *snaps = nullptr;
	size_t snaps_checked = 0;
	size_t cursor_of(int slot) const { return (size_t)slot < cursor.size() ? cursor[slot] : 0; }
	SymReader(Models &m, RangeDecoder &r) : md(m), rc(r) {}
	uint32_t pop(int slot)
	{
		if (cursor.size() < planes->size()) cursor.resize(planes->size(), 0);
		if (cursor[slot] >= (*planes)[slot].size()) throw std::runtime_error("oracle: chunked stream exhausted");
		return (*planes)[slot][cursor[slot]++];
	}
	uint32_t code(int ctx)   // coder.h:154-162
	{
		FreqTable &f = md.tab[ctx];
		uint64_t l, h, t = f.total();
		uint32_t s = f.find(rc.target(t), l, h);
		rc.consume(l, h, t);
		return s;
	}
	uint32_t sym(int ctx)
	{
		if (planes) return pop(ctx);
		uint32_t s = code(ctx); md.tab[ctx].inc(s); return s;
	}
	void bytes(int ctx, uint8_t *p, int n) { for (int i = 0; i < n; ++i) p[i] = (uint8_t)sym(ctx + i); }
	uint32_t iop() { return sym(CTX_IOP); }
	uint32_t op()
	{
		if (planes) return pop(REC_OP0 + md.order_class());
		md.op_prepare(); uint32_t s = code(CTX_OP); md.op_update(s); return s;
	}
	int elem() { uint32_t z; bytes(CTX_ELEM, (uint8_t*)&z, 4); return (int)((z >> 1) ^ ((z & 1) ? 0xffffffffu : 0u)); }
	uint16_t part() { uint16_t v; bytes(CTX_PART, (uint8_t*)&v, 2); return v; }
	uint32_t vertid() { uint32_t v; bytes(CTX_VERT, (uint8_t*)&v, 4); return v; }
	uint16_t numtri() { if (fixed_numtri >= 0) return (uint16_t)fixed_numtri; uint16_t v; bytes(CTX_NUMTRI, (uint8_t*)&v, 2); return v; }
	// chunked container: planes that carry no information are not stored (one region; the PLY layout's references are all DATA)
	bool stored(int slot) const { return slot < (int)planes->size() && !(*planes)[slot].empty(); }
	uint16_t reg_face() { if (planes) return stored(CTX_REGFACE) ? (uint16_t)pop(CTX_REGFACE) : 0; uint16_t v; bytes(CTX_REGFACE, (uint8_t*)&v, 2); return v; }
	uint16_t reg_vtx() { if (planes) return stored(CTX_REGVTX) ? (uint16_t)pop(CTX_REGVTX) : 0; uint16_t v; bytes(CTX_REGVTX, (uint8_t*)&v, 2); return v; }
	uint8_t attr_type(int l) { if (planes && !stored(md.attr_base[l] + ATTR_TYPE)) return A_DATA; return (uint8_t)sym(md.attr_base[l] + ATTR_TYPE); }
	uint32_t attr_ghist(int l) { uint32_t v; bytes(md.attr_base[l] + ATTR_GHIST, (uint8_t*)&v, 4); return v; }
	uint16_t attr_lhist(int l) { uint16_t v; bytes(md.attr_base[l] + ATTR_LHIST, (uint8_t*)&v, 2); return v; }
};

// ------------------------------------------------------------------------------------------------
// cut-border data structure (cbm/cutborder.h:49-333)
// ------------------------------------------------------------------------------------------------
static const uint32_t NOVTX = 0xffffffffu;
struct Elem { uint32_t v = NOVTX; he_t a = 0; };
struct Part { std::list<Elem> el; bool edge_begin = true; size_t num_edges() const { return el.size() - (edge_begin ? 0 : 1); } };

struct CutBorder {
	std::deque<Part> parts;          // top of the stack = back()
	Elem *first = nullptr, *second = nullptr;
	std::vector<uint8_t> on_border;  // occurrence counter per vertex (cutborder.h:69,98-112)
	explicit CutBorder(uint32_t nv) : on_border(nv, 0) {}
	Part &top() { return parts.back(); }
	bool empty() const { return parts.empty(); }
	void act(uint32_t v) { ++on_border[v]; }
	void deact(uint32_t v) { --on_border[v]; }

	void start(Elem a, Elem b, Elem c)   // cutborder.h:157-164
	{
		parts.emplace_back();
		Part &p = top();
		p.el.push_back(a); act(a.v);
		p.el.push_back(b); act(b.v);
		p.el.push_back(c); act(c.v);
	}
	void new_vertex(Elem e)   // :166-172
	{
		Part &p = top();
		first = &p.el.back();
		p.el.push_back(e); act(e.v);
		second = &p.el.back();
	}
	bool is_tri() { Part &p = top(); return p.num_edges() == 3 && p.el.size() == 3; }
	Op border()   // :217-248
	{
		Part &p = top();
		if (p.num_edges() == 1) {
			auto it = p.el.begin();
			deact((it++)->v);
			deact((it++)->v);
			parts.pop_back();
			return OP_BORDER;
		}
		Elem endv = p.el.back();
		bool rename = !p.edge_begin;
		deact(p.el.back().v);
		p.el.pop_back();
		if (!p.edge_begin) { deact(p.el.front().v); p.el.pop_front(); }
		p.el.push_front(endv); act(endv.v);
		p.edge_begin = false;
		return rename ? OP_CONNFWD : OP_BORDER;
	}
	Elem connect_fwd(Op &op)   // :173-195
	{
		Part &p = top();
		Elem d = *std::next(p.el.begin());
		if (!p.edge_begin) { op = border(); return Elem(); }
		if (is_tri()) {
			auto it = p.el.begin();
			deact((it++)->v); deact((it++)->v); deact((it++)->v);
			parts.pop_back();
			op = OP_CLOSE;
		} else {
			deact(p.el.front().v);
			p.el.pop_front();
			op = OP_CONNFWD;
			first = &top().el.back();
		}
		return d;
	}
	Elem connect_bwd(Op &op)   // :196-209
	{
		Part &p = top();
		deact(p.el.back().v);
		p.el.pop_back();
		op = OP_CONNBWD;
		first = &p.el.back();
		return p.el.back();
	}
	std::list<Elem>::iterator at(int i, int p = 0)   // :114-123
	{
		Part &pt = parts[parts.size() - 1 - p];
		if (i > 0) return std::next(pt.el.begin(), i - 1);
		return std::prev(pt.el.end(), -i + 1);
	}
	// :124-155 two-ended search through the stack of parts, front hit tested first
	std::list<Elem>::iterator locate(uint32_t v, int &i, int &p)
	{
		auto part = parts.rbegin();
		auto fw = part->el.begin();
		auto bw = std::prev(part->el.end());
		i = 0; p = 0;
		for (;;) {
			if (fw->v == v) { ++i; return fw; }
			if (bw->v == v) { i = -i; return bw; }
			if (bw == fw || std::next(bw) == fw) {
				++p; ++part;
				fw = part->el.begin();
				bw = std::prev(part->el.end());
				i = 0;
			} else { ++fw; --bw; ++i; }
		}
	}
	Elem split(std::list<Elem>::iterator it)   // :250-268
	{
		Part &p = top();
		Elem gate = p.el.back();
		deact(gate.v);
		p.el.pop_back();
		parts.emplace_back();
		Part &np = top();
		Part &old = parts[parts.size() - 2];
		np.el.splice(np.el.begin(), old.el, old.el.begin(), it);
		old.el.push_back(gate); act(gate.v);
		np.el.push_back(*it); act(it->v);
		std::swap(old.edge_begin, np.edge_begin);
		second = &np.el.back();
		first = &old.el.back();
		return *it;
	}
	Elem unite(std::list<Elem>::iterator it, int p)   // :274-297
	{
		Part &cur = top();
		Elem gate = cur.el.back();
		deact(gate.v);
		cur.el.pop_back();
		size_t oi = parts.size() - 1 - p;
		Part &other = parts[oi];
		cur.el.push_back(gate); act(gate.v);
		first = &cur.el.back();
		cur.el.splice(cur.el.end(), other.el, it, other.el.end());
		cur.el.splice(cur.el.end(), other.el, other.el.begin(), other.el.end());
		cur.el.push_back(*it); act(it->v);
		second = &cur.el.back();
		Elem hit = *it;
		// drop the emptied part, keeping the order of the others (the reference bubbles it to the top)
		for (size_t k = oi; k + 1 < parts.size(); ++k) {
			parts[k].el.swap(parts[k + 1].el);
			std::swap(parts[k].edge_begin, parts[k + 1].edge_begin);
		}
		parts.pop_back();
		return hit;
	}
	// :303-332 (encoder side classification)
	bool find_and_update(uint32_t v, int &i, int &p, Op &op)
	{
		if (on_border[v] == 0) return false;
		auto it = locate(v, i, p);
		if (p > 0) { op = OP_UNION; unite(it, p); }
		else {
			Part &pt = top();
			if (pt.edge_begin && std::next(pt.el.begin())->v == v) connect_fwd(op);
			else if (std::next(pt.el.rbegin())->v == v) connect_bwd(op);
			else { op = OP_SPLIT; split(it); }
		}
		return true;
	}
};

// ------------------------------------------------------------------------------------------------
// attribute prediction shared by encoder and decoder (formats/hry/attrcode.h:83-289)
// ------------------------------------------------------------------------------------------------
struct Cand { uint32_t v0, v1, vo; };

struct Predictor {
	Mesh &m;
	std::vector<bool> vdone;
	std::vector<Cand> cands;
	// identity maps for the encoder; for the decoder vertex -> attribute index assigned in decode order
	std::vector<uint32_t> *vtx_attr = nullptr;
	explicit Predictor(Mesh &mesh) : m(mesh), vdone(mesh.nv, false) {}

	void offer(uint32_t v0, uint32_t v1, uint32_t vo)   // attrcode.h:117-134 (single region: region test always passes)
	{
		if (!vdone[v0] || !vdone[v1] || !vdone[vo]) return;
		cands.push_back(Cand{ v0, v1, vo });
	}
	void face_parallelograms(he_t ein)   // attrcode.h:155-171
	{
		he_t e = ein;
		int d = m.deg(m.eface[e]);
		if (d == 3) {
			e = m.next(e);
			he_t t = m.twin[e];
			if (t == e) return;
			e = m.next(m.next(t));
			offer(m.org[t], m.dest(t), m.org[e]);
			return;
		}
		he_t e0 = m.next(e), e1 = m.prev(e);
		offer(m.org[e0], m.org[e1], m.dest(e0));
		if (d > 4) offer(m.org[e0], m.org[e1], m.org[m.prev(e)]);
	}
	void fan(he_t ein)   // attrcode.h:83-106 (TFAN_IT): forward until border, then backward sweep
	{
		he_t e = ein, t;
		for (;;) {
			face_parallelograms(e);
			t = m.twin[e];
			if (t == e) break;          // border: go backward
			e = m.next(t);
			if (e == ein) return;       // full circle
		}
		e = m.prev(ein);
		t = m.twin[e];
		if (e == t) return;
		e = t;
		do {
			face_parallelograms(e);
			e = m.prev(e);
			t = m.twin[e];
			if (e == t) break;
			e = t;
		} while (e != ein);
	}
	void collect_vertex(he_t e)   // attrcode.h:209-218
	{
		cands.clear();
		fan(e);
		vdone[m.org[e]] = true;
	}
	uint32_t attr_of(uint32_t v) const { return vtx_attr ? (*vtx_attr)[v] : v; }

	// attrcode.h:182-208: mean of the candidates, then (floats) the candidate nearest to the mean
	template <typename T> T predict_component(const List &L, int c, int q) const
	{
		typedef typename big_of<T>::type B;
		size_t n = cands.size();
		if (n == 0) return T(0);
		int off = L.fmt.off[c];
		B acc = 0;
		std::vector<T> pv(n);
		for (size_t k = 0; k < n; ++k) {
			T a = ld<T>(L.rec(attr_of(cands[k].v0)) + off), b = ld<T>(L.rec(attr_of(cands[k].v1)) + off), o = ld<T>(L.rec(attr_of(cands[k].vo)) + off);
			pv[k] = paral_predict<T>(a, b, o, q, std::is_floating_point<T>());
			acc = acc + (B)pv[k];
		}
		acc = div_round(acc, (B)n);
		T avg = (T)acc;
		if (!std::is_floating_point<T>::value) return avg;
		T res = std::numeric_limits<T>::max();
		for (size_t k = 0; k < n; ++k) {
			T rd = avg > res ? avg - res : res - avg;
			T pd = avg > pv[k] ? avg - pv[k] : pv[k] - avg;
			res = rd < pd ? res : pv[k];
		}
		return res;
	}
};

// ------------------------------------------------------------------------------------------------
// header (formats/hry/writer.cc:104-198, reader.cc:60-177)
// ------------------------------------------------------------------------------------------------
struct ByteWriter {
	std::vector<uint8_t> &o;
	template <typename T> void put(T v) { const uint8_t *p = (const uint8_t*)&v; o.insert(o.end(), p, p + sizeof(T)); }
	void raw(const void *p, size_t n) { o.insert(o.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};
static void write_header(const Mesh &m, std::vector<uint8_t> &out, int ver_minor = 1)
{
	ByteWriter w{ out };
	const uint8_t magic[6] = { 0xfa, 0xff, 0xaf, 0xaf, 0, (uint8_t)ver_minor };   // big-endian magic, version 0.1 (common.h:15-16)
	w.raw(magic, 6);
	const bool sh = m.is_shard();   // a shard announces the whole mesh
	w.put<uint32_t>(sh ? m.g_nv : m.nv); w.put<uint32_t>(sh ? m.g_nf : m.nf); w.put<uint32_t>(sh ? m.g_ne : m.num_edge());
	std::vector<char> seen(m.lists.size(), 1);
	if (m.bind.on) {   // writer.cc:124-151: regions with the lists bound to them; lists no region names are not written
		const Mesh::Bind &b = m.bind;
		seen.assign(m.lists.size(), 0);
		w.put<uint16_t>((uint16_t)b.nregs_face()); w.put<uint16_t>((uint16_t)b.nregs_vtx());
		for (int r = 0; r < b.nregs_face(); ++r) {
			w.put<uint16_t>((uint16_t)b.nfacelists(r)); w.put<uint16_t>((uint16_t)b.ncornerlists(r));
			for (int a = 0; a < b.nfacelists(r); ++a) { seen[b.facelist(r, a)] = 1; w.put<uint16_t>((uint16_t)b.facelist(r, a)); }
			for (int a = 0; a < b.ncornerlists(r); ++a) { seen[b.cornerlist(r, a)] = 1; w.put<uint16_t>((uint16_t)b.cornerlist(r, a)); }
		}
		for (int r = 0; r < b.nregs_vtx(); ++r) {
			w.put<uint16_t>((uint16_t)b.nvtxlists(r));
			for (int a = 0; a < b.nvtxlists(r); ++a) { seen[b.vtxlist(r, a)] = 1; w.put<uint16_t>((uint16_t)b.vtxlist(r, a)); }
		}
	} else {
		w.put<uint16_t>(1); w.put<uint16_t>(1);                 // one face region, one vertex region
		w.put<uint16_t>(1); w.put<uint16_t>(0); w.put<uint16_t>(0);   // face region: 1 face list (id 0), 0 corner lists
		w.put<uint16_t>(1); w.put<uint16_t>(1);                 // vertex region: 1 list (id 1)
	}
	for (size_t i = 0; i < m.lists.size(); ++i) {
		const List &L = m.lists[i];
		if (!seen[i]) continue;
		w.put<uint32_t>(sh ? (L.target == TG_FACE ? m.g_nf : m.g_nv) : L.count);
		w.put<uint16_t>((uint16_t)L.fmt.size());
		for (int j = 0; j < L.fmt.size(); ++j) { w.put<uint8_t>(L.fmt.type[j]); w.put<uint8_t>((uint8_t)L.fmt.quant[j]); }
		w.put<uint16_t>((uint16_t)L.interps.size());
		for (int j = 0; j < L.interps.size(); ++j) {
			w.put<uint16_t>((uint16_t)L.interps.len[j]);
			if (j >= I_OTHER) {
				const std::string &nm = L.interps.names[j - I_OTHER];
				w.put<uint32_t>((uint32_t)nm.size());
				w.raw(nm.data(), nm.size());
			}
		}
		w.raw(L.bmin.data(), L.bmin.size());
		w.raw(L.bmax.data(), L.bmax.size());
	}
	uint16_t cnt = 0;
	for (char c : m.have_deg) cnt += c ? 1 : 0;
	w.put<uint16_t>(cnt);
	for (size_t d = 0; d < m.have_deg.size(); ++d) if (m.have_deg[d]) w.put<uint16_t>((uint16_t)d);
}

struct ByteReader {
	const uint8_t *p, *end;
	template <typename T> T get() { need(sizeof(T)); T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
	void raw(void *d, size_t n) { need(n); memcpy(d, p, n); p += n; }
	void need(size_t n) { if ((size_t)(end - p) < n) throw std::runtime_error("oracle: truncated header"); }
};
static int dequant_bytes(const Fmt &f) { return f.bytes(); }
static void read_header(ByteReader &r, Mesh &m, int want_minor = 1, uint32_t *declared_ne = nullptr)
{
	uint8_t magic[6];
	r.raw(magic, 6);
	if (magic[0] != 0xfa || magic[1] != 0xff || magic[2] != 0xaf || magic[3] != 0xaf) throw std::runtime_error("Invalid magic number");
	if (magic[4] != 0 || magic[5] != want_minor) throw std::runtime_error("File format version incompatible to decoder format version 0.1");
	m.nv = r.get<uint32_t>(); m.nf = r.get<uint32_t>();
	{ uint32_t ne = r.get<uint32_t>(); if (declared_ne) *declared_ne = ne; }
	uint16_t nrf = r.get<uint16_t>(), nrv = r.get<uint16_t>();
	std::vector<int> targets;
	auto mark = [&](uint16_t b, int t) { if (b >= targets.size()) targets.resize(b + 1, TG_NONE); targets[b] = t; return b; };
	Mesh::Bind b;   // reader.cc:86-126
	for (int i = 0; i < nrf; ++i) {
		uint16_t nbf = r.get<uint16_t>(), nbc = r.get<uint16_t>();
		int reg = b.add_face_region(nbf, nbc);
		b.nb_face = std::max<int>(b.nb_face, nbf); b.nb_corner = std::max<int>(b.nb_corner, nbc);
		for (int a = 0; a < nbf; ++a) b.reg_facelist[b.off_facelist[reg] + a] = mark(r.get<uint16_t>(), TG_FACE);
		for (int a = 0; a < nbc; ++a) b.reg_cornerlist[b.off_cornerlist[reg] + a] = mark(r.get<uint16_t>(), TG_CORNER);
	}
	for (int i = 0; i < nrv; ++i) {
		uint16_t nbv = r.get<uint16_t>();
		int reg = b.add_vtx_region(nbv);
		b.nb_vtx = std::max<int>(b.nb_vtx, nbv);
		for (int a = 0; a < nbv; ++a) b.reg_vtxlist[b.off_vtxlist[reg] + a] = mark(r.get<uint16_t>(), TG_VTX);
	}
	const bool ply_layout = nrf == 1 && nrv == 1 && b.nfacelists(0) == 1 && b.ncornerlists(0) == 0 && b.nvtxlists(0) == 1 &&
	                        b.facelist(0, 0) == 0 && b.vtxlist(0, 0) == 1;
	if (!ply_layout) {
		if (nrf > 128 || nrv > 128) throw std::runtime_error("oracle: more than 128 regions overflow the reference's model seeding (model.h:49-55)");
		b.on = true;
		b.face_reg.assign(m.nf, 0); b.vtx_reg.assign(m.nv, 0);
		b.face_attr.assign((size_t)m.nf * b.nb_face, 0); b.vtx_attr.assign((size_t)m.nv * b.nb_vtx, 0);
		m.bind = std::move(b);   // corner slots follow the connectivity
	}
	for (size_t i = 0; i < targets.size(); ++i) {
		List L;
		L.target = targets[i];
		if (L.target == TG_NONE) { m.lists.push_back(std::move(L)); continue; }   // reader.cc:128-166: nothing stored for it
		L.count = r.get<uint32_t>();
		uint16_t nf = r.get<uint16_t>();
		for (int j = 0; j < nf; ++j) { uint8_t t = r.get<uint8_t>(), q = r.get<uint8_t>(); if (t >= T_NONE) throw std::runtime_error("oracle: bad type"); L.fmt.add((Type)t, q); }
		uint16_t ni = r.get<uint16_t>();
		int off = 0;
		for (int j = 0; j < ni; ++j) {
			uint16_t len = r.get<uint16_t>();
			for (int k = 0; k < len; ++k) L.interps.append(j, off + k);
			off += len;
			if (j >= I_OTHER) {
				uint32_t sl = r.get<uint32_t>();
				std::string nm(sl, '\0');
				r.raw(&nm[0], sl);
				L.interps.describe(j, nm);
			}
		}
		L.data.assign((size_t)L.count * L.fmt.bytes(), 0);
		L.bmin.resize(dequant_bytes(L.fmt)); L.bmax.resize(dequant_bytes(L.fmt));
		r.raw(L.bmin.data(), L.bmin.size());
		r.raw(L.bmax.data(), L.bmax.size());
		m.lists.push_back(std::move(L));
	}
	uint16_t cnt = r.get<uint16_t>();
	for (int i = 0; i < cnt; ++i) {
		uint16_t d = r.get<uint16_t>();
		if (d >= m.have_deg.size()) m.have_deg.resize(d + 1, 0);
		m.have_deg[d] = 1;
	}
}

// ------------------------------------------------------------------------------------------------
// encoder: cut-border walk + attribute pass (cbm/encoder.h:54-217, attrcode.h:291-417, writer.cc:200-214)
// ------------------------------------------------------------------------------------------------
struct Result {
	std::vector<uint8_t> bytes;
	size_t header_size = 0;
	std::vector<ho_sym> trace;
	std::vector<uint32_t> order_v, order_f;
};

// Start-face choice: face 0 if unvisited, else the first unvisited face in the iteration order of a
// std::unordered_set<uint32_t> that received 0..F-1 in order (writer.cc:28-46).
struct FacePool {
	std::vector<uint32_t> iter_order;
	std::vector<uint8_t> gone;
	size_t cursor = 0, left;
	bool seeded = false;
	explicit FacePool(uint32_t nf, const std::vector<uint32_t> &seeds = std::vector<uint32_t>()) : gone(nf, 0), left(nf)
	{
		// a shard brings the start faces of its components along, in the coding order they have in the whole mesh (the rule
		// below depends on the face count of the whole mesh, which a shard does not have)
		if (!seeds.empty()) { iter_order = seeds; seeded = true; return; }
		std::unordered_set<uint32_t> s;
		for (uint32_t i = 0; i < nf; ++i) s.insert(i);
		iter_order.assign(s.begin(), s.end());
	}
	void take(uint32_t f) { gone[f] = 1; --left; }
	uint32_t choose()
	{
		uint32_t f;
		if (!seeded && !gone[0]) f = 0;
		else {
			while (gone[iter_order[cursor]]) ++cursor;
			f = iter_order[cursor];
		}
		take(f);
		return f;
	}
};

static void cbm_encode(Mesh &m, SymWriter &wr, std::vector<uint32_t> &order_v, std::vector<uint32_t> &order_f)
{
	CutBorder cb(m.nv);
	FacePool pool(m.num_face(), m.seeds);
	std::vector<uint32_t> perm(m.nv, NOVTX);
	std::vector<uint16_t> seen(m.nv, 0);
	uint32_t next_id = 0;
	auto mapped = [&](uint32_t v) { return perm[v] != NOVTX; };
	auto add_vtx = [&](he_t e) { order_v.push_back(e); perm[m.org[e]] = next_id++; };

	int curtri = 0, ntri = 0;
	he_t e0 = 0, e1 = 0, e2 = 0;
	uint32_t f = 0;
	do {
		curtri = 0;
		wr.mark_component(next_id, (uint32_t)order_f.size());
		f = pool.choose();
		e0 = m.foff[f]; e1 = m.next(e0); e2 = m.next(e1);
		uint32_t a = m.org[e0], b = m.org[e1], c = m.org[e2];
		bool m0 = mapped(a), m1 = mapped(b), m2 = mapped(c);
		ntri = m.deg(f) - 2;
		// encoder.h:76-123; ids are written as transmitted indices, new vertices recorded in this order
		if (m0 && m1 && m2) { wr.iop(IOP_TRI111); wr.name(perm[a], seen[a]); wr.name(perm[b], seen[b]); wr.name(perm[c], seen[c]); wr.numtri(ntri); }
		else if (m0 && m1) { wr.iop(IOP_TRI110); wr.name(perm[a], seen[a]); wr.name(perm[b], seen[b]); wr.numtri(ntri); add_vtx(e2); }
		else if (m1 && m2) { wr.iop(IOP_TRI011); wr.name(perm[b], seen[b]); wr.name(perm[c], seen[c]); wr.numtri(ntri); add_vtx(e0); }
		else if (m2 && m0) { wr.iop(IOP_TRI101); wr.name(perm[c], seen[c]); wr.name(perm[a], seen[a]); wr.numtri(ntri); add_vtx(e1); }
		else if (m0) { wr.iop(IOP_TRI100); wr.name(perm[a], seen[a]); wr.numtri(ntri); add_vtx(e1); add_vtx(e2); }
		else if (m1) { wr.iop(IOP_TRI010); wr.name(perm[b], seen[b]); wr.numtri(ntri); add_vtx(e2); add_vtx(e0); }
		else if (m2) { wr.iop(IOP_TRI001); wr.name(perm[c], seen[c]); wr.numtri(ntri); add_vtx(e0); add_vtx(e1); }
		else { wr.iop(IOP_INIT); wr.numtri(ntri); add_vtx(e0); add_vtx(e1); add_vtx(e2); }
		order_f.push_back(e0);
		++seen[a]; ++seen[b]; ++seen[c];
		cb.start(Elem{ a, e0 }, Elem{ b, e1 }, Elem{ c, e2 });
		++curtri;
		const size_t comp_face0 = order_f.size() - 1;
		uint64_t next_snap = wr.record && wr.snapshot_faces ? wr.snapshot_faces : ~0ull;

		while (!cb.empty()) {
			if (curtri == ntri && order_f.size() - comp_face0 >= next_snap) {   // a border snapshot (SymWriter::Snap)
				next_snap += wr.snapshot_faces;
				SymWriter::Snap sn;
				sn.mark = (uint32_t)wr.marks.size() - 1;
				const auto &rec = *wr.record;
				sn.n_grp[0] = (uint32_t)rec[CTX_IOP].size(); sn.n_grp[1] = (uint32_t)rec[CTX_ELEM].size(); sn.n_grp[2] = (uint32_t)rec[CTX_PART].size();
				sn.n_grp[3] = (uint32_t)rec[CTX_VERT].size(); sn.n_grp[4] = (uint32_t)rec[CTX_NUMTRI].size();
				for (int i = 0; i < 8; ++i) sn.n_op[i] = (uint32_t)rec[REC_OP0 + i].size();
				sn.first_vertex = next_id; sn.first_face = (uint32_t)order_f.size();
				for (const Part &q : cb.parts) {   // bottom of the stack first; elements front (the gate's head side) to back (the gate)
					sn.parts.push_back((uint32_t)q.el.size() << 1 | (q.edge_begin ? 1u : 0u));
					for (const Elem &el : q.el) { sn.vtx.push_back(perm[el.v]); sn.seen.push_back((uint8_t)std::min<uint32_t>(seen[el.v], 9u)); }
				}
				wr.snaps.push_back(std::move(sn));
				++wr.snaps_in_component;
			}
			Part &pt = cb.top();
			Elem g0 = pt.el.back(), g1 = pt.el.front();
			he_t gate = g0.a;
			he_t gateprev = std::next(pt.el.rbegin())->a;
			he_t gatenext = pt.el.front().a;
			bool seq_first = curtri == ntri;
			bool valid = true;
			if (seq_first) {   // writer.cc:48-58: cross the gate unless it is a border or its neighbour is consumed
				he_t t = m.twin[gate];
				if (t == gate || pool.gone[m.eface[t]]) valid = false;
				else { pool.take(m.eface[t]); e0 = t; }
			}
			wr.md.order = seen[g1.v];
			if (seq_first && !valid) {
				Op bop = cb.border();
				if (m.twin[gate] != gate) m.merge(gate, gate);   // one-sided: the old twin keeps pointing here (writer.cc:81-84)
				wr.op(bop);
				continue;
			}
			if (seq_first) {
				curtri = 0;
				f = m.eface[e0];
				ntri = m.deg(f) - 2;
				e1 = m.next(e0);
			} else e1 = m.next(e1);
			e2 = m.next(e1);
			uint32_t v2 = m.org[e2];
			bool seq_last = curtri + 1 == ntri;
			int nt = seq_first ? ntri : 0;
			if (!mapped(v2)) {
				cb.new_vertex(Elem{ v2, 0 });
				cb.first->a = e1; cb.second->a = e2;
				wr.op(OP_NEWVTX); wr.numtri(nt);
				add_vtx(e2);
			} else {
				int i, p;
				Op op;
				if (!cb.find_and_update(v2, i, p, op)) {
					cb.new_vertex(Elem{ v2, 0 });
					cb.first->a = e1; cb.second->a = e2;
					wr.op(OP_NM); wr.name(perm[v2], seen[v2]); wr.numtri(nt);
				} else if (op == OP_UNION) {
					wr.op(OP_UNION); wr.elem(i); wr.part(p); wr.numtri(nt);
					cb.first->a = e1; cb.second->a = e2;
				} else if (op == OP_CONNFWD || op == OP_CLOSE) {
					if (seq_last && m.twin[gatenext] != e2) m.merge(gatenext, e2);
					if (op == OP_CLOSE && m.twin[gateprev] != e1) m.merge(gateprev, e1);
					wr.op(OP_CONNFWD); wr.numtri(nt);
					if (op == OP_CONNFWD) cb.first->a = e1;
				} else if (op == OP_CONNBWD) {
					if (m.twin[gateprev] != e1) m.merge(gateprev, e1);
					wr.op(OP_CONNBWD); wr.numtri(nt);
					cb.first->a = e2;
				} else {
					wr.op(OP_SPLIT); wr.elem(i); wr.numtri(nt);
					cb.first->a = e1; cb.second->a = e2;
				}
			}
			++seen[g0.v]; ++seen[g1.v]; ++seen[v2];
			if (seq_first) order_f.push_back(e0);
			++curtri;
		}
	} while (pool.left != 0);
	wr.iop(IOP_EOM);
}

static void encode_attrs(Mesh &m, SymWriter &wr, const std::vector<uint32_t> &order_v, const std::vector<uint32_t> &order_f)
{
	Predictor pr(m);
	const int LV = 1, LF = 0;
	std::vector<uint8_t> resid;
	// vertices in traversal order (attrcode.h:321-344,398-403)
	{
		List &L = m.lists[LV];
		resid.assign(std::max(L.fmt.bytes(), 1), 0);
		for (he_t e : order_v) {
			uint32_t v = m.org[e];
			pr.collect_vertex(e);
			wr.reg_vtx(0);
			int ctx = wr.md.attr_base[LV] + ATTR_DATA;
			for (int c = 0; c < L.fmt.size(); ++c) {
				with_type(L.fmt.stype[c], [&](auto tag) {
					typedef decltype(tag) T;
					T pred = pr.predict_component<T>(L, c, L.fmt.quant[c]);
					T raw = ld<T>(L.rec(v) + L.fmt.off[c]);
					st<T>(resid.data() + L.fmt.off[c], fold_residual<T>(raw, pred, L.fmt.quant[c], std::is_floating_point<T>()));
				});
			}
			wr.attr_type(LV, A_DATA);   // io.h:90-94
			for (int c = 0; c < L.fmt.size(); ++c) {
				int nb = TSIZE[L.fmt.stype[c]];
				wr.bytes(ctx, resid.data() + L.fmt.off[c], nb);
				ctx += nb;
			}
		}
	}
	// faces in traversal order; face prediction never has candidates (attrcode.h:227-254)
	{
		List &L = m.lists[LF];
		resid.assign(std::max(L.fmt.bytes(), 1), 0);
		for (he_t e : order_f) {
			uint32_t f = m.eface[e];
			wr.reg_face(0);
			int ctx = wr.md.attr_base[LF] + ATTR_DATA;
			for (int c = 0; c < L.fmt.size(); ++c) {
				with_type(L.fmt.stype[c], [&](auto tag) {
					typedef decltype(tag) T;
					T raw = ld<T>(L.rec(f) + L.fmt.off[c]);
					st<T>(resid.data() + L.fmt.off[c], fold_residual<T>(raw, T(0), L.fmt.quant[c], std::is_floating_point<T>()));
				});
			}
			wr.attr_type(LF, A_DATA);
			for (int c = 0; c < L.fmt.size(); ++c) {
				int nb = TSIZE[L.fmt.stype[c]];
				wr.bytes(ctx, resid.data() + L.fmt.off[c], nb);
				ctx += nb;
			}
		}
	}
}

static void check_general(const Mesh &m);
static void encode_attrs_general(Mesh &m, SymWriter &wr, const std::vector<uint32_t> &order_v, const std::vector<uint32_t> &order_f);

static void check_supported(const Mesh &m)
{
	if (m.bind.on) throw std::runtime_error("oracle: this path takes the PLY layout only (list 0 = face attributes, list 1 = vertex attributes)");
	if (m.lists.size() != 2 || m.lists[0].target != TG_FACE || m.lists[1].target != TG_VTX)
		throw std::runtime_error("oracle: mesh must have list 0 = face attributes and list 1 = vertex attributes");
	if (m.lists[0].count != m.nf || m.lists[1].count != m.nv) throw std::runtime_error("oracle: list sizes must equal element counts");
}

static Result *encode(Mesh &m, bool trace)
{
	if (m.bind.on) check_general(m); else check_supported(m);
	Result *res = new Result();
	try {
		write_header(m, res->bytes);
		res->header_size = res->bytes.size();
		RangeEncoder rc(res->bytes);
		Models md(m);
		SymWriter wr(md, rc, trace ? &res->trace : nullptr);
		cbm_encode(m, wr, res->order_v, res->order_f);
		if (m.bind.on) encode_attrs_general(m, wr, res->order_v, res->order_f);
		else encode_attrs(m, wr, res->order_v, res->order_f);
		rc.finish();
	} catch (...) { delete res; throw; }
	return res;
}

// ------------------------------------------------------------------------------------------------
// decoder (cbm/decoder.h:27-211, attrcode.h:420-551, reader.cc:179-193)
// ------------------------------------------------------------------------------------------------
static void cbm_decode(Mesh &m, SymReader &rd, std::vector<uint32_t> &order_v)
{
	CutBorder cb(m.nv);
	std::vector<uint16_t> seen(m.nv, 0);
	uint32_t next_id = 0;
	int curtri = 0, ntri = 0;
	he_t e0 = 0, e1 = 0, e2 = 0;
	uint32_t f = 0;
	for (;;) {
		uint32_t iop = rd.iop();
		if (iop == IOP_EOM) break;
		uint32_t a = 0, b = 0, c = 0;
		curtri = 0;
		switch (iop) {   // decoder.h:46-77
		case IOP_INIT: a = next_id++; b = next_id++; c = next_id++; break;
		case IOP_TRI100: a = rd.vertid(); b = next_id++; c = next_id++; break;
		case IOP_TRI010: c = next_id++; b = rd.vertid(); a = next_id++; break;
		case IOP_TRI001: a = next_id++; b = next_id++; c = rd.vertid(); break;
		case IOP_TRI110: a = rd.vertid(); b = rd.vertid(); c = next_id++; break;
		case IOP_TRI101: c = rd.vertid(); b = next_id++; a = rd.vertid(); break;
		case IOP_TRI011: a = next_id++; b = rd.vertid(); c = rd.vertid(); break;
		case IOP_TRI111: a = rd.vertid(); b = rd.vertid(); c = rd.vertid(); break;
		default: throw std::runtime_error("oracle: bad init op");
		}
		if (a >= m.nv || b >= m.nv || c >= m.nv) throw std::runtime_error("oracle: corrupt stream (vertex id)");
		ntri = rd.numtri();
		++seen[a]; ++seen[b]; ++seen[c];
		f = m.add_face(ntri + 2);
		e0 = m.foff[f]; e1 = m.next(e0); e2 = m.next(e1);
		m.set_org(e0, a); m.set_org(e1, b); m.set_org(e2, c);
		++curtri;
		switch (iop) {   // decoder.h:86-110
		case IOP_INIT: order_v.push_back(e0); order_v.push_back(e1); order_v.push_back(e2); break;
		case IOP_TRI100: order_v.push_back(e1); order_v.push_back(e2); break;
		case IOP_TRI010: order_v.push_back(e2); order_v.push_back(e0); break;
		case IOP_TRI001: order_v.push_back(e0); order_v.push_back(e1); break;
		case IOP_TRI110: order_v.push_back(e2); break;
		case IOP_TRI101: order_v.push_back(e1); break;
		case IOP_TRI011: order_v.push_back(e0); break;
		default: break;
		}
		cb.start(Elem{ a, e0 }, Elem{ b, e1 }, Elem{ c, e2 });

		while (!cb.empty()) {
			// a border snapshot of the directory that claims this moment -- between two operations, the polygon in hand complete, this
			// many faces made (the first such moment) -- must describe exactly the state this replay is in: cursors, next vertex / face /
			// half-edge, parts, vertices, triangle counts; and the counters listed with it must be those of vertices off the border
			if (rd.snaps && rd.snaps_checked < rd.snaps->size() && curtri == ntri && m.num_face() >= (*rd.snaps)[rd.snaps_checked].first_face) {
				const SymReader::Snap &S = (*rd.snaps)[rd.snaps_checked++];
				bool ok = m.num_face() == S.first_face && next_id == S.first_vertex && m.num_edge() == S.first_halfedge;
				ok = ok && rd.cursor_of(CTX_IOP) == S.n_grp[0] && rd.cursor_of(CTX_ELEM) == S.n_grp[1] && rd.cursor_of(CTX_PART) == S.n_grp[2] && rd.cursor_of(CTX_VERT) == S.n_grp[3];
				if (rd.fixed_numtri < 0) ok = ok && rd.cursor_of(CTX_NUMTRI) == S.n_grp[4];
				for (int i = 0; i < 8; ++i) ok = ok && rd.cursor_of(REC_OP0 + i) == S.n_op[i];
				std::vector<uint32_t> parts, vtx; std::vector<uint8_t> sn;
				for (const Part &q : cb.parts) {
					parts.push_back((uint32_t)q.el.size() << 1 | (q.edge_begin ? 1u : 0u));
					for (const Elem &el : q.el) { vtx.push_back(el.v); sn.push_back((uint8_t)std::min<uint32_t>(seen[el.v], 9u)); }
				}
				ok = ok && parts == S.parts && vtx == S.vtx && sn == S.seen;
				for (const auto &c : S.counters) ok = ok && c.first < S.first_vertex && seen[c.first] == c.second && std::find(vtx.begin(), vtx.end(), c.first) == vtx.end();
				if (!ok) throw std::runtime_error("oracle: a border snapshot of the directory does not match the replay");
			}
			Part &pt = cb.top();
			Elem g0 = pt.el.back(), g1 = pt.el.front();
			he_t gate = g0.a;
			he_t gateprev = std::next(pt.el.rbegin())->a;
			he_t gatenext = pt.el.front().a;
			rd.md.order = seen[g1.v];
			Op op = (Op)rd.op(), realop = op;
			bool seq_first = curtri == ntri;
			Elem v2;
			switch (op) {   // decoder.h:133-166
			case OP_CONNFWD: v2 = cb.connect_fwd(realop); break;
			case OP_CONNBWD: v2 = cb.connect_bwd(realop); break;
			case OP_SPLIT: { int i = rd.elem(); v2 = cb.split(cb.at(i)); break; }
			case OP_UNION: { int i = rd.elem(); int p = rd.part(); v2 = cb.unite(cb.at(i, p), p); break; }
			case OP_NEWVTX: v2 = Elem{ next_id++, 0 }; cb.new_vertex(v2); break;
			case OP_NM: v2 = Elem{ rd.vertid(), 0 }; cb.new_vertex(v2); break;
			case OP_BORDER: cb.border(); v2 = Elem(); break;
			default: throw std::runtime_error("oracle: bad op");
			}
			if (v2.v == NOVTX) continue;
			if (v2.v >= m.nv) throw std::runtime_error("oracle: corrupt stream (vertex id)");
			if (seq_first) {
				ntri = rd.numtri();
				curtri = 0;
				f = m.add_face(ntri + 2);
				e0 = m.foff[f]; e1 = m.next(e0); e2 = m.next(e1);
				m.set_org(e0, g1.v); m.set_org(e1, g0.v); m.set_org(e2, v2.v);
			} else {
				e1 = m.next(e1);
				e2 = m.next(e1);
				m.set_org(e2, v2.v);
			}
			bool seq_last = curtri + 1 == ntri;
			switch (realop) {   // decoder.h:182-197
			case OP_CONNFWD: cb.first->a = e1; break;
			case OP_CONNBWD: cb.first->a = e2; break;
			case OP_SPLIT: case OP_UNION: case OP_NEWVTX: case OP_NM: cb.first->a = e1; cb.second->a = e2; break;
			default: break;
			}
			++seen[g0.v]; ++seen[g1.v]; ++seen[v2.v];
			if (op == OP_NEWVTX) order_v.push_back(m.foff[f] + curtri + 2);
			++curtri;
			if (seq_first) m.merge(gate, e0);
			if (op == OP_CONNFWD) {
				if (seq_last && realop != OP_BORDER) m.merge(gatenext, e2);
				if (realop == OP_CLOSE) m.merge(gateprev, e1);
			} else if (op == OP_CONNBWD) m.merge(gateprev, e1);
		}
	}
}

static void decode_attrs(Mesh &m, SymReader &rd, const std::vector<uint32_t> &order_v)
{
	Predictor pr(m);
	std::vector<uint32_t> vattr(m.nv, 0);
	pr.vtx_attr = &vattr;
	const int LV = 1, LF = 0;
	uint32_t cur = 0;
	{
		List &L = m.lists[LV];
		for (he_t e : order_v) {   // attrcode.h:443-470
			uint32_t v = m.org[e];
			(void)rd.reg_vtx();
			pr.collect_vertex(e);
			uint8_t ty = rd.attr_type(LV);
			if (ty != A_DATA) throw std::runtime_error("oracle: shared-attribute history is not supported");
			if (cur >= L.count) throw std::runtime_error("oracle: corrupt stream (attribute overflow)");
			uint32_t idx = cur++;
			int ctx = rd.md.attr_base[LV] + ATTR_DATA;
			for (int c = 0; c < L.fmt.size(); ++c) {
				int nb = TSIZE[L.fmt.stype[c]];
				rd.bytes(ctx, L.rec(idx) + L.fmt.off[c], nb);
				ctx += nb;
			}
			vattr[v] = idx;
			for (int c = 0; c < L.fmt.size(); ++c) {
				with_type(L.fmt.stype[c], [&](auto tag) {
					typedef decltype(tag) T;
					T pred = pr.predict_component<T>(L, c, L.fmt.quant[c]);
					T d = ld<T>(L.rec(idx) + L.fmt.off[c]);
					st<T>(L.rec(idx) + L.fmt.off[c], unfold_residual<T>(d, pred, L.fmt.quant[c], std::is_floating_point<T>()));
				});
			}
		}
	}
	{
		List &L = m.lists[LF];
		uint32_t curf = 0;
		for (uint32_t f = 0; f < m.nf; ++f) {   // attrcode.h:476-501, faces in index order
			(void)rd.reg_face();
			uint8_t ty = rd.attr_type(LF);
			if (ty != A_DATA) throw std::runtime_error("oracle: shared-attribute history is not supported");
			uint32_t idx = curf++;
			int ctx = rd.md.attr_base[LF] + ATTR_DATA;
			for (int c = 0; c < L.fmt.size(); ++c) {
				int nb = TSIZE[L.fmt.stype[c]];
				rd.bytes(ctx, L.rec(idx) + L.fmt.off[c], nb);
				ctx += nb;
			}
			for (int c = 0; c < L.fmt.size(); ++c) {
				with_type(L.fmt.stype[c], [&](auto tag) {
					typedef decltype(tag) T;
					T d = ld<T>(L.rec(idx) + L.fmt.off[c]);
					st<T>(L.rec(idx) + L.fmt.off[c], unfold_residual<T>(d, T(0), L.fmt.quant[c], std::is_floating_point<T>()));
				});
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// general bindings: several regions, records shared between elements (global / per-vertex history) and corner
// attributes -- what the OBJ reader creates (attrcode.h:23-80 histories, :108-289 prediction, :321-393 encoder,
// :443-531 decoder).  The PLY layout above is the special case with private records and no corner lists.
// ------------------------------------------------------------------------------------------------
static const uint32_t UNSET = 0xffffffffu;

template <typename F> static void fan_each(const Mesh &m, he_t ein, F &&cb)   // attrcode.h:83-106 (TFAN_IT)
{
	he_t e = ein, t;
	for (;;) {
		cb(e);
		t = m.twin[e];
		if (t == e) break;
		e = m.next(t);
		if (e == ein) return;
	}
	e = m.prev(ein);
	t = m.twin[e];
	if (e == t) return;
	e = t;
	do {
		cb(e);
		e = m.prev(e);
		t = m.twin[e];
		if (e == t) break;
		e = t;
	} while (e != ein);
}

// attrcode.h:182-208: mean of the parts (rounded division in the wide type), then for floats the part nearest to the mean
template <typename T> static T combine_parts(const std::vector<T> &pv)
{
	typedef typename big_of<T>::type B;
	if (pv.empty()) return T(0);
	B acc = 0;
	for (T x : pv) acc = acc + (B)x;
	acc = div_round(acc, (B)pv.size());
	T avg = (T)acc;
	if (!std::is_floating_point<T>::value) return avg;
	T res = std::numeric_limits<T>::max();
	for (T x : pv) {
		T rd = avg > res ? avg - res : res - avg;
		T pd = avg > x ? avg - x : x - avg;
		res = rd < pd ? res : x;
	}
	return res;
}

struct GenAttr {
	Mesh &m;
	Mesh::Bind &b;
	std::vector<char> vdone, fdone;
	std::vector<Cand> cands;      // parallelograms around the vertex being coded
	std::vector<he_t> srcs;       // corners (half-edges) of already coded faces of the same region around the corner's vertex
	explicit GenAttr(Mesh &mesh) : m(mesh), b(mesh.bind), vdone(mesh.nv, 0), fdone(mesh.nf, 0) {}

	void offer(uint32_t v0, uint32_t v1, uint32_t vo, int r)   // attrcode.h:117-134
	{
		if (!vdone[v0] || !vdone[v1] || !vdone[vo]) return;
		if (b.vtx_reg[v0] != r || b.vtx_reg[v1] != r || b.vtx_reg[vo] != r) return;
		cands.push_back(Cand{ v0, v1, vo });
	}
	void paral(he_t ein, int r)   // attrcode.h:155-171
	{
		he_t e = ein;
		int d = m.deg(m.eface[e]);
		if (d == 3) {
			e = m.next(e);
			he_t t = m.twin[e];
			if (t == e) return;
			e = m.next(m.next(t));
			offer(m.org[t], m.dest(t), m.org[e], r);
			return;
		}
		he_t e0 = m.next(e), e1 = m.prev(e);
		offer(m.org[e0], m.org[e1], m.dest(e0), r);
		if (d > 4) offer(m.org[e0], m.org[e1], m.org[m.prev(e)], r);
	}
	void vertex(he_t e, int r)   // attrcode.h:209-225
	{
		cands.clear();
		fan_each(m, e, [&](he_t x) { paral(x, r); });
		vdone[m.org[e]] = 1;
	}
	void corner(uint32_t f, he_t e)   // attrcode.h:135-154,272-288: the face itself does not count
	{
		int r = b.face_reg[f];
		srcs.clear();
		fdone[f] = 0;
		fan_each(m, e, [&](he_t x) { uint32_t g = m.eface[x]; if (fdone[g] && b.face_reg[g] == r) srcs.push_back(x); });
		fdone[f] = 1;
	}
	template <typename T> T predict_vtx(const List &L, int a, int c) const
	{
		std::vector<T> pv;
		const int off = L.fmt.off[c], q = L.fmt.quant[c];
		for (const Cand &k : cands) {
			T x = ld<T>(L.rec(b.vtx_attr[(size_t)k.v0 * b.nb_vtx + a]) + off), y = ld<T>(L.rec(b.vtx_attr[(size_t)k.v1 * b.nb_vtx + a]) + off),
			  o = ld<T>(L.rec(b.vtx_attr[(size_t)k.vo * b.nb_vtx + a]) + off);
			pv.push_back(paral_predict<T>(x, y, o, q, std::is_floating_point<T>()));
		}
		return combine_parts<T>(pv);
	}
	template <typename T> T predict_corner(const List &L, int a, int c) const   // prediction.h:149-164: the value itself
	{
		std::vector<T> pv;
		for (he_t x : srcs) pv.push_back(ld<T>(L.rec(b.corner_attr[(size_t)x * b.nb_corner + a]) + L.fmt.off[c]));
		return combine_parts<T>(pv);
	}
};

struct LocalHist {   // attrcode.h:54-80, one per corner slot
	std::vector<std::vector<uint32_t>> hist;
	uint32_t insert(uint32_t v, uint32_t idx)
	{
		std::vector<uint32_t> &h = hist[v];
		for (size_t i = 0; i < h.size(); ++i) if (h[i] == idx) return (uint32_t)(h.size() - 1 - i);
		h.push_back(idx);
		return UNSET;
	}
	uint32_t find(uint32_t v, uint32_t off) const
	{
		const std::vector<uint32_t> &h = hist[v];
		if (off >= h.size()) throw std::runtime_error("oracle: corrupt stream (per-vertex history)");
		return h[h.size() - 1 - off];
	}
};

static void check_general(const Mesh &m)
{
	const Mesh::Bind &b = m.bind;
	if (m.is_shard()) throw std::runtime_error("oracle: a shard holds the PLY layout only");
	if (b.nregs_face() > 128 || b.nregs_vtx() > 128) throw std::runtime_error("oracle: more than 128 regions overflow the reference's model seeding (model.h:49-55)");
	if (b.face_reg.size() != m.nf || b.vtx_reg.size() != m.nv) throw std::runtime_error("oracle: region tables do not match the element counts");
	if (b.face_attr.size() != (size_t)m.nf * b.nb_face || b.vtx_attr.size() != (size_t)m.nv * b.nb_vtx ||
	    b.corner_attr.size() != (size_t)m.num_edge() * b.nb_corner) throw std::runtime_error("oracle: binding tables do not match the element counts");
}

static void encode_attrs_general(Mesh &m, SymWriter &wr, const std::vector<uint32_t> &order_v, const std::vector<uint32_t> &order_f)
{
	GenAttr g(m);
	const Mesh::Bind &b = m.bind;
	std::vector<std::vector<uint32_t>> seen_at(m.lists.size());   // GlobalHistory::tidxlist (attrcode.h:23-53)
	std::vector<uint32_t> nseen(m.lists.size(), 0);
	for (size_t l = 0; l < m.lists.size(); ++l) seen_at[l].assign(m.lists[l].count, UNSET);
	std::vector<LocalHist> lh(b.nb_corner);
	for (auto &h : lh) h.hist.resize(m.nv);
	std::vector<uint8_t> resid;
	auto seen_before = [&](int l, uint32_t idx) -> uint32_t {   // lget_set
		if (idx >= seen_at[l].size()) throw std::runtime_error("oracle: binding outside its list");
		uint32_t at = seen_at[l][idx];
		if (at == UNSET) { seen_at[l][idx] = nseen[l]++; return UNSET; }
		return nseen[l] - 1 - at;
	};
	auto code_data = [&](int l, uint32_t idx, auto &&pred) {   // attrcode.h:340-342 + io.h:90-94
		List &L = m.lists[l];
		resid.assign(std::max(L.fmt.bytes(), 1), 0);
		for (int c = 0; c < L.fmt.size(); ++c)
			with_type(L.fmt.stype[c], [&](auto tag) {
				typedef decltype(tag) T;
				T p = pred(tag, L, c);
				T raw = ld<T>(L.rec(idx) + L.fmt.off[c]);
				st<T>(resid.data() + L.fmt.off[c], fold_residual<T>(raw, p, L.fmt.quant[c], std::is_floating_point<T>()));
			});
		wr.attr_type(l, A_DATA);
		int ctx = wr.md.attr_base[l] + ATTR_DATA;
		for (int c = 0; c < L.fmt.size(); ++c) {
			int nb = TSIZE[L.fmt.stype[c]];
			wr.bytes(ctx, resid.data() + L.fmt.off[c], nb);
			ctx += nb;
		}
	};
	for (he_t e : order_v) {   // attrcode.h:321-344
		uint32_t v = m.org[e];
		int r = b.vtx_reg[v];
		g.vertex(e, r);
		wr.reg_vtx((uint16_t)r);
		for (int a = 0; a < b.nvtxlists(r); ++a) {
			int l = b.vtxlist(r, a);
			uint32_t idx = b.vtx_attr[(size_t)v * b.nb_vtx + a];
			uint32_t d = seen_before(l, idx);
			if (d != UNSET) { wr.attr_ghist(l, d); continue; }
			code_data(l, idx, [&](auto tag, const List &L, int c) { return g.predict_vtx<decltype(tag)>(L, a, c); });
		}
	}
	for (he_t e : order_f) {   // attrcode.h:345-393,405-414
		uint32_t f = m.eface[e];
		int r = b.face_reg[f];
		g.fdone[f] = 1;          // face prediction never finds a coded neighbour (attrcode.h:245-254 tests the face itself)
		wr.reg_face((uint16_t)r);
		for (int a = 0; a < b.nfacelists(r); ++a) {
			int l = b.facelist(r, a);
			uint32_t idx = b.face_attr[(size_t)f * b.nb_face + a];
			uint32_t d = seen_before(l, idx);
			if (d != UNSET) { wr.attr_ghist(l, d); continue; }
			code_data(l, idx, [&](auto tag, const List &, int) { return decltype(tag)(0); });
		}
		he_t c = e;
		do {
			g.corner(f, c);
			uint32_t v = m.org[c];
			for (int a = 0; a < b.ncornerlists(r); ++a) {
				int l = b.cornerlist(r, a);
				uint32_t idx = b.corner_attr[(size_t)c * b.nb_corner + a];
				uint32_t ld_ = lh[a].insert(v, idx);
				if (ld_ != UNSET) { wr.attr_lhist(l, (uint16_t)ld_); continue; }
				uint32_t d = seen_before(l, idx);
				if (d != UNSET) { wr.attr_ghist(l, d); continue; }
				code_data(l, idx, [&](auto tag, const List &L, int cc) { return g.predict_corner<decltype(tag)>(L, a, cc); });
			}
			c = m.next(c);
		} while (c != e);
	}
}

static void decode_attrs_general(Mesh &m, SymReader &rd, const std::vector<uint32_t> &order_v)
{
	GenAttr g(m);
	Mesh::Bind &b = m.bind;
	b.corner_attr.assign((size_t)m.num_edge() * b.nb_corner, 0);
	std::vector<uint32_t> cur(m.lists.size(), 0);
	std::vector<LocalHist> lh(b.nb_corner);
	for (auto &h : lh) h.hist.resize(m.nv);
	auto read_data = [&](int l, auto &&pred) -> uint32_t {   // attrcode.h:457-462
		List &L = m.lists[l];
		if (cur[l] >= L.count) throw std::runtime_error("oracle: corrupt stream (attribute overflow)");
		uint32_t idx = cur[l]++;
		int ctx = rd.md.attr_base[l] + ATTR_DATA;
		for (int c = 0; c < L.fmt.size(); ++c) {
			int nb = TSIZE[L.fmt.stype[c]];
			rd.bytes(ctx, L.rec(idx) + L.fmt.off[c], nb);
			ctx += nb;
		}
		for (int c = 0; c < L.fmt.size(); ++c)
			with_type(L.fmt.stype[c], [&](auto tag) {
				typedef decltype(tag) T;
				T p = pred(tag, L, c);
				T d = ld<T>(L.rec(idx) + L.fmt.off[c]);
				st<T>(L.rec(idx) + L.fmt.off[c], unfold_residual<T>(d, p, L.fmt.quant[c], std::is_floating_point<T>()));
			});
		return idx;
	};
	auto read_hist = [&](int l) -> uint32_t {   // attrcode.h:463-465
		uint32_t d = rd.attr_ghist(l);
		if (d >= cur[l]) throw std::runtime_error("oracle: corrupt stream (history)");
		return cur[l] - 1 - d;
	};
	for (he_t e : order_v) {   // attrcode.h:443-470
		uint32_t v = m.org[e];
		int r = rd.reg_vtx();
		if (r >= b.nregs_vtx()) throw std::runtime_error("oracle: corrupt stream (vertex region)");
		b.vtx_reg[v] = (uint16_t)r;
		g.vertex(e, r);
		for (int a = 0; a < b.nvtxlists(r); ++a) {
			int l = b.vtxlist(r, a);
			uint8_t ty = rd.attr_type(l);
			uint32_t idx;
			if (ty == A_DATA) idx = read_data(l, [&](auto tag, const List &L, int c) { return g.predict_vtx<decltype(tag)>(L, a, c); });
			else if (ty == A_HIST) idx = read_hist(l);
			else throw std::runtime_error("oracle: corrupt stream (attribute type)");
			b.vtx_attr[(size_t)v * b.nb_vtx + a] = idx;
		}
	}
	for (uint32_t f = 0; f < m.nf; ++f) {   // attrcode.h:476-531,543-548: faces in index order, corners from 0
		int r = rd.reg_face();
		if (r >= b.nregs_face()) throw std::runtime_error("oracle: corrupt stream (face region)");
		b.face_reg[f] = (uint16_t)r;
		g.fdone[f] = 1;
		for (int a = 0; a < b.nfacelists(r); ++a) {
			int l = b.facelist(r, a);
			uint8_t ty = rd.attr_type(l);
			uint32_t idx;
			if (ty == A_DATA) idx = read_data(l, [&](auto tag, const List &, int) { return decltype(tag)(0); });
			else if (ty == A_HIST) idx = read_hist(l);
			else throw std::runtime_error("oracle: corrupt stream (attribute type)");
			b.face_attr[(size_t)f * b.nb_face + a] = idx;
		}
		for (he_t c = m.foff[f]; c < m.foff[f + 1]; ++c) {
			g.corner(f, c);
			uint32_t v = m.org[c];
			for (int a = 0; a < b.ncornerlists(r); ++a) {
				int l = b.cornerlist(r, a);
				uint8_t ty = rd.attr_type(l);
				uint32_t idx;
				if (ty == A_DATA) { idx = read_data(l, [&](auto tag, const List &L, int cc) { return g.predict_corner<decltype(tag)>(L, a, cc); }); lh[a].insert(v, idx); }
				else if (ty == A_HIST) { idx = read_hist(l); lh[a].insert(v, idx); }
				else if (ty == A_LHIST) idx = lh[a].find(v, rd.attr_lhist(l));
				else throw std::runtime_error("oracle: corrupt stream (attribute type)");
				b.corner_attr[(size_t)c * b.nb_corner + a] = idx;
			}
		}
	}
}

static Mesh *decode(const uint8_t *p, size_t n)
{
	Mesh *m = new Mesh();
	try {
		ByteReader br{ p, p + n };
		read_header(br, *m);
		RangeDecoder rc(br.p, p + n);
		Models md(*m);
		SymReader rd(md, rc);
		std::vector<uint32_t> order_v;
		m->org.reserve(m->nf * 3); m->twin.reserve(m->nf * 3); m->eface.reserve(m->nf * 3); m->foff.reserve(m->nf + 1);
		// the models need the degree table from the header, but add_face() also records degrees: keep header's
		std::vector<char> hdr_deg = m->have_deg;
		cbm_decode(*m, rd, order_v);
		m->have_deg = hdr_deg;
		if (m->num_face() != m->nf) throw std::runtime_error("oracle: face count mismatch");
		if (m->bind.on) decode_attrs_general(*m, rd, order_v);
		else decode_attrs(*m, rd, order_v);
	} catch (...) { delete m; throw; }
	return m;
}

// ------------------------------------------------------------------------------------------------
// chunked profile (.hry v0.2, this implementation's parallel container; NOT a reference format).
// Same header (minor version 2), same symbols as the v0.1 stream, but every context plane is cut into chunks of
// `chunk_syms` symbols and every (plane, chunk) is an independent stream: fresh adaptive model + fresh coder with
// 32-bit registers (arith::Encoder<uint32_t>: the reference's coder template instantiated on uint32_t) + 32-bit flush,
// using exactly the reference's model/coder arithmetic.  A chunk holds at most 2^20 symbols, so every total stays far
// below 2^30 = QUARTER (the coder's requirement) and the interval keeps >= 10 bits of resolution per count.  Symbols without information are
// not stored: reg_face/reg_vtx (single region), attr_type (always DATA), numtri for single-degree meshes.
// Operations are split into one plane per order class (models.h:101-105) with a plain adaptive 7-symbol model.
//   u32 chunk_syms, u32 conn_chunk_syms, u32 n_planes, n_planes x u32 n_symbols, n_planes x prior (see plane_prior / write_prior),
//   u32 n_restart, n_restart x 17 u32 (restart points, see encode_chunked), per restart point u32 n + n x (u32 vertex, u32 counter),
//   per stream u32 n_bytes, then the streams.
// Plane order: iop, elem[4], part[2], vertid[4], numtri[2], op class[8], vertex data bytes, face data bytes.
// The first 21 planes (connectivity) are cut every conn_chunk_syms symbols, the attribute planes every chunk_syms: the
// decoder needs the connectivity first and a stream is a serial chain, so short connectivity streams shorten its start-up
// latency; their small alphabets re-adapt within a few symbols, which keeps the size cost below 1 %.
// ------------------------------------------------------------------------------------------------
struct PlaneDef { int slot; int init_kind; };   // init_kind: 0 = 256 ones, 1 = iop (9 ones), 2/3 = numtri byte 0/1, 4 = op (7 ones)

static std::vector<PlaneDef> chunked_planes(const Mesh &m, const Models &md)
{
	std::vector<PlaneDef> p;
	p.push_back({ CTX_IOP, 1 });
	for (int i = 0; i < 4; ++i) p.push_back({ CTX_ELEM + i, 0 });
	for (int i = 0; i < 2; ++i) p.push_back({ CTX_PART + i, 0 });
	for (int i = 0; i < 4; ++i) p.push_back({ CTX_VERT + i, 0 });
	p.push_back({ CTX_NUMTRI, 2 }); p.push_back({ CTX_NUMTRI + 1, 3 });
	for (int i = 0; i < 8; ++i) p.push_back({ REC_OP0 + i, 4 });
	if (m.bind.on) {
		// general bindings: the region of every vertex / face (low byte; only when there is more than one region), then for every
		// list a region binds, in list order: the kind of every reference, the creation-order distances (4 bytes), at corner
		// lists the per-vertex distances (2 bytes), and the residual bytes of the records coded as data
		if (m.bind.nregs_vtx() > 1) p.push_back({ CTX_REGVTX, 5 });
		if (m.bind.nregs_face() > 1) p.push_back({ CTX_REGFACE, 6 });
		for (size_t l = 0; l < m.lists.size(); ++l) {
			if (m.lists[l].target == TG_NONE) continue;
			p.push_back({ md.attr_base[l] + ATTR_TYPE, m.lists[l].target == TG_CORNER ? 8 : 7 });
			for (int b = 0; b < 4; ++b) p.push_back({ md.attr_base[l] + ATTR_GHIST + b, 0 });
			if (m.lists[l].target == TG_CORNER) for (int b = 0; b < 2; ++b) p.push_back({ md.attr_base[l] + ATTR_LHIST + b, 0 });
			int n = 0;
			for (int c = 0; c < m.lists[l].fmt.size(); ++c) n += TSIZE[m.lists[l].fmt.stype[c]];
			for (int b = 0; b < n; ++b) p.push_back({ md.attr_base[l] + ATTR_DATA + b, 0 });
		}
		return p;
	}
	for (int l : { 1, 0 }) {
		int n = 0;
		for (int c = 0; c < m.lists[l].fmt.size(); ++c) n += TSIZE[m.lists[l].fmt.stype[c]];
		for (int b = 0; b < n; ++b) p.push_back({ md.attr_base[l] + ATTR_DATA + b, 0 });
	}
	return p;
}
static void seed_table(FreqTable &f, int kind, const Mesh &m)
{
	switch (kind) {
	case 0: for (uint32_t j = 0; j < 256; ++j) f.inc(j); break;
	case 1: for (uint32_t j = 0; j <= IOP_EOM; ++j) f.inc(j); break;
	case 2: for (size_t d = 0; d < m.have_deg.size(); ++d) if (m.have_deg[d]) f.inc((uint32_t)((d - 2) & 0xff)); break;
	case 3: for (size_t d = 0; d < m.have_deg.size(); ++d) if (m.have_deg[d]) f.inc((uint32_t)((d - 2) >> 8)); break;
	case 4: for (uint32_t j = 0; j <= OP_CONNFWD; ++j) f.inc(j); break;
	case 5: for (int r = 0; r < m.bind.nregs_vtx(); ++r) f.inc((uint32_t)r); break;    // models.h:212-217
	case 6: for (int r = 0; r < m.bind.nregs_face(); ++r) f.inc((uint32_t)r); break;
	case 7: f.inc(A_DATA); f.inc(A_HIST); break;                                       // models.h:201-203
	case 8: f.inc(A_DATA); f.inc(A_HIST); f.inc(A_LHIST); break;
	}
}
// Static prior of a plane (chunked container): every chunk of the plane starts its adaptive table from the plane's own
// histogram scaled to about PRIOR_K counts instead of the reference's flat initial counts -- a fresh table per chunk costs
// ~250 bytes of learning on a 256-symbol plane, which is what made short chunks expensive.  Planes with fewer than
// PRIOR_MIN_SYMS symbols keep the reference's initial counts.  table[s] = 0 for symbols that do not occur in the plane.
enum { PRIOR_K = 1024, PRIOR_MIN_SYMS = 1024 };
static bool plane_prior(const std::vector<uint8_t> &sy, uint32_t table[256])
{
	for (int s = 0; s < 256; ++s) table[s] = 0;
	if (sy.size() < PRIOR_MIN_SYMS) return false;
	uint64_t hist[256] = { 0 };
	for (uint8_t b : sy) ++hist[b];
	const uint64_t n = sy.size();
	for (int s = 0; s < 256; ++s)
		if (hist[s]) table[s] = (uint32_t)std::max<uint64_t>(1, (hist[s] * PRIOR_K + n / 2) / n);
	return true;
}
// directory form: u8 mode (0 = reference initial counts, 1 = prior); prior: 32-byte bitmap of the symbols present (bit s & 7
// of byte s >> 3), then one value per present symbol in symbol order: u8 if < 255, else 255 followed by u16
static void write_prior(ByteWriter &w, bool use, const uint32_t table[256])
{
	w.put<uint8_t>(use ? 1 : 0);
	if (!use) return;
	uint8_t bm[32] = { 0 };
	for (int s = 0; s < 256; ++s) if (table[s]) bm[s >> 3] |= (uint8_t)(1u << (s & 7));
	w.raw(bm, 32);
	for (int s = 0; s < 256; ++s) {
		if (!table[s]) continue;
		if (table[s] < 255) w.put<uint8_t>((uint8_t)table[s]);
		else { w.put<uint8_t>(255); w.put<uint16_t>((uint16_t)table[s]); }
	}
}
static bool read_prior(ByteReader &br, uint32_t table[256])
{
	for (int s = 0; s < 256; ++s) table[s] = 0;
	uint8_t mode = br.get<uint8_t>();
	if (mode == 0) return false;
	if (mode != 1) throw std::runtime_error("oracle: bad prior mode");
	uint8_t bm[32];
	for (int i = 0; i < 32; ++i) bm[i] = br.get<uint8_t>();
	for (int s = 0; s < 256; ++s) {
		if (!(bm[s >> 3] & (1u << (s & 7)))) continue;
		uint32_t v = br.get<uint8_t>();
		if (v == 255) v = br.get<uint16_t>();
		if (v == 0) throw std::runtime_error("oracle: zero count in a prior");
		table[s] = v;
	}
	return true;
}
static int count_degrees(const Mesh &m) { int n = 0; for (char c : m.have_deg) n += c ? 1 : 0; return n; }
enum { CONN_PLANES = 21, RESTART_FACES = 8192 };
static uint32_t default_conn_chunk(uint32_t chunk_syms) { return std::min(chunk_syms, std::max(chunk_syms / 8, 512u)); }
// chunks of an attribute plane grow with their position: length = position / 16 rounded down to a power of two, at least 1024,
// at most the container's chunk size (a decoder walks the vertices in order and should find the early chunks decoded early);
// connectivity planes keep one size
static size_t attr_chunk_len(size_t pos, size_t chunk_syms)
{
	size_t len = 1024;
	while (len < chunk_syms && len * 2 <= pos / 16) len *= 2;
	return std::min(len, chunk_syms);
}

// border snapshots in the directory (round 6): LEB128 varints, zigzag-folded differences in runs (see encode_chunked)
static void put_varint(std::vector<uint8_t> &o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
static uint64_t zigzag64(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
static int64_t unzigzag64(uint64_t z) { return (int64_t)(z >> 1) ^ -(int64_t)(z & 1); }
static void put_runs(std::vector<uint8_t> &o, const std::vector<int64_t> &d)   // `length of a run of zeros`, then the non-zero value that ends it (nothing behind the last one)
{
	for (size_t i = 0; i < d.size();) {
		uint64_t run = 0;
		while (i < d.size() && d[i] == 0) { ++run; ++i; }
		put_varint(o, run);
		if (i < d.size()) { put_varint(o, zigzag64(d[i])); ++i; }
	}
}
static uint64_t get_varint(const uint8_t *&p, const uint8_t *end)
{
	uint64_t v = 0;
	for (int sh = 0; sh < 64; sh += 7) {
		if (p == end) throw std::runtime_error("oracle: truncated border snapshot");
		const uint8_t b = *p++;
		v |= (uint64_t)(b & 0x7f) << sh;
		if (!(b & 0x80)) return v;
	}
	throw std::runtime_error("oracle: bad varint");
}
static std::vector<int64_t> get_runs(const uint8_t *&p, const uint8_t *end, size_t n)
{
	std::vector<int64_t> d;
	while (d.size() < n) {
		uint64_t run = get_varint(p, end);
		if (run > n - d.size()) throw std::runtime_error("oracle: bad run in a border snapshot");
		d.insert(d.end(), (size_t)run, 0);
		if (d.size() < n) d.push_back(unzigzag64(get_varint(p, end)));
	}
	return d;
}
// spacing of the border snapshots for a container that describes nf faces: at least 2^17 faces, at most some sixty snapshots
static uint32_t default_snapshot_faces(uint32_t nf) { uint32_t sp = 1u << 17; while ((uint64_t)sp * 64u < nf) sp <<= 1; return sp; }
enum : uint32_t { SNAPSHOT_DEFAULT = 0xffffffffu };

static Result *encode_chunked(Mesh &m, uint32_t chunk_syms, uint32_t snapshot_faces = SNAPSHOT_DEFAULT)
{
	if (m.bind.on) check_general(m); else check_supported(m);
	if (chunk_syms == 0) chunk_syms = 8192;
	chunk_syms = std::min(chunk_syms, 1u << 20);
	Result *res = new Result();
	try {
		// sharded container (.hry v0.3): the header of the WHOLE mesh, u32 n_segments, n_segments x u64 bytes, then per segment
		// u32 n_runs, the runs (6 x u32), and the v0.2 body of the shard in its own numbering.  A shard writes one segment
		// (none when it holds no face); segments of different shards are concatenated without re-coding.
		write_header(m, res->bytes, m.is_shard() ? 3 : 2);
		res->header_size = res->bytes.size();
		size_t seg_len_at = 0, seg_begin = 0;
		if (m.is_shard()) {
			ByteWriter w0{ res->bytes };
			if (m.num_face() == 0) { w0.put<uint32_t>(0); return res; }
			w0.put<uint32_t>(1);
			seg_len_at = res->bytes.size();
			w0.put<uint64_t>(0);
			seg_begin = res->bytes.size();
			w0.put<uint32_t>((uint32_t)m.runs.size());
			for (const auto &r : m.runs) for (uint32_t x : r) w0.put<uint32_t>(x);
		}
		std::vector<uint8_t> dummy;
		RangeEncoder rc(dummy);
		Models md(m);
		SymWriter wr(md, rc, nullptr);
		std::vector<std::vector<uint8_t>> rec(std::max<size_t>(REC_SLOTS, md.tab.size()));
		wr.record = &rec;
		wr.snapshot_faces = snapshot_faces == SNAPSHOT_DEFAULT ? default_snapshot_faces((uint32_t)m.num_face()) : snapshot_faces;
		cbm_encode(m, wr, res->order_v, res->order_f);
		if (m.bind.on) encode_attrs_general(m, wr, res->order_v, res->order_f);
		else encode_attrs(m, wr, res->order_v, res->order_f);
		if (count_degrees(m) <= 1) { rec[CTX_NUMTRI].clear(); rec[CTX_NUMTRI + 1].clear(); }
		std::vector<PlaneDef> planes = chunked_planes(m, md);
		ByteWriter w{ res->bytes };
		const uint32_t conn_chunk = default_conn_chunk(chunk_syms);
		w.put<uint32_t>(chunk_syms);
		w.put<uint32_t>(conn_chunk);
		w.put<uint32_t>((uint32_t)planes.size());
		for (const PlaneDef &pd : planes) w.put<uint32_t>((uint32_t)rec[pd.slot].size());
		std::vector<std::array<uint32_t, 256>> prior(planes.size());
		std::vector<char> has_prior(planes.size(), 0);
		for (size_t pi = 0; pi < planes.size(); ++pi) {
			has_prior[pi] = plane_prior(rec[planes[pi].slot], prior[pi].data()) ? 1 : 0;
			write_prior(w, has_prior[pi] != 0, prior[pi].data());
		}
		// restart points of the connectivity replay: the coder state at the first component start that lies at least
		// RESTART_FACES faces after the previous point: symbols consumed per plane group (iop, elem, part, vertid, numtri)
		// and per operation class, first vertex index / face / half-edge, flags (bit 0: a component up to the next point
		// names a vertex created before this point).  A decoder may replay from several points at once.
		{
			wr.finish_marks();
			const bool numtri_coded = count_degrees(m) > 1;
			std::vector<uint32_t> he_before(res->order_f.size() + 1, 0);
			for (size_t i = 0; i < res->order_f.size(); ++i) he_before[i + 1] = he_before[i] + (uint32_t)m.deg(m.eface[res->order_f[i]]);
			std::vector<std::vector<uint32_t>> pts;
			uint32_t last_face = 0;
			for (size_t k = 1; k < wr.marks.size(); ++k) {
				const SymWriter::Mark &mk = wr.marks[k];
				if (mk.first_face - last_face < RESTART_FACES) {
					if (!pts.empty() && mk.min_ref < pts.back()[13]) pts.back()[16] |= 1u;
					continue;
				}
				std::vector<uint32_t> r(17);
				for (int g = 0; g < 5; ++g) r[g] = mk.n_grp[g];
				if (!numtri_coded) r[4] = 0;
				for (int i = 0; i < 8; ++i) r[5 + i] = mk.n_op[i];
				r[13] = mk.first_vertex; r[14] = mk.first_face; r[15] = he_before[mk.first_face];
				r[16] = mk.min_ref < mk.first_vertex ? 1u : 0u;
				pts.push_back(r);
				last_face = mk.first_face;
			}
			w.put<uint32_t>((uint32_t)pts.size() | (wr.snaps.empty() ? 0u : 0x80000000u));   // (top bit: a section of border snapshots follows the counters)
			for (auto &r : pts) for (uint32_t x : r) w.put<uint32_t>(x);
			// per restart point: the older vertices its span names (first naming in the span) with their counters at that moment
			// = at the start of the span (a vertex is only touched after it has been named in the component at hand)
			std::vector<uint32_t> span_of_mark(wr.marks.size(), 0xffffffffu);   // mark -> index of its restart span, or none (span before the first point)
			{
				for (size_t k = 1; k < wr.marks.size(); ++k) {
					if (pts.empty() || wr.marks[k].first_face < pts[0][14]) continue;
					size_t q = 0;
					while (q + 1 < pts.size() && wr.marks[k].first_face >= pts[q + 1][14]) ++q;
					span_of_mark[k] = (uint32_t)q;
				}
			}
			// A span ends where the next point lies, of either kind: a naming behind a border snapshot belongs to the span that starts
			// at the snapshot (the last one at or before it in the stream), which brings the counters of the vertices ON its border
			// along; of the older vertices it names only the others are listed with it.
			std::vector<std::vector<std::pair<uint32_t, uint32_t>>> counters(pts.size()), snap_counters(wr.snaps.size());
			std::vector<uint32_t> nth(wr.snaps.size(), 0);   // a snapshot's number inside its component, from 1
			for (size_t i = 0; i < wr.snaps.size(); ++i) nth[i] = i && wr.snaps[i - 1].mark == wr.snaps[i].mark ? nth[i - 1] + 1 : 1u;
			for (const SymWriter::Named &ev : wr.named) {
				const uint32_t sp = span_of_mark[ev.mark];
				long q = -1;   // the last snapshot at or before the naming
				for (size_t i = 0; i < wr.snaps.size(); ++i)
					if (wr.snaps[i].mark < ev.mark || (wr.snaps[i].mark == ev.mark && nth[i] <= ev.snap)) q = (long)i;
				if (q >= 0 && (sp == 0xffffffffu || wr.snaps[q].first_face > pts[sp][14])) {
					const SymWriter::Snap &S = wr.snaps[q];
					if (ev.id >= S.first_vertex || std::find(S.vtx.begin(), S.vtx.end(), ev.id) != S.vtx.end()) continue;
					bool dup = false;
					for (auto &c : snap_counters[q]) if (c.first == ev.id) { dup = true; break; }
					if (!dup) snap_counters[q].push_back({ ev.id, ev.count });
					continue;
				}
				if (sp == 0xffffffffu || ev.id >= pts[sp][13]) continue;
				bool dup = false;
				for (auto &c : counters[sp]) if (c.first == ev.id) { dup = true; break; }
				if (!dup) counters[sp].push_back({ ev.id, ev.count });
			}
			for (auto &cs : counters) {
				w.put<uint32_t>((uint32_t)cs.size());
				for (auto &c : cs) { w.put<uint32_t>(c.first); w.put<uint32_t>(c.second); }
			}
			// the border snapshots: u32 spacing, u32 n, u32 bytes, then `bytes` bytes of varints, then zero bytes up to a multiple of
			// four counted from the section's first byte.  Per snapshot: symbols consumed per plane group (5) and operation class (8),
			// next vertex / face / half-edge, the number of counters + the counters, the number of parts and of elements, per part
			// size << 1 | edge_begin; the vertices as differences from "the vertex before + 1" (before the first: the snapshot's
			// next vertex) in runs; the triangle counts as differences from the count before (before the first: 3) in runs
			if (!wr.snaps.empty()) {
				const size_t sec0 = res->bytes.size();
				std::vector<uint8_t> body;
				for (size_t k = 0; k < wr.snaps.size(); ++k) {
					const SymWriter::Snap &S = wr.snaps[k];
					for (int g = 0; g < 5; ++g) put_varint(body, g == 4 && !numtri_coded ? 0u : S.n_grp[g]);
					for (int i = 0; i < 8; ++i) put_varint(body, S.n_op[i]);
					put_varint(body, S.first_vertex); put_varint(body, S.first_face); put_varint(body, he_before[S.first_face]);
					put_varint(body, snap_counters[k].size());
					for (auto &c : snap_counters[k]) { put_varint(body, c.first); put_varint(body, c.second); }
					put_varint(body, S.parts.size()); put_varint(body, S.vtx.size());
					for (uint32_t pt : S.parts) put_varint(body, pt);
					std::vector<int64_t> d(S.vtx.size());
					for (size_t i = 0; i < d.size(); ++i) d[i] = (int64_t)S.vtx[i] - ((i ? (int64_t)S.vtx[i - 1] : (int64_t)S.first_vertex - 1) + 1);
					put_runs(body, d);
					for (size_t i = 0; i < d.size(); ++i) d[i] = (int64_t)S.seen[i] - (i ? (int64_t)S.seen[i - 1] : 3);
					put_runs(body, d);
				}
				w.put<uint32_t>(wr.snapshot_faces); w.put<uint32_t>((uint32_t)wr.snaps.size()); w.put<uint32_t>((uint32_t)body.size());
				w.raw(body.data(), body.size());
				while ((res->bytes.size() - sec0) & 3) w.put<uint8_t>(0);
			}
		}
		std::vector<std::vector<uint8_t>> streams;
		for (size_t pi = 0; pi < planes.size(); ++pi) {
			const PlaneDef &pd = planes[pi];
			const std::vector<uint8_t> &sy = rec[pd.slot];
			for (size_t first = 0, step; first < sy.size(); first += step) {
				step = pi < CONN_PLANES ? conn_chunk : attr_chunk_len(first, chunk_syms);
				size_t end = std::min(sy.size(), first + step);
				std::vector<uint8_t> out;
				ChunkEncoder enc(out);
				FreqTable f(256);
				if (has_prior[pi]) { for (uint32_t j = 0; j < 256; ++j) if (prior[pi][j]) f.inc(j, prior[pi][j]); }
				else seed_table(f, pd.init_kind, m);
				for (size_t j = first; j < end; ++j) {
					uint64_t l, h, t = f.total();
					f.range_of(sy[j], l, h);
					enc.encode((uint32_t)l, (uint32_t)h, (uint32_t)t);
					f.inc(sy[j]);
				}
				enc.finish();
				streams.push_back(std::move(out));
			}
		}
		for (auto &st : streams) w.put<uint32_t>((uint32_t)st.size());
		for (auto &st : streams) w.raw(st.data(), st.size());
		if (m.is_shard()) { const uint64_t len = res->bytes.size() - seg_begin; memcpy(res->bytes.data() + seg_len_at, &len, 8); }
	} catch (...) { delete res; throw; }
	return res;
}

// body of a v0.2 container (or of one segment of a v0.3 container) -> *m, whose header fields are set
static void decode_chunked_body(Mesh *m, const uint8_t *p, size_t n)
{
	{
		ByteReader br{ p, p + n };
		Models md(*m);
		uint32_t chunk_syms = br.get<uint32_t>(), conn_chunk = br.get<uint32_t>(), np = br.get<uint32_t>();
		std::vector<PlaneDef> planes = chunked_planes(*m, md);
		if (np != planes.size() || chunk_syms == 0 || conn_chunk == 0) throw std::runtime_error("oracle: bad chunked directory");
		std::vector<uint32_t> nsym(np);
		for (auto &x : nsym) x = br.get<uint32_t>();
		size_t nstreams = 0;
		for (size_t k = 0; k < np; ++k) {
			if (k < CONN_PLANES) nstreams += (size_t)((nsym[k] + (uint64_t)conn_chunk - 1) / conn_chunk);
			else for (size_t f = 0; f < nsym[k]; f += attr_chunk_len(f, chunk_syms)) ++nstreams;
		}
		std::vector<std::array<uint32_t, 256>> prior(np);
		std::vector<char> has_prior(np, 0);
		for (size_t k = 0; k < np; ++k) has_prior[k] = read_prior(br, prior[k].data()) ? 1 : 0;
		uint32_t n_restart = br.get<uint32_t>();   // restart points: an aid for parallel decoders, not needed here
		const bool has_snaps = (n_restart & 0x80000000u) != 0;
		n_restart &= 0x7fffffffu;
		for (uint64_t i = 0; i < (uint64_t)n_restart * 17; ++i) (void)br.get<uint32_t>();
		for (uint32_t i = 0; i < n_restart; ++i) { uint32_t nc = br.get<uint32_t>(); for (uint64_t j = 0; j < 2ull * nc; ++j) (void)br.get<uint32_t>(); }
		// border snapshots (round 6): not needed for this sequential decode either -- but it CHECKS every one of them against the
		// state its own replay is in at that moment (SymReader::snaps, cbm_decode): what a parallel decoder would start from
		std::vector<SymReader::Snap> snaps;
		if (has_snaps) {
			const uint8_t *sec0 = br.p;
			(void)br.get<uint32_t>();   // spacing
			const uint32_t ns = br.get<uint32_t>(), nb = br.get<uint32_t>();
			br.need(nb);
			const uint8_t *q = br.p, *qe = br.p + nb;
			auto get32 = [&]() { return (uint32_t)get_varint(q, qe); };
			for (uint32_t k = 0; k < ns; ++k) {
				SymReader::Snap S;
				for (int g = 0; g < 5; ++g) S.n_grp[g] = get32();
				for (int i = 0; i < 8; ++i) S.n_op[i] = get32();
				S.first_vertex = get32(); S.first_face = get32(); S.first_halfedge = get32();
				const uint32_t nc = get32();
				for (uint32_t j = 0; j < nc; ++j) { uint32_t v = get32(), c = get32(); S.counters.push_back({ v, c }); }
				const uint32_t n_parts = get32(), n_elems = get32();
				for (uint32_t i = 0; i < n_parts; ++i) S.parts.push_back(get32());
				std::vector<int64_t> d = get_runs(q, qe, n_elems);
				for (size_t i = 0; i < d.size(); ++i) S.vtx.push_back((uint32_t)((i ? (int64_t)S.vtx[i - 1] : (int64_t)S.first_vertex - 1) + 1 + d[i]));
				d = get_runs(q, qe, n_elems);
				for (size_t i = 0; i < d.size(); ++i) S.seen.push_back((uint8_t)((i ? (int64_t)S.seen[i - 1] : 3) + d[i]));
				snaps.push_back(std::move(S));
			}
			if (q != qe) throw std::runtime_error("oracle: bytes left in the border snapshots' section");
			br.p += nb;
			while ((br.p - sec0) & 3) (void)br.get<uint8_t>();
		}
		std::vector<uint32_t> nbytes(nstreams);
		for (auto &x : nbytes) x = br.get<uint32_t>();
		std::vector<std::vector<uint8_t>> rec(std::max<size_t>(REC_SLOTS, md.tab.size()));
		size_t si = 0;
		const uint8_t *q = br.p;
		for (size_t k = 0; k < planes.size(); ++k) {
			std::vector<uint8_t> &sy = rec[planes[k].slot];
			sy.resize(nsym[k]);
			for (size_t first = 0, step; first < sy.size(); first += step, ++si) {
				step = k < CONN_PLANES ? conn_chunk : attr_chunk_len(first, chunk_syms);
				size_t end = std::min(sy.size(), first + step);
				if ((size_t)(p + n - q) < nbytes[si]) throw std::runtime_error("oracle: truncated chunked stream");
				ChunkDecoder dec(q, q + nbytes[si]);
				FreqTable f(256);
				if (has_prior[k]) { for (uint32_t j = 0; j < 256; ++j) if (prior[k][j]) f.inc(j, prior[k][j]); }
				else seed_table(f, planes[k].init_kind, *m);
				for (size_t j = first; j < end; ++j) {
					uint64_t l, h, t = f.total();
					uint32_t s = f.find(dec.target((uint32_t)t), l, h);
					dec.consume((uint32_t)l, (uint32_t)h, (uint32_t)t);
					f.inc(s);
					sy[j] = (uint8_t)s;
				}
				q += nbytes[si];
			}
		}
		std::vector<uint8_t> none;
		RangeDecoder rc(none.data(), none.data());
		SymReader rd(md, rc);
		rd.planes = &rec;
		rd.snaps = &snaps;
		if (count_degrees(*m) <= 1) {
			int d = 0;
			for (size_t i = 0; i < m->have_deg.size(); ++i) if (m->have_deg[i]) d = (int)i;
			rd.fixed_numtri = d - 2;
		}
		std::vector<uint32_t> order_v;
		std::vector<char> hdr_deg = m->have_deg;
		cbm_decode(*m, rd, order_v);
		m->have_deg = hdr_deg;
		if (rd.snaps_checked != snaps.size()) throw std::runtime_error("oracle: a border snapshot of the directory lies where the replay never stood");
		if (m->num_face() != m->nf) throw std::runtime_error("oracle: face count mismatch");
		if (m->bind.on) decode_attrs_general(*m, rd, order_v);
		else decode_attrs(*m, rd, order_v);
	}
}

// v0.2: one body.  v0.3 (sharded): every segment is decoded as a mesh of its own -- vertex ids inside a segment are the
// segment's -- and then placed into the numbering of the whole mesh run by run.
static Mesh *decode_chunked(const uint8_t *p, size_t n)
{
	Mesh *m = new Mesh();
	try {
		if (n < 6) throw std::runtime_error("oracle: truncated header");
		ByteReader br{ p, p + n };
		if (p[5] != 3) {
			read_header(br, *m, 2);
			decode_chunked_body(m, br.p, (size_t)(p + n - br.p));
			return m;
		}
		uint32_t gne = 0;
		read_header(br, *m, 3, &gne);
		const uint32_t gnv = m->nv, gnf = m->nf;
		const bool general = m->bind.on;
		const size_t nl = m->lists.size();
		m->foff.assign((size_t)gnf + 1, 0); m->org.assign(gne, 0); m->twin.assign(gne, 0); m->eface.assign(gne, 0);
		if (general) m->bind.corner_attr.assign((size_t)gne * m->bind.nb_corner, 0);   // (the header sized the face / vertex tables)
		std::vector<char> face_seen(gnf, 0);
		const uint32_t nseg = br.get<uint32_t>();
		std::vector<uint64_t> len(nseg);
		for (auto &x : len) x = br.get<uint64_t>();
		const uint8_t *q = br.p;
		for (uint32_t si = 0; si < nseg; ++si) {
			if ((uint64_t)(p + n - q) < len[si]) throw std::runtime_error("oracle: truncated segment");
			ByteReader sr{ q, q + len[si] };
			const uint32_t nr = sr.get<uint32_t>();
			std::vector<std::array<uint32_t, 6>> runs(nr);
			// general bindings: every run is followed by its place in the record numbering of every list (first record, records):
			// the decoder numbers the records of a list in the order they are first coded (attrcode.h:443-531), over all components
			std::vector<std::vector<std::array<uint32_t, 2>>> rrec(nr, std::vector<std::array<uint32_t, 2>>(general ? nl : 0));
			Mesh lm;
			uint32_t lne = 0;
			std::vector<uint32_t> nrec(nl, 0);
			for (uint32_t j = 0; j < nr; ++j) {
				auto &r = runs[j];
				for (auto &x : r) x = sr.get<uint32_t>();
				if ((uint64_t)r[0] + r[3] > gnv || (uint64_t)r[1] + r[4] > gnf || (uint64_t)r[2] + r[5] > gne) throw std::runtime_error("oracle: run outside the mesh");
				lm.nv += r[3]; lm.nf += r[4]; lne += r[5];
				for (size_t l = 0; l < rrec[j].size(); ++l) {
					rrec[j][l][0] = sr.get<uint32_t>(); rrec[j][l][1] = sr.get<uint32_t>();
					if ((uint64_t)rrec[j][l][0] + rrec[j][l][1] > m->lists[l].count) throw std::runtime_error("oracle: records outside their list");
					nrec[l] += rrec[j][l][1];
				}
			}
			lm.have_deg = m->have_deg;
			for (size_t l = 0; l < nl; ++l) {
				const List &L = m->lists[l];
				List D;
				D.target = L.target; D.fmt = L.fmt; D.interps = L.interps; D.bmin = L.bmin; D.bmax = L.bmax;
				D.count = general ? nrec[l] : L.target == TG_FACE ? lm.nf : lm.nv;
				D.data.assign((size_t)D.count * D.fmt.bytes(), 0);
				lm.lists.push_back(std::move(D));
			}
			if (general) {
				lm.bind = m->bind;   // the regions of the whole mesh; the element tables in the segment's sizes
				lm.bind.face_reg.assign(lm.nf, 0); lm.bind.vtx_reg.assign(lm.nv, 0);
				lm.bind.face_attr.assign((size_t)lm.nf * lm.bind.nb_face, 0); lm.bind.vtx_attr.assign((size_t)lm.nv * lm.bind.nb_vtx, 0);
				lm.bind.corner_attr.clear();
			}
			decode_chunked_body(&lm, sr.p, (size_t)(sr.end - sr.p));
			if (lm.num_edge() != lne) throw std::runtime_error("oracle: segment edge count mismatch");
			std::vector<uint32_t> l2g(lm.nv);
			uint32_t cv = 0, cf = 0, ch = 0;
			for (const auto &r : runs) { for (uint32_t i = 0; i < r[3]; ++i) l2g[cv + i] = r[0] + i; cv += r[3]; }
			std::vector<std::vector<uint32_t>> rl2g(general ? nl : 0);
			for (size_t l = 0; l < rl2g.size(); ++l) {
				const size_t st = (size_t)m->lists[l].fmt.bytes();
				uint32_t at = 0;
				for (uint32_t j = 0; j < nr; ++j) {
					for (uint32_t i = 0; i < rrec[j][l][1]; ++i) rl2g[l].push_back(rrec[j][l][0] + i);
					if (st && rrec[j][l][1]) memcpy(m->lists[l].data.data() + (size_t)rrec[j][l][0] * st, lm.lists[l].data.data() + (size_t)at * st, (size_t)rrec[j][l][1] * st);
					at += rrec[j][l][1];
				}
			}
			cv = 0;
			for (const auto &r : runs) {
				for (uint32_t i = 0; i < r[4]; ++i) {
					if (face_seen[r[1] + i]) throw std::runtime_error("oracle: overlapping runs");
					face_seen[r[1] + i] = 1;
					m->foff[(size_t)r[1] + i + 1] = lm.foff[(size_t)cf + i + 1] - ch + r[2];
				}
				m->foff[r[1]] = r[2];
				for (uint32_t i = 0; i < r[5]; ++i) {
					m->org[(size_t)r[2] + i] = l2g.at(lm.org[ch + i]);
					const uint32_t tw = lm.twin[ch + i];
					if (tw < ch || tw >= ch + r[5]) throw std::runtime_error("oracle: twin outside its run");
					m->twin[(size_t)r[2] + i] = tw - ch + r[2];
					m->eface[(size_t)r[2] + i] = lm.eface[ch + i] - cf + r[1];
				}
				if (general) {
					Mesh::Bind &b = m->bind;
					const Mesh::Bind &d = lm.bind;
					for (uint32_t i = 0; i < r[3]; ++i) {
						const int reg = d.vtx_reg[cv + i];
						b.vtx_reg[r[0] + i] = (uint16_t)reg;
						for (int a = 0; a < b.nvtxlists(reg); ++a) b.vtx_attr[(size_t)(r[0] + i) * b.nb_vtx + a] = rl2g[b.vtxlist(reg, a)].at(d.vtx_attr[(size_t)(cv + i) * b.nb_vtx + a]);
					}
					for (uint32_t i = 0; i < r[4]; ++i) {
						const int reg = d.face_reg[cf + i];
						b.face_reg[r[1] + i] = (uint16_t)reg;
						for (int a = 0; a < b.nfacelists(reg); ++a) b.face_attr[(size_t)(r[1] + i) * b.nb_face + a] = rl2g[b.facelist(reg, a)].at(d.face_attr[(size_t)(cf + i) * b.nb_face + a]);
						for (uint32_t h = lm.foff[cf + i]; h < lm.foff[(size_t)cf + i + 1]; ++h)
							for (int a = 0; a < b.ncornerlists(reg); ++a)
								b.corner_attr[(size_t)(h - ch + r[2]) * b.nb_corner + a] = rl2g[b.cornerlist(reg, a)].at(d.corner_attr[(size_t)h * b.nb_corner + a]);
					}
				} else
				for (const List &S : lm.lists) {
					List &D = m->lists[&S - &lm.lists[0]];
					const size_t st = (size_t)S.fmt.bytes();
					if (!st) continue;
					if (S.target == TG_FACE) memcpy(D.data.data() + (size_t)r[1] * st, S.data.data() + (size_t)cf * st, (size_t)r[4] * st);
					else memcpy(D.data.data() + (size_t)r[0] * st, S.data.data() + (size_t)cv * st, (size_t)r[3] * st);
				}
				cv += r[3]; cf += r[4]; ch += r[5];
			}
			q += len[si];
		}
		m->conn_nv = gnv;
	} catch (...) { delete m; throw; }
	return m;
}

// ------------------------------------------------------------------------------------------------
// bounds + requantisation (structs/quant.h:30-242)
// ------------------------------------------------------------------------------------------------
static void set_bounds(List &L)   // quant.h:30-38; max starts at numeric_limits<T>::min() (FLT_MIN for floats, App. B-2)
{
	Fmt d;
	for (int c = 0; c < L.fmt.size(); ++c) d.add(L.fmt.type[c]);
	L.bmin.assign(d.bytes(), 0); L.bmax.assign(d.bytes(), 0);
	for (int c = 0; c < d.size(); ++c) {
		with_type(d.type[c], [&](auto tag) {
			typedef decltype(tag) T;
			T mn = std::numeric_limits<T>::max(), mx = std::numeric_limits<T>::min();
			for (uint32_t i = 0; i < L.count; ++i) {
				T e = ld<T>(L.rec(i) + L.fmt.off[c]);
				mn = e < mn ? e : mn;
				mx = e > mx ? e : mx;
			}
			st<T>(L.bmin.data() + d.off[c], mn);
			st<T>(L.bmax.data() + d.off[c], mx);
		});
	}
}
// read component j of a record as type T with C++ conversion (mixing.h:245-270 get<T>)
template <typename T> static T get_as(const uint8_t *rec, const Fmt &f, int j)
{
	T out = T();
	with_type(f.type[j], [&](auto tag) { typedef decltype(tag) S; out = (T)ld<S>(rec + f.off[j]); });
	return out;
}
static std::vector<uint8_t> compute_scale(const List &L)   // quant.h:46-96
{
	Fmt d;
	for (int c = 0; c < L.fmt.size(); ++c) d.add(L.fmt.type[c]);
	std::vector<uint8_t> spre(d.bytes(), 0), s(d.bytes(), 0);
	int n = d.size();
	for (int c = 0; c < n; ++c)
		with_type(d.type[c], [&](auto tag) {
			typedef decltype(tag) T;
			st<T>(spre.data() + d.off[c], (T)(ld<T>(L.bmax.data() + d.off[c]) - ld<T>(L.bmin.data() + d.off[c])));
			st<T>(s.data() + d.off[c], std::numeric_limits<T>::min());
		});
	std::vector<int> group(n, 0);   // components not covered by any interpretation fall into group 0 (zero-initialised in the reference)
	for (int i = 0; i < L.interps.size(); ++i)
		for (int j = 0; j < L.interps.len[i]; ++j) group[L.interps.off[i] + j] = L.interps.off[i];
	for (int j = 0; j < n; ++j) {
		int k = group[j];
		with_type(d.type[k], [&](auto tag) {
			typedef decltype(tag) T;
			T cur = ld<T>(s.data() + d.off[k]), v = get_as<T>(spre.data(), d, j);
			st<T>(s.data() + d.off[k], std::max(cur, v));
		});
	}
	for (int j = 0; j < n; ++j) {
		int k = group[j];
		with_type(d.type[j], [&](auto tag) { typedef decltype(tag) T; st<T>(s.data() + d.off[j], get_as<T>(s.data(), d, k)); });
	}
	return s;
}
template <typename T> static T rescale(T val, T from, T to, std::true_type) { return val / from * to; }                       // quant.h:98-102
template <typename T> static T rescale(T val, T from, T to, std::false_type) { return val / from * to + val % from * to / from; }   // quant.h:103-107
template <typename T> static T rescale(T val, T from, T to) { return rescale<T>(val, from, to, std::is_floating_point<T>()); }

static uint64_t quantise_scalar_f32(float v, float mn, float sc, int q)   // quant.h:134-136
{
	return (uint64_t)(rescale<float>(v - mn, sc, (float)((1 << (uint32_t)q) - 1)) + 0.5f);
}

static void requant_list(List &L, const Fmt &nf)   // quant.h:114-221: in place, record by record
{
	std::vector<uint8_t> scale = compute_scale(L);
	Fmt d;
	for (int c = 0; c < L.fmt.size(); ++c) d.add(L.fmt.type[c]);
	const Fmt &of = L.fmt;
	for (uint32_t i = 0; i < L.count; ++i) {
		uint8_t *rec = L.rec(i);
		for (int j = 0; j < of.size(); ++j) {
			bool sq = of.quant[j] != 0, dq = nf.quant[j] != 0;
			if (!sq && !dq) continue;   // copy onto itself
			uint64_t q = 0;
			uint8_t *slot = rec + of.off[j];
			const uint8_t *mn = L.bmin.data() + d.off[j], *sc = scale.data() + d.off[j];
			if (sq) {
				switch (of.stype[j]) {
				case T_ULONG: q = ld<uint64_t>(slot); break;
				case T_UINT: q = ld<uint32_t>(slot); break;
				case T_USHORT: q = ld<uint16_t>(slot); break;
				case T_UCHAR: q = ld<uint8_t>(slot); break;
				default: throw std::runtime_error("Invalid quantization type");
				}
			} else {
				int nq = nf.quant[j];
				switch (of.stype[j]) {
				case T_FLOAT: q = (uint64_t)(rescale<float>(ld<float>(slot) - ld<float>(mn), ld<float>(sc), (float)((1 << (uint32_t)nq) - 1)) + 0.5f); break;
				case T_DOUBLE: q = (uint64_t)(rescale<double>(ld<double>(slot) - ld<double>(mn), ld<double>(sc), (double)((1 << (uint64_t)nq) - 1)) + 0.5); break;
				case T_ULONG: q = rescale<uint64_t>(ld<uint64_t>(slot) - ld<uint64_t>(mn), ld<uint64_t>(sc), (uint64_t)((1 << (uint64_t)nq) - 1)); break;
				case T_LONG: q = (uint64_t)rescale<int64_t>(ld<int64_t>(slot) - ld<int64_t>(mn), ld<int64_t>(sc), (int64_t)((1 << (uint64_t)nq) - 1)); break;
				case T_UINT: q = rescale<uint32_t>(ld<uint32_t>(slot) - ld<uint32_t>(mn), ld<uint32_t>(sc), (uint32_t)((1 << (uint32_t)nq) - 1)); break;
				case T_INT: q = (uint64_t)rescale<int32_t>(ld<int32_t>(slot) - ld<int32_t>(mn), ld<int32_t>(sc), (int32_t)((1 << (uint32_t)nq) - 1)); break;
				case T_USHORT: q = rescale<uint16_t>((uint16_t)(ld<uint16_t>(slot) - ld<uint16_t>(mn)), ld<uint16_t>(sc), (uint16_t)((1 << (uint32_t)nq) - 1)); break;
				case T_SHORT: q = (uint64_t)rescale<int16_t>((int16_t)(ld<int16_t>(slot) - ld<int16_t>(mn)), ld<int16_t>(sc), (int16_t)((1 << (uint32_t)nq) - 1)); break;
				case T_UCHAR: q = rescale<uint8_t>((uint8_t)(ld<uint8_t>(slot) - ld<uint8_t>(mn)), ld<uint8_t>(sc), (uint8_t)((1 << (uint32_t)nq) - 1)); break;
				case T_CHAR: q = (uint64_t)rescale<int8_t>((int8_t)(ld<int8_t>(slot) - ld<int8_t>(mn)), ld<int8_t>(sc), (int8_t)((1 << (uint32_t)nq) - 1)); break;
				default: break;
				}
			}
			if (sq && dq) q = rescale<uint64_t>(q, (uint64_t)((1 << (uint32_t)of.quant[j]) - 1), (uint64_t)((1 << (uint32_t)nf.quant[j]) - 1));
			if (dq) {
				switch (nf.stype[j]) {
				case T_ULONG: st<uint64_t>(slot, q); break;
				case T_UINT: st<uint32_t>(slot, (uint32_t)q); break;
				case T_USHORT: st<uint16_t>(slot, (uint16_t)q); break;
				case T_UCHAR: st<uint8_t>(slot, (uint8_t)q); break;
				default: throw std::runtime_error("Invalid quantization type");
				}
			} else {   // dequantise (quant.h:180-212)
				int oq = of.quant[j];
				switch (nf.stype[j]) {
				case T_FLOAT: st<float>(slot, rescale<float>((float)q, (float)((1 << (uint32_t)oq) - 1), ld<float>(sc)) + ld<float>(mn)); break;
				case T_DOUBLE: st<double>(slot, rescale<double>((double)q, (double)((1 << (uint64_t)oq) - 1), ld<double>(sc)) + ld<double>(mn)); break;
				case T_UINT: st<uint32_t>(slot, rescale<uint32_t>((uint32_t)q, (uint32_t)((1 << (uint32_t)oq) - 1), ld<uint32_t>(sc)) + ld<uint32_t>(mn)); break;
				case T_INT: st<int32_t>(slot, rescale<int32_t>((int32_t)q, (int32_t)((1 << (uint32_t)oq) - 1), ld<int32_t>(sc)) + ld<int32_t>(mn)); break;
				case T_USHORT: st<uint16_t>(slot, (uint16_t)(rescale<uint16_t>((uint16_t)q, (uint16_t)((1 << (uint32_t)oq) - 1), ld<uint16_t>(sc)) + ld<uint16_t>(mn))); break;
				case T_SHORT: st<int16_t>(slot, (int16_t)(rescale<int16_t>((int16_t)q, (int16_t)((1 << (uint32_t)oq) - 1), ld<int16_t>(sc)) + ld<int16_t>(mn))); break;
				case T_UCHAR: st<uint8_t>(slot, (uint8_t)(rescale<uint8_t>((uint8_t)q, (uint8_t)((1 << (uint32_t)oq) - 1), ld<uint8_t>(sc)) + ld<uint8_t>(mn))); break;
				case T_CHAR: st<int8_t>(slot, (int8_t)(rescale<int8_t>((int8_t)q, (int8_t)((1 << (uint32_t)oq) - 1), ld<int8_t>(sc)) + ld<int8_t>(mn))); break;
				default: throw std::runtime_error("oracle: dequantisation of 8-byte integer types is not restated");
				}
			}
		}
	}
	L.fmt = nf;
}

static void requant(Mesh &m, const int *tr, int n, bool clear)   // main.cc:74-91 + quant.h:222-242
{
	std::vector<Fmt> nf;
	for (auto &L : m.lists) nf.push_back(L.fmt);
	struct Q { int l, o, q; };
	std::vector<Q> qs;
	for (int i = 0; i < n; ++i) {
		int l = tr[3 * i], o = tr[3 * i + 1], q = tr[3 * i + 2];
		if (q < 0) throw std::runtime_error("Invalid quantization bits");
		if (l < 0 || l >= (int)m.lists.size()) throw std::runtime_error("Invalid list index");
		const Fmt &f = m.lists[l].fmt;
		if (o == -1) {
			for (int c = 0; c < f.size(); ++c) {
				if (q > TSIZE[f.type[c]] * 8) throw std::runtime_error("Invalid quantization bits");
				qs.push_back(Q{ l, c, q });
			}
		} else {
			if (o < 0 || o >= f.size()) throw std::runtime_error("Invalid attribute index");
			if (q > TSIZE[f.type[o]] * 8) throw std::runtime_error("Invalid quantization bits");
			qs.push_back(Q{ l, o, q });
		}
	}
	if (clear) for (auto &f : nf) for (int c = 0; c < f.size(); ++c) f.setquant(c, 0);
	for (const Q &q : qs) nf[q.l].setquant(q.o, q.q);
	for (size_t l = 0; l < m.lists.size(); ++l) requant_list(m.lists[l], nf[l]);
}

// ------------------------------------------------------------------------------------------------
// PLY subset reader (formats/ply/reader.cc:36-429) + half-edge twin matching (structs/conn.h:172-234)
// ------------------------------------------------------------------------------------------------
struct PlyProp { std::string name; Type type; Type len_type; };
struct PlyElem { std::string name; long len; std::vector<PlyProp> props; };

static Type ply_type(const std::string &s)   // reader.cc:70-81
{
	if (s == "float" || s == "float32") return T_FLOAT;
	if (s == "double" || s == "float64") return T_DOUBLE;
	if (s == "uint" || s == "uint32") return T_UINT;
	if (s == "int" || s == "int32") return T_INT;
	if (s == "ushort" || s == "uint16") return T_USHORT;
	if (s == "short" || s == "int16") return T_SHORT;
	if (s == "uchar" || s == "uint8") return T_UCHAR;
	if (s == "char" || s == "int8") return T_CHAR;
	throw std::runtime_error("Invalid data type");
}
static const std::unordered_map<std::string, int> &well_known()   // reader.cc:36-57: canonical order of well-known names
{
	static const std::unordered_map<std::string, int> m = {
		{ "x", 0 }, { "y", 1 }, { "z", 2 }, { "w", 3 }, { "nx", 4 }, { "ny", 5 }, { "nz", 6 }, { "nw", 7 },
		{ "red", 8 }, { "green", 9 }, { "blue", 10 }, { "alpha", 11 },
		{ "ambient_red", 12 }, { "ambient_green", 13 }, { "ambient_blue", 14 }, { "ambient_alpha", 15 }, { "ambient_coeff", 16 },
		{ "diffuse_red", 17 }, { "diffuse_green", 18 }, { "diffuse_blue", 19 }, { "diffuse_alpha", 20 }, { "diffuse_coeff", 21 },
		{ "specular_red", 22 }, { "specular_green", 23 }, { "specular_blue", 24 }, { "specular_alpha", 25 }, { "specular_power", 26 }, { "specular_coeff", 27 },
		{ "u", 28 }, { "tu", 28 }, { "v", 29 }, { "tv", 29 }, { "tw", 30 }, { "value", 31 }, { "scale", 31 }, { "confidence", 32 },
	};
	return m;
}
static int interp_of_weight(int w)   // reader.cc:58-68
{
	if (w < 4) return I_POS;
	if (w < 8) return I_NORMAL;
	if (w < 12) return I_COLOR;
	if (w < 17) return I_COLOR_AMBIENT;
	if (w < 22) return I_COLOR_DIFFUSE;
	if (w < 28) return I_COLOR_SPECULAR;
	if (w < 31) return I_TEX;
	if (w == 31) return I_SCALE;
	return I_CONFIDENCE;
}
static const int WKEND = 33;

// reader.cc:130-168: attribute properties sorted by canonical weight; unknown names keep file order after them
static void layout_element(const PlyElem &el, std::vector<int> &perm, Fmt &fmt, Interps &interps)
{
	int other = WKEND, valid = 0;
	std::vector<int> weight(el.props.size(), std::numeric_limits<int>::max());
	for (size_t i = 0; i < el.props.size(); ++i) {
		if (el.props[i].len_type != T_NONE) continue;
		auto it = well_known().find(el.props[i].name);
		weight[i] = it == well_known().end() ? other++ : it->second;
		++valid;
	}
	std::vector<int> inv(el.props.size());
	for (size_t i = 0; i < inv.size(); ++i) inv[i] = (int)i;
	std::stable_sort(inv.begin(), inv.end(), [&](int a, int b) { return weight[a] < weight[b]; });
	perm.assign(el.props.size(), -1);
	int oid = 0;
	for (int i = 0; i < valid; ++i) {
		int t = inv[i];
		fmt.add(el.props[t].type);
		int w = weight[t];
		int interp = w >= WKEND ? I_OTHER + oid++ : interp_of_weight(w);
		interps.append(interp, i);
		if (interp >= I_OTHER) interps.describe(interp, el.props[t].name);
		perm[t] = i;
	}
}

struct Cursor {
	const uint8_t *p, *end;
	bool eof() const { return p >= end; }
	void skip_ws() { while (p < end && isspace(*p)) ++p; }
	std::string token() { skip_ws(); const uint8_t *b = p; while (p < end && !isspace(*p)) ++p; return std::string((const char*)b, (const char*)p); }
	void skip_line() { while (p < end && *p != '\n') ++p; if (p < end) ++p; }
};
template <typename T> static T bswap(T v) { uint8_t *b = (uint8_t*)&v; std::reverse(b, b + sizeof(T)); return v; }

struct ValueReader {   // reader.cc:245-321
	int mode;   // 0 ascii, 1 LE, 2 BE
	Cursor &c;
	uint64_t read(uint8_t *dst, Type t)
	{
		if (t == T_NONE) return 1;
		if (mode == 0) {
			std::string tok = c.token();
			if (tok.empty()) throw std::runtime_error("oracle: truncated PLY");
			switch (t) {
			case T_CHAR: { int8_t v = (int8_t)strtoll(tok.c_str(), 0, 10); st(dst, v); return (uint64_t)v; }
			case T_UCHAR: { uint8_t v = (uint8_t)strtoull(tok.c_str(), 0, 10); st(dst, v); return v; }
			case T_SHORT: { int16_t v = (int16_t)strtoll(tok.c_str(), 0, 10); st(dst, v); return (uint64_t)v; }
			case T_USHORT: { uint16_t v = (uint16_t)strtoull(tok.c_str(), 0, 10); st(dst, v); return v; }
			case T_INT: { int32_t v = (int32_t)strtoll(tok.c_str(), 0, 10); st(dst, v); return (uint64_t)v; }
			case T_UINT: { uint32_t v = (uint32_t)strtoull(tok.c_str(), 0, 10); st(dst, v); return v; }
			case T_FLOAT: { float v = (float)strtod(tok.c_str(), 0); st(dst, v); return (uint64_t)v; }
			case T_DOUBLE: { double v = strtod(tok.c_str(), 0); st(dst, v); return (uint64_t)v; }
			default: throw std::runtime_error("oracle: bad PLY type");
			}
		}
		int n = TSIZE[t];
		if ((size_t)(c.end - c.p) < (size_t)n) throw std::runtime_error("oracle: truncated PLY");
		uint8_t tmp[8];
		memcpy(tmp, c.p, n);
		c.p += n;
		if (mode == 2) std::reverse(tmp, tmp + n);
		memcpy(dst, tmp, n);
		switch (t) {
		case T_CHAR: return (uint64_t)ld<int8_t>(tmp);
		case T_UCHAR: return ld<uint8_t>(tmp);
		case T_SHORT: return (uint64_t)ld<int16_t>(tmp);
		case T_USHORT: return ld<uint16_t>(tmp);
		case T_INT: return (uint64_t)ld<int32_t>(tmp);
		case T_UINT: return ld<uint32_t>(tmp);
		case T_FLOAT: return (uint64_t)ld<float>(tmp);
		case T_DOUBLE: return (uint64_t)ld<double>(tmp);
		default: return 0;
		}
	}
};

struct PairHash { size_t operator()(const std::pair<uint32_t, uint32_t> &x) const { return std::hash<uint32_t>()(x.first) * 0x9e3779b97f4a7c15ull + x.second; } };

// conn.h:201-232: a directed edge (a,b) is matched with a pending (b,a); a duplicate pending key is NOT replaced
struct TwinMatcher {
	std::unordered_map<std::pair<uint32_t, uint32_t>, he_t, PairHash> pending;
	Mesh &m;
	explicit TwinMatcher(Mesh &mm) : m(mm) {}
	void edge(uint32_t a, uint32_t b, he_t e)
	{
		auto it = pending.find(std::make_pair(b, a));
		if (it != pending.end()) { m.merge(it->second, e); pending.erase(it); }
		else pending.insert(std::make_pair(std::make_pair(a, b), e));
	}
};

static Mesh *read_ply(const uint8_t *buf, size_t n)
{
	Cursor c{ buf, buf + n };
	std::vector<PlyElem> elems;
	int mode = -1;
	std::string id;
	do {   // reader.cc:197-243
		id = c.token();
		if (c.eof() && id.empty()) throw std::runtime_error("oracle: PLY header without end_header");
		if (id == "ply") {}
		else if (id == "format") {
			std::string f = c.token();
			if (f == "ascii") mode = 0; else if (f == "binary_little_endian") mode = 1; else if (f == "binary_big_endian") mode = 2;
			else throw std::runtime_error("Invlaid format");
			c.skip_line();
		} else if (id == "comment") c.skip_line();
		else if (id == "element") { PlyElem e; e.name = c.token(); e.len = atol(c.token().c_str()); elems.push_back(e); }
		else if (id == "property") {
			if (elems.empty()) throw std::runtime_error("Invlaid property");
			std::string ty = c.token();
			Type lt = T_NONE;
			if (ty == "list") { lt = ply_type(c.token()); ty = c.token(); }
			std::string nm = c.token();
			elems.back().props.push_back(PlyProp{ nm, ply_type(ty), lt });
		} else if (id != "end_header") c.skip_line();
	} while (id != "end_header");
	c.skip_line();
	if (mode < 0) throw std::runtime_error("oracle: PLY without format line");

	int fi = -1, vi = -1;
	for (size_t i = 0; i < elems.size(); ++i) { if (elems[i].name == "face") fi = (int)i; if (elems[i].name == "vertex") vi = (int)i; }
	if (fi < 0 || vi < 0) throw std::runtime_error("oracle: PLY needs vertex and face elements");
	Mesh *m = new Mesh();
	try {
		std::vector<int> perm[2];
		int idx[2] = { fi, vi };
		for (int k = 0; k < 2; ++k) {   // reader.cc:388-400: list 0 = face attributes, list 1 = vertex attributes
			List L;
			layout_element(elems[idx[k]], perm[k], L.fmt, L.interps);
			L.target = k == 0 ? TG_FACE : TG_VTX;
			L.count = (uint32_t)elems[idx[k]].len;
			L.data.assign((size_t)L.count * L.fmt.bytes(), 0);
			m->lists.push_back(std::move(L));
		}
		m->nf = (uint32_t)elems[fi].len;
		m->nv = (uint32_t)elems[vi].len;
		int vidx = -1;
		for (size_t k = 0; k < elems[fi].props.size(); ++k) if (elems[fi].props[k].name == "vertex_indices") vidx = (int)k;
		ValueReader rd{ mode, c };
		TwinMatcher tm(*m);
		tm.pending.reserve((size_t)m->nf * 3);
		uint8_t ign[8];
		for (size_t ei = 0; ei < elems.size(); ++ei) {   // reader.cc:323-380
			const PlyElem &el = elems[ei];
			bool attr_el = (int)ei == fi || (int)ei == vi;
			int list = (int)ei == fi ? 0 : 1;
			for (long j = 0; j < el.len; ++j) {
				for (size_t k = 0; k < el.props.size(); ++k) {
					const PlyProp &pp = el.props[k];
					if (!attr_el || perm[list][k] == -1) {
						uint64_t len = rd.read(ign, pp.len_type);
						if (attr_el && pp.len_type != T_NONE && (int)ei == fi && (int)k == vidx) {
							uint32_t f = m->add_face((int)len);
							uint32_t first = 0, last = NOVTX;
							for (uint64_t l = 0; l < len; ++l) {
								uint32_t v = (uint32_t)rd.read(ign, pp.type);
								he_t e = m->foff[f] + (uint32_t)l;
								m->set_org(e, v);
								if (last != NOVTX) tm.edge(last, v, e - 1); else first = v;
								last = v;
							}
							tm.edge(last, first, m->foff[f] + (uint32_t)len - 1);
						} else for (uint64_t l = 0; l < len; ++l) rd.read(ign, pp.type);
					} else {
						List &L = m->lists[list];
						rd.read(L.rec((uint32_t)j) + L.fmt.off[perm[list][k]], pp.type);
					}
				}
			}
		}
		if (m->num_face() != m->nf) throw std::runtime_error("oracle: PLY face element without vertex_indices");
		for (uint32_t v : m->org) if (v >= m->nv) throw std::runtime_error("oracle: PLY vertex index out of range");
		for (auto &L : m->lists) set_bounds(L);   // reader.cc:428
	} catch (...) { delete m; throw; }
	return m;
}

// ------------------------------------------------------------------------------------------------
// OBJ reader (formats/obj/reader.rl:27-80 grammar, :108-299 actions).  The grammar is restated by hand, with its quirks:
//   * the sign of an exponent is required and then ignored ("1e-2" reads as 100; "1e2" does not parse) -- rl:36-37,44
//   * "v" takes 3, 4, 6, 7 or 8 numbers; "vt" 2 or 3; "vn" 3; "o s g l p" need at least one blank after the keyword
//   * anything else, and any stray '\r', is "Unable to parse this OBJ file"; a last line without '\n' is dropped
//   * material names keep trailing blanks; "usemtl" of an unknown material falls back to material 0
//   * the face-region lookup key (tex_list << 3) | normal_list with 9 = none collides for some list pairs (rl:229)
//   * a corner written "v/t" names normal t as well ("n too big" when there are fewer normals); "v/t/" names the texture only
// Not restated: "nan" / "inf" literals (the sign state of "inf" is undefined in the reference), faces with fewer than 3 corners
// and faces whose corners do not all carry the same kinds of index (the reference reads past its index arrays there).
// ------------------------------------------------------------------------------------------------
struct ObjLine {
	const uint8_t *p, *end;
	bool at_end() const { return p == end; }
	bool blank(uint8_t c) const { return c == ' ' || c == '\t'; }
	bool sp() { if (p == end || !blank(*p)) return false; while (p != end && blank(*p)) ++p; return true; }
	void fail() const { throw std::runtime_error("Unable to parse this OBJ file"); }
	bool digit() const { return p != end && *p >= '0' && *p <= '9'; }
	float number()   // rl:32-50
	{
		double sign = 1, val = 0, fraction = 0, denom = 1, ex = 0, expmul = 1;
		if (p != end && (*p == 'n' || *p == 'N' || *p == 'i' || *p == 'I')) throw std::runtime_error("oracle: nan / inf literals in an OBJ file are not restated");
		if (p != end && (*p == '+' || *p == '-')) { sign = *p == '-' ? -1 : 1; ++p; }
		if (digit()) {
			while (digit()) { val *= 10; val += *p - '0'; ++p; }
			if (p != end && *p == '.') { ++p; while (digit()) { fraction *= 10; fraction += *p - '0'; denom *= 10; ++p; } }
		} else if (p != end && *p == '.') {
			++p;
			if (!digit()) fail();
			while (digit()) { fraction *= 10; fraction += *p - '0'; denom *= 10; ++p; }
		} else fail();
		if (p != end && (*p == 'e' || *p == 'E')) {
			++p;
			if (p == end || (*p != '+' && *p != '-')) fail();
			++p;
			if (!digit()) fail();
			while (digit()) { ex *= 10; ex += *p - '0'; ++p; }
			expmul = std::pow(10.0, ex);
		}
		val += fraction / denom;
		val *= sign * expmul;
		return (float)val;
	}
	bool index(int &out)   // rl:52
	{
		const uint8_t *q = p;
		bool neg = false;
		if (q != end && *q == '-') { neg = true; ++q; }
		if (q == end || *q < '0' || *q > '9') return false;
		int v = 0;
		while (q != end && *q >= '0' && *q <= '9') { v = (int)((unsigned)v * 10u + (unsigned)(*q - '0')); ++q; }
		p = q;
		out = neg ? -v : v;
		return true;
	}
};

static int obj_index(int n, int size)   // rl:91-103
{
	if (n == 0) throw std::runtime_error("index cannot be 0");
	if (n > size) throw std::runtime_error("n too big");
	if (n < 0) {
		if (size + n < 0) throw std::runtime_error("n too small");
		return size + n;
	}
	return n - 1;
}

struct ObjBuilder {
	enum { VERTEX, TEX, NORMAL, IL = 9 };
	Mesh &m;
	TwinMatcher tm;
	int attr_lists[3][9];
	int vtx_reg[9];
	std::vector<std::array<int, 256>> face_regs;
	std::vector<std::pair<int, uint32_t>> tex_loc, normal_loc;
	std::unordered_map<std::string, int> mtls;
	int cur_mtl = 0;
	std::string base;
	explicit ObjBuilder(Mesh &mesh, const std::string &dir) : m(mesh), tm(mesh), base(dir)
	{
		for (auto &row : attr_lists) for (int &x : row) x = IL;
		for (int &x : vtx_reg) x = -1;
		face_regs.emplace_back();
		face_regs[0].fill(-1);
		m.bind.on = true;
		m.bind.nb_face = 0; m.bind.nb_vtx = 1; m.bind.nb_corner = 2;   // rl:257
	}
	int init_attr(int attr, int n)   // rl:132-148
	{
		int &list = attr_lists[attr][n];
		if (list == IL) {
			static const int lut[3] = { I_POS, I_TEX, I_NORMAL };
			List L;
			for (int i = 0; i < n; ++i) {
				L.fmt.add(T_FLOAT);
				int interp = lut[attr];
				if (attr == VERTEX && n > 4) interp = ((n == 8 && i >= 4) || i >= 3) ? I_COLOR : interp;
				L.interps.append(interp, i);
			}
			L.target = attr == VERTEX ? TG_VTX : TG_CORNER;
			m.lists.push_back(std::move(L));
			list = (int)m.lists.size() - 1;
		}
		return list;
	}
	uint32_t write_attr(int list, const float *c, int n)
	{
		List &L = m.lists[list];
		L.data.resize(L.data.size() + (size_t)n * 4);
		memcpy(L.rec(L.count), c, (size_t)n * 4);
		return L.count++;
	}
	void usemtl(const std::string &name) { auto it = mtls.find(name); cur_mtl = it == mtls.end() ? 0 : it->second; }   // rl:160-169
	void mtllib(const std::string &name)   // rl:170-190
	{
		std::ifstream is(base + "/" + name);
		if (!is) return;
		while (!is.eof()) {
			std::string id;
			is >> id;
			if (id == "newmtl") {
				std::string nm;
				is >> nm;
				mtls[nm] = (int)face_regs.size();
				face_regs.emplace_back();
				face_regs.back().fill(-1);
			} else is.ignore(std::numeric_limits<std::streamsize>::max(), '\n');
		}
	}
	void vertex(const float *c, int n)   // rl:191-204
	{
		int list = init_attr(VERTEX, n);
		uint32_t aidx = write_attr(list, c, n);
		if (vtx_reg[n] < 0) { vtx_reg[n] = m.bind.add_vtx_region(1); m.bind.reg_vtxlist[m.bind.off_vtxlist[vtx_reg[n]]] = (uint16_t)list; }
		m.bind.vtx_reg.push_back((uint16_t)vtx_reg[n]);
		m.bind.vtx_attr.push_back(aidx);
		++m.nv;
	}
	void tex(const float *c, int n) { int l = init_attr(TEX, n); tex_loc.push_back(std::make_pair(l, write_attr(l, c, n))); }
	void normal(const float *c, int n) { int l = init_attr(NORMAL, n); normal_loc.push_back(std::make_pair(l, write_attr(l, c, n))); }
	void face(const std::vector<int> &vi, const std::vector<int> &ti, const std::vector<int> &ni)   // rl:217-249
	{
		const int corners = (int)vi.size();
		const bool has_t = !ti.empty(), has_n = !ni.empty();
		if (corners < 3) throw std::runtime_error("oracle: OBJ faces with fewer than 3 corners are not restated");
		if ((has_t && (int)ti.size() != corners) || (has_n && (int)ni.size() != corners)) throw std::runtime_error("oracle: OBJ face with unevenly indexed corners is not restated");
		int tex_l = IL, normal_l = IL;
		if (has_t) tex_l = tex_loc[ti[0]].first;
		if (has_n) normal_l = normal_loc[ni[0]].first;
		for (int c = 1; c < corners; ++c) {
			if (has_t && tex_loc[ti[c]].first != tex_l) throw std::runtime_error("Inconsistent texture attribute types in face");
			if (has_n && normal_loc[ni[c]].first != normal_l) throw std::runtime_error("Inconsistent normal attribute types in face");
		}
		int key = (tex_l << 3) | normal_l;
		int r = face_regs[cur_mtl][key];
		const int tex_a = 0, normal_a = has_t ? 1 : 0;
		if (r < 0) {
			r = face_regs[cur_mtl][key] = m.bind.add_face_region(0, (has_t ? 1 : 0) + (has_n ? 1 : 0));
			if (has_t) m.bind.reg_cornerlist[m.bind.off_cornerlist[r] + tex_a] = (uint16_t)tex_l;
			if (has_n) m.bind.reg_cornerlist[m.bind.off_cornerlist[r] + normal_a] = (uint16_t)normal_l;
		}
		uint32_t f = m.add_face(corners);
		m.bind.face_reg.push_back((uint16_t)r);
		m.bind.corner_attr.resize(m.bind.corner_attr.size() + (size_t)corners * 2, 0);
		uint32_t first = 0, last = NOVTX;
		for (int i = 0; i < corners; ++i) {
			he_t e = m.foff[f] + (uint32_t)i;
			uint32_t v = (uint32_t)vi[i];
			m.set_org(e, v);
			if (has_t) m.bind.corner_attr[(size_t)e * 2 + tex_a] = tex_loc[ti[i]].second;
			if (has_n) m.bind.corner_attr[(size_t)e * 2 + normal_a] = normal_loc[ni[i]].second;
			if (last != NOVTX) tm.edge(last, v, e - 1); else first = v;
			last = v;
		}
		tm.edge(last, first, m.foff[f] + (uint32_t)corners - 1);
		++m.nf;
	}
};

static Mesh *read_obj(const uint8_t *buf, size_t n, const std::string &dir)
{
	Mesh *m = new Mesh();
	try {
		ObjBuilder ob(*m, dir);
		int count[3] = { 0, 0, 0 };
		std::vector<int> fi[3];
		float coords[8];
		const uint8_t *p = buf, *end = buf + n;
		while (p != end) {
			const uint8_t *nl = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));
			if (!nl) break;                                   // a last line without '\n' never completes
			const uint8_t *le = nl;
			if (le != p && le[-1] == '\r') --le;              // eol = sp? '\r'? '\n'
			ObjLine L{ p, le };
			p = nl + 1;
			for (const uint8_t *q = L.p; q != L.end; ++q) if (*q == '\r') L.fail();
			auto eol = [&]() { L.sp(); if (!L.at_end()) L.fail(); };
			auto keyword = [&](const char *kw) {
				size_t k = strlen(kw);
				if ((size_t)(L.end - L.p) < k || memcmp(L.p, kw, k) != 0) return false;
				if ((size_t)(L.end - L.p) > k && !L.blank(L.p[k])) return false;
				return true;
			};
			if (L.at_end()) continue;
			if (L.blank(*L.p)) { eol(); continue; }           // '' eol with blanks only
			if (*L.p == '#') continue;
			if (keyword("usemtl") || keyword("mtllib")) {
				const bool use = L.p[0] == 'u';
				L.p += 6;
				const uint8_t *a = L.p;
				if (!L.sp()) L.fail();
				std::string s((const char*)L.p, (const char*)L.end);
				if (s.empty()) {                               // only blanks followed the keyword: one is sp, the last one is the name
					if (L.end - a < 2) L.fail();
					s.assign(1, (char)L.end[-1]);
				}
				if (use) ob.usemtl(s); else ob.mtllib(s);
				continue;
			}
			if (keyword("vt") || keyword("vn") || keyword("v")) {
				int kind = keyword("vt") ? ObjBuilder::TEX : keyword("vn") ? ObjBuilder::NORMAL : ObjBuilder::VERTEX;
				L.p += kind == ObjBuilder::VERTEX ? 1 : 2;
				int nc = 0;
				while (L.sp()) {
					if (L.at_end()) break;
					if (nc == 8) L.fail();
					coords[nc++] = L.number();
				}
				if (!L.at_end()) L.fail();
				if (kind == ObjBuilder::VERTEX) { if (nc != 3 && nc != 4 && nc != 6 && nc != 7 && nc != 8) L.fail(); ++count[0]; ob.vertex(coords, nc); }
				else if (kind == ObjBuilder::TEX) { if (nc != 2 && nc != 3) L.fail(); ++count[1]; ob.tex(coords, nc); }
				else { if (nc != 3) L.fail(); ++count[2]; ob.normal(coords, nc); }
				continue;
			}
			if (keyword("f")) {
				L.p += 1;
				fi[0].clear(); fi[1].clear(); fi[2].clear();
				while (L.sp()) {
					if (L.at_end()) break;
					int idx;
					if (!L.index(idx)) L.fail();
					fi[0].push_back(obj_index(idx, count[0]));
					int slashes = 0, last_with_index = 0;
					for (int k = 1; k <= 2; ++k) {
						if (L.at_end() || *L.p != '/') break;
						++L.p; ++slashes;
						if (L.index(idx)) { fi[k].push_back(obj_index(idx, count[k])); last_with_index = k; }
					}
					// "v/t" also matches as "v" + no texture + "/n": the generated scanner runs both and so the index lands in the
					// normals as well (rl:54-56); "v/t/" does not
					if (slashes == 1 && last_with_index == 1) fi[2].push_back(obj_index(idx, count[2]));
				}
				if (!L.at_end() || fi[0].size() < 2) L.fail();
				ob.face(fi[0], fi[1], fi[2]);
				continue;
			}
			if (keyword("o") || keyword("s") || keyword("g") || keyword("l") || keyword("p")) {
				if (L.end - L.p < 2) L.fail();                // the keyword needs a blank after it
				continue;
			}
			L.fail();
		}
		m->conn_nv = std::max(m->conn_nv, m->nv);
		for (auto &L : m->lists) set_bounds(L);   // rl:294
	} catch (...) { delete m; throw; }
	return m;
}

}   // namespace ho

// ------------------------------------------------------------------------------------------------
// C interface
// ------------------------------------------------------------------------------------------------
struct ho_mesh { ho::Mesh m; };
struct ho_result { ho::Result r; };

#define HO_TRY try {
#define HO_CATCH(ret) } catch (const std::exception &e) { ho::g_err = e.what(); return ret; }

extern "C" {

const char *ho_last_error(void) { return ho::g_err.c_str(); }

ho_mesh *ho_mesh_from_ply(const uint8_t *ply, size_t n)
{
	HO_TRY
	ho::Mesh *m = ho::read_ply(ply, n);
	ho_mesh *h = new ho_mesh{ std::move(*m) };
	delete m;
	return h;
	HO_CATCH(nullptr)
}
ho_mesh *ho_mesh_from_hry(const uint8_t *hry, size_t n)
{
	HO_TRY
	ho::Mesh *m = ho::decode(hry, n);
	ho_mesh *h = new ho_mesh{ std::move(*m) };
	delete m;
	return h;
	HO_CATCH(nullptr)
}
ho_mesh *ho_mesh_from_obj(const uint8_t *obj, size_t n, const char *dir)
{
	HO_TRY
	ho::Mesh *m = ho::read_obj(obj, n, dir ? dir : "");
	ho_mesh *h = new ho_mesh{ std::move(*m) };
	delete m;
	return h;
	HO_CATCH(nullptr)
}
int ho_mesh_general(const ho_mesh *m) { return m->m.bind.on ? 1 : 0; }
void ho_mesh_make_general(ho_mesh *m) { m->m.make_general(); }
int ho_mesh_nregions(const ho_mesh *m, int which) { return !m->m.bind.on ? 1 : which == 0 ? m->m.bind.nregs_face() : m->m.bind.nregs_vtx(); }
int ho_mesh_region_lists(const ho_mesh *m, int kind, int r, uint16_t *out, int cap)
{
	const ho::Mesh::Bind &b = m->m.bind;
	if (!b.on) { if (kind == 2) return 0; if (cap > 0) out[0] = kind == 0 ? 0 : 1; return 1; }
	int n = kind == 0 ? b.nfacelists(r) : kind == 1 ? b.nvtxlists(r) : b.ncornerlists(r);
	for (int a = 0; a < n && a < cap; ++a) out[a] = (uint16_t)(kind == 0 ? b.facelist(r, a) : kind == 1 ? b.vtxlist(r, a) : b.cornerlist(r, a));
	return n;
}
const uint16_t *ho_mesh_regions_of(const ho_mesh *m, int which) { return which == 0 ? m->m.bind.face_reg.data() : m->m.bind.vtx_reg.data(); }
int ho_mesh_nslots(const ho_mesh *m, int kind) { const ho::Mesh::Bind &b = m->m.bind; return kind == 0 ? b.nb_face : kind == 1 ? b.nb_vtx : b.nb_corner; }
const uint32_t *ho_mesh_bindings(const ho_mesh *m, int kind) { const ho::Mesh::Bind &b = m->m.bind; return kind == 0 ? b.face_attr.data() : kind == 1 ? b.vtx_attr.data() : b.corner_attr.data(); }
ho_mesh *ho_mesh_clone(const ho_mesh *m) { return new ho_mesh{ m->m }; }
void ho_mesh_set_shard(ho_mesh *m, uint32_t g_nv, uint32_t g_nf, uint32_t g_ne, const uint32_t *seeds, size_t nseeds, const uint32_t *runs, size_t nruns)
{
	m->m.g_nv = g_nv; m->m.g_nf = g_nf; m->m.g_ne = g_ne;
	m->m.seeds.assign(seeds, seeds + nseeds);
	m->m.runs.resize(nruns);
	for (size_t i = 0; i < nruns; ++i) for (int k = 0; k < 6; ++k) m->m.runs[i][k] = runs[6 * i + k];
}
void ho_mesh_set_bounds(ho_mesh *m, int l, const uint8_t *mn, const uint8_t *mx)
{
	ho::List &L = m->m.lists[l];
	L.bmin.assign(mn, mn + L.fmt.bytes()); L.bmax.assign(mx, mx + L.fmt.bytes());
}
void ho_mesh_set_degrees(ho_mesh *m, const uint16_t *deg, size_t n)
{
	m->m.have_deg.assign(m->m.have_deg.size(), 0);
	for (size_t i = 0; i < n; ++i) { if (deg[i] >= m->m.have_deg.size()) m->m.have_deg.resize(deg[i] + 1, 0); m->m.have_deg[deg[i]] = 1; }
}
size_t ho_mesh_degrees(const ho_mesh *m, uint16_t *out, size_t cap)
{
	size_t n = 0;
	for (size_t d = 0; d < m->m.have_deg.size(); ++d) if (m->m.have_deg[d]) { if (n < cap) out[n] = (uint16_t)d; ++n; }
	return n;
}
void ho_mesh_free(ho_mesh *m) { delete m; }

int ho_requant(ho_mesh *m, const int *triples, int n, int clear)
{
	HO_TRY
	ho::requant(m->m, triples, n, clear != 0);
	return 0;
	HO_CATCH(-1)
}
ho_result *ho_encode(ho_mesh *m, int trace)
{
	HO_TRY
	ho::Result *r = ho::encode(m->m, trace != 0);
	ho_result *h = new ho_result{ std::move(*r) };
	delete r;
	return h;
	HO_CATCH(nullptr)
}
ho_result *ho_encode_chunked(ho_mesh *m, uint32_t chunk_syms) { return ho_encode_chunked2(m, chunk_syms, 0xffffffffu); }
ho_result *ho_encode_chunked2(ho_mesh *m, uint32_t chunk_syms, uint32_t snapshot_faces)
{
	HO_TRY
	ho::Result *r = ho::encode_chunked(m->m, chunk_syms, snapshot_faces);
	ho_result *h = new ho_result{ std::move(*r) };
	delete r;
	return h;
	HO_CATCH(nullptr)
}
ho_mesh *ho_mesh_from_hry_chunked(const uint8_t *hry, size_t n)
{
	HO_TRY
	ho::Mesh *m = ho::decode_chunked(hry, n);
	ho_mesh *h = new ho_mesh{ std::move(*m) };
	delete m;
	return h;
	HO_CATCH(nullptr)
}
void ho_result_free(ho_result *r) { delete r; }
size_t ho_result_size(const ho_result *r) { return r->r.bytes.size(); }
const uint8_t *ho_result_data(const ho_result *r) { return r->r.bytes.data(); }
size_t ho_result_header_size(const ho_result *r) { return r->r.header_size; }
size_t ho_result_trace_len(const ho_result *r) { return r->r.trace.size(); }
const ho_sym *ho_result_trace(const ho_result *r) { return r->r.trace.data(); }
size_t ho_result_order_vtx(const ho_result *r, const uint32_t **out) { *out = r->r.order_v.data(); return r->r.order_v.size(); }
size_t ho_result_order_face(const ho_result *r, const uint32_t **out) { *out = r->r.order_f.data(); return r->r.order_f.size(); }

uint32_t ho_mesh_nv(const ho_mesh *m) { return m->m.nv; }
uint32_t ho_mesh_nf(const ho_mesh *m) { return m->m.nf; }
uint32_t ho_mesh_ne(const ho_mesh *m) { return m->m.num_edge(); }
uint64_t ho_mesh_ntri(const ho_mesh *m) { return m->m.num_tri(); }
const uint32_t *ho_mesh_face_offsets(const ho_mesh *m) { return m->m.foff.data(); }
const uint32_t *ho_mesh_org(const ho_mesh *m) { return m->m.org.data(); }
const uint32_t *ho_mesh_twin(const ho_mesh *m) { return m->m.twin.data(); }
int ho_mesh_nlists(const ho_mesh *m) { return (int)m->m.lists.size(); }
int ho_list_ncomp(const ho_mesh *m, int l) { return m->m.lists[l].fmt.size(); }
int ho_list_target(const ho_mesh *m, int l) { return m->m.lists[l].target; }
uint32_t ho_list_count(const ho_mesh *m, int l) { return m->m.lists[l].count; }
int ho_list_stride(const ho_mesh *m, int l) { return m->m.lists[l].fmt.bytes(); }
int ho_list_type(const ho_mesh *m, int l, int c) { return m->m.lists[l].fmt.type[c]; }
int ho_list_quant(const ho_mesh *m, int l, int c) { return m->m.lists[l].fmt.quant[c]; }
int ho_list_offset(const ho_mesh *m, int l, int c) { return m->m.lists[l].fmt.off[c]; }
const uint8_t *ho_list_data(const ho_mesh *m, int l) { return m->m.lists[l].data.data(); }
const uint8_t *ho_list_min(const ho_mesh *m, int l) { return m->m.lists[l].bmin.data(); }
const uint8_t *ho_list_max(const ho_mesh *m, int l) { return m->m.lists[l].bmax.data(); }

int ho_ctx_count(const ho_mesh *m) { ho::Models md(m->m); return (int)md.tab.size(); }
int ho_ctx_attr_base(const ho_mesh *m, int l) { ho::Models md(m->m); return md.attr_base[l]; }

uint32_t ho_kat_encode_delta_f32(uint32_t raw, uint32_t pred)
{
	float r = ho::bitcast<uint32_t, float>(raw), p = ho::bitcast<uint32_t, float>(pred);
	return ho::bitcast<float, uint32_t>(ho::fold_residual<float>(r, p, 0, std::true_type()));
}
uint32_t ho_kat_decode_delta_f32(uint32_t d, uint32_t pred)
{
	float r = ho::bitcast<uint32_t, float>(d), p = ho::bitcast<uint32_t, float>(pred);
	return ho::bitcast<float, uint32_t>(ho::unfold_residual<float>(r, p, 0, std::true_type()));
}
uint32_t ho_kat_encode_delta_u(uint32_t raw, uint32_t pred, int bytes, int q)
{
	switch (bytes) {
	case 1: return ho::fold_residual<uint8_t>((uint8_t)raw, (uint8_t)pred, q, std::false_type());
	case 2: return ho::fold_residual<uint16_t>((uint16_t)raw, (uint16_t)pred, q, std::false_type());
	default: return ho::fold_residual<uint32_t>(raw, pred, q, std::false_type());
	}
}
uint32_t ho_kat_decode_delta_u(uint32_t d, uint32_t pred, int bytes, int q)
{
	switch (bytes) {
	case 1: return ho::unfold_residual<uint8_t>((uint8_t)d, (uint8_t)pred, q, std::false_type());
	case 2: return ho::unfold_residual<uint16_t>((uint16_t)d, (uint16_t)pred, q, std::false_type());
	default: return ho::unfold_residual<uint32_t>(d, pred, q, std::false_type());
	}
}
uint32_t ho_kat_predict_u(uint32_t v0, uint32_t v1, uint32_t v2, int bytes, int q)
{
	switch (bytes) {
	case 1: return ho::paral_predict<uint8_t>((uint8_t)v0, (uint8_t)v1, (uint8_t)v2, q, std::false_type());
	case 2: return ho::paral_predict<uint16_t>((uint16_t)v0, (uint16_t)v1, (uint16_t)v2, q, std::false_type());
	default: return ho::paral_predict<uint32_t>(v0, v1, v2, q, std::false_type());
	}
}
uint32_t ho_kat_predict_f32(uint32_t v0, uint32_t v1, uint32_t v2)
{
	using ho::bitcast;
	return bitcast<float, uint32_t>(ho::paral_predict<float>(bitcast<uint32_t, float>(v0), bitcast<uint32_t, float>(v1), bitcast<uint32_t, float>(v2), 0, std::true_type()));
}
uint64_t ho_kat_requant_f32(uint32_t v, uint32_t mn, uint32_t sc, int q)
{
	using ho::bitcast;
	return ho::quantise_scalar_f32(bitcast<uint32_t, float>(v), bitcast<uint32_t, float>(mn), bitcast<uint32_t, float>(sc), q);
}
size_t ho_kat_range_encode_bytes(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
	std::vector<uint8_t> out;
	ho::RangeEncoder rc(out);
	ho::FreqTable f(256);
	for (uint32_t i = 0; i < 256; ++i) f.inc(i);
	for (size_t i = 0; i < n; ++i) {
		uint64_t l, h, t = f.total();
		f.range_of(src[i], l, h);
		rc.encode(l, h, t);
		f.inc(src[i]);
	}
	rc.finish();
	if (out.size() <= cap) memcpy(dst, out.data(), out.size());
	return out.size();
}
size_t ho_kat_range_decode_bytes(const uint8_t *src, size_t nsrc, uint8_t *dst, size_t nsym)
{
	ho::RangeDecoder rc(src, src + nsrc);
	ho::FreqTable f(256);
	for (uint32_t i = 0; i < 256; ++i) f.inc(i);
	for (size_t i = 0; i < nsym; ++i) {
		uint64_t l, h, t = f.total();
		uint32_t s = f.find(rc.target(t), l, h);
		rc.consume(l, h, t);
		f.inc(s);
		dst[i] = (uint8_t)s;
	}
	return nsym;
}
// the same with the 32-bit instantiation (the streams of the chunked container)
size_t ho_kat_range_encode_bytes32(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
	std::vector<uint8_t> out;
	ho::ChunkEncoder rc(out);
	ho::FreqTable f(256);
	for (uint32_t i = 0; i < 256; ++i) f.inc(i);
	for (size_t i = 0; i < n; ++i) {
		uint64_t l, h, t = f.total();
		f.range_of(src[i], l, h);
		rc.encode((uint32_t)l, (uint32_t)h, (uint32_t)t);
		f.inc(src[i]);
	}
	rc.finish();
	if (out.size() <= cap) memcpy(dst, out.data(), out.size());
	return out.size();
}
size_t ho_kat_range_decode_bytes32(const uint8_t *src, size_t nsrc, uint8_t *dst, size_t nsym)
{
	ho::ChunkDecoder rc(src, src + nsrc);
	ho::FreqTable f(256);
	for (uint32_t i = 0; i < 256; ++i) f.inc(i);
	for (size_t i = 0; i < nsym; ++i) {
		uint64_t l, h, t = f.total();
		uint32_t s = f.find(rc.target((uint32_t)t), l, h);
		rc.consume((uint32_t)l, (uint32_t)h, (uint32_t)t);
		f.inc(s);
		dst[i] = (uint8_t)s;
	}
	return nsym;
}
size_t ho_kat_range_encode_lht32(const uint64_t *lht, size_t n, uint8_t *dst, size_t cap)
{
	std::vector<uint8_t> out;
	ho::ChunkEncoder rc(out);
	for (size_t i = 0; i < n; ++i) rc.encode((uint32_t)lht[3 * i], (uint32_t)lht[3 * i + 1], (uint32_t)lht[3 * i + 2]);
	rc.finish();
	if (out.size() <= cap) memcpy(dst, out.data(), out.size());
	return out.size();
}
size_t ho_kat_range_encode_lht(const uint64_t *lht, size_t n, uint8_t *dst, size_t cap)
{
	std::vector<uint8_t> out;
	ho::RangeEncoder rc(out);
	for (size_t i = 0; i < n; ++i) rc.encode(lht[3 * i], lht[3 * i + 1], lht[3 * i + 2]);
	rc.finish();
	if (out.size() <= cap) memcpy(dst, out.data(), out.size());
	return out.size();
}

}   // extern "C"
