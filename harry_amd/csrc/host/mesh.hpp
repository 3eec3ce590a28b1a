// Host-side mesh model of the product: flat arrays only (what the kernels consume).
// Reference counterparts: structs/mesh.h:19-40, structs/conn.h:72-170, structs/attr.h:24-189, structs/mixing.h:41-200.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace hry {

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
	int target = 0;                        // 0 face, 1 vertex
	std::vector<CompType> type;
	std::vector<uint8_t> quant;
	std::vector<int> offset{0};            // ncomp + 1
	std::vector<int> interp_off, interp_len;   // indexed by interpretation id
	std::vector<std::string> interp_name;      // id - kInterpOther
	uint32_t count = 0;
	std::vector<uint8_t> data;
	std::vector<uint8_t> bmin, bmax;       // records in original types; empty until computed
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

struct Mesh {
	uint32_t nv = 0, nf = 0;
	std::vector<uint32_t> face_off{0};   // nf + 1
	std::vector<uint32_t> org;           // per half-edge
	std::vector<uint32_t> twin;          // per half-edge, flat id; self = border
	std::vector<uint8_t> have_degree;    // have_degree[d] != 0 iff a polygon with d edges exists (faces.h:44-56)
	AttrList lists[2];                   // [0] face attributes, [1] vertex attributes (formats/ply/reader.cc:388-400)
	uint64_t device_token = 0;           // identity of the HBM-resident copy, 0 = none
	uint32_t declared_ne = 0;            // half-edge count announced by a .hry header (the connectivity follows later)

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
