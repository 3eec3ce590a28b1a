// Cut-border replay from entropy-decoded symbol planes (chunked profile).  See cbm_replay.hpp.
#include "cbm_replay.hpp"

namespace hry {
namespace {
struct Planes {
	const std::vector<uint8_t> *pl;   // container order: iop, elem[4], part[2], vertid[4], numtri[2], op[8]
	size_t cur[21] = { 0 };
	int fixed_numtri;
	uint32_t byte(int plane)
	{
		const std::vector<uint8_t> &v = pl[plane];
		if (cur[plane] >= v.size()) throw Error(HRY_E_FORMAT, "corrupt stream (connectivity plane exhausted)");
		return v[cur[plane]++];
	}
	uint32_t iop() { return byte(0); }
	uint32_t u32(int first) { uint32_t v = byte(first); v |= byte(first + 1) << 8; v |= byte(first + 2) << 16; v |= byte(first + 3) << 24; return v; }
	int elem() { uint32_t z = u32(1); return (int)((z >> 1) ^ ((z & 1) ? 0xffffffffu : 0u)); }   // transform.h:31-36
	int part() { uint32_t v = byte(5); v |= byte(6) << 8; return (int)v; }
	uint32_t vertid() { return u32(7); }
	int numtri() { if (fixed_numtri >= 0) return fixed_numtri; uint32_t v = byte(11); v |= byte(12) << 8; return (int)v; }
	uint32_t op(int order) { int k = order - 1; if (k > 7) k = 7; if (k < 0) k = 0; return byte(13 + k); }
};

}   // namespace

// planes: 21 connectivity planes in container order.  Fills m.face_off / org / twin and returns the decode order
// (one half-edge per vertex; vertex ids are assigned in this order, cbm/decoder.h:48-75,145).
void cut_border_replay(Mesh &m, const std::vector<uint8_t> *conn_planes, const std::vector<RestartPoint> &restarts,
                       std::vector<uint32_t> &order_v, std::vector<uint32_t> &seg_start, std::vector<uint32_t> &seg_level)
{
	(void)restarts;
	int ndeg = 0, onlydeg = 0;
	for (size_t d = 0; d < m.have_degree.size(); ++d) if (m.have_degree[d]) { ++ndeg; onlydeg = (int)d; }
	Planes rd{ conn_planes, { 0 }, ndeg <= 1 ? onlydeg - 2 : -1 };
	cut_border_replay_with(m, rd, order_v, seg_start, seg_level);
}

}   // namespace hry
