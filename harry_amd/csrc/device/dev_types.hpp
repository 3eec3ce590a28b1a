// Plain structs shared between host launch code and HIP kernels.
#pragma once
#include <cstdint>

namespace hry {
namespace dev {

constexpr int kMaxComp = 32;
constexpr int kChunk = 2048;        // symbols per model-evaluation chunk (one wavefront walks one chunk)
constexpr uint32_t kNoRank = 0xffffffffu;

// one attribute list as the kernels see it
struct ListDesc {
	int32_t ncomp, stride, nplanes;
	uint8_t stype[kMaxComp];   // storage type (mixing::Type value) of each component
	uint8_t otype[kMaxComp];   // original type
	uint8_t quant[kMaxComp];
	uint16_t off[kMaxComp];    // byte offset of the component's slot in a record
	uint16_t plane[kMaxComp];  // index of the component's first byte plane
};

struct ConnView {
	const uint32_t *org, *twin, *foff, *eface;   // eface == nullptr when every polygon has udeg edges
	uint32_t udeg, nf, ne;
};

// general bindings as the kernels see them (mesh.hpp Bindings): element -> region, element x slot -> record of the bound list
struct GenView {
	const uint16_t *vtx_reg, *face_reg;
	const uint32_t *vtx_attr, *corner_attr;
	int32_t nb_vtx, nb_corner;
};
// which lists a region binds at its slots (Bindings: off_* / reg_*: region r's slots are [off[r], off[r + 1])), and the faces' own
// records -- what the device needs beside GenView to say which record every element names (events.hip)
struct EvRegions {
	const int32_t *off_face, *off_vtx, *off_corner;
	const uint16_t *face_lists, *vtx_lists, *corner_lists;
	const uint32_t *face_attr;
	int32_t nb_face;
};

// per-symbol record consumed by the serial range recurrence (16 bytes, one dwordx4 per lane)
struct alignas(16) SymRec {
	uint64_t magic;    // reciprocal of the context total t (round-up method, 65-bit magic with implicit top bit)
	uint32_t x;        // multiplier: h - l, or l when the symbol is the last one with non-zero count (h == t)
	uint32_t meta;     // bits 0-5: post-shift, bit 6: h == t form (R' = R - r*x), bit 7: exact no-op (l == 0 && h == t)
};
constexpr uint32_t kMetaSub = 1u << 6, kMetaNoop = 1u << 7;

// reciprocals of a context total t: 65-bit magic + shift for 64-bit coder registers (reference stream), and m32 / sh32
// (cm::make_magic32) for the 32-bit registers of the chunked container's streams
struct alignas(16) MagicEnt { uint64_t magic; uint32_t shift; uint32_t m32; };
constexpr uint32_t kMagicSh32Shift = 8;   // MagicEnt::shift bits 8..15 hold sh32

// a byte plane whose adaptive model is evaluated by counting
struct PlaneJob {
	const uint8_t *sym;        // n symbols
	const uint32_t *pos_tab;   // global position of symbol j is pos_tab[j] + pos_add, or pos_base + j * pos_stride when null
	const uint32_t *init;      // 256 initial counts
	uint32_t n, t0;            // t0 = sum of init
	uint32_t pos_add, pos_base, pos_stride;
	uint32_t chunk0;           // index of this job's first chunk in the global chunk list
};
struct ChunkRef { uint32_t job, first; };

// chunked profile: one independent stream = one chunk of one context plane
struct StreamJob {
	const uint8_t *sym;    // symbols of the chunk (encode: input, decode: output)
	uint32_t n;            // symbols in the chunk
	uint32_t init;         // index of the 256-entry initial count table
	uint32_t t0;           // sum of the initial counts
	uint32_t word_base;    // first 32-bit word of this stream's big-number accumulator / byte image
};

// a slice of a plane for the histogram pass (static priors)
struct HistSlice { const uint8_t *sym; uint32_t n, plane; };

struct SplitBase { uint32_t b[8]; };

// the components of a list for the bounds reduction
struct BoundsPlan { int32_t n, stride; uint16_t off[kMaxComp]; uint8_t type[kMaxComp]; };

// one component of an in-place requantisation: mn / scale carry the raw bits of the component's original type
struct RequantComp { int32_t off, src_type, src_bits, dst_bits, dst_type, pad; uint64_t mn, scale; };   // dst_bits 0: dequantise into dst_type
struct RequantPlan { int32_t n; int32_t pad; RequantComp c[kMaxComp]; };

}   // namespace dev
}   // namespace hry
