// Kernels of the general-bindings path (OBJ input, regions, records shared between elements,
// corner attributes).  The host resolves WHICH record every vertex / face / corner names (history symbols are integer
// bookkeeping, device/general.cpp); these kernels do the value arithmetic for the records that are coded as data:
//
//   encode   k_gen_vtx_resid     parallelogram prediction restricted to the vertex' region + residual bytes   attrcode.h:117-134,209-225,321-344
//            k_gen_face_resid    residual against 0 (face prediction never finds a neighbour, App. B-16)       attrcode.h:245-270,345-366
//            k_gen_corner_resid  mean / nearest of the same slot's records at the already coded faces of the
//                                same region around the corner's vertex + residual bytes                      attrcode.h:135-154,272-288,367-393
//   decode   k_gen_sources       which earlier records every record's prediction reads (connectivity only: all records at once)
//            k_gen_chain         the inverse of the above, record by record in creation order.  A record's prediction reads records
//                                created before it (any of them: a corner may name a record that was created at another vertex),
//                                so the records of one list form a DAG with edges to smaller indices only: one wavefront per list
//                                takes 64 records at a time and relaxes the batch in LDS until every lane's sources are final.
// All byte / integer work next to dependent gathers: bounded by memory latency, not by bandwidth.
#include <hip/hip_runtime.h>

#include "codec_math.hpp"
#include "dev_types.hpp"
#include "fan.hpp"
#include "kernels.hpp"

#include <type_traits>

namespace hry {
namespace dev {

__global__ __launch_bounds__(256) void k_face_rank(ConnView cv, const uint32_t *order_f, uint32_t n, uint32_t *frank)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	Topo tp{ cv };
	frank[tp.face(order_f[j])] = j;
}

// mean of the parts in visiting order, then (floats) the part nearest to the mean, first wins (attrcode.h:182-208).
// `parts(f)` calls f(value) for every part; it is evaluated twice for floats, as the reference sweeps twice.
template <typename T, typename P>
__device__ __forceinline__ T combine_parts(P &&parts)
{
	typedef typename cm::wide<T>::type W;
	W acc = 0;
	uint32_t n = 0;
	parts([&](T x) { acc = acc + (W)x; ++n; });
	if (n == 0) return T(0);
	const T avg = (T)cm::mean_of(acc, (W)n);
	if constexpr (!cm::is_fp<T>::value) return avg;
	else {
		T best = 3.402823466e+38f;   // numeric_limits<float>::max()
		parts([&](T x) {
			T db = avg > best ? avg - best : best - avg;
			T dx = avg > x ? avg - x : x - avg;
			best = db < dx ? best : x;
		});
		return best;
	}
}

template <typename T>
__device__ __forceinline__ void put_code(uint8_t *planes, const ListDesc &ld, int c, uint32_t n, uint32_t j, T raw, T pred)
{
	auto code = cm::residual_bits<T>(raw, pred, ld.quant[c]);
	for (int b = 0; b < (int)sizeof(T); ++b) planes[(size_t)(ld.plane[c] + b) * n + j] = (uint8_t)(code >> (8 * b));
}

// one thread per vertex record coded as data: ev_he = the vertex' half-edge in the coding order, ev_slot = the slot of this list
// in the vertex' region, ev_idx = the record
__global__ __launch_bounds__(256) void k_gen_vtx_resid(ConnView cv, GenView gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                                                       const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, ListDesc ld, uint8_t *planes)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	Topo tp{ cv };
	const uint32_t e = ev_he[j], v = cv.org[e], my_rank = rank[v];
	const uint32_t r = gv.vtx_reg[v];
	const int a = ev_slot[j];
	for (int c = 0; c < ld.ncomp; ++c)
		with_stype(ld.stype[c], [&](auto tag) {
			typedef decltype(tag) T;
			const int off = ld.off[c], q = ld.quant[c];
			auto value = [&](uint32_t x) { return ldg<T>(rec + (size_t)gv.vtx_attr[(size_t)x * gv.nb_vtx + a] * ld.stride + off); };
			const T pred = combine_parts<T>([&](auto &&f) {
				fan_candidates(tp, rank, e, my_rank, 0, [&](uint32_t v0, uint32_t v1, uint32_t vo) {
					if (gv.vtx_reg[v0] != r || gv.vtx_reg[v1] != r || gv.vtx_reg[vo] != r) return;   // attrcode.h:120-121
					f(cm::parallelogram<T>(value(v0), value(v1), value(vo), q));
				});
			});
			put_code<T>(planes, ld, c, n, j, ldg<T>(rec + (size_t)ev_idx[j] * ld.stride + off), pred);
		});
}

__global__ __launch_bounds__(256) void k_gen_face_resid(const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, ListDesc ld, uint8_t *planes)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	for (int c = 0; c < ld.ncomp; ++c)
		with_stype(ld.stype[c], [&](auto tag) {
			typedef decltype(tag) T;
			put_code<T>(planes, ld, c, n, j, ldg<T>(rec + (size_t)ev_idx[j] * ld.stride + ld.off[c]), T(0));
		});
}

// one thread per corner record coded as data: ev_he = the corner (half-edge); frank = coding rank of every face
__global__ __launch_bounds__(256) void k_gen_corner_resid(ConnView cv, GenView gv, const uint32_t *frank, const uint32_t *ev_he, const uint8_t *ev_slot,
                                                          const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, ListDesc ld, uint8_t *planes)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	Topo tp{ cv };
	const uint32_t e = ev_he[j], f = tp.face(e), my_rank = frank[f];
	const uint32_t r = gv.face_reg[f];
	const int a = ev_slot[j];
	for (int c = 0; c < ld.ncomp; ++c)
		with_stype(ld.stype[c], [&](auto tag) {
			typedef decltype(tag) T;
			const int off = ld.off[c];
			const T pred = combine_parts<T>([&](auto &&use) {
				fan_each(tp, e, [&](uint32_t x) {
					const uint32_t g = tp.face(x);
					if (frank[g] >= my_rank || gv.face_reg[g] != r) return;   // attrcode.h:139-141 (the face itself is not coded yet)
					use(ldg<T>(rec + (size_t)gv.corner_attr[(size_t)x * gv.nb_corner + a] * ld.stride + off));
				});
			});
			put_code<T>(planes, ld, c, n, j, ldg<T>(rec + (size_t)ev_idx[j] * ld.stride + off), pred);
		});
}

// ---------------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------------
// a value another workgroup (possibly on another XCD) wrote during this launch: read past the non-coherent caches
template <typename T> __device__ __forceinline__ T far_value(const uint8_t *p, bool aligned)
{
	typedef typename cm::word<sizeof(T)>::u U;
	U u;
	if (aligned) u = __hip_atomic_load((const U*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else {
		u = 0;
		for (int b = 0; b < (int)sizeof(T); ++b) u |= (U)((U)__hip_atomic_load(p + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << (8 * b));
	}
	return cm::bits<T>(u);
}

// The chain reads its job's tables through pointers it loaded from the job list, which the compiler can only take for generic
// ("flat") ones; flat loads may return out of order, so every wait for one of them waits for ALL loads in flight -- including the
// next batch's prefetch.  The chain therefore states the address space (global) itself.
#define HRY_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T far_value_g(const HRY_GLOBAL uint8_t *p, bool aligned)
{
	typedef typename cm::word<sizeof(T)>::u U;
	U u;
	if (aligned) u = __hip_atomic_load((const HRY_GLOBAL U*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else {
		u = 0;
		for (int b = 0; b < (int)sizeof(T); ++b) u |= (U)((U)__hip_atomic_load(p + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << (8 * b));
	}
	return cm::bits<T>(u);
}
template <typename T> __device__ __forceinline__ T load_g(const HRY_GLOBAL uint8_t *p, bool aligned)
{
	typedef typename cm::word<sizeof(T)>::u U;
	U u;
	if (aligned) u = *(const HRY_GLOBAL U*)p;
	else {
		u = 0;
		for (int b = 0; b < (int)sizeof(T); ++b) u |= (U)((U)p[b] << (8 * b));
	}
	return cm::bits<T>(u);
}
template <typename T> __device__ __forceinline__ void store_g(HRY_GLOBAL uint8_t *p, T v, bool aligned)
{
	typedef typename cm::word<sizeof(T)>::u U;
	const U u = cm::bits<U>(v);
	if (aligned) *(HRY_GLOBAL U*)p = u;
	else for (int b = 0; b < (int)sizeof(T); ++b) p[b] = (uint8_t)(u >> (8 * b));
}

// ---- sources: which earlier records the prediction of record i reads -----------------------------------------------
// KIND 0: vertex records (the three records of every parallelogram, in fan order), 1: corner records (one record per already
// decoded face of the region around the vertex).  Connectivity only, so every record at once: src[k * n + i] = k-th source id of
// record i, nsrc[i] = their number, kSrcOverflow when the fan has more than SrcCap<KIND> rows (the chain walks that fan itself).
template <int KIND> struct SrcCap { static constexpr int value = KIND == 0 ? 24 : 12; };   // 8 parallelograms / 12 faces around a vertex
constexpr uint8_t kSrcOverflow = 255;

template <int KIND, typename F>
__device__ __forceinline__ void walk_sources(const Topo &tp, const GenView &gv, const uint32_t *rank, uint32_t e, int a, F &&f)
{
	if constexpr (KIND == 0) {
		const uint32_t v = tp.c.org[e], my_rank = rank[v], r = gv.vtx_reg[v];
		fan_candidates(tp, rank, e, my_rank, 0, [&](uint32_t v0, uint32_t v1, uint32_t vo) {
			if (gv.vtx_reg[v0] != r || gv.vtx_reg[v1] != r || gv.vtx_reg[vo] != r) return;   // attrcode.h:120-121
			f(gv.vtx_attr[(size_t)v0 * gv.nb_vtx + a]); f(gv.vtx_attr[(size_t)v1 * gv.nb_vtx + a]); f(gv.vtx_attr[(size_t)vo * gv.nb_vtx + a]);
		});
	} else {
		const uint32_t face = tp.face(e), r = gv.face_reg[face];
		fan_each(tp, e, [&](uint32_t x) {
			const uint32_t g = tp.face(x);
			if (g >= face || gv.face_reg[g] != r) return;   // faces are decoded in index order (attrcode.h:543-548); not the face itself
			f(gv.corner_attr[(size_t)x * gv.nb_corner + a]);
		});
	}
}

template <int KIND>
__global__ __launch_bounds__(256) void k_gen_sources(ConnView cv, GenView gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                                                     uint32_t n, uint32_t *src, uint8_t *nsrc)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	Topo tp{ cv };
	int ns = 0;
	constexpr int CAP = SrcCap<KIND>::value;
	walk_sources<KIND>(tp, gv, rank, ev_he[i], ev_slot[i], [&](uint32_t id) { if (ns < CAP) src[(size_t)ns * n + i] = id; ++ns; });
	nsrc[i] = ns <= CAP ? (uint8_t)ns : kSrcOverflow;
}

// ---- the chain: one wavefront per list walks its records in creation order, 64 at a time -------------------------------------
// A record reads only records created before it, so inside a batch of 64 the values are found by relaxation: every lane
// evaluates its record from the current values of the batch (LDS) and the final values of everything before the batch; a lane
// whose in-batch sources are final becomes final; at most 64 rounds (a batch in which every record reads its predecessor), one
// round when no record reads inside the batch (texture atlases, smooth normals).  Same arithmetic, same order of the parts as
// the encoder (combine_parts above).  Records whose fan did not fit the source table are done by their lane alone, walking the fan.
struct GenChainJob {
	int32_t kind, comp;        // one job = one component of one list (the components of a record are predicted independently)
	uint32_t n, pad2;
	uint8_t *rec;
	const uint32_t *src, *ev_he;
	const uint8_t *nsrc, *ev_slot;
	ListDesc ld;
};

// largest value of the wavefront, for small non-negative values (five ballots)
__device__ __forceinline__ int wave_max_small(int v)
{
	int m = 0;
#pragma unroll
	for (int b = 4; b >= 0; --b) { const int t = m | (1 << b); if (__ballot(v >= t)) m = t; }
	return m;
}
template <typename T> __device__ __forceinline__ uint32_t as_u32(T v) { typename cm::word<sizeof(T)>::u u = cm::bits<typename cm::word<sizeof(T)>::u>(v); return (uint32_t)u; }
template <typename T> __device__ __forceinline__ T from_u32(uint32_t x) { return cm::bits<T>((typename cm::word<sizeof(T)>::u)x); }

// val[k] = value of the k-th source
// N <= CAP: only the first N slots can be in use (the batch's largest source count, rounded up: straight-line code per bucket)
template <int KIND, typename T, int CAP, int N = CAP>
__device__ __forceinline__ T predict_from(int ns, int q, const T (&val)[CAP])
{
	if constexpr (KIND == 0)
		return combine_parts<T>([&](auto &&use) {
#pragma unroll
			for (int k = 0; k + 2 < N; k += 3) if (k + 2 < ns) use(cm::parallelogram<T>(val[k], val[k + 1], val[k + 2], q));
		});
	else
		return combine_parts<T>([&](auto &&use) {
#pragma unroll
			for (int k = 0; k < N; ++k) if (k < ns) use(val[k]);
		});
}

// ---- one step of the short form of a float corner record (chain_component below), hand-scheduled -------------------------------
// Every lane: mean of its NN sources (weights 1 / 0: slots past the record's own count hold 3e38 with weight 0), the LAST source
// at the smallest distance from the mean (the reference's sweep keeps the later one on equal distances, attrcode.h:193-205), the
// residual code on that source's bits (near: +- delta by its sign; constant lanes: one of two constants by its sign), then lane
// `step` broadcasts its value and every lane takes it into the slots that wait for that record.  Written as one block so that no
// wait state is spent: a VALU-written SGPR needs two other instructions before the VALU reads it (the compiler's version of this
// loop carried nine s_nop and 68 issue slots at 6 sources; this one 49).  Temporaries: v236-v248, s86-s96, vcc.
#define HRY_GS_SUM(k)   "v_fma_f32 v236, %[v" #k "], %[w" #k "], v236\n\t"
#define HRY_GS_DIST(k)  "v_sub_f32 v24" #k ", v237, %[v" #k "]\n\t"
#define HRY_GS_MIN(k)   "v_min_f32_e64 v238, v238, |v24" #k "|\n\t"
#define HRY_GS_EQ(k, sp)   "v_cmp_eq_f32_e64 " sp ", |v24" #k "|, v238\n\t"
#define HRY_GS_TAKE(k, sp) "v_cndmask_b32_e64 v239, v239, %[v" #k "], " sp "\n\t"
#define HRY_GS_HIT(k, sp)  "v_cmp_eq_u32_e64 " sp ", %[step], %[s" #k "]\n\t"
#define HRY_GS_PICK(k, sp) "v_cndmask_b32_e64 %[v" #k "], %[v" #k "], v246, " sp "\n\t"
#define HRY_GS_HEAD     "v_mul_f32 v236, %[v0], %[w0]\n\t"
#define HRY_GS_MEAN     "v_mul_f32 v237, v236, %[rcp]\n\t"
#define HRY_GS_VALUE \
	"v_ashrrev_i32 v247, 31, v239\n\t" \
	"v_xad_u32 v248, %[delta], v247, v239\n\t" \
	"v_bfi_b32 v236, v247, %[aneg], %[apos]\n\t" \
	"v_sub_u32 v248, v248, v247\n\t" \
	"v_bfi_b32 %[out], %[cmask], v236, v248\n\t"
#define HRY_GS_CAST     "v_readlane_b32 s96, %[out], %[step]\n\t"
#define HRY_GS_CLOBBER : "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", \
	"s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "vcc"
#define HRY_GS_IO(k) [v##k] "+v"(v[k])
#define HRY_GS_IN(k) [w##k] "v"(w[k]), [s##k] "v"(slot[k])
template <int NN, int CV, int CW>
__device__ __forceinline__ uint32_t short_step(float (&v)[CV], const uint32_t (&slot)[CV], const float (&w)[CW], float rcp, uint32_t delta, uint32_t cmask,
                                               uint32_t apos, uint32_t aneg, int step)
{
	static_assert(NN >= 2 && NN <= 6 && NN <= CV && NN <= CW, "sources");
	uint32_t out;
	if constexpr (NN == 2)
		asm volatile(HRY_GS_HEAD HRY_GS_SUM(1) HRY_GS_MEAN HRY_GS_DIST(0) HRY_GS_DIST(1) "v_min_f32_e64 v238, |v240|, |v241|\n\t" "v_mov_b32 v239, %[v0]\n\t"
		             HRY_GS_EQ(1, "vcc") "s_nop 1\n\t" HRY_GS_TAKE(1, "vcc") HRY_GS_VALUE
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") "s_nop 0\n\t" "v_mov_b32 v246, s96\n\t" HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1)
		             : HRY_GS_IN(0), HRY_GS_IN(1), [rcp] "v"(rcp), [delta] "v"(delta), [cmask] "v"(cmask), [apos] "v"(apos), [aneg] "v"(aneg), [step] "s"(step) HRY_GS_CLOBBER);
	else if constexpr (NN == 3)
		asm volatile(HRY_GS_HEAD HRY_GS_SUM(1) HRY_GS_SUM(2) HRY_GS_MEAN HRY_GS_DIST(0) HRY_GS_DIST(1) HRY_GS_DIST(2) "v_min_f32_e64 v238, |v240|, |v241|\n\t" HRY_GS_MIN(2)
		             "v_mov_b32 v239, %[v0]\n\t" HRY_GS_EQ(1, "vcc") HRY_GS_EQ(2, "s[88:89]") "s_nop 0\n\t" HRY_GS_TAKE(1, "vcc") HRY_GS_TAKE(2, "s[88:89]") HRY_GS_VALUE
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") "v_mov_b32 v246, s96\n\t"
		             HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2)
		             : HRY_GS_IN(0), HRY_GS_IN(1), HRY_GS_IN(2), [rcp] "v"(rcp), [delta] "v"(delta), [cmask] "v"(cmask), [apos] "v"(apos), [aneg] "v"(aneg), [step] "s"(step) HRY_GS_CLOBBER);
	else if constexpr (NN == 4)
		asm volatile(HRY_GS_HEAD HRY_GS_SUM(1) HRY_GS_SUM(2) HRY_GS_SUM(3) HRY_GS_MEAN HRY_GS_DIST(0) HRY_GS_DIST(1) HRY_GS_DIST(2) HRY_GS_DIST(3)
		             "v_min_f32_e64 v238, |v240|, |v241|\n\t" HRY_GS_MIN(2) HRY_GS_MIN(3) "v_mov_b32 v239, %[v0]\n\t"
		             HRY_GS_EQ(1, "vcc") HRY_GS_EQ(2, "s[88:89]") HRY_GS_EQ(3, "s[90:91]") HRY_GS_TAKE(1, "vcc") HRY_GS_TAKE(2, "s[88:89]") HRY_GS_TAKE(3, "s[90:91]") HRY_GS_VALUE
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") HRY_GS_HIT(3, "s[92:93]") "v_mov_b32 v246, s96\n\t"
		             HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]") HRY_GS_PICK(3, "s[92:93]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2), HRY_GS_IO(3)
		             : HRY_GS_IN(0), HRY_GS_IN(1), HRY_GS_IN(2), HRY_GS_IN(3), [rcp] "v"(rcp), [delta] "v"(delta), [cmask] "v"(cmask), [apos] "v"(apos), [aneg] "v"(aneg), [step] "s"(step) HRY_GS_CLOBBER);
	else if constexpr (NN == 5)
		asm volatile(HRY_GS_HEAD HRY_GS_SUM(1) HRY_GS_SUM(2) HRY_GS_SUM(3) HRY_GS_SUM(4) HRY_GS_MEAN HRY_GS_DIST(0) HRY_GS_DIST(1) HRY_GS_DIST(2) HRY_GS_DIST(3) HRY_GS_DIST(4)
		             "v_min_f32_e64 v238, |v240|, |v241|\n\t" HRY_GS_MIN(2) HRY_GS_MIN(3) HRY_GS_MIN(4) "v_mov_b32 v239, %[v0]\n\t"
		             HRY_GS_EQ(1, "vcc") HRY_GS_EQ(2, "s[88:89]") HRY_GS_EQ(3, "s[90:91]") HRY_GS_EQ(4, "s[92:93]")
		             HRY_GS_TAKE(1, "vcc") HRY_GS_TAKE(2, "s[88:89]") HRY_GS_TAKE(3, "s[90:91]") HRY_GS_TAKE(4, "s[92:93]") HRY_GS_VALUE
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") HRY_GS_HIT(3, "s[92:93]") HRY_GS_HIT(4, "s[94:95]") "v_mov_b32 v246, s96\n\t"
		             HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]") HRY_GS_PICK(3, "s[92:93]") HRY_GS_PICK(4, "s[94:95]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2), HRY_GS_IO(3), HRY_GS_IO(4)
		             : HRY_GS_IN(0), HRY_GS_IN(1), HRY_GS_IN(2), HRY_GS_IN(3), HRY_GS_IN(4), [rcp] "v"(rcp), [delta] "v"(delta), [cmask] "v"(cmask), [apos] "v"(apos), [aneg] "v"(aneg), [step] "s"(step) HRY_GS_CLOBBER);
	else
		asm volatile(HRY_GS_HEAD HRY_GS_SUM(1) HRY_GS_SUM(2) HRY_GS_SUM(3) HRY_GS_SUM(4) HRY_GS_SUM(5) HRY_GS_MEAN
		             HRY_GS_DIST(0) HRY_GS_DIST(1) HRY_GS_DIST(2) HRY_GS_DIST(3) HRY_GS_DIST(4) HRY_GS_DIST(5)
		             "v_min_f32_e64 v238, |v240|, |v241|\n\t" HRY_GS_MIN(2) HRY_GS_MIN(3) HRY_GS_MIN(4) HRY_GS_MIN(5) "v_mov_b32 v239, %[v0]\n\t"
		             HRY_GS_EQ(1, "vcc") HRY_GS_EQ(2, "s[88:89]") HRY_GS_EQ(3, "s[90:91]") HRY_GS_EQ(4, "s[92:93]") HRY_GS_EQ(5, "s[94:95]")
		             HRY_GS_TAKE(1, "vcc") HRY_GS_TAKE(2, "s[88:89]") HRY_GS_TAKE(3, "s[90:91]") HRY_GS_TAKE(4, "s[92:93]") HRY_GS_TAKE(5, "s[94:95]") HRY_GS_VALUE
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") HRY_GS_HIT(3, "s[92:93]") HRY_GS_HIT(4, "s[94:95]") HRY_GS_HIT(5, "vcc")
		             "v_mov_b32 v246, s96\n\t"
		             HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]") HRY_GS_PICK(3, "s[92:93]") HRY_GS_PICK(4, "s[94:95]") HRY_GS_PICK(5, "vcc")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2), HRY_GS_IO(3), HRY_GS_IO(4), HRY_GS_IO(5)
		             : HRY_GS_IN(0), HRY_GS_IN(1), HRY_GS_IN(2), HRY_GS_IN(3), HRY_GS_IN(4), HRY_GS_IN(5), [rcp] "v"(rcp), [delta] "v"(delta), [cmask] "v"(cmask), [apos] "v"(apos), [aneg] "v"(aneg), [step] "s"(step) HRY_GS_CLOBBER);
	return out;
}

// ---- the same for unsigned integer records: rounded mean by a multiplication (see chain_component), residual code without branches --
#define HRY_GI_UNFOLD \
	"v_mul_hi_u32 v237, v236, %[magic]\n\t" \
	"v_lshrrev_b32 v237, %[shift], v237\n\t" \
	"v_sub_u32 v238, %[top], v237\n\t" \
	"v_add_u32 v239, -1, v237\n\t" \
	"v_min_u32 v239, v239, v238\n\t" \
	"v_cmp_gt_u32_e64 vcc, %[half], v239\n\t" \
	"v_cmp_ge_u32_e64 s[88:89], v238, v237\n\t" \
	"v_cmp_eq_u32_e64 s[90:91], 0, v237\n\t" \
	"v_sub_u32 v240, v237, v239\n\t" \
	"v_add3_u32 v240, v240, %[c32], -1\n\t" \
	"v_sub_u32 v241, v237, %[c32]\n\t" \
	"v_add_u32 v241, v241, v239\n\t" \
	"v_add_u32 v242, v237, %[dnear]\n\t" \
	"v_cndmask_b32_e64 v240, v241, v240, s[88:89]\n\t" \
	"v_cndmask_b32_e64 v240, v242, v240, vcc\n\t" \
	"v_cndmask_b32_e64 v240, v240, %[c32], s[90:91]\n\t" \
	"v_and_b32 %[out], v240, %[tmask]\n\t"
#define HRY_GI_CONSTS [rnd] "v"(rnd), [magic] "v"(magic), [shift] "v"(shift), [top] "v"(top), [half] "v"(half), [c32] "v"(c32), [dnear] "v"(dnear), [tmask] "v"(tmask), [step] "s"(step)
template <int NN, int CV>
__device__ __forceinline__ uint32_t short_step_int(uint32_t (&v)[CV], const uint32_t (&slot)[CV], uint32_t rnd, uint32_t magic, uint32_t shift, uint32_t top, uint32_t half,
                                                   uint32_t c32, uint32_t dnear, uint32_t tmask, int step)
{
	static_assert(NN >= 2 && NN <= 6 && NN <= CV, "sources");
	uint32_t out;
	if constexpr (NN == 2)
		asm volatile("v_add3_u32 v236, %[rnd], %[v0], %[v1]\n\t" HRY_GI_UNFOLD
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") "s_nop 0\n\t" "v_mov_b32 v246, s96\n\t" HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1)
		             : [s0] "v"(slot[0]), [s1] "v"(slot[1]), HRY_GI_CONSTS HRY_GS_CLOBBER);
	else if constexpr (NN == 3)
		asm volatile("v_add3_u32 v236, %[rnd], %[v0], %[v1]\n\t" "v_add_u32 v236, v236, %[v2]\n\t" HRY_GI_UNFOLD
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") "v_mov_b32 v246, s96\n\t" HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2)
		             : [s0] "v"(slot[0]), [s1] "v"(slot[1]), [s2] "v"(slot[2]), HRY_GI_CONSTS HRY_GS_CLOBBER);
	else if constexpr (NN == 4)
		asm volatile("v_add3_u32 v236, %[rnd], %[v0], %[v1]\n\t" "v_add3_u32 v236, v236, %[v2], %[v3]\n\t" HRY_GI_UNFOLD
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") HRY_GS_HIT(3, "s[92:93]") "v_mov_b32 v246, s96\n\t" HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]") HRY_GS_PICK(3, "s[92:93]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2), HRY_GS_IO(3)
		             : [s0] "v"(slot[0]), [s1] "v"(slot[1]), [s2] "v"(slot[2]), [s3] "v"(slot[3]), HRY_GI_CONSTS HRY_GS_CLOBBER);
	else if constexpr (NN == 5)
		asm volatile("v_add3_u32 v236, %[rnd], %[v0], %[v1]\n\t" "v_add3_u32 v236, v236, %[v2], %[v3]\n\t" "v_add_u32 v236, v236, %[v4]\n\t" HRY_GI_UNFOLD
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") HRY_GS_HIT(3, "s[92:93]") HRY_GS_HIT(4, "s[94:95]") "v_mov_b32 v246, s96\n\t" HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]") HRY_GS_PICK(3, "s[92:93]") HRY_GS_PICK(4, "s[94:95]")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2), HRY_GS_IO(3), HRY_GS_IO(4)
		             : [s0] "v"(slot[0]), [s1] "v"(slot[1]), [s2] "v"(slot[2]), [s3] "v"(slot[3]), [s4] "v"(slot[4]), HRY_GI_CONSTS HRY_GS_CLOBBER);
	else
		asm volatile("v_add3_u32 v236, %[rnd], %[v0], %[v1]\n\t" "v_add3_u32 v236, v236, %[v2], %[v3]\n\t" "v_add3_u32 v236, v236, %[v4], %[v5]\n\t" HRY_GI_UNFOLD
		             HRY_GS_HIT(0, "s[86:87]") HRY_GS_CAST HRY_GS_HIT(1, "s[88:89]") HRY_GS_HIT(2, "s[90:91]") HRY_GS_HIT(3, "s[92:93]") HRY_GS_HIT(4, "s[94:95]") HRY_GS_HIT(5, "vcc") "v_mov_b32 v246, s96\n\t" HRY_GS_PICK(0, "s[86:87]") HRY_GS_PICK(1, "s[88:89]") HRY_GS_PICK(2, "s[90:91]") HRY_GS_PICK(3, "s[92:93]") HRY_GS_PICK(4, "s[94:95]") HRY_GS_PICK(5, "vcc")
		             : [out] "=&v"(out), HRY_GS_IO(0), HRY_GS_IO(1), HRY_GS_IO(2), HRY_GS_IO(3), HRY_GS_IO(4), HRY_GS_IO(5)
		             : [s0] "v"(slot[0]), [s1] "v"(slot[1]), [s2] "v"(slot[2]), [s3] "v"(slot[3]), [s4] "v"(slot[4]), [s5] "v"(slot[5]), HRY_GI_CONSTS HRY_GS_CLOBBER);
	return out;
}

#ifdef HRY_GEN_CLOCKS
#define GEN_CLK(...) __VA_ARGS__
#else
#define GEN_CLK(...)
#endif
template <int KIND, typename T>
__device__ void chain_component(const ConnView &cv, const GenView &gv, const uint32_t *rank, const GenChainJob &jb, uint32_t *s_val)
{
	typedef typename cm::word<sizeof(T)>::u U;
	constexpr int CAP = SrcCap<KIND>::value;
	const int lane = threadIdx.x;
	const ListDesc &ld = jb.ld;
	const int c = jb.comp, off = ld.off[c], q = ld.quant[c];
	const bool aligned = (ld.stride % (int)sizeof(T)) == 0 && (off % (int)sizeof(T)) == 0;
	Topo tp{ cv };
	HRY_GLOBAL uint8_t *const g_rec = (HRY_GLOBAL uint8_t*)jb.rec;
	const HRY_GLOBAL uint32_t *const g_src = (const HRY_GLOBAL uint32_t*)jb.src;
	const HRY_GLOBAL uint8_t *const g_nsrc = (const HRY_GLOBAL uint8_t*)jb.nsrc;
	GEN_CLK(unsigned long long ck_load = 0, ck_depth = 0, ck_steps = 0, ck_verify = 0, ck_exact = 0, ck_shallow = 0, ck_store = 0, ck_runs = 0, ck_retry = 0, ck_exact_n = 0, ck_shallow_n = 0, ck_single = 0, ck_evals = 0, ck_t, ck_n[4] = {0, 0, 0, 0}; const unsigned long long ck_begin = __builtin_amdgcn_s_memtime();)
	// a batch's inputs (source count, the table's rows -- all of them, so that no load waits for the count --, the residual code
	// that lies in the record's own slot) are loaded one batch ahead: three dependent trips to memory a batch became one
	int nx_ns = 0;
	uint32_t nx_id[CAP];
	U nx_code = U(0);
	auto fetch = [&](uint32_t at) {
		const uint32_t j = at + lane;
		const bool there = j < jb.n;
		nx_ns = there ? g_nsrc[j] : 0;
#pragma unroll
		for (int k = 0; k < CAP; ++k) nx_id[k] = there ? g_src[(size_t)k * jb.n + j] : 0u;
		nx_code = there ? cm::bits<U>(load_g<T>(g_rec + (size_t)j * ld.stride + off, aligned)) : U(0);
	};
	fetch(0);
	for (uint32_t base = 0; base < jb.n; base += 64) {
		GEN_CLK(ck_t = __builtin_amdgcn_s_memtime();)
		const uint32_t i = base + lane;
		const bool live = i < jb.n;
		const int ns_raw = nx_ns;
		const bool heavy = ns_raw == kSrcOverflow;
		const int ns = heavy ? 0 : ns_raw;
		uint32_t id[CAP];
#pragma unroll
		for (int k = 0; k < CAP; ++k) id[k] = k < ns ? nx_id[k] : 0u;
		uint8_t *mine = jb.rec + (size_t)i * ld.stride + off;   // (the heavy lanes' path)
		HRY_GLOBAL uint8_t *const g_mine = g_rec + (size_t)i * ld.stride + off;
		const U code = nx_code;
		bool fetched = false;
		const unsigned long long heavy_mask = __ballot(heavy);
		// the batch in runs of lanes between the heavy ones
		int lo = 0;
		while (lo < 64) {
			const unsigned long long rest = heavy_mask >> lo;
			const int hi = rest ? lo + __builtin_ctzll(rest) : 64;   // [lo, hi): ordinary lanes; hi: a heavy lane (or the end)
			const uint32_t first = base + lo;                        // records before `first` are final in memory
			const bool in_run = live && lane >= lo && lane < hi;
			if (hi > lo) {
				T val[CAP];
				uint32_t slot[CAP];          // LDS slot of a source inside the run, 64 = outside
				bool inside = false;
				const int most = (int)__builtin_amdgcn_readfirstlane(wave_max_small(in_run ? ns : 0));
				// the sources before the run, from memory: every lane loads every slot the run uses (a lane without that source reads
				// record 0 and drops it), so the loads go out back to back and are waited for once -- one trip to memory a run; with a
				// branch per source and lane the compiler had serialised them
#pragma unroll
				for (int k = 0; k < CAP; ++k) {
					val[k] = T(0); slot[k] = 64u;
					if (k < most) {
						const bool used = in_run && k < ns;
						const bool before = used && id[k] < first;
						const T x = far_value_g<T>(g_rec + (size_t)(before ? id[k] : 0u) * ld.stride + off, aligned);
						val[k] = before ? x : T(0);
						if (used && !before && id[k] < i) { slot[k] = id[k] - base; inside = true; }   // (id >= i: damaged input, reads as 0)
					}
				}
				// the next batch's inputs: asked for AFTER this run's far values (loads return in order: the wait for the far values
				// then leaves these in flight), in every run so that the wait can be counted (harmless when a batch has two)
				fetch(base + 64);
				fetched = true;
				T out = T(0);
				GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); if (__ballot(val[0] == val[1] || true)) ck_load += n - ck_t; ck_t = n; ++ck_runs; })
				// the run, compiled for N source slots (a batch whose records read at most N sources runs the N-slot code).  Records read
				// earlier records only, so the run is evaluated SYSTOLICALLY: at step i every lane evaluates its record from what it
				// holds, lane i -- whose sources inside the run are all final by then -- broadcasts its value (v_readlane), and every
				// lane that waits for record i picks it up.  Exactly hi - lo steps of one evaluation each, no LDS, no barrier; round 2
				// relaxed the batch through LDS until nothing changed: up to 64 rounds of evaluate + two barriers + a ballot + a scan of
				// the slots (0.7 us per record when every record reads its predecessor -- one normal per face).
				auto rounds = [&](auto n_slots) {
					constexpr int N = decltype(n_slots)::value;
					if (!__ballot(inside)) {   // nothing reads inside the run (private texture coordinates): one evaluation
						if (in_run) out = cm::value_from_residual<T>(code, predict_from<KIND, T, CAP, N>(ns, q, val), q);
						GEN_CLK(++ck_single;)
						return;
					}
					GEN_CLK(++ck_n[N <= 3 ? 0 : N <= 6 ? 1 : N <= 12 ? 2 : 3];)
					// how deep the dependencies inside the run go (connectivity only: a relaxation on small integers through the LDS
					// crossbar, ~100 cycles a round): level = 1 + the deepest source inside the run
					int level = 0, depth = 0;
					bool settled = false;
					// (a run in which most records read the record right before them is deep without asking)
					bool reads_prev = false;
#pragma unroll
					for (int k = 0; k < N; ++k) reads_prev |= slot[k] + 1u == (uint32_t)lane;
					const bool chain_like = 2 * __builtin_popcountll(__ballot(reads_prev)) > hi - lo;
					for (int r = 0; r < 24 && !settled && !chain_like; ++r) {
						int mx = 0;
#pragma unroll
						for (int k = 0; k < N; ++k) {
							const int sl = __builtin_amdgcn_ds_bpermute((int)((slot[k] & 63u) * 4u), level);
							mx = slot[k] < 64u ? max(mx, sl + 1) : mx;
						}
						settled = !__ballot(mx != level);
						level = mx;
						++depth;
					}
					// shallow (smooth normals, texture atlases: a handful of levels): that many rounds of "every lane evaluates from the
					// current values of the run" through LDS; deep (one normal per face: every record reads the one before it): the
					// systolic form, one evaluation per record
					GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_depth += n - ck_t; ck_t = n; })
					constexpr bool kShort = KIND == 1 && std::is_same<T, float>::value && N <= 6;
					if (settled && depth * 3 < hi - lo) {
						GEN_CLK(++ck_shallow_n;)
						for (int round = 0; round < depth; ++round) {
							if (round) {
#pragma unroll
								for (int k = 0; k < N; ++k) {   // unconditional reads: issued back to back, one wait
									const uint32_t x = s_val[slot[k] & 63u];
									val[k] = slot[k] < 64u ? from_u32<T>(x) : val[k];
								}
							}
							if (in_run) out = cm::value_from_residual<T>(code, predict_from<KIND, T, CAP, N>(ns, q, val), q);
							__syncthreads();   // (one wavefront: orders the LDS traffic of the rounds)
							if (in_run) s_val[lane] = as_u32<T>(out);
							__syncthreads();
						}
						GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_shallow += n - ck_t; ck_t = n; })
						return;
					}
					if constexpr (kShort) {
						// Float corner records (normals, texture coordinates), the deep case -- one normal per face: every record reads the
						// one before it, and which of two sources is nearer to their mean hangs on the last bit, so nothing but the
						// record's final sources will do (a round-3 trial that solved the run from estimated values -- the picks as
						// pointers, the residual codes as offsets, pointer jumping -- needed seven rounds a run for exactly that reason).
						// So the step itself is SHORT: the mean in float (exact for one and two sources; a sum of three or more may round
						// differently from the reference's double, which only matters when two sources are about equally near), the sweep
						// for the nearest source without its FLT_MAX start, the residual code applied on the value's bits: "near" codes
						// move the bits by +- delta by the prediction's own sign, codes from 2^31 are "far" for every prediction up to 2.0
						// in magnitude -- a change of sign -- and give one of two constants by the prediction's sign
						// (attrcode.h:135-154,182-208, prediction.h:46-64).  Then it is VERIFIED exactly, all lanes at once: after the
						// run every lane holds the final values of its sources, so it evaluates the reference arithmetic on them; the
						// broadcast values were what `out` is now, hence all are right iff every lane agrees (induction over the lanes).
						// The first lane that disagrees gets its exact value and the run is repeated from there, six times at most; then
						// the exact systolic form below takes the run.
						const bool none = ns == 0;
						const float rcp = none ? 0.0f : 1.0f / (float)ns;
						float w[N];
#pragma unroll
						for (int k = 0; k < N; ++k) { w[k] = k < ns ? 1.0f : 0.0f; val[k] = k < ns ? val[k] : 3.0e38f; }
						const uint32_t c32 = (uint32_t)code, half = c32 >> 1, delta = (c32 & 1u) ? 0u - half - 1u : half;
						// constant lanes: a far code (ordered value = the code after a prediction >= +0, its complement after a negative one),
						// a record without sources, a lane that was put right
						uint32_t const_mask = (c32 >> 31) ? ~0u : 0u;
						uint32_t after_pos = as_u32<T>(cm::f32_from_ordered(c32)), after_neg = as_u32<T>(cm::f32_from_ordered(~c32));
						if (none) { const_mask = ~0u; after_pos = after_neg = as_u32<T>(cm::value_from_residual<T>(code, T(0), q)); }
						int from = lo;
						bool exact_form = false;
						uint32_t outb = 0;
						for (int tries = 0;; ++tries) {
							// (compiled per source count of the run: a step is 7 issue slots per source + 7)
							auto steps = [&](auto nn) {
								constexpr int NN = decltype(nn)::value;
								if constexpr (NN <= N)
									for (int i = from; i < hi; ++i) outb = short_step<NN>(val, slot, w, rcp, delta, const_mask, after_pos, after_neg, i);
							};
							if (most <= 2) steps(std::integral_constant<int, 2>());
							else if (most == 3) steps(std::integral_constant<int, 3>());
							else if (most == 4) steps(std::integral_constant<int, 4>());
							else if (most == 5) steps(std::integral_constant<int, 5>());
							else steps(std::integral_constant<int, 6>());
							out = from_u32<T>(outb);
							GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); if (__ballot(out == out)) ck_steps += n - ck_t; ck_t = n; })
							const uint32_t ref = as_u32<T>(cm::value_from_residual<T>(code, predict_from<KIND, T, CAP, N>(ns, q, val), q));
							const unsigned long long bad = __ballot(in_run && ref != as_u32<T>(out));
							GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_verify += n - ck_t; ck_t = n; ++ck_evals; })
							if (!bad) break;
							if (tries >= 6) { exact_form = true; break; }
							const int f = __builtin_ctzll(bad);
							if (lane == f) { const_mask = ~0u; after_pos = after_neg = ref; }
							from = f;
						}
						if (!exact_form) return;
					}
					if constexpr (KIND == 1 && std::is_unsigned<T>::value && sizeof(T) <= 4) {
						// Integer corner records (quantised normals and texture coordinates), the deep case: the reference's step is the
						// rounded mean of the sources in 64 bits and the residual code against it (attrcode.h:182-190, transform.h:91,
						// prediction.h:46-64).  With at most 12 sources of at most 27 bits the sum fits 31 bits, and the division by the
						// source count is a multiplication: q = (x M) >> (31 + s), s = ceil(log2 n), M = floor(2^(31+s) / n) + 1 is exact
						// for x < 2^31 (M < 2^32 for n >= 2); the residual code without its branches.  Exact by construction; verified
						// like the float form all the same (a damaged stream may wrap a narrow type where 32 bits do not).
						const int nbits = cm::width_bits<T>(q);
						if (sizeof(T) < 4 || nbits <= 27) {
							const uint32_t tmask = sizeof(T) == 4 ? ~0u : (1u << (8 * (sizeof(T) & 3))) - 1u;
							const uint32_t top = (uint32_t)cm::ones<T>(nbits);
							const uint32_t c32 = (uint32_t)code, half = c32 >> 1, dnear = (c32 & 1u) ? ~half : half;
							// (one source: x = mulhi(x + 1, 2^32 - 1); none: magic 0, prediction 0)
							const uint32_t un = (uint32_t)ns, rnd = un == 1 ? 1u : un >> 1;
							const int lg = un >= 2 ? 32 - __builtin_clz(un - 1) : 1;               // ceil(log2 n)
							// (2^(31+s) / n in double: exact to the integer part -- the fraction is a multiple of 1/n, far from the rounding)
							const uint32_t magic = un >= 2 ? (uint32_t)((double)(1ull << (31 + lg)) / (double)un) + 1u : un == 1 ? ~0u : 0u;
							const uint32_t shift = (uint32_t)(lg - 1);
							uint32_t vi[CAP];
#pragma unroll
							for (int k = 0; k < CAP; ++k) vi[k] = k < N && k < ns ? (uint32_t)val[k] : 0u;
							uint32_t ob = 0;
							auto steps = [&](auto nn) {
								constexpr int NN = decltype(nn)::value;
								if constexpr (NN <= N)
									for (int i = lo; i < hi; ++i) ob = short_step_int<NN>(vi, slot, rnd, magic, shift, top, half, c32, dnear, tmask, i);
							};
							if (N > 6 && most > 6) {   // (more than six sources: the compiler's version of the step)
								for (int i = lo; i < hi; ++i) {
									uint32_t sum = rnd;
#pragma unroll
									for (int k = 0; k < N; ++k) sum += vi[k];
									const uint32_t pr = __umulhi(sum, magic) >> shift;
									const uint32_t room = top - pr, bal = min(pr - 1u, room);
									const uint32_t farv = room >= pr ? pr + c32 - bal - 1u : pr - c32 + bal;
									uint32_t v = half > bal ? farv : pr + dnear;
									v = pr == 0u ? c32 : v;
									ob = v & tmask;
									const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)ob, i);
#pragma unroll
									for (int k = 0; k < N; ++k) vi[k] = slot[k] == (uint32_t)i ? x : vi[k];
								}
							}
							else if (most <= 2) steps(std::integral_constant<int, 2>());
							else if (most == 3) steps(std::integral_constant<int, 3>());
							else if (most == 4) steps(std::integral_constant<int, 4>());
							else if (most == 5) steps(std::integral_constant<int, 5>());
							else steps(std::integral_constant<int, 6>());
#pragma unroll
							for (int k = 0; k < N; ++k) val[k] = k < ns ? (T)vi[k] : val[k];
							out = (T)ob;
							GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); if (__ballot(out == out)) ck_steps += n - ck_t; ck_t = n; })
							const uint32_t ref = as_u32<T>(cm::value_from_residual<T>(code, predict_from<KIND, T, CAP, N>(ns, q, val), q));
							const bool all_right = !__ballot(in_run && ref != as_u32<T>(out));
							GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_verify += n - ck_t; ck_t = n; ++ck_evals; })
							if (all_right) return;
						}
					}
					GEN_CLK(++ck_exact_n;)
					for (int i = lo; i < hi; ++i) {
						out = cm::value_from_residual<T>(code, predict_from<KIND, T, CAP, N>(ns, q, val), q);
						const T x = from_u32<T>((uint32_t)__builtin_amdgcn_readlane((int)as_u32<T>(out), i));
#pragma unroll
						for (int k = 0; k < N; ++k) val[k] = slot[k] == (uint32_t)i ? x : val[k];
					}
					// (lane j's `out` is final from step j on: its picks were complete then and nothing it holds changes afterwards)
					GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); if (__ballot(out == out)) ck_exact += n - ck_t; ck_t = n; })
				};
				constexpr int N1 = CAP / 4 >= 3 ? (CAP / 4 / 3) * 3 : CAP / 4, N2 = CAP / 2;   // multiples of three for the parallelograms of KIND 0
				if (most <= N1) rounds(std::integral_constant<int, N1>());
				else if (most <= N2) rounds(std::integral_constant<int, N2>());
				else rounds(std::integral_constant<int, CAP>());
				GEN_CLK(ck_t = __builtin_amdgcn_s_memtime();)
				if (in_run) store_g<T>(g_mine, out, aligned);
			}
			__threadfence();   // the run's records are in memory before anything later reads them
			GEN_CLK({ const unsigned long long n = __builtin_amdgcn_s_memtime(); ck_store += n - ck_t; ck_t = n; })
			if (hi < 64) {
				if (lane == hi && live) {   // a fan too large for the table: this lane alone, every source from memory
					auto value = [&](uint32_t x) { return x < i ? far_value<T>(jb.rec + (size_t)x * ld.stride + off, aligned) : T(0); };
					const uint32_t e = jb.ev_he[i];
					const int a = jb.ev_slot[i];
					const T pred = combine_parts<T>([&](auto &&use) {
						if constexpr (KIND == 0) {
							T tri[3];
							int k = 0;
							walk_sources<0>(tp, gv, rank, e, a, [&](uint32_t x) { tri[k++] = value(x); if (k == 3) { use(cm::parallelogram<T>(tri[0], tri[1], tri[2], q)); k = 0; } });
						} else walk_sources<1>(tp, gv, rank, e, a, [&](uint32_t x) { use(value(x)); });
					});
					stg<T>(mine, cm::value_from_residual<T>(code, pred, q));
				}
				__threadfence();
			}
			lo = hi + 1;
		}
		if (!fetched) fetch(base + 64);
	}
	GEN_CLK(if (lane == 0 && jb.n > 10000) printf("gen chain kind %d comp %d: %u records, %llu runs (single %llu, shallow %llu, exact %llu, evaluations %llu; N<=3 %llu, <=6 %llu, <=12 %llu, more %llu) | per run: load %llu depth %llu steps %llu verify %llu exact %llu shallow %llu store %llu | total %llu per record %llu\n",
	        KIND, c, jb.n, ck_runs, ck_single, ck_shallow_n, ck_exact_n, ck_evals, ck_n[0], ck_n[1], ck_n[2], ck_n[3], ck_load / ck_runs, ck_depth / ck_runs, ck_steps / ck_runs, ck_verify / ck_runs, ck_exact / ck_runs, ck_shallow / ck_runs, ck_store / ck_runs,
	        __builtin_amdgcn_s_memtime() - ck_begin, (__builtin_amdgcn_s_memtime() - ck_begin) / jb.n);)
}

// one instantiation per kind and storage type: the jobs of a launch all have that kind and type (the host groups them)
template <int KIND, typename T>
__global__ __launch_bounds__(64) void k_gen_chain(ConnView cv, GenView gv, const uint32_t *rank, const GenChainJob *jobs)
{
	__shared__ uint32_t s_val[64];
	// the components of a list write into the same records: every chain of the launch on ONE XCD (workgroups go round-robin over the
	// eight; only every eighth carries a chain), so that their stores meet in one L2 -- as the chains of the PLY layout do
	if (blockIdx.x & 7u) return;
	chain_component<KIND, T>(cv, gv, rank, jobs[blockIdx.x >> 3], s_val);
}

// ---------------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

void launch_face_rank(hipStream_t st, const ConnView &cv, const uint32_t *order_f, uint32_t n, uint32_t *frank)
{
	if (n) hipLaunchKernelGGL(k_face_rank, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, order_f, n, frank);
}
void launch_gen_vtx_resid(hipStream_t st, const ConnView &cv, const GenView &gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                          const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes)
{
	if (n) hipLaunchKernelGGL(k_gen_vtx_resid, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, gv, rank, ev_he, ev_slot, ev_idx, n, rec, ld, planes);
}
void launch_gen_face_resid(hipStream_t st, const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes)
{
	if (n) hipLaunchKernelGGL(k_gen_face_resid, dim3(blocks_for(n, 256)), dim3(256), 0, st, ev_idx, n, rec, ld, planes);
}
void launch_gen_corner_resid(hipStream_t st, const ConnView &cv, const GenView &gv, const uint32_t *frank, const uint32_t *ev_he, const uint8_t *ev_slot,
                             const uint32_t *ev_idx, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes)
{
	if (n) hipLaunchKernelGGL(k_gen_corner_resid, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, gv, frank, ev_he, ev_slot, ev_idx, n, rec, ld, planes);
}
void launch_gen_sources(hipStream_t st, int kind, const ConnView &cv, const GenView &gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                        uint32_t n, uint32_t *src, uint8_t *nsrc)
{
	if (!n) return;
	if (kind == 0) hipLaunchKernelGGL(k_gen_sources<0>, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, gv, rank, ev_he, ev_slot, n, src, nsrc);
	else hipLaunchKernelGGL(k_gen_sources<1>, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, gv, rank, ev_he, ev_slot, n, src, nsrc);
}
template <int KIND>
static void launch_chain_kind(hipStream_t st, int stype, const ConnView &cv, const GenView &gv, const uint32_t *rank, const GenChainJob *jobs, uint32_t njobs)
{
	switch (stype) {
	case 0: hipLaunchKernelGGL((k_gen_chain<KIND, float>), dim3((njobs - 1) * 8 + 1), dim3(64), 0, st, cv, gv, rank, jobs); break;
	case 4: hipLaunchKernelGGL((k_gen_chain<KIND, uint32_t>), dim3((njobs - 1) * 8 + 1), dim3(64), 0, st, cv, gv, rank, jobs); break;
	case 5: hipLaunchKernelGGL((k_gen_chain<KIND, int32_t>), dim3((njobs - 1) * 8 + 1), dim3(64), 0, st, cv, gv, rank, jobs); break;
	case 6: hipLaunchKernelGGL((k_gen_chain<KIND, uint16_t>), dim3((njobs - 1) * 8 + 1), dim3(64), 0, st, cv, gv, rank, jobs); break;
	case 7: hipLaunchKernelGGL((k_gen_chain<KIND, int16_t>), dim3((njobs - 1) * 8 + 1), dim3(64), 0, st, cv, gv, rank, jobs); break;
	case 8: hipLaunchKernelGGL((k_gen_chain<KIND, uint8_t>), dim3((njobs - 1) * 8 + 1), dim3(64), 0, st, cv, gv, rank, jobs); break;
	case 9: hipLaunchKernelGGL((k_gen_chain<KIND, int8_t>), dim3((njobs - 1) * 8 + 1), dim3(64), 0, st, cv, gv, rank, jobs); break;
	default: break;   // 8-byte storage types are rejected on the host
	}
}
// jobs: device array of njobs jobs that all have this kind and storage type
void launch_gen_chain(hipStream_t st, int kind, int stype, const ConnView &cv, const GenView &gv, const uint32_t *rank, const GenChainJob *jobs, uint32_t njobs)
{
	if (!njobs) return;
	if (kind == 0) launch_chain_kind<0>(st, stype, cv, gv, rank, jobs, njobs);
	else launch_chain_kind<1>(st, stype, cv, gv, rank, jobs, njobs);
}

}   // namespace dev
}   // namespace hry
