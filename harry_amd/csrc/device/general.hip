// Kernels of the general-bindings path (OBJ input, regions, records shared between elements,
// corner attributes).  The host resolves WHICH record every vertex / face / corner names (history symbols are integer
// bookkeeping, device/general.cpp); these kernels do the value arithmetic for the records that are coded as data:
//
//   encode   k_gen_vtx_resid     parallelogram prediction restricted to the vertex' region + residual bytes   attrcode.h:117-134,209-225,321-344
//            k_gen_face_resid    residual against 0 (face prediction never finds a neighbour, App. B-16)       attrcode.h:245-270,345-366
//            k_gen_corner_resid  mean / nearest of the same slot's records at the already coded faces of the
//                                same region around the corner's vertex + residual bytes                      attrcode.h:135-154,272-288,367-393
//   decode   k_gen_unpredict     the inverse, record by record in coding order.  A record's prediction reads records coded
//                                before it (any of them: a corner may name a record that was created at another vertex), so the
//                                records of one list form a DAG with edges to smaller indices only: one thread per record
//                                waits for its sources' flags, computes, raises its own.  Workgroups start in index order and
//                                every wait is for a smaller index, so the waits cannot deadlock; they are bounded anyway.
// All byte / integer work next to dependent gathers: bounded by memory latency, not by bandwidth.
#include <hip/hip_runtime.h>

#include "codec_math.hpp"
#include "dev_types.hpp"
#include "fan.hpp"
#include "kernels.hpp"

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
__device__ uint32_t g_gen_timeout;
constexpr uint32_t kGenSpinLimit = 1u << 22;   // x s_sleep(4): seconds; a healthy wait is microseconds

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

// KIND 0: vertex records (sources: the three records of every parallelogram), 1: corner records (sources: one record per
// already decoded face of the region around the vertex).  rec holds the residual codes on entry (record layout), values on exit.
// Record i of the list was created by the i-th data symbol of the list; ev_he / ev_slot say where.
constexpr int kGenCap = 24;   // source ids kept per thread; fans with more are walked again whenever they are needed
template <int KIND>
__global__ __launch_bounds__(256) void k_gen_unpredict(ConnView cv, GenView gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                                                       uint32_t n, uint8_t *rec, ListDesc ld, uint32_t *done)
{
	__shared__ uint32_t s_src[kGenCap][256];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const int t = threadIdx.x;
	const bool live = i < n;
	Topo tp{ cv };
	uint32_t e = 0, my_rank = 0, r = 0;
	int a = 0, ns = 0;
	// every source id of this record, in the reference's visiting order: f(id) -- three consecutive calls per parallelogram for KIND 0
	auto walk = [&](auto &&f) {
		if constexpr (KIND == 0)
			fan_candidates(tp, rank, e, my_rank, 0, [&](uint32_t v0, uint32_t v1, uint32_t vo) {
				if (gv.vtx_reg[v0] != r || gv.vtx_reg[v1] != r || gv.vtx_reg[vo] != r) return;
				f(gv.vtx_attr[(size_t)v0 * gv.nb_vtx + a]); f(gv.vtx_attr[(size_t)v1 * gv.nb_vtx + a]); f(gv.vtx_attr[(size_t)vo * gv.nb_vtx + a]);
			});
		else
			fan_each(tp, e, [&](uint32_t x) {
				const uint32_t g = tp.face(x);
				if (g >= my_rank || gv.face_reg[g] != r) return;   // faces are decoded in index order (attrcode.h:543-548)
				f(gv.corner_attr[(size_t)x * gv.nb_corner + a]);
			});
	};
	if (live) {
		e = ev_he[i]; a = ev_slot[i];
		if constexpr (KIND == 0) { const uint32_t v = cv.org[e]; my_rank = rank[v]; r = gv.vtx_reg[v]; }
		else { my_rank = tp.face(e); r = gv.face_reg[my_rank]; }
		walk([&](uint32_t id) { if (ns < kGenCap) s_src[ns][t] = id; ++ns; });
	}
	auto sources = [&](auto &&f) {
		if (ns <= kGenCap) { for (int k = 0; k < ns; ++k) f(s_src[k][t]); }
		else walk(f);
	};
	bool pending = live;
	uint32_t spins = 0;
	for (;;) {
		if (pending) {
			bool ready = true;
			sources([&](uint32_t id) { if (id < i && __hip_atomic_load(done + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ready = false; });   // id >= i: damaged input, read as 0
			if (ready) {
				uint8_t *mine = rec + (size_t)i * ld.stride;
				for (int c = 0; c < ld.ncomp; ++c)
					with_stype(ld.stype[c], [&](auto tag) {
						typedef decltype(tag) T;
						typedef typename cm::word<sizeof(T)>::u U;
						const int off = ld.off[c], q = ld.quant[c];
						const bool aligned = (ld.stride % (int)sizeof(T)) == 0 && (off % (int)sizeof(T)) == 0;
						T pred;
						if constexpr (KIND == 0) {
							pred = combine_parts<T>([&](auto &&use) {
								T tri[3];
								int k = 0;
								sources([&](uint32_t id) {
									tri[k++] = id < i ? far_value<T>(rec + (size_t)id * ld.stride + off, aligned) : T(0);
									if (k == 3) { use(cm::parallelogram<T>(tri[0], tri[1], tri[2], q)); k = 0; }
								});
							});
						} else {
							pred = combine_parts<T>([&](auto &&use) {
								sources([&](uint32_t id) { use(id < i ? far_value<T>(rec + (size_t)id * ld.stride + off, aligned) : T(0)); });
							});
						}
						const U code = cm::bits<U>(ldg<T>(mine + off));
						stg<T>(mine + off, cm::value_from_residual<T>(code, pred, q));
					});
				__threadfence();   // the record is visible device-wide before its flag
				__hip_atomic_store(done + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				pending = false;
			}
		}
		if (__ballot(pending) == 0ull) break;
		__builtin_amdgcn_s_sleep(4);
		if (++spins > kGenSpinLimit) { atomicOr(&g_gen_timeout, 1u); break; }
	}
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
void launch_gen_unpredict(hipStream_t st, int kind, const ConnView &cv, const GenView &gv, const uint32_t *rank, const uint32_t *ev_he, const uint8_t *ev_slot,
                          uint32_t n, uint8_t *rec, const ListDesc &ld, uint32_t *done)
{
	if (!n) return;
	if (kind == 0) hipLaunchKernelGGL(k_gen_unpredict<0>, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, gv, rank, ev_he, ev_slot, n, rec, ld, done);
	else hipLaunchKernelGGL(k_gen_unpredict<1>, dim3(blocks_for(n, 256)), dim3(256), 0, st, cv, gv, rank, ev_he, ev_slot, n, rec, ld, done);
}
uint32_t gen_timeout_flags(hipStream_t st)
{
	uint32_t f = 0, zero = 0;
	if (hipMemcpyFromSymbolAsync(&f, HIP_SYMBOL(g_gen_timeout), 4, 0, hipMemcpyDeviceToHost, st) != hipSuccess) return 0;
	if (hipStreamSynchronize(st) != hipSuccess) return 0;
	if (f) { (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_gen_timeout), &zero, 4, 0, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
	return f;
}

}   // namespace dev
}   // namespace hry
