// Launch wrappers of the HIP kernels (kernels.hip, chunked.hip).  Everything is enqueued on the given stream.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "dev_types.hpp"

namespace hry {
namespace dev {

// every component of a list in two launches; out: 24 bytes per component { u64 min, u64 max, u32 min_at, u32 max_at }
void launch_bounds(hipStream_t st, const uint8_t *rec, uint32_t count, const BoundsPlan &plan,
                   uint8_t *part_min, uint8_t *part_max, uint32_t *part_idx, int nparts, uint8_t *out);
void launch_requant(hipStream_t st, uint8_t *rec, uint32_t count, int stride, const RequantPlan &plan);
void launch_rank(hipStream_t st, const uint32_t *order_v, uint32_t n, const uint32_t *org, uint32_t *rank);
void launch_predict_vtx(hipStream_t st, const ConnView &cv, const uint32_t *order_v, uint32_t n, const uint32_t *rank, const uint8_t *rec,
                        const ListDesc &ld, uint8_t *planes);
void launch_face_planes(hipStream_t st, const ConnView &cv, const uint32_t *order_f, uint32_t n, const uint8_t *rec, const ListDesc &ld, uint8_t *planes);
// the same over a batch of runs of the coding order (chunked.cpp: EncodePipeline): start = exclusive scan of the runs' lengths
// (nruns + 1 entries, device), first = where each run begins in the coding order, total = start[nruns]; n = all coded elements;
// packed = the runs' entries of the host array back to back (they take their places in order_v / order_f on the way)
void launch_rank_runs(hipStream_t st, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *packed, uint32_t *order_v, const uint32_t *org, uint32_t *rank);
void launch_predict_vtx_runs(hipStream_t st, const ConnView &cv, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *order_v, uint32_t n,
                             const uint32_t *rank, const uint8_t *rec, const ListDesc &ld, uint8_t *planes);
void launch_face_planes_runs(hipStream_t st, const ConnView &cv, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *packed, uint32_t *order_f, uint32_t n,
                             const uint8_t *rec, const ListDesc &ld, uint8_t *planes);
void launch_split_bytes_runs(hipStream_t st, const uint32_t *start, const uint32_t *first, uint32_t nruns, uint32_t total, const uint32_t *packed, const uint32_t *val, uint32_t n, int nbytes, uint8_t *planes);
// (packed == nullptr in any of them: the runs' entries were copied to their places in order_v / order_f / val run by run)
void launch_edge_faces(hipStream_t st, const uint32_t *foff, uint32_t nf, uint32_t *eface, uint32_t first = 0);   // faces [first, nf)
void launch_magic_table(hipStream_t st, MagicEnt *tab, uint32_t from, uint32_t to);
void launch_split_bytes(hipStream_t st, const uint32_t *val, uint32_t n, int nbytes, uint8_t *planes);
// the order-conditioned operation model of the reference stream, by counting; op: symbol | class << 3 per operation, thr / cum: op_position_table
size_t op_model_scratch_bytes(uint32_t n);
void launch_op_model(hipStream_t st, const uint8_t *op, uint32_t n, const uint32_t *thr, const uint32_t *cum, uint32_t ngroups, void *scratch,
                     const MagicEnt *magic, SymRec *rec, uint32_t *sym_l);
void launch_op_records(hipStream_t st, const uint32_t *l, const uint32_t *h, const uint32_t *t, const uint32_t *pos, uint32_t n,
                       const MagicEnt *magic, SymRec *rec, uint32_t *sym_l);
void launch_type_records(hipStream_t st, uint32_t n, uint32_t pos_base, uint32_t pos_stride, const MagicEnt *magic, SymRec *rec, uint32_t *sym_l);
void launch_model(hipStream_t st, const PlaneJob *jobs, uint32_t njobs, const ChunkRef *chunks, uint32_t nchunks, uint32_t *hist,
                  const MagicEnt *magic, SymRec *rec, uint32_t *sym_l);
void launch_rchain(hipStream_t st, const SymRec *rec, uint32_t n, uint64_t *r_out, uint32_t *s_out, uint64_t *state);
void launch_low_accumulate(hipStream_t st, const uint64_t *r, const uint32_t *s, const uint32_t *sym_l, uint32_t n, uint64_t *acc);
struct PullRanges { uint32_t *dst[4]; const uint32_t *src[4]; uint32_t words[4]; };   // device destinations, pinned host sources (device-visible), 32-bit words each
void launch_pull_ranges(hipStream_t st, const PullRanges &r);
void launch_carry(hipStream_t st, const uint64_t *acc, uint32_t nw, uint64_t *v, uint32_t *summary, uint8_t *bytes, const StreamJob *jobs = nullptr, const uint32_t *stream_bits = nullptr, uint32_t ns = 0);

// half-edge twin matching (twins.hip): conn.org / foff (/ eface) resident, twin = output; ws: twin_workspace_bytes
size_t twin_workspace_bytes(uint32_t nv, uint32_t ne);
uint32_t twin_overflow_capacity();
// *over: device pointer (inside ws) of the list of vertices with too many half-edges for the kernel: count, then vertex ids
void launch_twins(hipStream_t st, const ConnView &cv, uint32_t nv, uint32_t *twin, void *ws, const uint32_t **over);
// connected components of the faces and their tables for the walk on several host threads (twins.hip; driver: analysis.cpp)
size_t components_workspace_bytes(uint32_t nv, uint32_t nf);
// where the workspace's parts lie (twins.hip decides; nobody else computes an offset into it): per face the component label (a
// root face after stage 1), the root flags, their exclusive scan (num[nf] = components) and the scan's block sums; per vertex the
// first component in coding order (stage 3); the list of (component, component) ties with its counter
struct ComponentsWorkspace { uint32_t *label, *flag, *num, *sums, *vfirst, *tie_count, *tie_pairs; };
ComponentsWorkspace components_workspace(void *ws, uint32_t nv, uint32_t nf);
void launch_components_label(hipStream_t st, const ConnView &cv, const ComponentsWorkspace &w);
void launch_components_faces(hipStream_t st, const ConnView &cv, uint32_t *label, const uint32_t *num, const uint32_t *spans, uint32_t nspans,
                             uint32_t *nfaces, uint32_t *nhe, uint32_t *flo, uint32_t *fhi, uint64_t *first_key);
void launch_components_vertices(hipStream_t st, const ConnView &cv, uint32_t nv, uint32_t ncomp, const ComponentsWorkspace &w, const uint32_t *rank_of,
                                uint32_t *tie, uint32_t *fresh, uint32_t *vlo, uint32_t *vhi);

// chunked profile (chunked.hip)
void launch_chunk_encode(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic, uint64_t *acc, uint32_t *stream_bits);
// the same streams in two kernels (thousands of streams: the model a wavefront per stream, the range registers a lane per stream):
// rec = 8 bytes per symbol of all streams (every stream: t0 + n <= 65535), rec_off[j] = symbols of the streams before j,
// order = stream indices, longest first
void launch_chunk_encode_split(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic, uint64_t *acc, uint32_t *stream_bits,
                               const uint64_t *rec_off, void *rec, const uint32_t *order);
void launch_stream_pack(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *stream_bits, const uint8_t *bytes,
                        uint32_t *nbytes, uint64_t *offsets, uint8_t *out, bool pack);
// stable partition of the operation bytes (symbol | class << 3) into one plane of symbols per class; base[c] = first byte of
// plane c in `planes`, scratch: 8 counters per unit of kSplitUnit operations
constexpr uint32_t kSplitUnit = 1024;
void launch_split_classes(hipStream_t st, const uint8_t *ops, uint32_t n, const uint32_t base[8], uint32_t *scratch, uint8_t *planes);
void launch_plane_hist(hipStream_t st, const HistSlice *slices, uint32_t nslices, uint32_t *hist);
void launch_scatter_u8(hipStream_t st, const uint8_t *src, const uint32_t *dst_index, uint32_t n, uint8_t *dst);
void launch_chunk_decode(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic,
                         const uint8_t *payload, const uint64_t *offsets, const uint32_t *nbytes);
void launch_chunk_decode_lanes(hipStream_t st, const StreamJob *jobs, uint32_t nstreams, const uint32_t *inits, const MagicEnt *magic,
                               const uint8_t *payload, const uint64_t *offsets, const uint32_t *nbytes, bool counts16);   // one lane per stream; every job with t0 > 128 (counts16: and t0 + n <= 65535)

// events.hip: which record every element names, per list, on the device (the host's loop: host/general_events.cpp)
size_t events_list_workspace_bytes(uint32_t n_order, uint32_t max_refs, uint32_t list_count);
size_t events_names_workspace_bytes(uint32_t fc, uint32_t corner_refs_max, uint32_t head_words);
void launch_corner_places(hipStream_t st, const ConnView &cv, const GenView &gv, const EvRegions &rg, const uint32_t *order_f, uint32_t fc, uint32_t corner_refs_max,
                          uint32_t head_words, uint32_t nv, void *names_ws);
void launch_list_refs(hipStream_t st, int kind, uint32_t list, uint32_t list_count, const ConnView &cv, const GenView &gv, const EvRegions &rg,
                      const uint32_t *order, uint32_t n_order, uint32_t max_refs, uint32_t nv, uint32_t fc, uint32_t corner_refs_max, uint32_t head_words, void *names_ws, void *list_ws,
                      uint32_t *counts, uint32_t *err);
void launch_list_kinds(hipStream_t st, int kind, uint32_t list_count, const ConnView &cv, uint32_t n_order, uint32_t max_refs, uint32_t nv, uint32_t fc, uint32_t corner_refs_max,
                       uint32_t head_words, void *names_ws, void *list_ws, uint8_t *type_sym, uint32_t *gh_val, uint32_t *lh_val, uint32_t *d_idx, uint32_t *d_he, uint8_t *d_slot,
                       uint32_t *counts, uint32_t *err);
void launch_region_symbols(hipStream_t st, int kind, const ConnView &cv, const GenView &gv, const uint32_t *order, uint32_t n, uint8_t *out);

}   // namespace dev
}   // namespace hry
