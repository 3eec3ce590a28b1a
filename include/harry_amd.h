/*
 * harry_amd.h -- C ABI of the MI355X-native .hry codec path (libharry_amd.so).
 *
 * Drop-in boundary for the attribute-quantisation + arithmetic-coding hot path of maxvonbuelow/harry.
 * The reference has no FFI; its seams are C++ free functions.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference tree):
 *
 *   hry_mesh_from_ply   <- ply::reader::read(std::istream&, mesh::Mesh&)          formats/ply/reader.cc:382-429
 *   hry_mesh_to_ply     <- ply::writer::write(std::ostream&, mesh::Mesh&, bool)   formats/ply/writer.cc:136-192
 *   hry_mesh_from_obj   <- obj::reader::read(std::istream&, const std::string&, mesh::Mesh&)  formats/obj/reader.rl:287-299
 *   hry_mesh_to_obj     <- obj::writer::write(std::ostream&, const std::string&, mesh::Mesh&) formats/obj/writer.cc:20-132
 *   hry_requant         <- quant::requant(Attrs&, const vector<Quant>&, bool)     structs/quant.h:222-242 (+ main.cc:74-91)
 *   hry_encode          <- hry::writer::write(std::ostream&, mesh::Mesh&)         formats/hry/writer.h:19, writer.cc:200-218
 *   hry_decode          <- hry::reader::read(std::istream&, mesh::Mesh&)          formats/hry/reader.h:19, reader.cc:179-193
 *   hry_bounds          <- quant::set_bounds(Attrs&)                              structs/quant.h:30-44 (called by ply/reader.cc:428)
 *
 * Plain pointers and sizes only; no C++/torch types.  All functions return HRY_OK (0) or a negative error
 * code; hry_last_error() returns the message of the calling thread's last failure (the reference throws
 * std::runtime_error with the same texts, e.g. "Invalid magic number", formats/hry/reader.cc:70).
 *
 * The compute stages run as hand-written HIP kernels on gfx950.  There is no CPU fallback: without a HIP
 * device hry_ctx_create() fails with HRY_E_NODEVICE and nothing can be encoded or decoded.
 */
#ifndef HARRY_AMD_H
#define HARRY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HRY_ABI_VERSION 6

enum {
    HRY_OK = 0,
    HRY_E_ARG = -1,         /* invalid argument */
    HRY_E_FORMAT = -2,      /* malformed PLY / .hry input */
    HRY_E_UNSUPPORTED = -3, /* valid input outside the supported subset */
    HRY_E_NODEVICE = -4,    /* no usable HIP device / HIP runtime error */
    HRY_E_NOMEM = -5,
    HRY_E_INTERNAL = -6
};

/* component types: numeric values of mixing::Type (structs/mixing.h:19), as stored in the .hry header */
enum { HRY_FLOAT = 0, HRY_DOUBLE, HRY_ULONG, HRY_LONG, HRY_UINT, HRY_INT, HRY_USHORT, HRY_SHORT, HRY_UCHAR, HRY_CHAR };

/* output profiles: COMPAT is byte-identical to the reference's single stream (v0.1);
 * CHUNKED is the version-tagged (v0.2) parallel container: same symbols, independent coder per chunk. */
enum { HRY_PROFILE_COMPAT = 0, HRY_PROFILE_CHUNKED = 1 };

typedef struct hry_ctx hry_ctx;   /* device context: HIP device, streams, workspace */
typedef struct hry_mesh hry_mesh; /* host-side mesh: flat arrays (see accessors) */

/* one -q request: list, component (-1 = every component of the list), bits (0 clears)  (main.cc:23-27,63-65) */
typedef struct hry_quant {
    int32_t list;
    int32_t comp;
    int32_t bits;
} hry_quant;

typedef struct hry_opts {
    int32_t profile;      /* HRY_PROFILE_* */
    int32_t chunk_syms;   /* CHUNKED: symbols per chunk and plane (0 = default) */
    int32_t keep_stages;  /* keep intermediate device buffers for hry_stage_get (tests) */
    int32_t flags;        /* HRY_FLAG_* */
    int32_t shard_index;  /* hry_decode of a sharded container (.hry v0.3): decode only the segments i with */
    int32_t shard_count;  /*   i % shard_count == shard_index (one process per GPU); 0 or 1 = every segment */
} hry_opts;

/* COMPAT only: the one strictly serial recurrence of the reference stream (the range register R of arith/coder.h:69-91,
 * SURVEY.md App. C-3) runs on a host core behind the device kernels (records stream down slice by slice); every parallel stage
 * stays on the device.  HRY_FLAG_DEVICE_RECURRENCE runs it on a single GPU wavefront instead (k_rchain: 8 times slower, same
 * bytes).  HRY_FLAG_HOST_RECURRENCE is accepted for older callers and changes nothing. */
#define HRY_FLAG_HOST_RECURRENCE 1
#define HRY_FLAG_DEVICE_RECURRENCE 2
/* hry_decode of a sharded container (.hry v0.3) that does not hold the whole mesh -- one rank's own part -- is an error unless this
 * flag (or a share, shard_count > 1) asks for a PARTIAL mesh: only the runs hry_mesh_runs lists are decoded, everything else is
 * filler (faces without half-edges, zero records).  The accessors read such a mesh; the writers, hry_encode, hry_mesh_upload,
 * hry_walk_run and hry_shard_plan refuse it (HRY_E_ARG). */
#define HRY_FLAG_PARTIAL 4
/* hry_encode_sharded: leave *m exactly as it is (do not store the combined bounds of the whole mesh in it) */
#define HRY_FLAG_KEEP_MESH 8

/* timings of the last hry_encode / hry_decode on this context, milliseconds */
typedef struct hry_timing {
    double host_walk_ms;   /* cut-border walk on the host (cbm/encoder.h:54-217 equivalent) */
    double h2d_ms;
    double device_ms;      /* all kernels, measured with HIP events on the codec stream */
    double d2h_ms;
    double total_ms;
    double k_rchain_ms;    /* compat: serial range recurrence kernel */
    double k_model_ms;     /* adaptive-model evaluation kernels */
    double k_predict_ms;   /* prediction + residual + symbolisation kernels; decode: candidates + chain records + the chain */
    double k_entropy_ms;   /* chunked: fused model+coder kernel */
    double k_chain_ms;     /* decode: the reconstruction chain kernels alone (k_unpredict2 / k_unpredict3) */
    uint64_t n_symbols;    /* coder invocations represented in the stream */
    uint64_t payload_bytes;
} hry_timing;

const char *hry_last_error(void);
int hry_abi_version(void);

/* ---- context ------------------------------------------------------------------------------------ */
int hry_device_count(void);   /* HIP devices this process sees (0: none) */
int hry_ctx_create(int device, hry_ctx **out);
void hry_ctx_destroy(hry_ctx *ctx);
int hry_ctx_timing(const hry_ctx *ctx, hry_timing *out);
/* the HIP stream all kernels of this context are launched on (hipStream_t as void*), for external event timing */
void *hry_ctx_stream(const hry_ctx *ctx);

/* ---- mesh (host) -------------------------------------------------------------------------------- */
int hry_mesh_from_ply(const uint8_t *ply, size_t n, hry_mesh **out);
/* Build from flat arrays: vertex records (AoS, one slot per component in its original type), polygon
 * degrees + flat vertex indices, optional face records.  Component names follow PLY conventions ("x","nx",
 * "red", ...; unknown names become named "other" interpretations, formats/ply/reader.cc:130-168). */
int hry_mesh_from_arrays(uint32_t nv, const uint8_t *vrec, int v_ncomp, const uint8_t *v_types, const char *const *v_names,
                         uint32_t nf, const uint8_t *degrees, const uint32_t *indices,
                         const uint8_t *frec, int f_ncomp, const uint8_t *f_types, const char *const *f_names,
                         hry_mesh **out);
/* flags: HRY_PLY_ASCII (the reference's --ply-ascii), HRY_PLY_PACKED: binary values of QUANTISED components in the width of the
 * storage type the header declares (a well-formed PLY).  Without it the binary writer does what the reference does: it announces
 * the storage type and dumps the whole original-width record (formats/ply/writer.cc:72-75,168) -- readable only by knowing
 * that.  `-c` (hry_requant with clear) before writing gives dequantised values in the original types instead. */
#define HRY_PLY_ASCII 1
#define HRY_PLY_PACKED 2
int hry_mesh_to_ply(const hry_mesh *m, int flags, uint8_t **out, size_t *out_len);
/* OBJ (formats/obj/reader.rl, writer.cc): positions (+ colours) per vertex; texture coordinates and normals per CORNER, shared
 * between corners; "usemtl" materials become face regions.  Such a mesh has GENERAL bindings (structs/attr.h:101-189): any
 * number of lists (hry_mesh_nlists / hry_list_target), faces and vertices belong to regions, a region names the lists its
 * elements (and, for face regions, their corners) carry, and every element holds one record index per list of its region.
 * The reader follows the reference's scanner where that differs from the OBJ specification (harry_amd/csrc/host/obj_io.cpp).
 * `dir`: where "mtllib" files are looked up (the reference passes the input path up to its last '/', formats/unified_reader.h:56).
 * hry_encode codes general bindings into the reference stream (HRY_PROFILE_COMPAT) and into the chunked container
 * (HRY_PROFILE_CHUNKED, .hry v0.2). */
int hry_mesh_from_obj(const uint8_t *obj, size_t n, const char *dir, hry_mesh **out);
int hry_mesh_to_obj(const hry_mesh *m, int flags, uint8_t **out, size_t *out_len);   /* flags: 0 */
int hry_mesh_general(const hry_mesh *m);                   /* 0: the PLY layout (list 0 = face, list 1 = vertex attributes, record i of element i) */
int hry_list_target(const hry_mesh *m, int l);             /* 0 face, 1 vertex, 2 corner, 3 none (structs/attr.h:22) */
int hry_mesh_nregions(const hry_mesh *m, int which);       /* which: 0 face regions, 1 vertex regions */
/* lists bound to region r: kind 0 = face lists, 1 = vertex lists, 2 = corner lists of face region r; returns their number */
int hry_mesh_region_lists(const hry_mesh *m, int kind, int r, uint16_t *out, int cap);
/* general bindings only: region of every face (which 0) / vertex (which 1); returns the element count */
size_t hry_mesh_regions_of(const hry_mesh *m, int which, const uint16_t **out);
/* general bindings only: record index per element and slot (kind 0 faces, 1 vertices, 2 corners = half-edges), row-major with
 * *slots entries per element; returns the element count */
size_t hry_mesh_bindings(const hry_mesh *m, int kind, const uint32_t **out, int *slots);
void hry_mesh_free(hry_mesh *m);
hry_mesh *hry_mesh_clone(const hry_mesh *m);

uint32_t hry_mesh_nv(const hry_mesh *m);
uint32_t hry_mesh_nf(const hry_mesh *m);
uint32_t hry_mesh_ne(const hry_mesh *m);
uint64_t hry_mesh_ntri(const hry_mesh *m);              /* sum(ne - 2), structs/conn.h:87 */
const uint32_t *hry_mesh_face_offsets(const hry_mesh *m); /* nf + 1 */
const uint32_t *hry_mesh_org(const hry_mesh *m);          /* ne: origin vertex of each half-edge */
const uint32_t *hry_mesh_twin(const hry_mesh *m);         /* ne: flat id of the opposite half-edge (self = border) */
int hry_mesh_nlists(const hry_mesh *m);                   /* PLY layout: 2 (list 0 = face attributes, list 1 = vertex attributes) */
int hry_list_ncomp(const hry_mesh *m, int l);
uint32_t hry_list_count(const hry_mesh *m, int l);
int hry_list_stride(const hry_mesh *m, int l);
int hry_list_type(const hry_mesh *m, int l, int c);
int hry_list_quant(const hry_mesh *m, int l, int c);
int hry_list_offset(const hry_mesh *m, int l, int c);
const uint8_t *hry_list_data(const hry_mesh *m, int l);
const uint8_t *hry_list_min(const hry_mesh *m, int l);
const uint8_t *hry_list_max(const hry_mesh *m, int l);

/* ---- codec (device) ----------------------------------------------------------------------------- */
/* min/max per component on the GPU (k_bounds); hry_mesh_from_ply leaves bounds unset until first needed */
int hry_bounds(hry_ctx *ctx, hry_mesh *m);
int hry_requant(hry_ctx *ctx, hry_mesh *m, const hry_quant *q, size_t nq, int clear);
/* Keep the mesh's attribute records and connectivity resident in HBM for subsequent hry_encode calls -- for a mesh with general
 * bindings (OBJ) its region and record tables too. */
int hry_mesh_upload(hry_ctx *ctx, hry_mesh *m);
/* mesh -> .hry.  *out is allocated by the library, release with hry_free.  The mesh's twin array is updated
 * exactly as the reference's encoder mutates it (cbm/encoder.h:150,193-198). */
int hry_encode(hry_ctx *ctx, hry_mesh *m, const hry_opts *opts, uint8_t **out, size_t *out_len);
int hry_decode(hry_ctx *ctx, const uint8_t *hry, size_t n, const hry_opts *opts, hry_mesh **out);
/* every buffer the library hands out (*out of the encoders, writers and hry_merge) goes back through hry_free and nothing else:
 * large ones belong to the library's recycling pool (their pages serve the next call), not to the C library's heap */
void hry_free(void *p);
/* host-only: what a .hry file is, without decoding it.  info[0] minor version (1 reference stream, 2 chunked, 3 sharded chunked),
 * [1] header bytes, [2] vertices, [3] faces, [4] half-edges, [5] symbols per chunk and plane (first segment; 0 for v0.1),
 * [6] the same for the connectivity planes, [7] segments (v0.3; else 1) */
int hry_container_info(const uint8_t *hry, size_t n, uint32_t info[8]);

/* ---- one mesh over several GPUs ---------------------------------------------------- */
/* The reference has no multi-device path; what a split must honour is its numbering: vertices, faces and half-edges of the
 * decoded mesh are numbered in coding order across ALL connected components (cbm/encoder.h:61-68,215; cbm/decoder.h:48,75,
 * 145,162), components that share a vertex name it by that number (cbm/encoder.h:79-113,187), and the order of the
 * components is the start-face sequence over the whole mesh (formats/hry/writer.cc:28-46).
 *   hry_shard_plan     host analysis of the connectivity (no walk): components, their coding order, groups of components tied
 *                      by shared vertices, exclusive scans of the vertices / faces / half-edges each introduces, and the
 *                      distribution of the groups over n_shards (balanced by triangle count).  Deterministic: every rank
 *                      that holds the mesh computes the same plan.
 *   hry_shard_extract  the sub-mesh of one shard in its own compact numbering; it carries the seed face of each of its
 *                      components and the place of its runs of components in the whole numbering.  hry_encode (CHUNKED) of
 *                      a shard writes a one-segment sharded container (.hry v0.3) whose header describes the WHOLE mesh:
 *                      give the shard the bounds of the whole mesh first (hry_list_set_bounds) -- they are in that header
 *                      and scale the quantisation (structs/quant.h:30-96).
 *   hry_merge          concatenates the segments of several such containers into ONE .hry v0.3 (no re-coding).
 *   hry_decode         reads it on one GPU (all segments) or, with opts->shard_index / shard_count, a share of the segments
 *                      per process; hry_mesh_runs lists the runs of the whole numbering that the returned mesh holds. */
typedef struct hry_plan hry_plan;
int hry_shard_plan(const hry_mesh *m, int n_shards, hry_plan **out);
void hry_plan_free(hry_plan *p);
uint32_t hry_plan_ncomponents(const hry_plan *p);
uint32_t hry_plan_ngroups(const hry_plan *p);
uint64_t hry_plan_triangles(const hry_plan *p, int shard);
int hry_shard_extract(const hry_mesh *m, const hry_plan *p, int shard, hry_mesh **out);
int hry_merge(const uint8_t *const *parts, const size_t *sizes, size_t n, uint8_t **out, size_t *out_len);
/* runs as 6 x u32 each: first_vertex, first_face, first_halfedge, n_vertices, n_faces, n_halfedges (numbering of the whole
 * mesh).  For a shard: where its components go; for a mesh decoded from a sharded container: what was decoded. */
size_t hry_mesh_runs(const hry_mesh *m, const uint32_t **runs);
/* for a shard: index in the whole mesh of every vertex (which = 1) / face (which = 0) of the shard; which = 2: the start face of
 * each of its components (shard numbering) in coding order; which = 16 + l (general bindings): index in the whole mesh of every
 * record of list l */
size_t hry_shard_elements(const hry_mesh *m, int which, const uint32_t **idx);
/* bounds of a list as records in the original component types (what hry_list_min / hry_list_max return) */
int hry_list_set_bounds(hry_mesh *m, int l, const uint8_t *min_rec, const uint8_t *max_rec);
/* after hry_bounds: 1 + index of the first element that holds the minimum / maximum of component c, 0 = the initial value of
 * the reference's scan (structs/quant.h:33) -- what a combination of per-shard bounds needs to break ties (+-0.0) like one scan */
uint32_t hry_list_min_at(const hry_mesh *m, int l, int c);
uint32_t hry_list_max_at(const hry_mesh *m, int l, int c);
int hry_mesh_partial(const hry_mesh *m);   /* 1: decoded with HRY_FLAG_PARTIAL / as a share; only its runs are real */

/* ---- the same from ONE process over several devices ---------------------------------------------------------------
 * The reference's single entry (main.cc:93-123 -> quant::requant -> hry::writer::write, formats/hry/writer.cc:200-214) with
 * N device contexts behind it: what scales is "host thread + context" (the sequential cut-border walk of a shard on a host
 * core, the kernels of that shard on the context's device), so every context gets a worker thread of its own, confined to the
 * memory node of its device.  Contexts may sit on different devices (one per GPU of a node) or share one.
 *   hry_encode_sharded  plans once, extracts the shards on the workers, combines the shards' k_bounds results into the bounds of
 *                       the whole mesh with the tie rule of ONE scan (structs/quant.h:30-44), quantises (quant / n_quant / clear
 *                       as hry_requant; the shards are quantised, *m keeps its values and receives the bounds) and codes every
 *                       shard (CHUNKED) on its worker, and concatenates the segments in host memory: ONE .hry v0.3, byte-identical
 *                       to hry_shard_plan / hry_shard_extract / hry_encode / hry_merge run shard by shard.  opts->shard_count =
 *                       number of shards (0: one per context); shard s is coded by context s % n_ctx.
 *   hry_decode_sharded  decodes the segments of a sharded container on the contexts (segment i on context i % n_ctx) into ONE
 *                       mesh in the numbering of the whole; opts->shard_index / shard_count select a share as in hry_decode.
 * hry_ctx_timing of each context holds the sums over the shards / segments it processed. */
typedef struct hry_shard_timing {
    double twins_ms;     /* encode: half-edge twin matching of a freshly read mesh (first context's device), part of plan_ms */
    double plan_ms;      /* encode: twins + components + coding order + scans + distribution; decode: directory checks */
    double extract_ms;   /* encode: hry_shard_extract (decode: placement into the whole numbering), max over the workers */
    double bounds_ms;    /* upload + k_bounds per shard, max over the workers */
    double combine_ms;   /* bounds of the whole mesh from the shards' */
    double quant_ms;     /* quantisation: of the whole lists on the first context (device plan), of the extracted shards (max over the workers); coded in place on other devices it is part of extract_ms */
    double encode_ms;    /* hry_encode (decode: the segments' decode), max over the workers */
    double merge_ms;     /* concatenation of the segments (decode: filler for what no decoded run covers) */
    double phase_a_ms;   /* wall clock: extraction + bounds on all workers */
    double phase_b_ms;   /* wall clock: quantisation + encode (decode: decode + placement) on all workers */
    double host_walk_ms; /* the longest worker's cut-border walks / replays */
    double total_ms;
    uint32_t n_shards, n_contexts, n_segments, n_components, n_groups;
} hry_shard_timing;
int hry_encode_sharded(hry_ctx *const *ctx, int n_ctx, hry_mesh *m, const hry_quant *quant, size_t n_quant, int clear,
                       const hry_opts *opts, uint8_t **out, size_t *out_len, hry_shard_timing *timing /* may be NULL */);
int hry_decode_sharded(hry_ctx *const *ctx, int n_ctx, const uint8_t *hry, size_t n, const hry_opts *opts, hry_mesh **out,
                       hry_shard_timing *timing /* may be NULL */);
/* host-only: the directory of a sharded container checked against its header (segment extents, run tables, runs inside the mesh,
 * no overlap, order of faces and half-edges consistent); *complete = every face and half-edge lies in some run */
int hry_container_check(const uint8_t *hry, size_t n, int *complete);

/* ---- stage-level access for parity tests (valid after hry_encode/hry_decode with keep_stages) ----- */
/* names: "order_v","order_f","twin","vplanes","fplanes","rec","sym_l","r","S","payload", ... */
int hry_stage_get(hry_ctx *ctx, const char *name, void **host_copy, size_t *bytes);

/* host-only: the sequential cut-border walk of the encoder (cbm::encode, cbm/encoder.h:54-217) with a recording
 * writer.  Mutates the mesh's twins like hry_encode.  Arrays by name: "order_v", "order_f", "op_sym"(u8), "op_class"(u8),
 * "op_l", "op_h", "op_t", "op_pos", "grp<k>_val", "grp<k>_pos" with k = 0 iop, 1 elem, 2 part, 3 vertid, 4 numtri
 * (u32 unless noted), "info" = { n_conn, numtri_coded }.  hry_walk_get returns the element count. */
typedef struct hry_walk hry_walk;
int hry_walk_run(hry_mesh *m, hry_walk **out);
/* same walk without evaluating the operation model (what the chunked profile uses; then "op_l/op_h/op_t/op_pos" are empty and
 * components after the first may be walked on several host threads: HRY_HOST_THREADS, default min(16, cores)) */
int hry_walk_run_plain(hry_mesh *m, hry_walk **out);
/* host-only: the components of shard `shard` of a plan (hry_shard_plan), walked where they lie in the whole mesh -- what a worker
 * of hry_encode_sharded does instead of walking an extracted sub-mesh: same symbols as hry_walk_run_plain of hry_shard_extract's
 * mesh, "order_v" / "order_f" as half-edges of the WHOLE mesh.  Mutates the mesh's twins like hry_encode. */
int hry_walk_run_shard(hry_mesh *m, const hry_plan *plan, int shard, hry_walk **out);
/* development / tests: the component analysis of the mesh (connected components, coding order, sizes, new vertices, ties -- what
 * hry_encode of a large mesh computes on the device before its walk, analysis.cpp) by the device AND by the host, compared table by
 * table; 0 = equal (or fewer than two components), HRY_E_INTERNAL with the first difference otherwise.  Uploads the mesh. */
int hry_analysis_check(hry_ctx *ctx, hry_mesh *m);
size_t hry_walk_get(const hry_walk *w, const char *name, const void **ptr);
void hry_walk_free(hry_walk *w);

/* host-only: the serial half of reading a reference stream (.hry v0.1): entropy decoding of the single adaptive
 * stream interleaved with the cut-border replay (hry::reader::read, formats/hry/reader.cc:179-193; cbm::decode,
 * cbm/decoder.h:27-211; arith/coder.h:115-172).  *mesh gets the connectivity (attribute records still zero); the
 * result holds "order_v" (u32, decode order as half-edges), "vplanes"/"fplanes" (u8, residual byte planes, plane-major)
 * for the device reconstruction.  Free with hry_walk_free / hry_mesh_free. */
int hry_stream_read_host(const void *hry, size_t bytes, hry_mesh **mesh, hry_walk **out);

/* host-only: the decoder-side cut-border replay (cbm::decode, cbm/decoder.h:27-211) of the connectivity symbols a plain
 * walk recorded, as the chunked container carries them (21 byte planes).  use_restart_points 1: cut the replay at the
 * restart points the container directory would hold and replay the spans on several host threads (HRY_HOST_THREADS,
 * HRY_PARALLEL_MIN_FACES); 3: also at the border snapshots INSIDE the components (what hry_walk_run_plain noted every
 * HRY_SNAPSHOT_FACES faces of a component; through the directory's form and back): spans with placeholders for the border's
 * half-edges, joined afterwards.  *mesh gets nv/nf and the rebuilt connectivity; the result holds "order_v", "seg_start",
 * "seg_level" (u32) and "info" = { number of restart points, number of border snapshots }; hry_walk_get(walk, "snap_section")
 * is the directory section of a walk's snapshots. */
int hry_walk_replay(const hry_mesh *src, const hry_walk *walk, int use_restart_points, hry_mesh **mesh, hry_walk **out);

/* raw range-coder back end on explicit (l,h,t) triples (arith/coder.h:69-91 + flush :58-67), compat form */
int hry_range_encode_lht(hry_ctx *ctx, const uint64_t *lht, size_t n, uint8_t **out, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
