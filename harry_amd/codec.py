"""Host-side mirror of the reference's interface for the .hry path, on top of the C ABI (include/harry_amd.h).

Reference seams mirrored here (names and argument meaning follow the reference):
    unified::reader::read(fn, mesh)      formats/unified_reader.h:78-86   -> read_mesh(path)
    quant::requant(attrs, quants, clear) structs/quant.h:222-242          -> Codec.requant(mesh, quants, clear)
    hry::writer::write(os, mesh)         formats/hry/writer.h:19          -> Codec.write_hry(mesh)
    hry::reader::read(is, mesh)          formats/hry/reader.h:19          -> Codec.read_hry(data)
    ply::writer::write(os, mesh, ascii)  formats/ply/writer.cc:136-192    -> Mesh.to_ply(ascii)

All computation happens in libharry_amd.so (host C++ for the walk / PLY I/O, HIP kernels for the rest).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat
from ._native import FLAG_DEVICE_RECURRENCE, FLAG_HOST_RECURRENCE, FLAG_KEEP_MESH, FLAG_PARTIAL, HryError, PROFILE_CHUNKED, PROFILE_COMPAT  # noqa: F401

TYPE_NP = {0: "<f4", 1: "<f8", 2: "<u8", 3: "<i8", 4: "<u4", 5: "<i4", 6: "<u2", 7: "<i2", 8: "u1", 9: "i1"}
TYPE_SIZE = {0: 4, 1: 8, 2: 8, 3: 8, 4: 4, 5: 4, 6: 2, 7: 2, 8: 1, 9: 1}
NP_TYPE = {"f4": 0, "f8": 1, "u8": 2, "i8": 3, "u4": 4, "i4": 5, "u2": 6, "i2": 7, "u1": 8, "i1": 9}


def storage_type(t: int, q: int) -> int:
    """structs/mixing.h:101-108"""
    if q == 0:
        return t
    return 8 if q <= 8 else 6 if q <= 16 else 4 if q <= 32 else 2


class Mesh:
    """In-memory mesh (reference: mesh::Mesh, structs/mesh.h:19-40) as flat arrays owned by the native library."""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            nat.load().hry_mesh_free(self.h)
            self.h = None

    # ---- construction
    @classmethod
    def from_ply(cls, data: bytes) -> "Mesh":
        h = C.c_void_p()
        nat.check(nat.load().hry_mesh_from_ply(data, len(data), C.byref(h)))
        return cls(h)

    @classmethod
    def from_obj(cls, data: bytes, directory: str = "") -> "Mesh":
        """OBJ text -> mesh with general bindings (regions, shared records, corner lists); `directory`: where "mtllib" files are"""
        h = C.c_void_p()
        nat.check(nat.load().hry_mesh_from_obj(data, len(data), directory.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, verts: np.ndarray, degrees: np.ndarray, indices: np.ndarray, face_props: np.ndarray | None = None) -> "Mesh":
        """verts / face_props: numpy structured arrays (packed), one field per component."""
        L = nat.load()

        def pack(a):
            if a is None or a.dtype.names is None or len(a.dtype.names) == 0:
                return None, 0, None, None, []
            names = list(a.dtype.names)
            packed = np.empty(len(a), dtype=np.dtype([(n, a.dtype[n].newbyteorder("<")) for n in names]))
            for n in names:
                packed[n] = a[n]
            types = np.array([NP_TYPE[a.dtype[n].str[1:]] for n in names], np.uint8)
            cn = (C.c_char_p * len(names))(*[n.encode() for n in names])
            return packed, len(names), types, cn, names

        vp, vn, vt, vnames, _ = pack(verts)
        fp, fn, ft, fnames, _ = pack(face_props)
        degrees = np.ascontiguousarray(degrees, np.uint8)
        indices = np.ascontiguousarray(indices, np.uint32)
        h = C.c_void_p()
        nat.check(L.hry_mesh_from_arrays(
            len(verts), vp.ctypes.data if vp is not None else None, vn, vt.ctypes.data if vt is not None else None, vnames,
            len(degrees), degrees.ctypes.data, indices.ctypes.data,
            fp.ctypes.data if fp is not None else None, fn, ft.ctypes.data if ft is not None else None, fnames, C.byref(h)))
        return cls(h)

    def clone(self) -> "Mesh":
        h = nat.load().hry_mesh_clone(self.h)
        if not h:
            raise MemoryError("hry_mesh_clone")
        return Mesh(C.c_void_p(h))

    # ---- accessors
    nv = property(lambda s: nat.load().hry_mesh_nv(s.h))
    nf = property(lambda s: nat.load().hry_mesh_nf(s.h))
    ne = property(lambda s: nat.load().hry_mesh_ne(s.h))
    ntri = property(lambda s: nat.load().hry_mesh_ntri(s.h))

    def face_offsets(self):
        return nat.arr(nat.load().hry_mesh_face_offsets(self.h), self.nf + 1, np.uint32)

    def org(self):
        return nat.arr(nat.load().hry_mesh_org(self.h), self.ne, np.uint32)

    def twin(self):
        return nat.arr(nat.load().hry_mesh_twin(self.h), self.ne, np.uint32)

    def list_fmt(self, l):
        L = nat.load()
        return [(L.hry_list_type(self.h, l, c), L.hry_list_quant(self.h, l, c), L.hry_list_offset(self.h, l, c))
                for c in range(L.hry_list_ncomp(self.h, l))]

    def list_stride(self, l):
        return nat.load().hry_list_stride(self.h, l)

    def list_count(self, l):
        return nat.load().hry_list_count(self.h, l)

    def list_data(self, l) -> np.ndarray:
        n, s = self.list_count(l), self.list_stride(l)
        if n * s == 0:
            return np.zeros((n, s), np.uint8)
        return nat.arr(nat.load().hry_list_data(self.h, l), n * s, np.uint8).reshape(n, s)

    def list_min(self, l):
        p = nat.load().hry_list_min(self.h, l)
        return nat.arr(p, self.list_stride(l), np.uint8) if p else None

    def list_max(self, l):
        p = nat.load().hry_list_max(self.h, l)
        return nat.arr(p, self.list_stride(l), np.uint8) if p else None

    def component(self, l, c) -> np.ndarray:
        t, q, off = self.list_fmt(l)[c]
        st = storage_type(t, q)
        return self.list_data(l)[:, off:off + TYPE_SIZE[st]].copy().view(TYPE_NP[st]).reshape(-1)

    def set_bounds(self, l: int, mn: bytes, mx: bytes):
        """bounds of list l as records in the original component types (the whole mesh's, for a shard)"""
        nat.check(nat.load().hry_list_set_bounds(self.h, l, bytes(mn), bytes(mx)))

    def bounds_at(self, l: int):
        """after Codec.bounds: per component (1 + index of the first element holding the minimum, same for the maximum); 0 = the
        initial value of the reference's scan"""
        L = nat.load()
        n = L.hry_list_ncomp(self.h, l)
        return [(L.hry_list_min_at(self.h, l, c), L.hry_list_max_at(self.h, l, c)) for c in range(n)]

    partial = property(lambda s: bool(nat.load().hry_mesh_partial(s.h)))

    def runs(self) -> np.ndarray:
        """(n, 6) u32: first_vertex, first_face, first_halfedge, n_vertices, n_faces, n_halfedges in the numbering of the whole mesh.
        A shard: where its components go.  A mesh decoded from a sharded container: what was decoded."""
        p = C.c_void_p()
        n = nat.load().hry_mesh_runs(self.h, C.byref(p))
        if n == 0:
            return np.zeros((0, 6), np.uint32)
        return np.frombuffer(C.string_at(p, n * 24), dtype=np.uint32).reshape(n, 6).copy()

    def shard_elements(self, which: int) -> np.ndarray:
        """a shard: index in the whole mesh of every vertex (which = 1) / face (which = 0)"""
        p = C.c_void_p()
        n = nat.load().hry_shard_elements(self.h, which, C.byref(p))
        return np.frombuffer(C.string_at(p, n * 4), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32)

    def to_ply(self, ascii: bool = False, packed: bool = False) -> bytes:
        """packed: quantised components in the width of the storage type the header declares (see HRY_PLY_PACKED)"""
        p, n = C.c_void_p(), C.c_size_t()
        nat.check(nat.load().hry_mesh_to_ply(self.h, int(ascii) | (2 if packed else 0), C.byref(p), C.byref(n)))
        return nat.take_bytes(p, n.value)

    def to_obj(self) -> bytes:
        p, n = C.c_void_p(), C.c_size_t()
        nat.check(nat.load().hry_mesh_to_obj(self.h, 0, C.byref(p), C.byref(n)))
        return nat.take_bytes(p, n.value)

    # ---- general bindings (structs/attr.h:101-189)
    general = property(lambda s: bool(nat.load().hry_mesh_general(s.h)))
    nlists = property(lambda s: nat.load().hry_mesh_nlists(s.h))

    def list_target(self, l) -> int:
        return nat.load().hry_list_target(self.h, l)

    def nregions(self, which: int) -> int:
        return nat.load().hry_mesh_nregions(self.h, which)

    def region_lists(self, kind: int, r: int):
        out = np.zeros(256, np.uint16)
        n = nat.load().hry_mesh_region_lists(self.h, kind, r, out.ctypes.data, 256)
        return [int(x) for x in out[:n]]

    def regions_of(self, which: int) -> np.ndarray:
        p = C.c_void_p()
        n = nat.load().hry_mesh_regions_of(self.h, which, C.byref(p))
        return np.frombuffer(C.string_at(p, n * 2), dtype=np.uint16).copy() if n else np.zeros(0, np.uint16)

    def bindings(self, kind: int) -> np.ndarray:
        p, slots = C.c_void_p(), C.c_int()
        n = nat.load().hry_mesh_bindings(self.h, kind, C.byref(p), C.byref(slots))
        if not n or not slots.value:
            return np.zeros((n, slots.value), np.uint32)
        return np.frombuffer(C.string_at(p, n * slots.value * 4), dtype=np.uint32).copy().reshape(n, slots.value)

    def host_walk(self, plain: bool = False) -> dict:
        """Host-only cut-border walk with a recording writer (mutates twins like an encode).  plain: without the operation
        model (the chunked profile's walk; may use several host threads for multi-component meshes)."""
        L = nat.load()
        w = C.c_void_p()
        nat.check((L.hry_walk_run_plain if plain else L.hry_walk_run)(self.h, C.byref(w)))
        return _walk_arrays(w)


def _walk_arrays(w) -> dict:
    """every array of a recorded walk (hry_walk_get); frees the walk"""
    L = nat.load()
    try:
        out = {}
        names = [("order_v", np.uint32), ("order_f", np.uint32), ("op_sym", np.uint8), ("op_class", np.uint8), ("op_l", np.uint32),
                 ("op_h", np.uint32), ("op_t", np.uint32), ("op_pos", np.uint32), ("op_thr", np.uint32), ("op_cum", np.uint32), ("info", np.uint32),
                 ("marks", np.uint32), ("snap_section", np.uint8)]
        for g in range(5):
            names += [(f"grp{g}_val", np.uint32), (f"grp{g}_pos", np.uint32)]
        for name, dt in names:
            p = C.c_void_p()
            n = L.hry_walk_get(w, name.encode(), C.byref(p))
            out[name] = (np.frombuffer(C.string_at(p, n * np.dtype(dt).itemsize), dtype=dt).copy() if n else np.zeros(0, dt))
        return out
    finally:
        L.hry_walk_free(w)


class ShardPlan:
    """Distribution of one mesh over n shards by groups of connected components (include/harry_amd.h, hry_shard_plan)."""

    def __init__(self, mesh: Mesh, n_shards: int):
        self.h = C.c_void_p()
        self.n_shards = n_shards
        nat.check(nat.load().hry_shard_plan(mesh.h, n_shards, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            nat.load().hry_plan_free(self.h)
            self.h = None

    ncomponents = property(lambda s: nat.load().hry_plan_ncomponents(s.h))
    ngroups = property(lambda s: nat.load().hry_plan_ngroups(s.h))

    def triangles(self, shard: int) -> int:
        return nat.load().hry_plan_triangles(self.h, shard)

    def extract(self, mesh: Mesh, shard: int) -> Mesh:
        h = C.c_void_p()
        nat.check(nat.load().hry_shard_extract(mesh.h, self.h, shard, C.byref(h)))
        return Mesh(h)

    def walk_in_place(self, mesh: Mesh, shard: int) -> dict:
        """Host-only: the shard's components walked where they lie in `mesh` (hry_walk_run_shard; mutates its twins)."""
        w = C.c_void_p()
        nat.check(nat.load().hry_walk_run_shard(mesh.h, self.h, shard, C.byref(w)))
        return _walk_arrays(w)


def container_info(data: bytes) -> dict:
    """what a .hry file is, without decoding it (hry_container_info)"""
    info = (C.c_uint32 * 8)()
    nat.check(nat.load().hry_container_info(data, len(data), info))
    keys = ("minor", "header_bytes", "nv", "nf", "ne", "chunk_syms", "conn_chunk_syms", "segments")
    return dict(zip(keys, (int(x) for x in info)))


def merge(parts, as_buffer: bool = False):
    """several sharded containers (.hry v0.3) of the same mesh -> one (hry_merge).  The parts may be any buffers (bytes, numpy
    arrays, NativeBuffer): none is copied on the way in; as_buffer: the result stays in the library's buffer too."""
    where = [nat.buffer_address(p) for p in parts]
    arr = (C.c_void_p * len(parts))(*[w[0] for w in where])
    sizes = (C.c_size_t * len(parts))(*[w[1] for w in where])
    p, n = C.c_void_p(), C.c_size_t()
    nat.check(nat.load().hry_merge(arr, sizes, len(parts), C.byref(p), C.byref(n)))
    return nat.take(p, n.value, as_buffer)


def walk_and_replay(mesh: "Mesh", use_restart_points: bool):
    """Host-only: plain cut-border walk of `mesh` (mutates its twins), then the decoder-side replay of the recorded
    connectivity planes.  Returns (mesh with the rebuilt connectivity, order_v, seg_start, seg_level, n_restart_points)."""
    L = nat.load()
    w = C.c_void_p()
    nat.check(L.hry_walk_run_plain(mesh.h, C.byref(w)))
    try:
        mh, r = C.c_void_p(), C.c_void_p()
        nat.check(L.hry_walk_replay(mesh.h, w, int(use_restart_points), C.byref(mh), C.byref(r)))
        try:
            out = []
            for name in ("order_v", "seg_start", "seg_level", "info"):
                p = C.c_void_p()
                n = L.hry_walk_get(r, name.encode(), C.byref(p))
                out.append(np.frombuffer(C.string_at(p, n * 4), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32))
            # use_restart_points: False / True (the restart points at component starts), or 2 / 3: + the border snapshots inside components
            return (Mesh(mh), out[0], out[1], out[2], int(out[3][0]) if int(use_restart_points) < 2 else (int(out[3][0]), int(out[3][1])))
        finally:
            L.hry_walk_free(r)
    finally:
        L.hry_walk_free(w)


def read_stream_host(data: bytes):
    """Host-only serial half of reading a reference (v0.1) stream: (mesh with connectivity, order_v, vplanes, fplanes)."""
    L = nat.load()
    mh, w = C.c_void_p(), C.c_void_p()
    nat.check(L.hry_stream_read_host(data, len(data), C.byref(mh), C.byref(w)))
    try:
        out = []
        for name, dt in (("order_v", np.uint32), ("vplanes", np.uint8), ("fplanes", np.uint8)):
            p = C.c_void_p()
            n = L.hry_walk_get(w, name.encode(), C.byref(p))
            out.append(np.frombuffer(C.string_at(p, n * np.dtype(dt).itemsize), dtype=dt).copy() if n else np.zeros(0, dt))
        return (Mesh(mh), *out)
    finally:
        L.hry_walk_free(w)


class Codec:
    """Device context (one HIP device, one stream).  Raises HryError(E_NODEVICE) without a GPU: no CPU fallback."""

    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        self.device = int(device)
        nat.check(nat.load().hry_ctx_create(device, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            nat.load().hry_ctx_destroy(self.h)
            self.h = None

    def analysis_check(self, mesh: "Mesh") -> None:
        """Development / tests: the component analysis of `mesh` on the device against the host's, table by table (hry_analysis_check)."""
        nat.check(nat.load().hry_analysis_check(self.h, mesh.h))

    __del__ = close

    def bounds(self, mesh: Mesh):
        nat.check(nat.load().hry_bounds(self.h, mesh.h))

    def requant(self, mesh: Mesh, quants, clear: bool = False):
        """quants: iterable of (list, component or -1, bits) as produced by the reference CLI's -l/-a/-q flags."""
        qs = list(quants)
        arr = (nat.Quant * max(len(qs), 1))(*[nat.Quant(int(l), int(c), int(b)) for l, c, b in qs])
        nat.check(nat.load().hry_requant(self.h, mesh.h, arr, len(qs), int(clear)))

    def upload(self, mesh: Mesh):
        nat.check(nat.load().hry_mesh_upload(self.h, mesh.h))

    def write_hry(self, mesh: Mesh, profile: int = PROFILE_COMPAT, chunk_syms: int = 0, keep_stages: bool = False, flags: int = 0, as_buffer: bool = False):
        """as_buffer: return the library's buffer as it is (nat.NativeBuffer: what a C caller of hry_encode holds) instead of bytes"""
        o = nat.Opts(profile, chunk_syms, int(keep_stages), flags, 0, 0)
        p, n = C.c_void_p(), C.c_size_t()
        nat.check(nat.load().hry_encode(self.h, mesh.h, C.byref(o), C.byref(p), C.byref(n)))
        return nat.take(p, n.value, as_buffer)

    def read_hry(self, data: bytes, keep_stages: bool = False, shard=(0, 0), partial: bool = False) -> Mesh:
        """shard = (index, count): of a sharded container decode only the segments i with i % count == index.
        partial: accept a sharded container that does not hold the whole mesh (one rank's own part): the result is a partial mesh."""
        o = nat.Opts(0, 0, int(keep_stages), FLAG_PARTIAL if partial else 0, int(shard[0]), int(shard[1]))
        h = C.c_void_p()
        nat.check(nat.load().hry_decode(self.h, data, len(data), C.byref(o), C.byref(h)))
        return Mesh(h)

    def timing(self) -> dict:
        t = nat.Timing()
        nat.check(nat.load().hry_ctx_timing(self.h, C.byref(t)))
        return t.asdict()

    def stream(self) -> int:
        return nat.load().hry_ctx_stream(self.h) or 0

    def stage(self, name: str, dtype=np.uint8) -> np.ndarray:
        p, n = C.c_void_p(), C.c_size_t()
        nat.check(nat.load().hry_stage_get(self.h, name.encode(), C.byref(p), C.byref(n)))
        return np.frombuffer(nat.take_bytes(p, n.value), dtype=dtype).copy()

    def range_encode_lht(self, lht: np.ndarray) -> bytes:
        lht = np.ascontiguousarray(lht, np.uint64).reshape(-1, 3)
        p, n = C.c_void_p(), C.c_size_t()
        nat.check(nat.load().hry_range_encode_lht(self.h, lht.ctypes.data, len(lht), C.byref(p), C.byref(n)))
        return nat.take_bytes(p, n.value)


class MultiCodec:
    """Several device contexts driven from this one process (include/harry_amd.h: hry_encode_sharded / hry_decode_sharded): the
    reference's single entry with N devices behind it.  devices: one index per context; an index may repeat (contexts that share
    a device run side by side on it)."""

    def __init__(self, devices):
        self.ctx = [Codec(int(d)) for d in devices]
        self.last = {}

    def close(self):
        for c in getattr(self, "ctx", []):
            c.close()
        self.ctx = []

    __del__ = close

    def _handles(self):
        return (C.c_void_p * len(self.ctx))(*[c.h for c in self.ctx])

    def write_hry(self, mesh: Mesh, quants=(), clear: bool = False, n_shards: int = 0, chunk_syms: int = 0, keep_mesh: bool = False, as_buffer: bool = False):
        """plan + extract + bounds of the whole mesh + quantisation + encode of every shard on its context + merge: ONE .hry v0.3.
        keep_mesh: do not store the combined bounds in `mesh`"""
        qs = list(quants)
        arr = (nat.Quant * max(len(qs), 1))(*[nat.Quant(int(l), int(c), int(b)) for l, c, b in qs])
        o = nat.Opts(PROFILE_CHUNKED, chunk_syms, 0, FLAG_KEEP_MESH if keep_mesh else 0, 0, int(n_shards))
        p, n, t = C.c_void_p(), C.c_size_t(), nat.ShardTiming()
        nat.check(nat.load().hry_encode_sharded(self._handles(), len(self.ctx), mesh.h, arr, len(qs), int(clear), C.byref(o), C.byref(p), C.byref(n), C.byref(t)))
        self.last = t.asdict()
        return nat.take(p, n.value, as_buffer)

    def read_hry(self, data: bytes, shard=(0, 0), partial: bool = False) -> Mesh:
        o = nat.Opts(0, 0, 0, FLAG_PARTIAL if partial else 0, int(shard[0]), int(shard[1]))
        h, t = C.c_void_p(), nat.ShardTiming()
        nat.check(nat.load().hry_decode_sharded(self._handles(), len(self.ctx), data, len(data), C.byref(o), C.byref(h), C.byref(t)))
        self.last = t.asdict()
        return Mesh(h)

    def timings(self):
        return [c.timing() for c in self.ctx]


def container_check(data: bytes) -> bool:
    """host-only validation of a sharded container's directory; returns whether its runs cover the whole mesh"""
    c = C.c_int()
    nat.check(nat.load().hry_container_check(data, len(data), C.byref(c)))
    return bool(c.value)


def parse_quant_flags(flags):
    """The reference CLI's -l/-a/-q/-c state machine (main.cc:47-71) -> ([(list, comp, bits)], clear)."""
    cur_l, cur_a, out, clear = None, -1, [], False
    it = iter(flags)
    for f in it:
        if f in ("-l", "--list"):
            cur_l = int(next(it))
        elif f.startswith("-l") and f[2:].lstrip("-").isdigit():
            cur_l = int(f[2:])
        elif f in ("-a", "--attr"):
            cur_a = int(next(it))
        elif f.startswith("-a") and f[2:].lstrip("-").isdigit():
            cur_a = int(f[2:])
        elif f in ("-q", "--quant"):
            out.append((cur_l, cur_a, int(next(it)))); cur_a = -1
        elif f.startswith("-q") and f[2:].lstrip("-").isdigit():
            out.append((cur_l, cur_a, int(f[2:]))); cur_a = -1
        elif f in ("-c", "--clear-quant"):
            clear = True
        else:
            raise ValueError(f"unknown flag {f}")
    for l, _, _ in out:
        if l is None:
            raise ValueError("-q without a preceding -l (the reference reads an uninitialised list index here, main.cc:48)")
    return out, clear
