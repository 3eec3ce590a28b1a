"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (harry_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hry_oracle.cc")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "lib"], stdout=subprocess.DEVNULL)
    return _LIB


class HoSym(C.Structure):
    _fields_ = [("ctx", C.c_uint32), ("sym", C.c_uint32), ("l", C.c_uint64), ("h", C.c_uint64), ("t", C.c_uint64)]


SYM_DTYPE = np.dtype([("ctx", "<u4"), ("sym", "<u4"), ("l", "<u8"), ("h", "<u8"), ("t", "<u8")])

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, u8p, u32p, sz = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_size_t
    L.ho_last_error.restype = C.c_char_p
    L.ho_mesh_from_ply.restype = vp; L.ho_mesh_from_ply.argtypes = [C.c_char_p, sz]
    L.ho_mesh_from_hry.restype = vp; L.ho_mesh_from_hry.argtypes = [C.c_char_p, sz]
    L.ho_mesh_clone.restype = vp; L.ho_mesh_clone.argtypes = [vp]
    L.ho_mesh_from_obj.restype = vp; L.ho_mesh_from_obj.argtypes = [C.c_char_p, sz, C.c_char_p]
    L.ho_mesh_general.restype = C.c_int; L.ho_mesh_general.argtypes = [vp]
    L.ho_mesh_make_general.argtypes = [vp]
    L.ho_mesh_nregions.restype = C.c_int; L.ho_mesh_nregions.argtypes = [vp, C.c_int]
    L.ho_mesh_region_lists.restype = C.c_int; L.ho_mesh_region_lists.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.ho_mesh_regions_of.restype = C.POINTER(C.c_uint16); L.ho_mesh_regions_of.argtypes = [vp, C.c_int]
    L.ho_mesh_nslots.restype = C.c_int; L.ho_mesh_nslots.argtypes = [vp, C.c_int]
    L.ho_mesh_bindings.restype = u32p; L.ho_mesh_bindings.argtypes = [vp, C.c_int]
    L.ho_mesh_free.argtypes = [vp]
    L.ho_mesh_set_shard.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, sz, vp, sz]
    L.ho_mesh_set_bounds.argtypes = [vp, C.c_int, C.c_char_p, C.c_char_p]
    L.ho_mesh_set_degrees.argtypes = [vp, vp, sz]
    L.ho_mesh_degrees.restype = sz; L.ho_mesh_degrees.argtypes = [vp, vp, sz]
    L.ho_requant.restype = C.c_int; L.ho_requant.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.c_int]
    L.ho_encode.restype = vp; L.ho_encode.argtypes = [vp, C.c_int]
    L.ho_result_free.argtypes = [vp]
    L.ho_encode_chunked.restype = vp; L.ho_encode_chunked.argtypes = [vp, C.c_uint32]
    L.ho_encode_chunked2.restype = vp; L.ho_encode_chunked2.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.ho_mesh_from_hry_chunked.restype = vp; L.ho_mesh_from_hry_chunked.argtypes = [C.c_char_p, sz]
    L.ho_result_size.restype = sz; L.ho_result_size.argtypes = [vp]
    L.ho_result_data.restype = u8p; L.ho_result_data.argtypes = [vp]
    L.ho_result_header_size.restype = sz; L.ho_result_header_size.argtypes = [vp]
    L.ho_result_trace_len.restype = sz; L.ho_result_trace_len.argtypes = [vp]
    L.ho_result_trace.restype = C.POINTER(HoSym); L.ho_result_trace.argtypes = [vp]
    L.ho_result_order_vtx.restype = sz; L.ho_result_order_vtx.argtypes = [vp, C.POINTER(u32p)]
    L.ho_result_order_face.restype = sz; L.ho_result_order_face.argtypes = [vp, C.POINTER(u32p)]
    for name in ("nv", "nf", "ne"):
        f = getattr(L, "ho_mesh_" + name); f.restype = C.c_uint32; f.argtypes = [vp]
    L.ho_mesh_ntri.restype = C.c_uint64; L.ho_mesh_ntri.argtypes = [vp]
    for name in ("face_offsets", "org", "twin"):
        f = getattr(L, "ho_mesh_" + name); f.restype = u32p; f.argtypes = [vp]
    L.ho_mesh_nlists.restype = C.c_int; L.ho_mesh_nlists.argtypes = [vp]
    for name in ("ncomp", "target", "stride"):
        f = getattr(L, "ho_list_" + name); f.restype = C.c_int; f.argtypes = [vp, C.c_int]
    L.ho_list_count.restype = C.c_uint32; L.ho_list_count.argtypes = [vp, C.c_int]
    for name in ("type", "quant", "offset"):
        f = getattr(L, "ho_list_" + name); f.restype = C.c_int; f.argtypes = [vp, C.c_int, C.c_int]
    for name in ("data", "min", "max"):
        f = getattr(L, "ho_list_" + name); f.restype = u8p; f.argtypes = [vp, C.c_int]
    L.ho_ctx_count.restype = C.c_int; L.ho_ctx_count.argtypes = [vp]
    L.ho_ctx_attr_base.restype = C.c_int; L.ho_ctx_attr_base.argtypes = [vp, C.c_int]
    u32, i = C.c_uint32, C.c_int
    L.ho_kat_encode_delta_f32.restype = u32; L.ho_kat_encode_delta_f32.argtypes = [u32, u32]
    L.ho_kat_decode_delta_f32.restype = u32; L.ho_kat_decode_delta_f32.argtypes = [u32, u32]
    L.ho_kat_encode_delta_u.restype = u32; L.ho_kat_encode_delta_u.argtypes = [u32, u32, i, i]
    L.ho_kat_decode_delta_u.restype = u32; L.ho_kat_decode_delta_u.argtypes = [u32, u32, i, i]
    L.ho_kat_predict_u.restype = u32; L.ho_kat_predict_u.argtypes = [u32, u32, u32, i, i]
    L.ho_kat_predict_f32.restype = u32; L.ho_kat_predict_f32.argtypes = [u32, u32, u32]
    L.ho_kat_requant_f32.restype = C.c_uint64; L.ho_kat_requant_f32.argtypes = [u32, u32, u32, i]
    L.ho_kat_range_encode_bytes.restype = sz; L.ho_kat_range_encode_bytes.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
    L.ho_kat_range_decode_bytes.restype = sz; L.ho_kat_range_decode_bytes.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
    L.ho_kat_range_encode_lht.restype = sz; L.ho_kat_range_encode_lht.argtypes = [C.POINTER(C.c_uint64), sz, C.c_char_p, sz]
    L.ho_kat_range_encode_bytes32.restype = sz; L.ho_kat_range_encode_bytes32.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
    L.ho_kat_range_decode_bytes32.restype = sz; L.ho_kat_range_decode_bytes32.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
    L.ho_kat_range_encode_lht32.restype = sz; L.ho_kat_range_encode_lht32.argtypes = [C.POINTER(C.c_uint64), sz, C.c_char_p, sz]
    _lib = L
    return L


def _err():
    return RuntimeError(lib().ho_last_error().decode())


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy()


TYPE_NP = {0: "<f4", 1: "<f8", 2: "<u8", 3: "<i8", 4: "<u4", 5: "<i4", 6: "<u2", 7: "<i2", 8: "u1", 9: "i1"}
TYPE_SIZE = {0: 4, 1: 8, 2: 8, 3: 8, 4: 4, 5: 4, 6: 2, 7: 2, 8: 1, 9: 1}


def stype_of(t: int, q: int) -> int:
    if q == 0:
        return t
    return 8 if q <= 8 else 6 if q <= 16 else 4 if q <= 32 else 2


class Mesh:
    def __init__(self, h):
        if not h:
            raise _err()
        self.h = h

    @classmethod
    def from_ply(cls, data: bytes) -> "Mesh":
        return cls(lib().ho_mesh_from_ply(data, len(data)))

    @classmethod
    def from_obj(cls, data: bytes, directory: str = "") -> "Mesh":
        return cls(lib().ho_mesh_from_obj(data, len(data), directory.encode()))

    general = property(lambda s: bool(lib().ho_mesh_general(s.h)))

    def make_general(self):
        lib().ho_mesh_make_general(self.h)

    def nregions(self, which):
        return lib().ho_mesh_nregions(self.h, which)

    def region_lists(self, kind, r):
        """lists bound to region r: kind 0 = face lists, 1 = vertex lists, 2 = corner lists (of face region r)"""
        out = np.zeros(64, np.uint16)
        n = lib().ho_mesh_region_lists(self.h, kind, r, out.ctypes.data, 64)
        return [int(x) for x in out[:n]]

    def regions_of(self, which):
        return _arr(lib().ho_mesh_regions_of(self.h, which), self.nf if which == 0 else self.nv, np.uint16)

    def nslots(self, kind):
        return lib().ho_mesh_nslots(self.h, kind)

    def bindings(self, kind):
        rows, slots = (self.nf, self.nv, self.ne)[kind], self.nslots(kind)
        if not rows or not slots:
            return np.zeros((rows, slots), np.uint32)
        return _arr(lib().ho_mesh_bindings(self.h, kind), rows * slots, np.uint32).reshape(rows, slots)

    def list_target(self, l):
        return lib().ho_list_target(self.h, l)

    @classmethod
    def from_hry(cls, data: bytes) -> "Mesh":
        return cls(lib().ho_mesh_from_hry(data, len(data)))

    @classmethod
    def from_hry_chunked(cls, data: bytes) -> "Mesh":
        return cls(lib().ho_mesh_from_hry_chunked(data, len(data)))

    def encode_chunked(self, chunk_syms: int = 0, snapshot_faces=None) -> "Result":
        """snapshot_faces: faces of a component between two border snapshots of the directory (restart points inside components);
        None = the default rule, 0 = none"""
        r = lib().ho_encode_chunked2(self.h, chunk_syms, 0xffffffff if snapshot_faces is None else int(snapshot_faces))
        if not r:
            raise _err()
        return Result(r)

    def set_shard(self, g_nv, g_nf, g_ne, seeds, runs):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        runs = np.ascontiguousarray(runs, np.uint32).reshape(-1, 6)
        lib().ho_mesh_set_shard(self.h, g_nv, g_nf, g_ne, seeds.ctypes.data, len(seeds), runs.ctypes.data, len(runs))

    def set_bounds(self, l, mn: bytes, mx: bytes):
        lib().ho_mesh_set_bounds(self.h, l, bytes(mn), bytes(mx))

    def degrees(self):
        out = np.zeros(256, np.uint16)
        n = lib().ho_mesh_degrees(self.h, out.ctypes.data, 256)
        return out[:n].copy()

    def set_degrees(self, deg):
        deg = np.ascontiguousarray(deg, np.uint16)
        lib().ho_mesh_set_degrees(self.h, deg.ctypes.data, len(deg))

    def clone(self) -> "Mesh":
        return Mesh(lib().ho_mesh_clone(self.h))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:   # (at interpreter shutdown the module's globals may be gone already)
            lib().ho_mesh_free(self.h)
            self.h = None

    nv = property(lambda s: lib().ho_mesh_nv(s.h))
    nf = property(lambda s: lib().ho_mesh_nf(s.h))
    ne = property(lambda s: lib().ho_mesh_ne(s.h))
    ntri = property(lambda s: lib().ho_mesh_ntri(s.h))
    nlists = property(lambda s: lib().ho_mesh_nlists(s.h))

    def face_offsets(self):
        n = lib().ho_mesh_ne(self.h)  # noqa: F841
        nf = np.ctypeslib.as_array(lib().ho_mesh_face_offsets(self.h), shape=(self._nfaces_built() + 1,)).copy()
        return nf

    def _nfaces_built(self):
        # number of faces actually present in the connectivity (== nf for complete meshes)
        return self.nf

    def org(self):
        return _arr(lib().ho_mesh_org(self.h), self.ne, np.uint32)

    def twin(self):
        return _arr(lib().ho_mesh_twin(self.h), self.ne, np.uint32)

    def list_fmt(self, l):
        L = lib()
        n = L.ho_list_ncomp(self.h, l)
        return [(L.ho_list_type(self.h, l, c), L.ho_list_quant(self.h, l, c), L.ho_list_offset(self.h, l, c)) for c in range(n)]

    def list_stride(self, l):
        return lib().ho_list_stride(self.h, l)

    def list_count(self, l):
        return lib().ho_list_count(self.h, l)

    def list_data(self, l) -> np.ndarray:
        n = self.list_count(l) * self.list_stride(l)
        return _arr(lib().ho_list_data(self.h, l), n, np.uint8).reshape(self.list_count(l), self.list_stride(l)) if n else np.zeros((self.list_count(l), 0), np.uint8)

    def list_min(self, l):
        return _arr(lib().ho_list_min(self.h, l), self.list_stride(l), np.uint8)

    def list_max(self, l):
        return _arr(lib().ho_list_max(self.h, l), self.list_stride(l), np.uint8)

    def component(self, l, c) -> np.ndarray:
        """Values of component c of list l in its STORAGE type (quantised ints live in the low bytes)."""
        t, q, off = self.list_fmt(l)[c]
        st = stype_of(t, q)
        d = self.list_data(l)
        return d[:, off:off + TYPE_SIZE[st]].copy().view(TYPE_NP[st]).reshape(-1)

    def requant(self, triples, clear=False):
        flat = [int(x) for tr in triples for x in tr]
        arr = (C.c_int * max(len(flat), 1))(*flat)
        if lib().ho_requant(self.h, arr, len(triples), int(clear)) != 0:
            raise _err()

    def encode(self, trace=False) -> "Result":
        r = lib().ho_encode(self.h, int(trace))
        if not r:
            raise _err()
        return Result(r)

    def ctx_count(self):
        return lib().ho_ctx_count(self.h)

    def ctx_attr_base(self, l):
        return lib().ho_ctx_attr_base(self.h, l)


class Result:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            lib().ho_result_free(self.h)
            self.h = None

    @property
    def data(self) -> bytes:
        n = lib().ho_result_size(self.h)
        return C.string_at(lib().ho_result_data(self.h), n)

    @property
    def header_size(self) -> int:
        return lib().ho_result_header_size(self.h)

    def trace(self) -> np.ndarray:
        n = lib().ho_result_trace_len(self.h)
        if n == 0:
            return np.zeros(0, SYM_DTYPE)
        buf = C.string_at(lib().ho_result_trace(self.h), n * C.sizeof(HoSym))
        return np.frombuffer(buf, dtype=SYM_DTYPE).copy()

    def order_vtx(self) -> np.ndarray:
        p = C.POINTER(C.c_uint32)()
        n = lib().ho_result_order_vtx(self.h, C.byref(p))
        return _arr(p, n, np.uint32)

    def order_face(self) -> np.ndarray:
        p = C.POINTER(C.c_uint32)()
        n = lib().ho_result_order_face(self.h, C.byref(p))
        return _arr(p, n, np.uint32)


def encode_ply(ply: bytes, quant=(), clear=False, trace=False):
    m = Mesh.from_ply(ply)
    if quant or clear:
        m.requant(quant, clear)
    return m, m.encode(trace)


def range_encode_bytes(data: bytes, bits: int = 64) -> bytes:
    """adaptive 256-ary model + range coder with `bits`-bit registers (64: the reference stream, 32: chunked container)"""
    cap = len(data) * 2 + 64
    buf = C.create_string_buffer(cap)
    f = lib().ho_kat_range_encode_bytes if bits == 64 else lib().ho_kat_range_encode_bytes32
    n = f(data, len(data), buf, cap)
    return buf.raw[:n]


def range_decode_bytes(code: bytes, nsym: int, bits: int = 64) -> bytes:
    buf = C.create_string_buffer(max(nsym, 1))
    f = lib().ho_kat_range_decode_bytes if bits == 64 else lib().ho_kat_range_decode_bytes32
    f(code, len(code), buf, nsym)
    return buf.raw[:nsym]


def range_encode_lht(lht: np.ndarray, bits: int = 64) -> bytes:
    lht = np.ascontiguousarray(lht, dtype=np.uint64).reshape(-1, 3)
    cap = len(lht) * 9 + 64
    buf = C.create_string_buffer(cap)
    f = lib().ho_kat_range_encode_lht if bits == 64 else lib().ho_kat_range_encode_lht32
    n = f(lht.ctypes.data_as(C.POINTER(C.c_uint64)), len(lht), buf, cap)
    return buf.raw[:n]
