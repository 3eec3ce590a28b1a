"""ctypes binding of libharry_amd.so (C ABI: include/harry_amd.h).

The library is built in-tree by `make -C harry_amd/csrc` (driven by __graft_entry__.build()).  There is no Python or
CPU implementation behind this module: if the shared library is missing, importing fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRY_LIB") or os.path.join(_HERE, "libharry_amd.so")   # HRY_LIB: a development build (scripts/build_variant.sh)


class HryError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{msg} (code {code})")
        self.code = code
        self.msg = msg


class Quant(C.Structure):
    _fields_ = [("list", C.c_int32), ("comp", C.c_int32), ("bits", C.c_int32)]


class Opts(C.Structure):
    _fields_ = [("profile", C.c_int32), ("chunk_syms", C.c_int32), ("keep_stages", C.c_int32), ("flags", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("host_walk_ms", C.c_double), ("h2d_ms", C.c_double), ("device_ms", C.c_double), ("d2h_ms", C.c_double),
                ("total_ms", C.c_double), ("k_rchain_ms", C.c_double), ("k_model_ms", C.c_double), ("k_predict_ms", C.c_double),
                ("k_entropy_ms", C.c_double), ("k_chain_ms", C.c_double), ("n_symbols", C.c_uint64), ("payload_bytes", C.c_uint64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ShardTiming(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("twins_ms", "plan_ms", "extract_ms", "bounds_ms", "combine_ms", "quant_ms", "encode_ms", "merge_ms", "phase_a_ms",
                                           "phase_b_ms", "host_walk_ms", "total_ms")] + \
               [(k, C.c_uint32) for k in ("n_shards", "n_contexts", "n_segments", "n_components", "n_groups")]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


OK, E_ARG, E_FORMAT, E_UNSUPPORTED, E_NODEVICE, E_NOMEM, E_INTERNAL = 0, -1, -2, -3, -4, -5, -6
PROFILE_COMPAT, PROFILE_CHUNKED = 0, 1
FLAG_HOST_RECURRENCE = 1
FLAG_DEVICE_RECURRENCE = 2
FLAG_PARTIAL = 4
FLAG_KEEP_MESH = 8

_lib = None


def load():
    """Load the shared library; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C harry_amd/csrc` (hipcc, gfx950). "
                          "harry_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, u8p, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    L.hry_last_error.restype = C.c_char_p
    L.hry_abi_version.restype = C.c_int
    L.hry_ctx_create.restype = C.c_int; L.hry_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.hry_ctx_destroy.argtypes = [vp]
    L.hry_ctx_timing.restype = C.c_int; L.hry_ctx_timing.argtypes = [vp, C.POINTER(Timing)]
    L.hry_ctx_stream.restype = vp; L.hry_ctx_stream.argtypes = [vp]
    L.hry_mesh_from_ply.restype = C.c_int; L.hry_mesh_from_ply.argtypes = [C.c_char_p, sz, C.POINTER(vp)]
    L.hry_mesh_from_arrays.restype = C.c_int
    L.hry_mesh_from_arrays.argtypes = [C.c_uint32, vp, C.c_int, vp, C.POINTER(C.c_char_p), C.c_uint32, vp, vp, vp, C.c_int, vp,
                                       C.POINTER(C.c_char_p), C.POINTER(vp)]
    L.hry_mesh_to_ply.restype = C.c_int; L.hry_mesh_to_ply.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz)]
    L.hry_mesh_free.argtypes = [vp]
    L.hry_mesh_clone.restype = vp; L.hry_mesh_clone.argtypes = [vp]
    for n in ("nv", "nf", "ne"):
        f = getattr(L, "hry_mesh_" + n); f.restype = C.c_uint32; f.argtypes = [vp]
    L.hry_mesh_ntri.restype = C.c_uint64; L.hry_mesh_ntri.argtypes = [vp]
    for n in ("face_offsets", "org", "twin"):
        f = getattr(L, "hry_mesh_" + n); f.restype = u32p; f.argtypes = [vp]
    L.hry_mesh_nlists.restype = C.c_int; L.hry_mesh_nlists.argtypes = [vp]
    for n in ("ncomp", "stride"):
        f = getattr(L, "hry_list_" + n); f.restype = C.c_int; f.argtypes = [vp, C.c_int]
    L.hry_list_count.restype = C.c_uint32; L.hry_list_count.argtypes = [vp, C.c_int]
    for n in ("type", "quant", "offset"):
        f = getattr(L, "hry_list_" + n); f.restype = C.c_int; f.argtypes = [vp, C.c_int, C.c_int]
    for n in ("data", "min", "max"):
        f = getattr(L, "hry_list_" + n); f.restype = u8p; f.argtypes = [vp, C.c_int]
    L.hry_bounds.restype = C.c_int; L.hry_bounds.argtypes = [vp, vp]
    L.hry_requant.restype = C.c_int; L.hry_requant.argtypes = [vp, vp, C.POINTER(Quant), sz, C.c_int]
    L.hry_mesh_upload.restype = C.c_int; L.hry_mesh_upload.argtypes = [vp, vp]
    L.hry_encode.restype = C.c_int; L.hry_encode.argtypes = [vp, vp, C.POINTER(Opts), C.POINTER(vp), C.POINTER(sz)]
    L.hry_decode.restype = C.c_int; L.hry_decode.argtypes = [vp, vp, sz, C.POINTER(Opts), C.POINTER(vp)]
    L.hry_free.argtypes = [vp]
    L.hry_container_info.restype = C.c_int; L.hry_container_info.argtypes = [vp, sz, C.POINTER(C.c_uint32)]
    L.hry_stage_get.restype = C.c_int; L.hry_stage_get.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(sz)]
    L.hry_walk_run.restype = C.c_int; L.hry_walk_run.argtypes = [vp, C.POINTER(vp)]
    L.hry_walk_run_plain.restype = C.c_int; L.hry_walk_run_plain.argtypes = [vp, C.POINTER(vp)]
    L.hry_walk_get.restype = sz; L.hry_walk_get.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.hry_walk_free.argtypes = [vp]
    L.hry_stream_read_host.restype = C.c_int; L.hry_stream_read_host.argtypes = [C.c_char_p, sz, C.POINTER(vp), C.POINTER(vp)]
    L.hry_walk_replay.restype = C.c_int; L.hry_walk_replay.argtypes = [vp, vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    L.hry_range_encode_lht.restype = C.c_int; L.hry_range_encode_lht.argtypes = [vp, vp, sz, C.POINTER(vp), C.POINTER(sz)]
    L.hry_shard_plan.restype = C.c_int; L.hry_shard_plan.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.hry_plan_free.argtypes = [vp]
    L.hry_walk_run_shard.restype = C.c_int; L.hry_walk_run_shard.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.hry_analysis_check.restype = C.c_int; L.hry_analysis_check.argtypes = [vp, vp]
    L.hry_plan_ncomponents.restype = C.c_uint32; L.hry_plan_ncomponents.argtypes = [vp]
    L.hry_plan_ngroups.restype = C.c_uint32; L.hry_plan_ngroups.argtypes = [vp]
    L.hry_plan_triangles.restype = C.c_uint64; L.hry_plan_triangles.argtypes = [vp, C.c_int]
    L.hry_shard_extract.restype = C.c_int; L.hry_shard_extract.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.hry_merge.restype = C.c_int; L.hry_merge.argtypes = [C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(vp), C.POINTER(sz)]
    L.hry_mesh_runs.restype = sz; L.hry_mesh_runs.argtypes = [vp, C.POINTER(vp)]
    L.hry_shard_elements.restype = sz; L.hry_shard_elements.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.hry_list_set_bounds.restype = C.c_int; L.hry_list_set_bounds.argtypes = [vp, C.c_int, C.c_char_p, C.c_char_p]
    for n in ("min_at", "max_at"):
        f = getattr(L, "hry_list_" + n); f.restype = C.c_uint32; f.argtypes = [vp, C.c_int, C.c_int]
    L.hry_mesh_from_obj.restype = C.c_int; L.hry_mesh_from_obj.argtypes = [C.c_char_p, sz, C.c_char_p, C.POINTER(vp)]
    L.hry_mesh_to_obj.restype = C.c_int; L.hry_mesh_to_obj.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz)]
    L.hry_mesh_general.restype = C.c_int; L.hry_mesh_general.argtypes = [vp]
    L.hry_list_target.restype = C.c_int; L.hry_list_target.argtypes = [vp, C.c_int]
    L.hry_mesh_nregions.restype = C.c_int; L.hry_mesh_nregions.argtypes = [vp, C.c_int]
    L.hry_mesh_region_lists.restype = C.c_int; L.hry_mesh_region_lists.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.hry_mesh_regions_of.restype = sz; L.hry_mesh_regions_of.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.hry_mesh_bindings.restype = sz; L.hry_mesh_bindings.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int)]
    L.hry_mesh_partial.restype = C.c_int; L.hry_mesh_partial.argtypes = [vp]
    L.hry_encode_sharded.restype = C.c_int
    L.hry_encode_sharded.argtypes = [C.POINTER(vp), C.c_int, vp, C.POINTER(Quant), sz, C.c_int, C.POINTER(Opts), C.POINTER(vp), C.POINTER(sz), C.POINTER(ShardTiming)]
    L.hry_decode_sharded.restype = C.c_int
    L.hry_decode_sharded.argtypes = [C.POINTER(vp), C.c_int, vp, sz, C.POINTER(Opts), C.POINTER(vp), C.POINTER(ShardTiming)]
    L.hry_container_check.restype = C.c_int; L.hry_container_check.argtypes = [vp, sz, C.POINTER(C.c_int)]
    if L.hry_abi_version() != 6:
        raise ImportError(f"{LIB_PATH} has ABI version {L.hry_abi_version()}, this binding expects 6: rebuild it")
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise HryError(rc, load().hry_last_error().decode(errors="replace"))


def take_bytes(ptr, n) -> bytes:
    """Copy a library-allocated buffer and release it."""
    try:
        return C.string_at(ptr, n)
    finally:
        load().hry_free(ptr)


class NativeBuffer:
    """A buffer the library allocated (a container out of hry_encode / hry_merge), kept as it is: what a C or C++ caller of the
    boundary holds.  `bytes` of a 39 MB container cost this binding 12 ms of a 61 ms encode (fresh pages + the copy).  Goes back into
    the boundary's calls directly (`_as_parameter_`), has a length, compares with bytes, and gives a zero-copy `view()`; freed with
    the object."""

    def __init__(self, ptr, n: int):
        self.ptr = ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)
        self.n = int(n)
        self._as_parameter_ = C.c_void_p(self.ptr)

    def __len__(self):
        return self.n

    def view(self) -> memoryview:
        """zero-copy view; it keeps this object (and so the library's buffer) alive for as long as it exists"""
        if not self.n:
            return memoryview(b"")
        arr = (C.c_ubyte * self.n).from_address(self.ptr)
        arr._owner = self          # the view holds the array, the array holds the owner of the pointer
        return memoryview(arr).cast("B")

    def tobytes(self) -> bytes:
        return C.string_at(self.ptr, self.n)

    def __bytes__(self):
        return self.tobytes()

    def __eq__(self, other):
        if isinstance(other, NativeBuffer):
            other = other.view()
        try:
            return self.view() == memoryview(other)
        except TypeError:
            return NotImplemented

    def __del__(self):
        if getattr(self, "ptr", 0) and _lib is not None:
            _lib.hry_free(self.ptr)
            self.ptr = 0


def take(ptr, n, as_buffer: bool):
    return NativeBuffer(ptr, n) if as_buffer else take_bytes(ptr, n)


def buffer_address(obj):
    """(address, length, keep-alive) of bytes / bytearray / numpy array / memoryview / NativeBuffer, without a copy"""
    if isinstance(obj, NativeBuffer):
        return obj.ptr, obj.n, obj
    a = np.frombuffer(obj, dtype=np.uint8)
    return a.ctypes.data, a.size, (a, obj)


def arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy()
