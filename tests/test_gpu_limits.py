"""The limits of the reference stream's profile (.hry v0.1: ONE symbol sequence with 32-bit places and bit positions) and of the
general-bindings path, each refused with its reason AT its boundary -- one symbol / one bit below it the encode goes through.
The counts of a small mesh are topped up through HRY_TEST_EXTRA_SYMBOLS / HRY_TEST_EXTRA_BITS (a mesh that reaches 2^31 symbols by
itself has some 180 M triangles); structs/types.h:14-21 of the reference allows 2^32 - 1 elements, the chunked container has no such
limit."""
import ctypes as C

import numpy as np
import pytest

from harry_amd import _native as nat
from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import objgen as og

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cx():
    c = hc.Codec(0)
    yield c
    c.close()


def test_reference_stream_refuses_2_to_the_31_symbols_at_the_boundary(cx, monkeypatch):
    gen = mg.torus(30, 32, seed=3)
    a = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    want = cx.write_hry(a.clone())
    ns = cx.timing()["n_symbols"]
    assert 0 < ns < 1 << 20
    monkeypatch.setenv("HRY_TEST_EXTRA_SYMBOLS", str((1 << 31) - ns))
    with pytest.raises(hc.HryError, match="2\\^31 symbols"):
        cx.write_hry(a.clone())
    monkeypatch.setenv("HRY_TEST_EXTRA_SYMBOLS", str((1 << 31) - ns - 1))
    assert cx.write_hry(a.clone()) == want
    # the parallel container of the same mesh knows no such limit
    monkeypatch.setenv("HRY_TEST_EXTRA_SYMBOLS", str(1 << 32))
    assert len(cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)) > 0


def test_reference_stream_refuses_2_to_the_32_bits_at_the_boundary(cx, monkeypatch):
    gen = mg.torus(30, 32, seed=3)
    a = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    want = cx.write_hry(a.clone())
    nbytes = len(want) - hc.container_info(want)["header_bytes"]      # the coder's bytes: ceil(bits / 8)
    monkeypatch.setenv("HRY_TEST_EXTRA_BITS", str((1 << 32) - 8 * nbytes + 8))
    with pytest.raises(hc.HryError, match="2\\^32 bits"):
        cx.write_hry(a.clone())
    monkeypatch.setenv("HRY_TEST_EXTRA_BITS", str((1 << 32) - 8 * nbytes - 1))
    assert cx.write_hry(a.clone()) == want


def test_range_coder_entry_refuses_2_to_the_31_triples(cx):
    """hry_range_encode_lht looks at the count before it looks at the triples"""
    one = np.array([[0, 1, 2]], np.uint64)
    p, n = C.c_void_p(), C.c_size_t()
    rc = nat.load().hry_range_encode_lht(cx.h, one.ctypes.data, 1 << 31, C.byref(p), C.byref(n))
    assert rc != 0 and "too many symbols" in nat.load().hry_last_error().decode()
    assert len(cx.range_encode_lht(one)) > 0


def test_general_bindings_reference_stream_refuses_2_to_the_31_symbols(cx, monkeypatch):
    sc = og.scene(mg.torus(12, 14, polys="mixed"), normals="smooth", tex="atlas", charts=3)
    g = hc.Mesh.from_obj(sc.obj, "")
    want = cx.write_hry(g.clone())
    monkeypatch.setenv("HRY_TEST_EXTRA_SYMBOLS", str(1 << 31))
    with pytest.raises(hc.HryError, match="2\\^31 symbols"):
        cx.write_hry(g.clone())
    monkeypatch.delenv("HRY_TEST_EXTRA_SYMBOLS")
    assert cx.write_hry(g.clone()) == want


def test_general_lists_with_8_byte_storage_are_refused(cx):
    """A header that announces general bindings (an OBJ scene's) with a component quantised to more than 32 bits -- 8 bytes of
    storage (structs/mixing.h:101-108) -- is outside the supported subset: refused by the decoders, both formats, with the reason."""
    sc = og.scene(mg.torus(10, 12), normals="smooth", tex="atlas", charts=2)
    g = hc.Mesh.from_obj(sc.obj, "")
    for profile in (hc.PROFILE_COMPAT, hc.PROFILE_CHUNKED):
        data = bytearray(cx.write_hry(g.clone(), profile=profile))
        hdr = hc.container_info(bytes(data))["header_bytes"]
        # the first list's first component: u8 type, u8 quantisation bits -- behind magic (6), three sizes (12), the regions' tables
        fmt = g.list_fmt(0)
        at = None
        for i in range(18, hdr - 1):          # (type, 0) pairs of a lossless float list: find the run of them that is as long as the list
            if all(data[i + 2 * k] == fmt[k][0] and data[i + 2 * k + 1] == 0 for k in range(len(fmt))) and data[i - 2] == len(fmt) and data[i - 1] == 0:
                at = i
                break
        assert at is not None
        data[at + 1] = 40                      # 40 bits: an 8-byte quantised type
        with pytest.raises(hc.HryError, match="8-byte|outside the supported subset|unsupported"):
            cx.read_hry(bytes(data))
