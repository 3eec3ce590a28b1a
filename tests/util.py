"""Shared helpers for the test-suite (CPU side)."""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "harry_ref")


def flags_to_quant(flags):
    """Emulate the reference CLI's -l/-a/-q/-c state machine (main.cc:47-71): returns (triples, clear)."""
    cur_l, cur_a, out, clear = None, -1, [], False
    it = iter(flags)
    for f in it:
        if f.startswith("-l") and f != "-l":
            cur_l = int(f[2:])
        elif f == "-l":
            cur_l = int(next(it))
        elif f.startswith("-a") and f != "-a":
            cur_a = int(f[2:])
        elif f == "-a":
            cur_a = int(next(it))
        elif f.startswith("-q") and f != "-q":
            out.append((cur_l, cur_a, int(f[2:])))
            cur_a = -1
        elif f == "-q":
            out.append((cur_l, cur_a, int(next(it))))
            cur_a = -1
        elif f == "-c":
            clear = True
        else:
            raise ValueError(f)
    return out, clear


def parse_ref_decoded_ply(data: bytes, vstride: int, fstride: int):
    """Parse a BINARY PLY written by the reference's ply writer (formats/ply/writer.cc:106-192).

    The writer declares quantised components with their storage type but dumps the whole original-width
    record (writer.cc:72-75), so records are sliced with the strides of the ORIGINAL
    types (supplied by the caller) instead of trusting the header.
    Returns (vertex records u8[nv, vstride], degrees, flat indices, face records u8[nf, fstride]).
    """
    end = data.index(b"end_header\n") + len(b"end_header\n")
    hdr = data[:end].decode().split("\n")
    assert "binary_little_endian" in hdr[1]
    nv = nf = None
    for line in hdr:
        if line.startswith("element vertex"):
            nv = int(line.split()[2])
        if line.startswith("element face"):
            nf = int(line.split()[2])
    body = np.frombuffer(data, dtype=np.uint8, offset=end)
    vrec = body[: nv * vstride].reshape(nv, vstride)
    p = nv * vstride
    degs = np.zeros(nf, np.uint8)
    idx = []
    frec = np.zeros((nf, fstride), np.uint8)
    raw = body[p:].tobytes()
    q = 0
    for f in range(nf):
        d = raw[q]
        degs[f] = d
        q += 1
        idx.append(np.frombuffer(raw, dtype="<u4", count=d, offset=q))
        q += 4 * d
        frec[f] = np.frombuffer(raw, dtype=np.uint8, count=fstride, offset=q)
        q += fstride
    assert q == len(raw), (q, len(raw))
    return vrec, degs, (np.concatenate(idx) if idx else np.zeros(0, "<u4")), frec


def canonical_faces(vrec: np.ndarray, degs: np.ndarray, idx: np.ndarray):
    """Order-independent description of a mesh: sorted list of faces, each a tuple of vertex records rotated
    to start at its smallest record (the codec permutes vertices and faces, SURVEY finding 0-4)."""
    out = []
    off = 0
    keys = [bytes(r) for r in vrec]
    for d in degs:
        d = int(d)
        vs = [keys[int(i)] for i in idx[off:off + d]]
        off += d
        k = min(range(d), key=lambda j: (vs[j], vs[(j + 1) % d]))
        out.append(tuple(vs[k:] + vs[:k]))
    out.sort()
    return out


def oracle_shard(shard, whole_oracle):
    """The CPU oracle's copy of a product shard (harry_amd.codec.Mesh from ShardPlan.extract): same elements in the same
    order (through PLY), the whole mesh's bounds and polygon degrees, the shard's seeds and runs."""
    from oracle import oracle_py as op
    if shard.nf == 0:
        o = whole_oracle.clone()   # formats only; no face is coded
        raise ValueError("empty shard: nothing to restate")
    o = op.Mesh.from_ply(shard.to_ply())
    for l in (0, 1):
        if whole_oracle.list_fmt(l):
            o.set_bounds(l, bytes(whole_oracle.list_min(l)), bytes(whole_oracle.list_max(l)))
    o.set_degrees(whole_oracle.degrees())
    o.set_shard(whole_oracle.nv, whole_oracle.nf, whole_oracle.ne, shard.shard_elements(2), shard.runs())
    return o
