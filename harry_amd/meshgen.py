"""Deterministic synthetic mesh generators + a minimal PLY writer.

These are the workloads named in BASELINE.md section 3 / SURVEY.md section 8(d): no mesh files exist
offline, so every configuration is a seeded synthetic stand-in produced here.  Everything is numpy,
vectorised so that the 1 M / 28 M / 100 M triangle configurations are practical.

A mesh is returned as a dict:
    verts : structured ndarray, one field per PLY vertex property (file order)
    faces : list of (deg, ndarray[n, deg] of uint32)  -- polygons grouped by degree, emitted in list order,
            OR a tuple (degrees u8[F], flat indices u32[sum deg]) under key "poly" for interleaved degrees
    face_props : optional structured ndarray with one row per face
"""
from __future__ import annotations

import io
import numpy as np

_PLY_T = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "u2": "ushort", "i2": "short",
          "u4": "uint", "i4": "int"}


def _vtx_struct(cols: dict) -> np.ndarray:
    names = list(cols)
    n = len(cols[names[0]])
    dt = np.dtype([(k, cols[k].dtype.newbyteorder("<")) for k in names])
    out = np.empty(n, dtype=dt)
    for k in names:
        out[k] = cols[k]
    return out


class Mesh:
    def __init__(self, verts: np.ndarray, degrees: np.ndarray, indices: np.ndarray, face_props=None):
        self.verts = verts
        self.degrees = np.ascontiguousarray(degrees, dtype=np.uint8)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.face_props = face_props
        assert int(self.degrees.astype(np.int64).sum()) == self.indices.size

    @property
    def nv(self):
        return len(self.verts)

    @property
    def nf(self):
        return len(self.degrees)

    @property
    def ntri(self):
        """Triangle count as the reference defines it: sum(ne - 2) (structs/conn.h:87)."""
        return int(self.degrees.astype(np.int64).sum()) - 2 * self.nf

    # ---------------------------------------------------------------- PLY output
    def to_ply(self, fmt: str = "binary_little_endian") -> bytes:
        """fmt in {"ascii", "binary_little_endian", "binary_big_endian"}."""
        hdr = ["ply", f"format {fmt} 1.0", "comment harry_amd synthetic", f"element vertex {self.nv}"]
        for name in self.verts.dtype.names:
            hdr.append(f"property {_PLY_T[self.verts.dtype[name].str[1:]]} {name}")
        hdr.append(f"element face {self.nf}")
        hdr.append("property list uchar int vertex_indices")
        if self.face_props is not None:
            for name in self.face_props.dtype.names:
                hdr.append(f"property {_PLY_T[self.face_props.dtype[name].str[1:]]} {name}")
        hdr.append("end_header")
        out = io.BytesIO()
        out.write(("\n".join(hdr) + "\n").encode())
        if fmt == "ascii":
            lines = []
            for row in self.verts:
                lines.append(" ".join(repr(float(x)) if isinstance(x, (np.floating, float)) else str(int(x))
                                      for x in row.tolist()))
            off = 0
            for f in range(self.nf):
                d = int(self.degrees[f])
                s = f"{d} " + " ".join(str(int(i)) for i in self.indices[off:off + d])
                off += d
                if self.face_props is not None:
                    s += " " + " ".join(repr(float(x)) if isinstance(x, float) else str(int(x))
                                        for x in self.face_props[f].tolist())
                lines.append(s)
            out.write(("\n".join(lines) + "\n").encode())
            return out.getvalue()
        big = fmt == "binary_big_endian"
        v = self.verts
        if big:
            v = v.astype(v.dtype.newbyteorder(">"))
        out.write(v.tobytes())
        out.write(self._face_bytes(big))
        return out.getvalue()

    def _face_bytes(self, big: bool) -> bytes:
        idt = ">i4" if big else "<i4"
        fp = self.face_props
        uniform = self.nf > 0 and bool((self.degrees == self.degrees[0]).all())
        if uniform:
            d = int(self.degrees[0])
            fields = [("n", "u1"), ("i", idt, (d,))]
            if fp is not None:
                for name in fp.dtype.names:
                    fields.append((name, fp.dtype[name].newbyteorder(">" if big else "<")))
            rec = np.empty(self.nf, dtype=np.dtype(fields))
            rec["n"] = d
            rec["i"] = self.indices.reshape(self.nf, d)
            if fp is not None:
                for name in fp.dtype.names:
                    rec[name] = fp[name]
            return rec.tobytes()
        # ragged: assemble byte buffer with offsets
        deg = self.degrees.astype(np.int64)
        fpb = 0 if fp is None else fp.dtype.itemsize
        rec_len = 1 + 4 * deg + fpb
        starts = np.concatenate(([0], np.cumsum(rec_len)))
        buf = np.zeros(int(starts[-1]), dtype=np.uint8)
        buf[starts[:-1]] = self.degrees
        idx_bytes = self.indices.astype(idt).view(np.uint8).reshape(-1, 4)
        foff = np.concatenate(([0], np.cumsum(deg)))[:-1]
        # position of each index' first byte
        within = np.arange(self.indices.size, dtype=np.int64) - np.repeat(foff, deg)
        pos = np.repeat(starts[:-1] + 1, deg) + 4 * within
        for b in range(4):
            buf[pos + b] = idx_bytes[:, b]
        if fp is not None:
            f2 = fp.astype(fp.dtype.newbyteorder(">")) if big else fp
            fb = f2.view(np.uint8).reshape(self.nf, fpb)
            p0 = starts[:-1] + 1 + 4 * deg
            for b in range(fpb):
                buf[p0 + b] = fb[:, b]
        return buf.tobytes()


# -------------------------------------------------------------------- helpers
def _tri_mesh(cols, tris, face_props=None):
    tris = np.ascontiguousarray(tris, dtype=np.uint32)
    return Mesh(_vtx_struct(cols), np.full(len(tris), 3, np.uint8), tris.reshape(-1), face_props)


def _noise(rng, n, sigma):
    return rng.normal(0.0, sigma, n).astype(np.float32) if sigma > 0 else np.zeros(n, np.float32)


def grid(n: int, m: int | None = None, seed: int = 1, sigma: float = 1e-4, quads: bool = False) -> Mesh:
    """Open n x m height field  z = 0.1 sin(6x) cos(5y) + N(0, sigma)  (the survey's probe mesh, App. D)."""
    m = m or n
    rng = np.random.default_rng(seed)
    j, i = np.meshgrid(np.arange(m), np.arange(n))
    x = (i / max(n - 1, 1)).astype(np.float32).ravel()
    y = (j / max(m - 1, 1)).astype(np.float32).ravel()
    z = (0.1 * np.sin(6 * x) * np.cos(5 * y)).astype(np.float32) + _noise(rng, n * m, sigma)
    a = (np.arange(n - 1)[:, None] * m + np.arange(m - 1)[None, :]).ravel().astype(np.uint32)
    b, c, d = a + 1, a + m + 1, a + m
    cols = {"x": x, "y": y, "z": z}
    if quads:
        q = np.stack([a, b, c, d], 1)
        return Mesh(_vtx_struct(cols), np.full(len(q), 4, np.uint8), q.reshape(-1))
    tris = np.stack([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 1).reshape(-1, 3)
    return _tri_mesh(cols, tris)


def torus(nu: int, nv: int | None = None, seed: int = 2, sigma: float = 1e-4, normals: bool = False,
          R: float = 1.0, r: float = 0.35, polys: str = "tri", center=(0.0, 0.0, 0.0)) -> Mesh:
    """Closed torus grid nu x nv; polys in {"tri", "quad", "mixed"} (mixed: quads + triangles + pentagons)."""
    nv = nv or nu
    rng = np.random.default_rng(seed)
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    u = (2 * np.pi * iu / nu).ravel()
    v = (2 * np.pi * iv / nv).ravel()
    cx, cy, cz = center
    nx_, ny_, nz_ = np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)
    x = ((R + r * np.cos(v)) * np.cos(u) + cx).astype(np.float32) + _noise(rng, nu * nv, sigma)
    y = ((R + r * np.cos(v)) * np.sin(u) + cy).astype(np.float32) + _noise(rng, nu * nv, sigma)
    z = (r * np.sin(v) + cz).astype(np.float32) + _noise(rng, nu * nv, sigma)
    cols = {"x": x, "y": y, "z": z}
    if normals:
        cols.update({"nx": nx_.astype(np.float32), "ny": ny_.astype(np.float32), "nz": nz_.astype(np.float32)})
    a = (iu * nv + iv).ravel().astype(np.uint32)
    b = (iu * nv + (iv + 1) % nv).ravel().astype(np.uint32)
    c = (((iu + 1) % nu) * nv + (iv + 1) % nv).ravel().astype(np.uint32)
    d = (((iu + 1) % nu) * nv + iv).ravel().astype(np.uint32)
    if polys == "tri":
        tris = np.stack([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 1).reshape(-1, 3)
        return _tri_mesh(cols, tris)
    if polys == "quad":
        q = np.stack([a, b, c, d], 1)
        return Mesh(_vtx_struct(cols), np.full(len(q), 4, np.uint8), q.reshape(-1))
    # mixed: per cell choose quad (40 %), two triangles (55 %), or pentagon+triangle pair (5 %, needs cell pairs)
    ncell = len(a)
    kind = rng.random(ncell)
    degs, idx = [], []
    # process vectorised by category, but keep cell order for determinism of the face order
    cat = np.where(kind < 0.40, 0, 1).astype(np.int8)
    # pentagons: merge triangle (a,c,d) of cell k with ... keep simple: split quad a,b,c,d into pentagon by
    # inserting no new vertices is impossible; instead emit quad cells whose neighbour in v is also quad as
    # pentagon+triangle over the 2-cell strip (a,b,b2,c2,c? ) -- done cellwise below for 5 % of even cells.
    pent = (kind > 0.95) & (iv.ravel() % 2 == 0) & (nv % 2 == 0)
    nxt = (iu * nv + (iv + 1) % nv).ravel()  # cell index of the neighbour in +v
    skip = np.zeros(ncell, bool)
    skip[nxt[pent]] = True
    pent &= ~skip
    skip[:] = False
    skip[nxt[pent]] = True
    b2 = (iu * nv + (iv + 2) % nv).ravel().astype(np.uint32)
    c2 = (((iu + 1) % nu) * nv + (iv + 2) % nv).ravel().astype(np.uint32)
    deg_cell = np.where(skip, 0, np.where(pent, 8, np.where(cat == 0, 4, 6)))
    starts = np.concatenate(([0], np.cumsum(deg_cell)))
    flat = np.zeros(int(starts[-1]), np.uint32)
    nfc = np.where(skip, 0, np.where(pent | (cat == 1), 2, 1))
    fstarts = np.concatenate(([0], np.cumsum(nfc)))
    degrees = np.zeros(int(fstarts[-1]), np.uint8)
    s = starts[:-1]
    fs = fstarts[:-1]
    mq = (~skip) & (~pent) & (cat == 0)
    for k, arr in enumerate((a, b, c, d)):
        flat[s[mq] + k] = arr[mq]
    degrees[fs[mq]] = 4
    mt = (~skip) & (~pent) & (cat == 1)
    for k, arr in enumerate((a, b, c, a, c, d)):
        flat[s[mt] + k] = arr[mt]
    degrees[fs[mt]] = 3
    degrees[fs[mt] + 1] = 3
    # pentagon (a, b, b2, c2, c) + triangle (a, c, d) covering the 2-cell strip [v, v+2)
    for k, arr in enumerate((a, b, b2, c2, c, a, c, d)):
        flat[s[pent] + k] = arr[pent]
    degrees[fs[pent]] = 5
    degrees[fs[pent] + 1] = 3
    return Mesh(_vtx_struct(cols), degrees, flat)


def icosphere(level: int, seed: int = 1, sigma: float = 1e-3, extra_props: bool = True) -> Mesh:
    """Icosphere (20 * 4^level triangles) with radial noise; optional Stanford-style confidence/intensity."""
    t = (1.0 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.sort(e, 1)
        uniq, inv = np.unique(key[:, 0] * (len(v) + 1) + key[:, 1], return_inverse=True)
        first = np.zeros(len(uniq), np.int64)
        first[inv[::-1]] = np.arange(len(e))[::-1]
        mid = v[key[first, 0]] + v[key[first, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        nf = len(f)
        m01, m12, m20 = base + inv[:nf], base + inv[nf:2 * nf], base + inv[2 * nf:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)])
    rng = np.random.default_rng(seed)
    rad = 1.0 + rng.normal(0, sigma, len(v))
    p = (v * rad[:, None]).astype(np.float32)
    cols = {"x": p[:, 0].copy(), "y": p[:, 1].copy(), "z": p[:, 2].copy()}
    if extra_props:
        cols["confidence"] = rng.random(len(v)).astype(np.float32)
        cols["intensity"] = (0.5 + 0.5 * np.sin(5 * v[:, 0])).astype(np.float32)
    return _tri_mesh(cols, f)


def soup(nv: int = 40, nf: int = 120, seed: int = 5) -> Mesh:
    """Random triangles over a few vertices: edges shared by any number of faces, in either direction.  The half-edge matching
    pairs them first come, first served (structs/conn.h:201-214) and the cut-border walk re-pairs some of them as it meets the
    faces (cbm/encoder.h:150,193-198): the meshes whose twins an encode repairs (seeds 5, 6, 13, 14, 16, 22 at the default size)."""
    rng = np.random.default_rng(seed)
    tris = set()
    while len(tris) < nf:
        a, b, c = rng.choice(nv, 3, replace=False)
        tris.add((int(a), int(b), int(c)))
    idx = np.array(sorted(tris), np.uint32).reshape(-1)
    v = np.zeros(nv, torus(4, 4).verts.dtype)
    for k in "xyz":
        v[k] = rng.normal(size=nv).astype(np.float32)
    return Mesh(v, np.full(nf, 3, np.uint8), idx)


def concat(meshes) -> Mesh:
    """Disjoint union (multi-component mesh); vertex layouts must agree."""
    verts = np.concatenate([m.verts for m in meshes])
    offs = np.cumsum([0] + [m.nv for m in meshes])[:-1]
    idx = np.concatenate([m.indices.astype(np.int64) + o for m, o in zip(meshes, offs)]).astype(np.uint32)
    deg = np.concatenate([m.degrees for m in meshes])
    return Mesh(verts, deg, idx)


def multi_component(ncomp: int, nu: int, nv: int, seed: int = 4, polys: str = "mixed") -> Mesh:
    parts = []
    for c in range(ncomp):
        parts.append(torus(nu, nv, seed=seed + c, polys=polys, center=(3.0 * (c % 32), 3.0 * (c // 32), 0.0)))
    return concat(parts)


def with_nonmanifold(m: Mesh, n_edges: int = 2, n_vtx: int = 1, seed: int = 7) -> Mesh:
    """Attach a third face to n_edges existing edges (non-manifold edges) and glue n_vtx extra cones at an
    existing vertex (non-manifold vertices).  New vertices are appended."""
    rng = np.random.default_rng(seed)
    verts = m.verts
    deg = m.degrees.astype(np.int64)
    offs = np.concatenate(([0], np.cumsum(deg)))
    new_v = []
    new_deg, new_idx = [], []
    names = verts.dtype.names
    nvtx = m.nv
    for _ in range(n_edges):
        f = int(rng.integers(0, m.nf))
        a, b = int(m.indices[offs[f]]), int(m.indices[offs[f] + 1])
        row = np.zeros(1, verts.dtype)
        for k in names:
            row[k] = (verts[k][a] + verts[k][b]) / 2
        row["z"] = row["z"] + np.float32(0.05)
        new_v.append(row)
        new_deg.append(3)
        new_idx += [a, b, nvtx]
        nvtx += 1
    for _ in range(n_vtx):
        a = int(rng.integers(0, m.nv))
        rows = np.zeros(2, verts.dtype)
        for k in names:
            rows[k] = verts[k][a]
        rows["x"] = rows["x"] + np.float32([0.03, 0.0])
        rows["y"] = rows["y"] + np.float32([0.0, 0.03])
        rows["z"] = rows["z"] + np.float32(0.07)
        new_v.append(rows)
        new_deg.append(3)
        new_idx += [a, nvtx, nvtx + 1]
        nvtx += 2
    if not new_v:
        return m
    return Mesh(np.concatenate([verts] + new_v), np.concatenate([m.degrees, np.array(new_deg, np.uint8)]),
                np.concatenate([m.indices, np.array(new_idx, np.uint32)]), None)


def with_colors(m: Mesh, seed: int = 9) -> Mesh:
    rng = np.random.default_rng(seed)
    cols = {k: m.verts[k] for k in m.verts.dtype.names}
    base = (127 + 120 * np.sin(4 * m.verts["x"].astype(np.float64))).astype(np.int64)
    for i, k in enumerate(("red", "green", "blue")):
        cols[k] = np.clip(base + rng.integers(-3, 4, m.nv) + 10 * i, 0, 255).astype(np.uint8)
    return Mesh(_vtx_struct(cols), m.degrees, m.indices, m.face_props)


def with_face_props(m: Mesh, seed: int = 11) -> Mesh:
    rng = np.random.default_rng(seed)
    fp = np.empty(m.nf, dtype=np.dtype([("red", "u1"), ("quality", "<f4")]))
    fp["red"] = rng.integers(0, 256, m.nf)
    fp["quality"] = rng.random(m.nf).astype(np.float32)
    return Mesh(m.verts, m.degrees, m.indices, fp)


def negated(m: Mesh) -> Mesh:
    """All coordinates strictly negative (exercises SURVEY App. B-2: max initialised with FLT_MIN)."""
    v = m.verts.copy()
    for k in ("x", "y", "z"):
        v[k] = -np.abs(v[k]) - np.float32(0.5)
    return Mesh(v, m.degrees, m.indices, m.face_props)


def doubles(m: Mesh) -> Mesh:
    """the same mesh with `double` vertex coordinates (8-byte sources of the quantiser, structs/quant.h:137-139)"""
    dt = np.dtype([(k, "<f8" if m.verts.dtype[k].kind == "f" else m.verts.dtype[k]) for k in m.verts.dtype.names])
    v = np.empty(m.nv, dt)
    for k in m.verts.dtype.names:
        v[k] = m.verts[k].astype(np.float64) * (1.0 + 1e-9) if m.verts.dtype[k].kind == "f" else m.verts[k]
    return Mesh(v, m.degrees, m.indices, m.face_props)


# the named configurations of BASELINE.json / SURVEY.md 8(d)
def cfg1_bunny_class() -> Mesh:
    return icosphere(6, seed=1, sigma=1e-3, extra_props=True)          # 81 920 tris


def cfg2_torus_1m() -> Mesh:
    return torus(708, 708, seed=2, sigma=1e-4)                          # 1 002 528 tris


def cfg3_lucy_class() -> Mesh:
    return torus(3742, 3742, seed=3, sigma=1e-4, normals=True)          # 28 005 128 tris


def cfg4_components(ncomp: int = 1024, nu: int = 221, nv: int = 222) -> Mesh:
    return multi_component(ncomp, nu, nv, seed=4, polys="mixed")        # ~1e8 tris at defaults
