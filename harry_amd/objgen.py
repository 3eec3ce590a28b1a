"""Deterministic synthetic OBJ scenes on top of meshgen's meshes: the input side of SURVEY.md section 8 row f3 (OBJ files, per-corner
texture coordinates / normals shared between corners, material regions).  No mesh files exist offline.

The text is written so that the reference's OBJ grammar (formats/obj/reader.rl:27-80) accepts it: fixed notation only (the
grammar ignores the sign of an exponent), one blank between tokens, every line ends in a line feed.
"""
from __future__ import annotations

import numpy as np

from . import meshgen as mg


def _num(x) -> str:
    s = f"{float(x):.6f}".rstrip("0")
    return s + "0" if s.endswith(".") else s


class Scene:
    """obj: the .obj bytes; files: {name: bytes} of the material libraries it names"""

    def __init__(self, obj: bytes, files: dict, nv: int, nf: int):
        self.obj, self.files, self.nv, self.nf = obj, files, nv, nf


def scene(mesh: mg.Mesh, normals: str | None = None, tex: str | None = None, charts: int = 3, materials: int = 0,
          colors: str | None = None, tex3: bool = False, interleave: bool = False, negative: bool = False, crlf: bool = False,
          chatter: bool = False, seed: int = 3, mtl_name: str = "scene.mtl") -> Scene:
    """normals: None | "smooth" (one per vertex, shared by all its corners) | "flat" (one per face, shared by its corners)
    tex:     None | "atlas" (one per vertex and chart: shared inside a chart, split along chart borders) | "corner" (all private)
    colors:  None | "all" (every vertex x y z r g b) | "some" (every third vertex: two vertex regions)
    interleave: emit each face's new v / vt / vn right before the face instead of block by block
    negative: faces index relatively (negative indices)"""
    rng = np.random.default_rng(seed)
    pos = np.stack([mesh.verts["x"], mesh.verts["y"], mesh.verts["z"]], axis=1).astype(np.float64)
    deg = mesh.degrees.astype(np.int64)
    off = np.concatenate([[0], np.cumsum(deg)])
    nf, nv = mesh.nf, mesh.nv
    eol = "\r\n" if crlf else "\n"
    chart_of = (np.arange(nf) * max(charts, 1)) // max(nf, 1)
    mat_of = (np.arange(nf) * max(materials, 1)) // max(nf, 1) if materials else np.zeros(nf, np.int64)

    vlines = []
    rgb = rng.random((nv, 3))
    for v in range(nv):
        vals = [_num(c) for c in pos[v]]
        if colors == "all" or (colors == "some" and v % 3 == 0):
            vals += [_num(c) for c in rgb[v]]
        vlines.append("v " + " ".join(vals))

    # per-corner indices into the vt / vn tables, created in order of first use
    vt_lines, vn_lines = [], []
    vt_key, vn_key = {}, {}
    ti = np.zeros(off[-1], np.int64)
    ni = np.zeros(off[-1], np.int64)
    ctr = pos.mean(axis=0)
    for f in range(nf):
        idx = mesh.indices[off[f]:off[f + 1]]
        if normals == "flat":
            p = pos[idx]
            n = np.cross(p[1] - p[0], p[2] - p[0])
            n = n / (np.linalg.norm(n) + 1e-30)
        for c, v in enumerate(idx):
            e = off[f] + c
            if tex is not None:
                key = (int(v), int(chart_of[f])) if tex == "atlas" else int(e)
                if key not in vt_key:
                    vt_key[key] = len(vt_lines)
                    u = [pos[v][0] * 0.25 + 0.5 + 0.125 * chart_of[f], pos[v][1] * 0.25 + 0.5]
                    if tex3:
                        u.append(0.5 * chart_of[f])
                    vt_lines.append("vt " + " ".join(_num(x) for x in u))
                ti[e] = vt_key[key]
            if normals is not None:
                key = int(v) if normals == "smooth" else ("f", f)
                if key not in vn_key:
                    vn_key[key] = len(vn_lines)
                    if normals == "smooth":
                        n = pos[v] - ctr
                        n = n / (np.linalg.norm(n) + 1e-30)
                    vn_lines.append("vn " + " ".join(_num(x) for x in n))
                ni[e] = vn_key[key]

    def face_line(f, nv_so_far, nt_so_far, nn_so_far):
        parts = []
        for c in range(int(deg[f])):
            e = off[f] + c
            v = int(mesh.indices[e])
            a = v - nv_so_far if negative else v + 1
            s = str(a)
            if tex is not None or normals is not None:
                t = (int(ti[e]) - nt_so_far if negative else int(ti[e]) + 1) if tex is not None else None
                n = (int(ni[e]) - nn_so_far if negative else int(ni[e]) + 1) if normals is not None else None
                s += "/" + (str(t) if t is not None else "")
                s += "/" + (str(n) if n is not None else "")   # "v/t/": a bare "v/t" also names normal t in the reference's scanner
            parts.append(s)
        return "f " + " ".join(parts)

    out = []
    if chatter:
        out += ["# synthetic scene", "", "o scene"]
    files = {}
    if materials:
        out.append("mtllib " + mtl_name)
        mt = []
        for k in range(materials):
            mt += [f"newmtl mat{k}", f"Kd {_num(rng.random())} {_num(rng.random())} {_num(rng.random())}", ""]
        files[mtl_name] = "\n".join(mt).encode()
    cur_mat = -1
    if not interleave:
        out += vlines + vt_lines + vn_lines
        for f in range(nf):
            if materials and mat_of[f] != cur_mat:
                cur_mat = int(mat_of[f])
                out.append(f"usemtl mat{cur_mat}")
                if chatter:
                    out += [f"g part{cur_mat}", "s 1"]
            out.append(face_line(f, len(vlines), len(vt_lines), len(vn_lines)))
    else:
        ev = et = en = 0   # emitted so far
        for f in range(nf):
            idx = mesh.indices[off[f]:off[f + 1]]
            need_v = int(idx.max()) + 1
            while ev < need_v:
                out.append(vlines[ev]); ev += 1
            if tex is not None:
                need = int(ti[off[f]:off[f + 1]].max()) + 1
                while et < need:
                    out.append(vt_lines[et]); et += 1
            if normals is not None:
                need = int(ni[off[f]:off[f + 1]].max()) + 1
                while en < need:
                    out.append(vn_lines[en]); en += 1
            if materials and mat_of[f] != cur_mat:
                cur_mat = int(mat_of[f])
                out.append(f"usemtl mat{cur_mat}")
            out.append(face_line(f, ev, et, en))
        while ev < nv:   # vertices no face names
            out.append(vlines[ev]); ev += 1
    return Scene((eol.join(out) + eol).encode(), files, nv, nf)
