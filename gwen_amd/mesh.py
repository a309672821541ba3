"""Synthetic inputs: Class-I geodesic icosahedral meshes (BASELINE.json configs).

Build-authored: the reference has no mesh (its graph is the complete graph over ensemble members,
/root/reference/src/gwen/utils.py:175-176); the meshes here are the synthetic inputs BASELINE.json
names.  ``edge_index`` follows the conventions of the reference's graph producer
(``erdos_renyi_graph`` output): int64 ``[2, E]``, both directions present, sorted by (row, col),
no self-loops, row 0 = source, row 1 = target.

    frequency nu:  N = 10 nu^2 + 2 nodes,  E = 60 nu^2 directed edges,  20 nu^2 triangles
    nu = 10  -> 1 002 / 6 000   (config c1)        nu = 100 -> 100 002 / 600 000 (configs c2..c5)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


def _icosahedron():
    p = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array(
        [[-1, p, 0], [1, p, 0], [-1, -p, 0], [1, -p, 0],
         [0, -1, p], [0, 1, p], [0, -1, -p], [0, 1, -p],
         [p, 0, -1], [p, 0, 1], [-p, 0, -1], [-p, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
         [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
         [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    return v, f


@dataclass
class Mesh:
    pos: np.ndarray          # [N, 3] float64, unit sphere
    edge_index: np.ndarray   # [2, E] int64, both directions, sorted by (row, col)
    faces: np.ndarray        # [20 nu^2, 3] int64
    nu: int
    perm: Optional[np.ndarray] = None   # new_id -> generator id, when reordered

    @property
    def num_nodes(self) -> int:
        return int(self.pos.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])


def _morton3(pos: np.ndarray, bits: int = 10) -> np.ndarray:
    q = np.clip(((pos + 1.0) * 0.5 * ((1 << bits) - 1)).round().astype(np.uint64), 0, (1 << bits) - 1)
    code = np.zeros(pos.shape[0], dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code


def _hilbert2(order: int, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Index of lattice point (x, y) on the 2-D Hilbert curve of the given order (vectorised xy -> d)."""
    x = x.astype(np.int64).copy()
    y = y.astype(np.int64).copy()
    n = 1 << order
    d = np.zeros_like(x)
    s = n >> 1
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flat = ry == 0
        flip = flat & (rx == 1)
        x[flip] = n - 1 - x[flip]
        y[flip] = n - 1 - y[flip]
        xt = x[flat].copy()
        x[flat] = y[flat]
        y[flat] = xt
        s >>= 1
    return d


def _cube_hilbert(pos: np.ndarray, order: int = 12) -> np.ndarray:
    """Sort key: (cube face, Hilbert index of the equi-angular cube-sphere coordinates on that face).
    Any run of consecutive nodes is a compact patch of the sphere -- 64 consecutive nodes of the nu = 100
    mesh name at most 121 distinct neighbours (3-D Morton: up to 167), which is what K8's tiles want."""
    n = pos.shape[0]
    ax = np.argmax(np.abs(pos), axis=1)
    rows = np.arange(n)
    major = pos[rows, ax]
    face = ax * 2 + (major < 0)
    others = np.array([[1, 2], [0, 2], [0, 1]])
    u = np.arctan(pos[rows, others[ax, 0]] / np.abs(major)) * (4.0 / np.pi)
    v = np.arctan(pos[rows, others[ax, 1]] / np.abs(major)) * (4.0 / np.pi)
    top = (1 << order) - 1
    xi = np.clip(((u + 1.0) * 0.5 * top).round(), 0, top).astype(np.int64)
    yi = np.clip(((v + 1.0) * 0.5 * top).round(), 0, top).astype(np.int64)
    return face.astype(np.int64) * (1 << (2 * order)) + _hilbert2(order, xi, yi)


def geodesic_mesh(nu: int, reorder: Optional[str] = None) -> Mesh:
    """Subdivide each icosahedron face into nu^2 triangles on an integer barycentric lattice.

    Shared corner/edge vertices are merged by integer lattice keys (never by float compare).
    ``reorder``: None (generator order: corners, edge vertices, face interiors), "morton" (3-D Morton
    order of the projected points) or "hilbert" (Hilbert curve on the faces of the cube-sphere: compact
    patches, no long jumps); both are locality-preserving relabellings, the permutation is kept in
    ``Mesh.perm``.
    """
    if nu < 1:
        raise ValueError("nu must be >= 1")
    v0, f0 = _icosahedron()
    # edge table of the base solid
    ekeys = {}
    for a, b, c in f0:
        for p, q in ((a, b), (b, c), (c, a)):
            k = (min(p, q), max(p, q))
            if k not in ekeys:
                ekeys[k] = len(ekeys)
    n_corner, n_edge_v = 12, 30 * (nu - 1)
    n_int_face = (nu - 1) * (nu - 2) // 2
    n = n_corner + n_edge_v + 20 * n_int_face
    pos = np.zeros((n, 3), dtype=np.float64)
    pos[:12] = v0

    # lattice (i, j, k), i + j + k = nu : weight i on A, j on B, k on C
    ii, jj = np.meshgrid(np.arange(nu + 1), np.arange(nu + 1), indexing="ij")
    m = (ii + jj) <= nu
    li, lj = ii[m], jj[m]
    lk = nu - li - lj
    lut = -np.ones((nu + 1, nu + 1), dtype=np.int64)        # (i, j) -> local slot
    lut[li, lj] = np.arange(li.size)
    interior = (li > 0) & (lj > 0) & (lk > 0)
    int_rank = np.cumsum(interior) - 1

    def edge_ids(p, q, t_from_p):
        """ids of the vertices at integer distance t from p on base edge (p, q)."""
        e = ekeys[(min(p, q), max(p, q))]
        t = t_from_p if p < q else nu - t_from_p
        return n_corner + e * (nu - 1) + (t - 1)

    tris = []
    for fi, (a, b, c) in enumerate(f0):
        gid = np.empty(li.size, dtype=np.int64)
        gid[interior] = n_corner + n_edge_v + fi * n_int_face + int_rank[interior]
        # corners
        gid[(li == nu)] = a
        gid[(lj == nu)] = b
        gid[(lk == nu)] = c
        # edges (exclude corners)
        on_ab = (lk == 0) & (li > 0) & (lj > 0)
        gid[on_ab] = edge_ids(a, b, lj[on_ab])              # distance from A = j
        on_bc = (li == 0) & (lj > 0) & (lk > 0)
        gid[on_bc] = edge_ids(b, c, lk[on_bc])              # distance from B = k
        on_ca = (lj == 0) & (li > 0) & (lk > 0)
        gid[on_ca] = edge_ids(c, a, li[on_ca])              # distance from C = i
        p = (li[:, None] * v0[a] + lj[:, None] * v0[b] + lk[:, None] * v0[c]) / nu
        pos[gid] = p / np.linalg.norm(p, axis=1, keepdims=True)
        # "up" triangles (i,j),(i+1,j),(i,j+1) for i+j <= nu-1 ; "down" for i+j <= nu-2
        up = (li + lj) <= nu - 1
        ui, uj = li[up], lj[up]
        tris.append(np.stack([gid[lut[ui, uj]], gid[lut[ui + 1, uj]], gid[lut[ui, uj + 1]]], 1))
        dn = (li + lj) <= nu - 2
        di, dj = li[dn], lj[dn]
        tris.append(np.stack([gid[lut[di + 1, dj]], gid[lut[di + 1, dj + 1]], gid[lut[di, dj + 1]]], 1))
    faces = np.concatenate(tris, 0)

    perm = None
    if reorder in ("morton", "hilbert"):
        perm = np.argsort(_morton3(pos) if reorder == "morton" else _cube_hilbert(pos), kind="stable")
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        pos = pos[perm]
        faces = inv[faces]
    elif reorder is not None:
        raise ValueError(f"unknown reorder {reorder!r}")

    u = np.concatenate([faces[:, 0], faces[:, 1], faces[:, 2]])
    w = np.concatenate([faces[:, 1], faces[:, 2], faces[:, 0]])
    lo, hi = np.minimum(u, w), np.maximum(u, w)
    und = np.unique(lo * n + hi)
    lo, hi = und // n, und % n
    row = np.concatenate([lo, hi])
    col = np.concatenate([hi, lo])
    order = np.lexsort((col, row))
    edge_index = np.stack([row[order], col[order]]).astype(np.int64)
    return Mesh(pos=pos, edge_index=edge_index, faces=faces, nu=nu, perm=perm)


def complete_graph(n: int) -> np.ndarray:
    """K_n as ``erdos_renyi_graph(n, edge_prob=1)`` emits it (/root/reference/src/gwen/utils.py:176):
    every ordered pair (i, j), i != j, sorted by (row, col)."""
    r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m = r != c
    return np.stack([r[m], c[m]]).astype(np.int64)
