"""Seeded synthetic terrain meshes (SURVEY.md §8d / BASELINE.md inputs).

The reference loads its meshes from HDF5/assimp files (mesh_map/src/mesh_map.cpp:149-452,
out of scope); these generators stand in for that loader and fix the id conventions
the rest of the repo relies on:

* vertex id  = row-major grid index  ``j * N + i``  (x = column i, y = row j);
* face ids   = cell-major, two triangles per cell split along the same diagonal,
  counter-clockwise seen from +z: ``(v00, v10, v11)`` then ``(v00, v11, v01)``;
* undirected edge ids = order of first appearance while iterating faces and, inside
  a face, the sides (v0,v1), (v1,v2), (v2,v0) -- the same convention the CPU checker uses
  (cross-checked in tests/test_meshgen.py).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class TerrainMesh:
    N: int
    h: float
    xyz: np.ndarray        # (V,3) float32
    faces: np.ndarray      # (F,3) uint32
    edges: np.ndarray      # (E,2) uint32, reference edge ids
    face_edges: np.ndarray # (F,3) uint32, edge between face vertex k and (k+1)%3

    @property
    def V(self) -> int:
        return int(self.xyz.shape[0])

    @property
    def F(self) -> int:
        return int(self.faces.shape[0])

    @property
    def E(self) -> int:
        return int(self.edges.shape[0])

    def vertex_at(self, fi: float, fj: float) -> int:
        """Vertex id at fractional grid position (fi, fj) in [0,1]^2."""
        i = min(self.N - 1, max(0, int(round(fi * (self.N - 1)))))
        j = min(self.N - 1, max(0, int(round(fj * (self.N - 1)))))
        return j * self.N + i


def grid_faces(N: int) -> np.ndarray:
    n = N - 1
    v00 = (np.arange(n, dtype=np.uint32)[:, None] * np.uint32(N) + np.arange(n, dtype=np.uint32)[None, :]).ravel()
    faces = np.empty((n * n, 6), dtype=np.uint32)          # per cell: (v00, v10, v11), (v00, v11, v01)
    faces[:, 0] = v00; faces[:, 1] = v00 + np.uint32(1); faces[:, 2] = v00 + np.uint32(N + 1)
    faces[:, 3] = v00; faces[:, 4] = faces[:, 2]; faces[:, 5] = v00 + np.uint32(N)
    return faces.reshape(2 * n * n, 3)


def grid_edges(N: int) -> tuple[np.ndarray, np.ndarray]:
    """edges_from_faces(grid_faces(N)) without the sort: the order of first appearance is known in closed form for
    the structured grid.  Cell (i, j) in row-major order adds, in this order: its bottom side (first row only --
    otherwise it is the top side of the cell below), right side, diagonal, top side, left side (first column only --
    otherwise the right side of the cell to the left); each oriented as first seen."""
    n = N - 1
    i, j = np.meshgrid(np.arange(n, dtype=np.int64), np.arange(n, dtype=np.int64))
    i, j = i.ravel(), j.ravel()
    v00 = j * N + i
    v10, v01, v11 = v00 + 1, v00 + N, v00 + N + 1
    has_b, has_l = (j == 0), (i == 0)
    cnt = 3 + has_b.astype(np.int64) + has_l.astype(np.int64)
    base = np.cumsum(cnt) - cnt
    e_right = base + has_b
    e_diag, e_top = e_right + 1, e_right + 2
    E = int(cnt.sum())
    edges = np.empty((E, 2), np.uint32)
    edges[e_right, 0], edges[e_right, 1] = v10, v11
    edges[e_diag, 0], edges[e_diag, 1] = v11, v00
    edges[e_top, 0], edges[e_top, 1] = v11, v01
    bi = np.nonzero(has_b)[0]
    edges[base[bi], 0], edges[base[bi], 1] = v00[bi], v10[bi]
    li = np.nonzero(has_l)[0]
    edges[e_top[li] + 1, 0], edges[e_top[li] + 1, 1] = v01[li], v00[li]
    cell = np.arange(n * n, dtype=np.int64)
    e_bottom = np.where(has_b, base, e_top[np.maximum(cell - n, 0)])
    e_left = np.where(has_l, e_top + 1, e_right[np.maximum(cell - 1, 0)])
    face_edges = np.empty((2 * n * n, 3), np.uint32)
    face_edges[0::2, 0], face_edges[0::2, 1], face_edges[0::2, 2] = e_bottom, e_right, e_diag
    face_edges[1::2, 0], face_edges[1::2, 1], face_edges[1::2, 2] = e_diag, e_top, e_left
    return edges, face_edges


def edges_from_faces(faces: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Undirected edges in order of first appearance + per-face side -> edge id."""
    f = faces.astype(np.int64)
    a = f[:, [0, 1, 2]].ravel()          # face-major, sides (0,1),(1,2),(2,0)
    b = f[:, [1, 2, 0]].ravel()
    lo = np.minimum(a, b)
    hi = np.maximum(a, b)
    key = (lo << 32) | hi
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # unique-key index -> rank by first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    face_edges = rank[inv].reshape(-1, 3).astype(np.uint32)
    fa = first[order]
    edges = np.stack([a[fa], b[fa]], axis=1).astype(np.uint32)   # oriented as first seen
    return edges, face_edges


def terrain(N: int, h: float = 0.1, seed: int = 0, amplitude: float = 2.0,
            base_freq: float = 1.0 / 50.0, jitter: float = 0.2, octaves: int = 5) -> TerrainMesh:
    """Terrain(N, h, seed) of SURVEY.md §8d: jittered grid, 5-octave sin*cos heights."""
    rng = np.random.default_rng(seed)
    jx = rng.uniform(-jitter * h, jitter * h, size=(N, N))
    jy = rng.uniform(-jitter * h, jitter * h, size=(N, N))
    phi = rng.uniform(0.0, 2.0 * np.pi, size=octaves)
    psi = rng.uniform(0.0, 2.0 * np.pi, size=octaves)
    i, j = np.meshgrid(np.arange(N, dtype=np.float64), np.arange(N, dtype=np.float64))
    x = i * h + jx
    y = j * h + jy
    z = np.zeros_like(x)
    for k in range(1, octaves + 1):
        z += (amplitude / 2.0 ** k) * np.sin(2.0 * np.pi * base_freq * 2.0 ** k * x + phi[k - 1]) \
             * np.cos(2.0 * np.pi * base_freq * 2.0 ** k * y + psi[k - 1])
    xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.float32)
    faces = grid_faces(N)
    edges, face_edges = grid_edges(N)
    return TerrainMesh(N=N, h=h, xyz=xyz, faces=faces, edges=edges, face_edges=face_edges)


def flat_grid(N: int, h: float = 1.0) -> TerrainMesh:
    """Un-jittered flat grid (tie-stress / analytic tests)."""
    i, j = np.meshgrid(np.arange(N, dtype=np.float64), np.arange(N, dtype=np.float64))
    xyz = np.stack([(i * h).ravel(), (j * h).ravel(), np.zeros(N * N)], axis=1).astype(np.float32)
    faces = grid_faces(N)
    edges, face_edges = edges_from_faces(faces)
    return TerrainMesh(N=N, h=h, xyz=xyz, faces=faces, edges=edges, face_edges=face_edges)


def edge_lengths(mesh: TerrainMesh) -> np.ndarray:
    """float32 Euclidean edge lengths with the oracle's operation order
    (dx*dx + dy*dy + dz*dz in float32, then sqrt)."""
    p = mesh.xyz[mesh.edges[:, 0]]
    q = mesh.xyz[mesh.edges[:, 1]]
    d = (p - q).astype(np.float32)
    s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
    s = (s + d[:, 2] * d[:, 2]).astype(np.float32)
    return np.sqrt(s, dtype=np.float32)


# Benchmark / parity configurations of BASELINE.md (C1..C5)
CONFIGS = {
    "C1": dict(N=224, seed=1),
    "C2": dict(N=1000, seed=2),
    "C3": dict(N=1000, seed=3),
    "C4": dict(N=3163, seed=4),
    "C5": dict(N=1000, seed=2),
}


def from_faces(xyz, faces, N: int = 0, h: float = 0.0) -> TerrainMesh:
    """Any triangle soup with shared vertices (edge ids = order of first appearance)."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
    edges, face_edges = edges_from_faces(faces)
    return TerrainMesh(N=N, h=h, xyz=xyz, faces=faces, edges=edges, face_edges=face_edges)


def punched(N: int, h: float = 0.1, seed: int = 0, drop: float = 0.08, amplitude: float = 0.5,
            cut_column: int | None = None) -> TerrainMesh:
    """Terrain with a random `drop` fraction of its faces removed: holes, boundary loops, vertices of
    valence 1..6, and a few vertices left without any face (ragged input for the planners).  With
    `cut_column` every face touching that grid column goes too: two components, a column of face-less
    vertices between them."""
    t = terrain(N, h, seed, amplitude=amplitude)
    rng = np.random.default_rng(seed + 7919)
    keep = rng.uniform(size=t.F) >= drop
    if cut_column is not None:
        keep &= ~((t.faces % N) == cut_column).any(axis=1)
    return from_faces(t.xyz, t.faces[keep], N=N, h=h)


def fan_field(spokes: int = 40, rings: int = 6, seed: int = 0) -> TerrainMesh:
    """A disc triangulated as concentric rings around ONE centre vertex of valence `spokes` (> 16: beyond
    the lanes-per-vertex fast paths of the kernels), slightly bumpy."""
    rng = np.random.default_rng(seed)
    pts = [(0.0, 0.0)]
    for r in range(1, rings + 1):
        for s in range(spokes):
            a = 2.0 * np.pi * (s + 0.5 * (r % 2)) / spokes
            pts.append((0.1 * r * np.cos(a), 0.1 * r * np.sin(a)))
    pts = np.asarray(pts)
    z = 0.02 * rng.standard_normal(len(pts))
    xyz = np.column_stack([pts, z]).astype(np.float32)
    def vid(r, s):
        return 0 if r == 0 else 1 + (r - 1) * spokes + (s % spokes)
    faces = []
    for s in range(spokes):
        faces.append((0, vid(1, s), vid(1, s + 1)))
    for r in range(1, rings):
        for s in range(spokes):
            a, b = vid(r, s), vid(r, s + 1)
            c, d = vid(r + 1, s), vid(r + 1, s + 1)
            faces.append((a, c, b)); faces.append((b, c, d))
    return from_faces(xyz, np.asarray(faces, np.uint32))
