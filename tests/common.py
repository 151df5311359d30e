"""Shared fixtures-as-functions for the parity tests (seeded inputs of BASELINE.md)."""
from __future__ import annotations

import functools

import numpy as np

from mesh_navigation_amd import meshgen
from oracle import oracle as O


class Case:
    """A mesh + costs + the derived oracle-side arrays."""

    def __init__(self, mesh: meshgen.TerrainMesh, vertex_costs=None, edge_cost_factor: float = 0.0, invalid=None):
        self.mesh = mesh
        self.om = O.OracleMesh(mesh.xyz, mesh.faces)
        assert self.om.E == mesh.E
        self.edge_dist = self.om.edge_distances()
        self.fn = self.om.face_normals()
        self.vn = self.om.vertex_normals(self.fn)
        self.costs = np.zeros(mesh.V, np.float32) if vertex_costs is None else np.asarray(vertex_costs, np.float32)
        self.factor = edge_cost_factor
        self.weights = self.om.edge_weights(self.edge_dist, self.costs, edge_cost_factor)
        self.invalid = None if invalid is None else np.asarray(invalid, np.uint8)

    def upload(self, ctx):
        ctx.upload_mesh(self.mesh.xyz, self.mesh.faces, self.mesh.edges, self.vn)
        ctx.upload_costs(self.costs, self.weights, self.invalid)


@functools.lru_cache(maxsize=4)
def terrain_case(N: int, seed: int) -> Case:
    return Case(meshgen.terrain(N, 0.1, seed))


def layered_costs(case: Case, mode: str = "avg"):
    """Config-3 cost stack: Steepness + Inflation combined."""
    steep, lethal = case.om.steepness(case.vn, 0.3)
    infl_cost, infl_dist, infl_vec = case.om.inflation(lethal, case.edge_dist)
    combined = O.combine([steep, infl_cost], [1.0, 1.0], mode)
    return combined, dict(steepness=steep, lethal=lethal, inflation=infl_cost, infl_dist=infl_dist, infl_vec=infl_vec)


def rel_err(a, b):
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), "reached sets differ"
    d = np.abs(a[fin].astype(np.float64) - b[fin].astype(np.float64))
    return float((d / np.maximum(np.abs(b[fin]), 1e-12)).max()) if fin.any() else 0.0
