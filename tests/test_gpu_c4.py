"""BASELINE config C4 on one GPU: the 10M-vertex terrain (N=3163, seed 4).  Parity with the oracle
(bit-exact potential, predecessors, vertex path) and the size-independent properties of the field."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_dijkstra_10m_vertices(gpu_ctx_factory):
    mesh = meshgen.terrain(3163, 0.1, 4)
    assert (mesh.V, mesh.F, mesh.E) == (10004569, 19996488, 30001056)
    w = meshgen.edge_lengths(mesh)
    costs = np.zeros(mesh.V, np.float32)
    ctx = gpu_ctx_factory()
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
    ctx.upload_costs(costs, w)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    out = ctx.plan_dijkstra(s, t)
    full = ctx.plan_dijkstra(s, t, goal_dist_offset=float("inf"))
    d, e = full.dist, mesh.edges
    assert d[s] == 0 and np.isfinite(d).all()
    assert (d[e[:, 0]] <= d[e[:, 1]] + w).all() and (d[e[:, 1]] <= d[e[:, 0]] + w).all()
    nz = np.arange(mesh.V) != s
    assert (d[full.pred[nz]] < d[nz]).all()
    assert full.stats["algorithmic_bytes"] == 24 * mesh.V + 24 * mesh.E           # 960.1 MB, BASELINE.md
    om = O.OracleMesh(mesh.xyz, mesh.faces)
    ref = om.dijkstra(w, costs, s, t)
    assert out.code == ref.code == 0
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(out.pred, ref.pred) and np.array_equal(out.path, ref.path)
