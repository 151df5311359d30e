"""The rule behind k_path_lazy (paths without the finalize pass): walking from the robot vertex and picking, at every vertex,
the neighbour u that minimises (dist[u] + w(u, v), dist[u], u) among the neighbours the reference would have expanded
(dist[u] <= goal_dist, cost within the limit, v valid) reproduces the reference's predecessor chain -- restated here in numpy
on the ORACLE's potential, so the rule itself is checked without a GPU (the device walk is covered by every GPU batch test,
which compare paths with the oracle)."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from tests.common import Case, layered_costs


def lazy_walk(mesh, weights, costs, dist, seed, target, offset=0.3, cost_limit=1.0, invalid=None):
    nbrs = [[] for _ in range(mesh.V)]
    for e, (a, b) in enumerate(mesh.edges):
        nbrs[int(a)].append((int(b), e)); nbrs[int(b)].append((int(a), e))
    goal_dist = np.float32(np.float64(dist[target]) + offset)
    path, v = [], int(target)
    while v != seed and len(path) <= mesh.V:
        best = None
        for u, e in nbrs[v]:
            du = dist[u]
            if not np.isfinite(du) or du > goal_dist or float(costs[u]) > cost_limit or (invalid is not None and invalid[v]):
                continue
            s = np.float32(du + weights[e])                                   # the float add of dijkstra :331
            key = (s, du, u)
            if best is None or key < best:
                best = key
        assert best is not None and best[0] == dist[v]                        # the fixed-point property, per hop
        v = best[2]
        path.append(v)
    return np.asarray(path[::-1], np.uint32)


@pytest.mark.parametrize("kind", ["terrain", "layered", "punched"])
def test_lazy_walk_reproduces_the_reference_path(kind):
    if kind == "terrain":
        case = Case(meshgen.terrain(64, 0.1, 4))
    elif kind == "layered":
        base = Case(meshgen.terrain(64, 0.1, 3, amplitude=0.8))
        costs, _ = layered_costs(base, "avg")
        case = Case(base.mesh, costs, 1.0)
    else:
        case = Case(meshgen.punched(64, 0.1, 6, drop=0.15))
    m = case.mesh
    deg = np.bincount(m.edges.ravel(), minlength=m.V)
    rng = np.random.default_rng(3)
    done = 0
    for _ in range(40):
        s, t = (int(x) for x in rng.choice(np.nonzero((deg > 0) & (case.costs < 0.5))[0], 2, replace=False))
        for offset in (0.3, np.inf):
            ref = case.om.dijkstra(case.weights, case.costs, s, t, goal_dist_offset=offset)
            if ref.code != 0:
                continue
            got = lazy_walk(m, case.weights, case.costs, ref.dist, s, t, offset=offset)
            assert np.array_equal(got, ref.path), (kind, s, t, offset)
            done += 1
    assert done > 20
