"""CPU: the rule the device uses to decide which sources the Dijkstra wave EXPANDS (mnav_eval.h goal_cut / expanded_source: the
finalize pass k_dij_finalize, the lazy path walks and the lazy vector entries share it) against the reference loop's behaviour
(dijkstra_mesh_planner.cpp:287-348 through the oracle), for every kind of goal_dist_offset the reference accepts -- the default,
zero, +inf, negative ones (the wave stops expanding AT the robot vertex: sources are the vertices popped before it in (value, id)
order), -inf, and a negative one that rounds away in float32.

From the FINAL potential d* (offset = inf) and the rule alone, the reference's outputs are rebuilt in numpy exactly as the finalize
pass derives them: a vertex at or below the cut keeps d*, any other vertex holds the smallest sum d*[u] + w over its EXPANDED
neighbours u (or +inf), and the predecessor is the first-popped neighbour attaining the value."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O
from tests.common import Case, terrain_case


def rebuild(case, s, t, offset):
    m = case.mesh
    full = case.om.dijkstra(case.weights, case.costs, s, t, goal_dist_offset=np.inf).dist
    exp, goal, cut = O.product_expanded_sources(full, t, offset)
    e = m.edges.astype(np.int64)
    src = np.concatenate([e[:, 0], e[:, 1]]); dst = np.concatenate([e[:, 1], e[:, 0]])
    w = np.concatenate([case.weights, case.weights]).astype(np.float32)
    ok = exp[src]
    cand = (full[src] + w).astype(np.float32)                        # the float add of dijkstra :331
    best = np.full(m.V, np.inf, np.float32)
    np.minimum.at(best, dst[ok], cand[ok])
    dist = np.where(full <= np.float32(cut), full, best).astype(np.float32)
    dist[s] = 0.0
    key = np.full(m.V, np.iinfo(np.int64).max, np.int64)
    att = ok & (cand == dist[dst]) & np.isfinite(cand)
    k = (full[src].view(np.uint32).astype(np.int64) << 32) | src     # first popped = smallest (value, id)
    np.minimum.at(key, dst[att], k[att])
    pred = np.arange(m.V)
    has = key != np.iinfo(np.int64).max
    pred[has] = key[has] & 0xFFFFFFFF
    pred[s] = s
    return dist, pred, goal


@pytest.mark.parametrize("offset", [0.3, 0.0, np.inf, -0.2, -1e-9, -50.0, -np.inf])
def test_expanded_set_rule_reproduces_the_reference(offset):
    jittered = terrain_case(48, 5)
    flat = Case(meshgen.flat_grid(40, 1.0))                          # many vertices tie with the robot vertex's value: the id half of the rule
    for case, pairs in ((jittered, ((3, 900), (2000, 77), (1500, 1501))), (flat, ((820, 831), (820, 207), (820, 3), (0, 1599)))):
        for s, t in pairs:
            ref = case.om.dijkstra(case.weights, case.costs, s, t, goal_dist_offset=offset)
            dist, pred, goal = rebuild(case, s, t, offset)
            assert ref.code == 0
            assert np.array_equal(dist.view(np.uint32), ref.dist.view(np.uint32)), (offset, s, t)
            assert np.array_equal(pred, ref.pred), (offset, s, t)
            assert np.float32(goal).view(np.uint32) == np.float32(ref.stats["goal_dist"]).view(np.uint32)
