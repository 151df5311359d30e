"""Cost preparation on the device: edge weights, Max/Avg combination and the INCREMENTAL update
(MeshMap::layerChanged + updateEdgeWeights(changed), mesh_map.cpp:454-493, :563-618) -- bit-equal to the restated
reference formula (which tests/test_ref_pins_oracle.py pins to the reference's own code, incl. its incremental path)."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from tests.common import Case

pytestmark = pytest.mark.gpu


def test_incremental_cost_update_matches_full_recompute(gpu_ctx_factory):
    mesh = meshgen.terrain(96, 0.1, 21)
    rng = np.random.default_rng(9)
    costs = rng.uniform(0, 0.9, mesh.V).astype(np.float32)
    base = Case(mesh, costs, 0.7)
    ctx = gpu_ctx_factory()
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, base.vn)
    w = ctx.compute_edge_weights(costs, base.edge_dist, 0.7)
    assert np.array_equal(w.view(np.uint32), base.weights.view(np.uint32))
    ids = rng.choice(mesh.V, 400, replace=False).astype(np.uint32)
    vals = rng.uniform(0, 1.5, ids.size).astype(np.float32)
    vals[:7] = np.inf                                          # lethal vertices: infinite weights around them
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    keep = ~np.isin(ids, [s, t])
    ids, vals = ids[keep], vals[keep]
    ctx.update_costs(ids, vals)                                # only the ids and values cross PCIe
    costs2 = costs.copy()
    costs2[ids] = vals
    want = Case(mesh, costs2, 0.7)
    vc, w2 = ctx.download_costs()
    assert np.array_equal(vc.view(np.uint32), costs2.view(np.uint32))
    assert np.array_equal(w2.view(np.uint32), want.weights.view(np.uint32))
    # the planners see the change (cost-limit folded copies are rebuilt)
    ref = want.om.dijkstra(want.weights, costs2, s, t)
    out = ctx.plan_dijkstra(s, t)
    assert out.code == ref.code
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.pred, ref.pred)
    # a second, overlapping update
    ids3 = np.concatenate([ids[:50], rng.choice(mesh.V, 50, replace=False).astype(np.uint32)])
    ids3 = np.unique(ids3[~np.isin(ids3, [s, t])])
    vals3 = rng.uniform(0, 0.5, ids3.size).astype(np.float32)
    ctx.update_costs(ids3, vals3)
    costs3 = costs2.copy()
    costs3[ids3] = vals3
    want3 = Case(mesh, costs3, 0.7)
    vc, w3 = ctx.download_costs()
    assert np.array_equal(w3.view(np.uint32), want3.weights.view(np.uint32))


def test_update_with_factor_zero_leaves_the_weights_alone(gpu_ctx_factory):
    """edge_cost_factor 0 (the reference default): "skipping edge cost update" (:568-572); only the cut-offs change."""
    case = Case(meshgen.terrain(64, 0.1, 5))
    ctx = gpu_ctx_factory()
    case.upload(ctx)                                           # weights uploaded by the caller
    m = case.mesh
    s, t = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    wall = np.array([m.vertex_at(0.5, f) for f in np.linspace(0.0, 0.8, 60)], np.uint32)
    wall = np.unique(wall)
    ctx.update_costs(wall, np.full(wall.size, 2.0, np.float32))   # above cost_limit: a wall the plan must go around
    costs = case.costs.copy()
    costs[wall] = 2.0
    _, w = ctx.download_costs()
    assert np.array_equal(w.view(np.uint32), case.weights.view(np.uint32))
    ref = case.om.dijkstra(case.weights, costs, s, t)
    out = ctx.plan_dijkstra(s, t)
    assert out.code == ref.code == 0 and np.array_equal(out.path, ref.path)
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32))


def test_combination_stays_on_the_device(gpu_ctx_factory):
    from tests.common import layered_costs
    base = Case(meshgen.terrain(48, 0.1, 3, amplitude=0.8))
    costs, parts = layered_costs(base, "avg")
    want = Case(base.mesh, costs, 1.0)
    ctx = gpu_ctx_factory()
    ctx.upload_mesh(base.mesh.xyz, base.mesh.faces, base.mesh.edges, base.vn)
    vc, ew = ctx.combine_costs([parts["steepness"], parts["inflation"]], [1.0, 1.0], base.edge_dist, 1.0, "avg")
    assert np.array_equal(vc.view(np.uint32), costs.view(np.uint32)) and np.array_equal(ew.view(np.uint32), want.weights.view(np.uint32))
    vc2, ew2 = ctx.download_costs()
    assert np.array_equal(vc2.view(np.uint32), vc.view(np.uint32)) and np.array_equal(ew2.view(np.uint32), ew.view(np.uint32))


def test_update_costs_with_the_callers_edge_weights(gpu_ctx_factory):
    """mnav_update_costs + mnav_update_edge_weights on UPLOADED weights: the caller's own updateEdgeWeights (mesh_map.cpp:563-618) ran on
    the host (here: the oracle's), the changed vertices and the weights of their edges go to the device -- the plans that follow equal
    the plans on a full upload of the new arrays (what the plugin does behind mesh_gpu_planners/CostObserverLayer)."""
    mesh = meshgen.terrain(160, 0.1, 23)
    rng = np.random.default_rng(8)
    c0 = rng.uniform(0.0, 0.5, mesh.V).astype(np.float32)
    case0 = Case(mesh, c0, 1.0)
    ctx = gpu_ctx_factory()
    case0.upload(ctx)
    s, t = mesh.vertex_at(0.1, 0.15), mesh.vertex_at(0.9, 0.85)
    assert np.array_equal(ctx.plan_dijkstra(s, t).path, case0.om.dijkstra(case0.weights, case0.costs, s, t).path)
    ids = rng.choice(mesh.V, 700, replace=False).astype(np.uint32)
    c1 = c0.copy()
    c1[ids] = rng.uniform(0.0, 0.95, ids.shape[0]).astype(np.float32)
    case1 = Case(mesh, c1, 1.0)                                        # the host's recomputed weights
    changed_edges = np.nonzero(np.isin(mesh.edges[:, 0], ids) | np.isin(mesh.edges[:, 1], ids))[0].astype(np.uint32)
    assert (case1.weights.view(np.uint32) != case0.weights.view(np.uint32)).sum() > 0
    assert not (case1.weights.view(np.uint32) != case0.weights.view(np.uint32))[np.setdiff1d(np.arange(mesh.E), changed_edges)].any()
    ctx.update_costs(ids, c1[ids])
    ctx.update_edge_weights(changed_edges, case1.weights[changed_edges])
    vc, w = ctx.download_costs()
    assert np.array_equal(vc.view(np.uint32), c1.view(np.uint32)) and np.array_equal(w.view(np.uint32), case1.weights.view(np.uint32))
    for offset in (0.3, float("inf")):
        ref = case1.om.dijkstra(case1.weights, case1.costs, s, t, goal_dist_offset=offset)
        out = ctx.plan_dijkstra(s, t, goal_dist_offset=offset)
        assert out.code == ref.code and np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.path, ref.path)
    tg = rng.choice(mesh.V, 64, replace=False).astype(np.uint32)      # a batch: the tile-batch engine's weights follow too
    b = ctx.plan_dijkstra_batch(np.full(64, s, np.uint32), tg, want_fields=False)
    for k in (0, 31, 63):
        refk = case1.om.dijkstra(case1.weights, case1.costs, s, int(tg[k]))
        assert b["codes"][k] == refk.code and np.array_equal(b["paths"][k], refk.path)
