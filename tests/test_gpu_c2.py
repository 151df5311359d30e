"""GPU parity at BASELINE's full single-GPU size (config C2 / C3 shape: 1M-vertex terrain)."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from tests.common import Case, layered_costs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(gpu_ctx_factory):
    case = Case(meshgen.terrain(1000, 0.1, 2))
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    return case, ctx


@pytest.mark.parametrize("engine", ["tiled", "band", "tile_batch", "async"])
def test_dijkstra_c2_bit_exact(c2, engine):
    case, ctx = c2
    ctx.set_dijkstra_engine(engine)
    m = case.mesh
    s, t = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, s, t)
    out = ctx.plan_dijkstra(s, t)
    assert out.code == ref.code == 0
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(out.path, ref.path)
    # predecessors: the device uses the documented (dist[u], u) tie rule, the oracle the emulated
    # Meap order; on jittered terrain they coincide (DESIGN.md "tie rule")
    assert np.array_equal(out.pred, ref.pred)
    ctx.set_dijkstra_engine("auto")


def test_dijkstra_c2_full_field_properties(c2):
    case, ctx = c2
    m = case.mesh
    s, t = m.vertex_at(0.5, 0.5), m.vertex_at(0.02, 0.97)
    out = ctx.plan_dijkstra(s, t, goal_dist_offset=float("inf"))
    d, e, w = out.dist, m.edges, case.weights
    assert d[s] == 0 and np.isfinite(d).all()
    assert (d[e[:, 0]] <= d[e[:, 1]] + w).all() and (d[e[:, 1]] <= d[e[:, 0]] + w).all()   # relaxation fixed point
    nz = np.arange(m.V) != s
    assert (d[out.pred[nz]] < d[nz]).all()                                                # descent along pred
    assert out.stats["algorithmic_bytes"] == 24 * m.V + 24 * m.E


def test_cvp_c2(c2):
    case, ctx = c2
    m = case.mesh
    s, t = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    sp = m.xyz[s] + np.array([0.03, 0.02, 0.0], np.float32)
    tp = m.xyz[t] + np.array([0.03, 0.02, 0.0], np.float32)
    sf, _ = case.om.containing_face(sp)
    tf, _ = case.om.containing_face(tp)
    ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf)
    out = ctx.plan_cvp(sp, sf, tf)
    assert out.code == ref.code == 0
    fin = np.isfinite(ref.dist)
    assert np.array_equal(np.isfinite(out.dist), fin)
    rel = np.abs(out.dist[fin] - ref.dist[fin]) / np.maximum(ref.dist[fin], 1e-12)
    assert rel.max() <= 1e-5                                         # north_star tolerance ...
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.pred, ref.pred)   # ... met bit for bit
    assert (out.pred != ref.pred).mean() < 1e-4


def test_cvp_c3_layered_costs_1m(gpu_ctx_factory):
    """BASELINE config 3: 1M vertices, Inflation + Steepness costs (Avg), edge_cost_factor 1."""
    base = Case(meshgen.terrain(1000, 0.1, 3, amplitude=0.8))
    costs, parts = layered_costs(base, "avg")
    case = Case(base.mesh, costs, 1.0)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    m = case.mesh
    free = np.where(costs < 0.5)[0]
    def near(fi, fj):
        v = m.vertex_at(fi, fj)
        return int(free[((m.xyz[free, :2] - m.xyz[v, :2]) ** 2).sum(1).argmin()])
    s, t = near(0.1, 0.1), near(0.9, 0.9)
    sp = m.xyz[s] + np.array([0.02, 0.01, 0], np.float32)
    tp = m.xyz[t] + np.array([0.02, 0.01, 0], np.float32)
    sf, _ = case.om.containing_face(sp)
    tf, _ = case.om.containing_face(tp)
    ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf)
    out = ctx.plan_cvp(sp, sf, tf)
    assert out.code == ref.code == 0
    fin = np.isfinite(ref.dist)
    assert np.array_equal(np.isfinite(out.dist), fin)
    rel = np.abs(out.dist[fin] - ref.dist[fin]) / np.maximum(ref.dist[fin], 1e-12)
    assert rel.max() <= 1e-5                                         # north_star tolerance ...
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.pred, ref.pred)   # ... met bit for bit
    refd = case.om.dijkstra(case.weights, case.costs, s, t)
    outd = ctx.plan_dijkstra(s, t)
    assert np.array_equal(outd.dist.view(np.uint32), refd.dist.view(np.uint32)) and np.array_equal(outd.path, refd.path)


def test_punched_terrain_1m_deep_cascades(gpu_ctx_factory):
    """1M-vertex terrain with 20 % of its faces removed: the CVP wave wraps around thousands of holes and
    fills their shadows backwards (cascades of pops below the main front, nested hundreds of levels deep).
    Both planners must still match the sequential oracle bit for bit."""
    mesh = meshgen.punched(1000, 0.1, 7, drop=0.2)
    case = Case(mesh)
    deg = np.bincount(mesh.edges.ravel(), minlength=mesh.V)
    assert (deg == 0).sum() > 0
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    while deg[s] == 0: s += 1
    while deg[t] == 0: t += 1
    ref = case.om.dijkstra(case.weights, case.costs, s, t)
    for engine in ("tiled", "tile_batch", "async"):
        ctx.set_dijkstra_engine(engine)
        out = ctx.plan_dijkstra(s, t)
        assert out.code == ref.code == 0
        assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(out.pred, ref.pred) and np.array_equal(out.path, ref.path)
    ctx.set_dijkstra_engine("auto")
    sf = int(np.where((mesh.faces == s).any(axis=1))[0][0])
    tf = int(np.where((mesh.faces == t).any(axis=1))[0][0])
    sp = mesh.xyz[mesh.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
    refc = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf)
    outc = ctx.plan_cvp(sp, sf, tf)
    assert outc.code == refc.code == 0
    assert np.array_equal(outc.dist.view(np.uint32), refc.dist.view(np.uint32))
    assert np.array_equal(outc.pred, refc.pred)
    upd = refc.pred != np.arange(mesh.V)
    assert np.array_equal(outc.cutface[upd], refc.cutface[upd])
    assert np.array_equal(outc.direction[upd].view(np.uint32), refc.direction[upd].view(np.uint32))


def test_cancel_stops_a_running_batch_on_the_tile_rounds(c2):
    """mnav_cancel during a long run of the tile rounds (640 full-field plans on the 1M mesh, ~0.3 s): the host loop
    looks at the flag between its graph replays, stops launching and every plan reports CANCELED (51), like the reference's
    `while (!pq.isEmpty() && !cancel_planning_)` + `:350-354`.  A later plan is unaffected (flag reset, :238)."""
    import threading
    import time
    from mesh_navigation_amd import capi
    case, ctx = c2
    m = case.mesh
    rng = np.random.default_rng(11)
    n = 640
    seeds = rng.choice(m.V, n, replace=False).astype(np.uint32)
    targets = np.full(n, m.vertex_at(0.5, 0.5), np.uint32)
    ctx.set_dijkstra_engine("tiled")
    t0 = time.perf_counter()
    full = ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=float("inf"), path_cap=4096)   # warm-up + reference time
    t_full = time.perf_counter() - t0
    assert full["rc"] == 0 and (full["codes"] == 0).all()
    t0 = time.perf_counter()
    full = ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=float("inf"), path_cap=4096)
    t_full = time.perf_counter() - t0
    out = {}
    th = threading.Thread(target=lambda: out.update(ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=float("inf"), path_cap=4096)))
    t0 = time.perf_counter()
    th.start()
    time.sleep(min(0.05, 0.25 * t_full))
    ctx.cancel()
    th.join(timeout=60)
    t_cancel = time.perf_counter() - t0
    assert not th.is_alive()
    assert out["rc"] == capi.CANCELED and (out["codes"] == capi.CANCELED).all(), (out["rc"], t_full, t_cancel)
    assert t_cancel < t_full, (t_cancel, t_full)
    again = ctx.plan_dijkstra_batch(seeds[:128], targets[:128], goal_dist_offset=0.3, path_cap=4096)
    assert again["rc"] == 0 and (again["codes"] == 0).all()
    ctx.set_dijkstra_engine("auto")
