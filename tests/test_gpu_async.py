"""GPU parity of the asynchronous tile engine (engine 'async', mesh_navigation_amd/csrc/mnav_async.h: a ticket queue of woken tiles,
one launch per call) -- what 'auto' runs for single plans and small batches, i.e. for every real MeshPlanner::makePlan call
(mbf_mesh_nav/src/mesh_planner_execution.cpp:55-66).  Against the sequential oracle (dijkstra_mesh_planner.cpp:287-373): potential,
predecessors and vertex path bit for bit; offsets 0 / 0.3 / inf / negative, invalid and over-limit vertices, unreachable targets,
batches with fields and paths only; mnav_cancel inside a call; the in-kernel watchdog; a ticket ring that runs out (the call is
re-run on the tile rounds).  Its protocol is also checked on the CPU model (tests/test_async_model.py)."""
import threading
import time

import numpy as np
import pytest

from mesh_navigation_amd import capi, meshgen
from tests.common import Case, terrain_case
from tests.test_gpu_planners import assert_dijkstra_equal

pytestmark = pytest.mark.gpu


def test_auto_takes_the_async_engine_for_single_plans_and_small_batches(gpu_ctx_factory):
    case = terrain_case(128, 3)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    m = case.mesh
    rng = np.random.default_rng(2)
    robot = m.vertex_at(0.85, 0.8)
    goals = rng.choice(m.V, 24, replace=False).astype(np.uint32)
    goals = goals[goals != robot]
    for engine in ("auto", "async"):
        ctx.set_dijkstra_engine(engine)
        for off in (0.3, 0.0, float("inf"), -0.2):
            for g in goals[:2]:
                o = ctx.plan_dijkstra(int(g), robot, goal_dist_offset=off)
                assert o.stats["launches"] == 1                          # ONE launch: the asynchronous engine, not rounds
                assert_dijkstra_equal(o, case.om.dijkstra(case.weights, case.costs, int(g), robot, goal_dist_offset=off))
        nb = 8 if engine == "auto" else goals.shape[0]                 # ('auto' gives the engine batches of up to 8 plans)
        targets = np.full(nb, robot, np.uint32)
        refs = [case.om.dijkstra(case.weights, case.costs, int(g), robot) for g in goals[:nb]]
        for fields in (True, False):
            b = ctx.plan_dijkstra_batch(goals[:nb], targets, want_fields=fields)
            assert b["stats"]["launches"] == 1
            for k, ref in enumerate(refs):
                assert b["codes"][k] == ref.code
                assert np.array_equal(b["paths"][k], ref.path), (fields, k)
                if fields:
                    assert np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32)), k
                    assert np.array_equal(b["pred"][k], ref.pred), k
    ctx.set_dijkstra_engine("auto")


def test_async_banded_solves_and_few_workgroups_give_the_same_bits(gpu_ctx_factory):
    """the engine's knobs change the schedule, never the result: banded solves (the tile wakes itself for the rest), 1 workgroup,
    more workgroups than tiles in flight"""
    case = terrain_case(160, 9)
    m = case.mesh
    s, t = m.vertex_at(0.1, 0.15), m.vertex_at(0.9, 0.85)
    ref = case.om.dijkstra(case.weights, case.costs, s, t)
    for opts in (dict(async_band_mult=0.25), dict(async_band_mult=2.0), dict(async_wg_per_plan=1), dict(async_wg_per_plan=3), dict(async_wg_per_cu=1, async_wg_per_plan=1000)):
        ctx = gpu_ctx_factory()
        for k, v in opts.items():
            ctx.set_option(k, v)
        case.upload(ctx)
        assert_dijkstra_equal(ctx.plan_dijkstra(s, t), ref)


def test_async_cost_limit_invalid_vertices_and_unreachable_targets(gpu_ctx_factory):
    mesh = meshgen.terrain(150, 0.1, 11)
    rng = np.random.default_rng(5)
    costs = rng.uniform(0.0, 1.4, mesh.V).astype(np.float32)
    inv = (rng.uniform(size=mesh.V) < 0.05).astype(np.uint8)
    case = Case(mesh, costs, 1.0, invalid=inv)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    ok = np.flatnonzero((inv == 0) & (costs <= 0.8))
    st = rng.choice(ok, 8, replace=False)
    for s, t in zip(st[:4], st[4:]):
        for off in (0.3, -0.1):
            ref = case.om.dijkstra(case.weights, case.costs, int(s), int(t), goal_dist_offset=off, cost_limit=0.8, invalid=inv)
            assert_dijkstra_equal(ctx.plan_dijkstra(int(s), int(t), goal_dist_offset=off, cost_limit=0.8), ref)
    # a wall of over-limit vertices: NO_PATH_FOUND after the component was swept
    N = 150
    costs2 = np.zeros(mesh.V, np.float32)
    costs2[np.arange(N) * N + N // 2] = 5.0
    case2 = Case(mesh, costs2, 0.0)
    ctx2 = gpu_ctx_factory()
    case2.upload(ctx2)
    s, t = mesh.vertex_at(0.2, 0.5), mesh.vertex_at(0.8, 0.5)
    ref = case2.om.dijkstra(case2.weights, case2.costs, s, t)
    o = ctx2.plan_dijkstra(s, t)
    assert o.code == ref.code != 0
    assert_dijkstra_equal(o, ref)


def test_async_single_plans_at_1m(gpu_ctx_factory):
    """C2-sized mesh: the four offsets, fields and vector map; a 47-plan batch"""
    case = terrain_case(1000, 21)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    m = case.mesh
    robot = m.vertex_at(0.9, 0.9)
    rng = np.random.default_rng(5)
    goals = rng.choice(m.V, 47, replace=False).astype(np.uint32)
    for g, off in zip(goals[:4], (0.3, 0.0, float("inf"), -0.5)):
        ref = case.om.dijkstra(case.weights, case.costs, int(g), robot, goal_dist_offset=off)
        o = ctx.plan_dijkstra(int(g), robot, goal_dist_offset=off, want_vecmap=True)
        assert o.stats["launches"] == 1
        assert_dijkstra_equal(o, ref)
        assert np.array_equal(o.vecmap.view(np.uint32), case.om.dijkstra_vector_map(ref.pred).view(np.uint32))
    ctx.set_dijkstra_engine("async")
    b = ctx.plan_dijkstra_batch(goals, np.full(47, robot, np.uint32))
    assert b["stats"]["launches"] == 1 and (b["codes"] == 0).all()
    for k in (0, 13, 46):
        assert np.array_equal(b["paths"][k], case.om.dijkstra(case.weights, case.costs, int(goals[k]), robot).path), k


def test_async_ring_overflow_falls_back_to_the_rounds_and_watchdog_reports(gpu_ctx_factory):
    case = terrain_case(128, 3)
    m = case.mesh
    s, t = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, s, t)
    ctx = gpu_ctx_factory()
    ctx.set_option("async_ring_cap", 8)                                 # eight tickets: the call gives up (abort 5) ...
    case.upload(ctx)
    o = ctx.plan_dijkstra(s, t)
    assert o.stats["launches"] > 1                                      # ... and was re-run on the tile rounds
    assert_dijkstra_equal(o, ref)
    ctx.set_option("async_ring_cap", None)
    assert_dijkstra_equal(ctx.plan_dijkstra(s, t), ref)
    # the in-kernel wall-clock guard (abort 2): reported as an internal error, never a hang, never a wrong plan
    big = terrain_case(600, 4)
    ctx3 = gpu_ctx_factory()
    ctx3.set_dijkstra_engine("async")
    ctx3.set_option("async_max_s", 2e-5)
    big.upload(ctx3)
    with pytest.raises(RuntimeError, match="gave up"):
        ctx3.plan_dijkstra(big.mesh.vertex_at(0.05, 0.05), big.mesh.vertex_at(0.95, 0.95), goal_dist_offset=float("inf"))
    ctx3.set_option("async_max_s", None)                               # and the context is usable afterwards
    refb = big.om.dijkstra(big.weights, big.costs, big.mesh.vertex_at(0.05, 0.05), big.mesh.vertex_at(0.95, 0.95))
    assert_dijkstra_equal(ctx3.plan_dijkstra(big.mesh.vertex_at(0.05, 0.05), big.mesh.vertex_at(0.95, 0.95)), refb)
    # when `auto` picked the engine, a watchdog give-up is not the caller's problem: the call is re-run on the tile rounds
    ctx3.set_option("dijkstra_engine", None)                           # NaN = the built-in default: auto (not "whatever was set last")
    ctx3.set_option("async_max_s", 2e-5)
    o = ctx3.plan_dijkstra(big.mesh.vertex_at(0.05, 0.05), big.mesh.vertex_at(0.95, 0.95))
    assert o.stats["launches"] > 1
    assert_dijkstra_equal(o, refb)
    ctx3.set_option("async_max_s", None)
    o = ctx3.plan_dijkstra(big.mesh.vertex_at(0.05, 0.05), big.mesh.vertex_at(0.95, 0.95))
    assert o.stats["launches"] == 1                                     # auto again: the asynchronous engine, one launch
    with pytest.raises(ValueError):                                     # the retired engines are refused by name and by option alike
        ctx3.set_option("dijkstra_engine", 2)


def test_cancel_stops_a_running_async_call(gpu_ctx_factory):
    """mnav_cancel from another thread (MBF's action server) while 40 full-field plans run: CANCELED, within milliseconds"""
    case = terrain_case(1000, 21)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    ctx.set_dijkstra_engine("async")
    m = case.mesh
    rng = np.random.default_rng(1)
    goals = rng.choice(m.V, 40, replace=False).astype(np.uint32)
    targets = np.full(40, m.vertex_at(0.9, 0.9), np.uint32)
    full = ctx.plan_dijkstra_batch(goals, targets, goal_dist_offset=float("inf"))
    assert (full["codes"] == 0).all()
    t_full = full["stats"]["ms_total"]
    res = {}
    def work():
        t0 = time.perf_counter()
        res["b"] = ctx.plan_dijkstra_batch(goals, targets, goal_dist_offset=float("inf"))
        res["ms"] = (time.perf_counter() - t0) * 1e3
    th = threading.Thread(target=work)
    th.start()
    time.sleep(max(t_full * 0.25e-3, 0.001))
    ctx.cancel()
    th.join()
    assert res["b"]["rc"] == capi.CANCELED and (res["b"]["codes"] == capi.CANCELED).all()
    assert res["ms"] < t_full
    ok = ctx.plan_dijkstra_batch(goals[:3], targets[:3])               # the context keeps working
    assert (ok["codes"] == 0).all()
