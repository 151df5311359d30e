"""GPU: the three paths `bench.py` runs that had no `-m gpu` test of their own (VERDICT r03 "missing" #3):

  (a) the tile-batch engine on the C4 mesh (10M vertices) with a batch large enough to take it (>= 1024 plans):
      vertex paths and popped potential of a sample against the oracle (dijkstra_mesh_planner.cpp:287-348, :358-373);
  (b) V-sized outputs of a batch at C2 scale through `k_dij_finalize<8, true>` with FULL groups of eight plans:
      potential / predecessors / vector map bits of >= 8 plans against the oracle (:189-209, :293-300);
  (c) `mnav_cancel` during a large tile-batch run: every plan CANCELED (:287, :350-354), a later batch unaffected (:238).
"""
import threading
import time

import numpy as np
import pytest

from mesh_navigation_amd import capi, meshgen
from oracle import oracle as O
from tests.common import terrain_case

pytestmark = pytest.mark.gpu


def _popped(full, t, offset):
    dt = full[t]
    gd = np.float32(np.float64(dt) + offset) if np.isfinite(dt) else np.float32(np.inf)
    return full <= gd


def test_c4_tile_batch_1024_plans_paths_and_popped_potential(gpu_ctx_factory):
    mesh = meshgen.terrain(3163, 0.1, 4)
    w = meshgen.edge_lengths(mesh)
    costs = np.zeros(mesh.V, np.float32)
    ctx = gpu_ctx_factory()
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
    ctx.upload_costs(costs, w)
    ctx.set_dijkstra_engine("tile_batch")
    rng = np.random.default_rng(41)
    n = 1024
    seeds = rng.choice(mesh.V, n, replace=False).astype(np.uint32)
    targets = rng.choice(mesh.V, n, replace=False).astype(np.uint32)
    targets[: n // 2] = mesh.vertex_at(0.9, 0.9)                      # the bench's common robot vertex and scattered ones
    b = ctx.plan_dijkstra_batch(seeds, targets, want_fields=False, path_cap=32768)
    assert b["rc"] == 0 and (b["codes"] == 0).all()
    assert b["stats"]["n_plans"] == n
    om = O.OracleMesh(mesh.xyz, mesh.faces)
    for k in (0, 511, 777, 1023):
        ref = om.dijkstra(w, costs, int(seeds[k]), int(targets[k]))
        assert ref.code == 0
        assert np.array_equal(b["paths"][k], ref.path), k
        full = om.dijkstra(w, costs, int(seeds[k]), int(targets[k]), goal_dist_offset=np.inf).dist
        pot = ctx.download_output("popped", k)
        m = _popped(full, int(targets[k]), 0.3)
        assert np.array_equal(pot[m].view(np.uint32), full[m].view(np.uint32)), k
        assert np.isinf(pot[~m]).all()
    # size-independent property over a spread of the plans: a path starts at the wave source and ends next to the robot
    # vertex (reference list order seed ... pred[target]), every hop is an edge of the grid mesh (meshgen: id = j * N + i,
    # cells split along v00 - v11)
    N = 3163
    for k in range(0, n, 37):
        p = b["paths"][k].astype(np.int64)
        assert p[0] == seeds[k]
        hops = np.abs(np.diff(np.concatenate([p, [int(targets[k])]])))
        assert np.isin(hops, (1, N, N + 1)).all(), k
    ctx.close()


def test_c2_tile_batch_fields_and_vector_map_full_groups(gpu_ctx_factory):
    """64 plans with V-sized outputs on the 1M mesh: eight full groups of `k_dij_finalize<8, true>`."""
    case = terrain_case(1000, 2)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    ctx.set_dijkstra_engine("tile_batch")
    ctx.set_resident_outputs(True)                                    # the vector map of every plan stays on the device
    m = case.mesh
    rng = np.random.default_rng(29)
    n = 64
    seeds = rng.choice(m.V, n, replace=False).astype(np.uint32)
    targets = rng.choice(m.V, n, replace=False).astype(np.uint32)
    targets[:32] = m.vertex_at(0.9, 0.9)
    b = ctx.plan_dijkstra_batch(seeds, targets, want_fields=True, path_cap=16384)
    assert b["rc"] == 0 and (b["codes"] == 0).all()
    for k in (0, 7, 8, 15, 16, 31, 32, 40, 55, 63):                   # first / last lanes of several groups
        ref = case.om.dijkstra(case.weights, case.costs, int(seeds[k]), int(targets[k]))
        assert ref.code == 0
        assert np.array_equal(b["paths"][k], ref.path), k
        assert np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32)), k
        assert np.array_equal(b["pred"][k], ref.pred), k
        vm = case.om.dijkstra_vector_map(ref.pred)                    # computeVectorMap :189-209
        got = ctx.download_output("vecmap", k)
        assert np.array_equal(got.view(np.uint32), vm.view(np.uint32)), k
    ctx.set_resident_outputs(False)
    ctx.close()


def test_cancel_stops_a_running_tile_batch(gpu_ctx_factory):
    case = terrain_case(1000, 2)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    ctx.set_dijkstra_engine("tile_batch")
    m = case.mesh
    rng = np.random.default_rng(13)
    n = 7168
    seeds = rng.choice(m.V, n, replace=False).astype(np.uint32)
    targets = np.full(n, m.vertex_at(0.5, 0.5), np.uint32)
    kw = dict(goal_dist_offset=float("inf"), want_fields=False, path_cap=4096, want_stats=False)
    full = ctx.plan_dijkstra_batch(seeds, targets, **kw)              # warm-up: builds the streams, allocates the batch
    assert full["rc"] == 0 and (full["codes"] == 0).all()
    t0 = time.perf_counter()
    full = ctx.plan_dijkstra_batch(seeds, targets, **kw)
    t_full = time.perf_counter() - t0
    ref_paths = [full["paths"][k].copy() for k in (0, 100, 7167)]
    out = {}
    th = threading.Thread(target=lambda: out.update(ctx.plan_dijkstra_batch(seeds, targets, **kw)))
    t0 = time.perf_counter()
    th.start()
    time.sleep(min(0.05, 0.25 * t_full))
    ctx.cancel()
    th.join(timeout=120)
    t_cancel = time.perf_counter() - t0
    assert not th.is_alive()
    assert out["rc"] == capi.CANCELED and (out["codes"] == capi.CANCELED).all(), (out["rc"], t_full, t_cancel)
    assert t_cancel < t_full, (t_cancel, t_full)
    st = ctx.stats()                                                  # must not touch the cancelled batch's freed / stale state
    assert st is not None
    again = ctx.plan_dijkstra_batch(seeds, targets, **kw)             # same size: the interrupted buffers are reused
    assert again["rc"] == 0 and (again["codes"] == 0).all()
    for k, p in zip((0, 100, 7167), ref_paths):
        assert np.array_equal(again["paths"][k], p)
    ctx.close()
