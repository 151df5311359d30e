"""GPU: the tile-batch SSSP engine (mnav_tb.h: one wave per (tile, <= 64 plans), one plan per lane) through the C ABI
against the CPU oracle -- vertex-index paths identical, popped potential (dist <= goal_dist) bit-exact -- and against
the tile rounds on a 1M-vertex batch (dijkstra_mesh_planner.cpp:287-348, :358-373)."""
import os

import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from tests.common import Case, terrain_case

pytestmark = pytest.mark.gpu


def popped(ref_full, t, offset):
    dt = ref_full[t]
    gd = np.float32(np.float64(dt) + offset) if np.isfinite(dt) else np.float32(np.inf)
    return ref_full <= gd


def check_against_oracle(case, ctx, b, seeds, targets, sample, offset=0.3, cost_limit=1.0):
    for k in sample:
        ref = case.om.dijkstra(case.weights, case.costs, int(seeds[k]), int(targets[k]), goal_dist_offset=offset, cost_limit=cost_limit,
                               invalid=case.invalid)
        assert b["codes"][k] == ref.code, (k, b["codes"][k], ref.code)
        assert np.array_equal(b["paths"][k], ref.path), k
        if ref.code == 0 and seeds[k] != targets[k]:
            full = case.om.dijkstra(case.weights, case.costs, int(seeds[k]), int(targets[k]), goal_dist_offset=np.inf, cost_limit=cost_limit,
                                    invalid=case.invalid).dist
            pot = ctx.download_output("popped", k)
            m = popped(full, int(targets[k]), offset)
            assert np.array_equal(pot[m].view(np.uint32), full[m].view(np.uint32)), k
            assert np.isinf(pot[~m]).all()


@pytest.mark.parametrize("tile", ["64", "128"])
def test_c1_batch_paths_and_popped_potential(gpu_ctx_factory, tile):
    case = terrain_case(224, 1)
    ctx = gpu_ctx_factory()
    ctx.set_option("tb_tile", int(tile))           # read when the engine builds its streams (first batch)
    case.upload(ctx)
    ctx.set_dijkstra_engine("tile_batch")
    m = case.mesh
    rng = np.random.default_rng(11)
    n = 300
    seeds = rng.choice(m.V, n, replace=False).astype(np.uint32)
    targets = np.full(n, m.vertex_at(0.9, 0.9), np.uint32)
    targets[:40] = rng.choice(m.V, 40, replace=False)                 # not only the common robot vertex
    seeds[7] = targets[7]                                             # seed == target inside the batch
    seeds[9] = seeds[8]                                               # duplicate wave source
    for offset in (0.3, 0.0, float("inf")):
        b = ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=offset, want_fields=False)
        assert b["stats"]["n_plans"] == n - 1
        check_against_oracle(case, ctx, b, seeds, targets, list(range(0, n, 23)) + [7, 8, 9], offset=offset)
    ctx.close()


def test_fields_through_the_finalize_pass(gpu_ctx_factory):
    """Calls that want the V-sized outputs: blocked distances -> vertex order -> k_dij_finalize; potential (incl. the
    tentative values beyond goal_dist), predecessors and paths bit-equal to the oracle."""
    case = terrain_case(224, 1)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    ctx.set_dijkstra_engine("tile_batch")
    m = case.mesh
    rng = np.random.default_rng(23)
    n = 40
    seeds = rng.choice(m.V, n, replace=False).astype(np.uint32)
    targets = rng.choice(m.V, n, replace=False).astype(np.uint32)
    b = ctx.plan_dijkstra_batch(seeds, targets, want_fields=True)
    for k in range(n):
        ref = case.om.dijkstra(case.weights, case.costs, int(seeds[k]), int(targets[k]))
        assert b["codes"][k] == ref.code
        assert np.array_equal(b["paths"][k], ref.path)
        assert np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(b["pred"][k], ref.pred)
    ctx.close()


def test_costs_limit_invalid_unreachable(gpu_ctx_factory):
    mesh = meshgen.terrain(160, 0.1, 11)
    rng = np.random.default_rng(5)
    costs = rng.uniform(0.0, 1.4, mesh.V).astype(np.float32)
    inv = (rng.uniform(size=mesh.V) < 0.04).astype(np.uint8)
    case = Case(mesh, costs, edge_cost_factor=1.0, invalid=inv)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    ctx.set_dijkstra_engine("tile_batch")
    n = 260
    seeds = rng.choice(mesh.V, n, replace=False).astype(np.uint32)    # some sources / targets are invalid or over the limit
    targets = rng.choice(mesh.V, n, replace=False).astype(np.uint32)
    b = ctx.plan_dijkstra_batch(seeds, targets, cost_limit=0.8, want_fields=False)
    codes = set(int(c) for c in b["codes"])
    assert 0 in codes and 54 in codes                                 # reachable and NO_PATH_FOUND plans in one batch
    check_against_oracle(case, ctx, b, seeds, targets, range(0, n, 7), cost_limit=0.8)
    ctx.close()


def test_c2_batch_equals_oracle_and_tile_rounds(gpu_ctx_factory):
    """1M vertices, 512 plans: every path equals the tile rounds', a sample equals the oracle's."""
    case = terrain_case(1000, 2)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    m = case.mesh
    rng = np.random.default_rng(17)
    n = 512
    seeds = rng.choice(m.V, n, replace=False).astype(np.uint32)
    targets = np.full(n, m.vertex_at(0.9, 0.9), np.uint32)
    ctx.set_dijkstra_engine("tile_batch")
    b = ctx.plan_dijkstra_batch(seeds, targets, want_fields=False, path_cap=16384)
    assert (b["codes"] == 0).all()
    st = b["stats"]
    check_against_oracle(case, ctx, b, seeds, targets, [0, 101, 255, 511])
    ctx.set_dijkstra_engine("tiled")
    p = ctx.plan_dijkstra_batch(seeds, targets, want_fields=False, path_cap=16384)
    assert (p["codes"] == 0).all()
    for k in range(n):
        assert np.array_equal(b["paths"][k], p["paths"][k]), k
    assert st["settled"] == p["stats"]["settled"]                     # popped vertices, counted by both engines
    ctx.close()


@pytest.mark.parametrize("kernel", [0, 1])
def test_both_solve_kernels_of_the_tile_batch_engine(gpu_ctx_factory, kernel):
    """The engine solves its tiles with k_tb_solve_q (16 plans per quarter of a wave, distances in LDS) or with k_tbv_solve (one wave per
    tile, <= 64 plans, distances and ghosts in a window of VGPRs addressed through the VGPR index mode, mnav_tbv.h) -- `auto` picks by
    the plans a tile sees per iteration, option "tb_kernel" forces one.  Both against the oracle on the same batch: paths, popped
    potential, the V-sized outputs through the finalize pass; costs, a cost limit, invalid vertices and unreachable targets in the
    batch; and mnav_last_engine says which one ran."""
    mesh = meshgen.terrain(160, 0.1, 6)
    rng = np.random.default_rng(17)
    costs = rng.uniform(0.0, 1.3, mesh.V).astype(np.float32)
    inv = (rng.uniform(size=mesh.V) < 0.03).astype(np.uint8)
    case = Case(mesh, costs, edge_cost_factor=1.0, invalid=inv)
    ctx = gpu_ctx_factory()
    ctx.set_option("tb_kernel", kernel)
    case.upload(ctx)
    ctx.set_dijkstra_engine("tile_batch")
    ok = np.flatnonzero((inv == 0) & (costs <= 0.9))
    n = 200
    seeds = rng.choice(ok, n, replace=False).astype(np.uint32)
    targets = rng.choice(ok, n, replace=False).astype(np.uint32)
    targets[:120] = targets[0]
    for offset in (0.3, 0.0):
        b = ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=offset, cost_limit=0.9, want_fields=False)
        assert ("k_tbv_solve" in ctx.last_engine()) == (kernel == 1), ctx.last_engine()
        check_against_oracle(case, ctx, b, seeds, targets, list(range(0, n, 17)), offset=offset, cost_limit=0.9)
    bf = ctx.plan_dijkstra_batch(seeds[:100], targets[:100], goal_dist_offset=0.3, cost_limit=0.9, want_fields=True)
    for k in (0, 33, 99):
        ref = case.om.dijkstra(case.weights, case.costs, int(seeds[k]), int(targets[k]), goal_dist_offset=0.3, cost_limit=0.9, invalid=case.invalid)
        assert bf["codes"][k] == ref.code
        assert np.array_equal(bf["dist"][k].view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(bf["pred"][k], ref.pred)
    ctx.close()
