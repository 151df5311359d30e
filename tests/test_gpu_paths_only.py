"""GPU: the PATHS-ONLY batch API at C2 scale -- the calls the batch bench and mbf's getPath make (no dist / pred / vector
map asked for, so no finalize pass: predecessors are derived along the returned path, k_path_lazy / k_tb_path) -- with
cost-limit-blocked, invalid and unreachable targets and goal_dist_offset in {0, 0.3, inf}: vertex paths equal to the
oracle's (dijkstra_mesh_planner.cpp:287-373) and to the finalize path's, on every engine."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from tests.common import Case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2_costs(gpu_ctx_factory):
    mesh = meshgen.terrain(1000, 0.1, 2)
    rng = np.random.default_rng(31)
    costs = rng.uniform(0.0, 0.6, mesh.V).astype(np.float32)
    N = 1000
    # a wall of over-limit vertices with one gap, a block of invalid vertices, an enclosed pocket (unreachable targets)
    wall = np.arange(100, 900) * N + 400
    costs[wall] = 5.0
    inv = np.zeros(mesh.V, np.uint8)
    jj, ii = np.meshgrid(np.arange(600, 640), np.arange(200, 260), indexing="ij")
    inv[(jj * N + ii).ravel()] = 1
    ring = []
    for k in range(700, 741):
        ring += [700 * N + k, 740 * N + k, k * N + 700, k * N + 740]
    costs[np.array(ring)] = 9.0                                       # closed ring: the pocket inside cannot be entered
    case = Case(mesh, costs, edge_cost_factor=1.0, invalid=inv)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    n = 288
    seeds = rng.choice(mesh.V, n, replace=False).astype(np.uint32)
    targets = np.full(n, mesh.vertex_at(0.9, 0.9), np.uint32)
    targets[:48] = rng.choice(mesh.V, 48, replace=False)
    targets[48] = 720 * N + 720                                        # inside the pocket: NO_PATH_FOUND
    targets[49] = 620 * N + 230                                        # invalid target vertex
    seeds[50] = 620 * N + 231                                          # invalid wave source
    targets[51] = wall[10]                                             # over the cost limit
    seeds[52] = wall[20]
    seeds[53] = targets[53]                                            # seed == target: SUCCESS, empty path
    return case, ctx, seeds, targets


@pytest.mark.parametrize("offset", [0.3, 0.0, float("inf")])
def test_paths_only_batches_equal_the_oracle_and_the_finalize_path(c2_costs, offset):
    case, ctx, seeds, targets = c2_costs
    n = len(seeds)
    sample = list(range(0, n, 12)) + [48, 49, 50, 51, 52, 53]
    refs = {k: case.om.dijkstra(case.weights, case.costs, int(seeds[k]), int(targets[k]), goal_dist_offset=offset, cost_limit=0.8,
                                invalid=case.invalid) for k in sample}
    assert {int(r.code) for r in refs.values()} >= {0, 54}
    results = {}
    for engine in ("tile_batch", "persistent", "tiled"):
        ctx.set_dijkstra_engine(engine)
        lazy = ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=offset, cost_limit=0.8, want_fields=False, path_cap=16384)
        assert ctx.device_output(0, 1) == 0                            # a paths-only call leaves no predecessor / potential array behind
        assert ctx.device_output(0, 0) == 0
        for k in sample:
            assert lazy["codes"][k] == refs[k].code, (engine, k)
            assert np.array_equal(lazy["paths"][k], refs[k].path), (engine, k)
        results[engine] = ([int(c) for c in lazy["codes"]], [p.tolist() for p in lazy["paths"]])
    # every engine returns the same paths for ALL plans ...
    for engine in ("persistent", "tiled"):
        assert results[engine] == results["tile_batch"], engine
    # ... and they equal the paths of the finalize path (V-sized fields asked for) on a subset that fits the host
    ctx.set_dijkstra_engine("tile_batch")
    sub = np.arange(0, 64)
    fin = ctx.plan_dijkstra_batch(seeds[sub], targets[sub], goal_dist_offset=offset, cost_limit=0.8, want_fields=True, path_cap=16384)
    for k in sub:
        assert fin["codes"][k] == results["tile_batch"][0][k]
        assert fin["paths"][k].tolist() == results["tile_batch"][1][k]
    ctx.set_dijkstra_engine("auto")
