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
    for engine in ("tile_batch", "tiled", "async"):
        ctx.set_dijkstra_engine(engine)
        lazy = ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=offset, cost_limit=0.8, want_fields=False, path_cap=16384)
        assert ctx.device_output(0, 1) == 0                            # a paths-only call leaves no predecessor / potential array behind
        assert ctx.device_output(0, 0) == 0
        for k in sample:
            assert lazy["codes"][k] == refs[k].code, (engine, k)
            assert np.array_equal(lazy["paths"][k], refs[k].path), (engine, k)
        results[engine] = ([int(c) for c in lazy["codes"]], [p.tolist() for p in lazy["paths"]])
    # every engine returns the same paths for ALL plans ...
    for engine in ("tiled", "async"):
        assert results[engine] == results["tile_batch"], engine
    # ... and they equal the paths of the finalize path (V-sized fields asked for) on a subset that fits the host
    ctx.set_dijkstra_engine("tile_batch")
    sub = np.arange(0, 64)
    fin = ctx.plan_dijkstra_batch(seeds[sub], targets[sub], goal_dist_offset=offset, cost_limit=0.8, want_fields=True, path_cap=16384)
    for k in sub:
        assert fin["codes"][k] == results["tile_batch"][0][k]
        assert fin["paths"][k].tolist() == results["tile_batch"][1][k]
    ctx.set_dijkstra_engine("auto")


def test_vector_at_after_a_paths_only_batch_equals_the_vector_map(c2_costs):
    """mnav_vector_at (MeshMap::directionAtPosition, mesh_map.cpp:625-650) after a paths-only tile-batch batch: the three vector-map
    entries (dijkstra :189-209) are derived from the blocked distances on demand (k_tb_vector3) -- the same numbers as sampling the
    vector map that the finalize pass writes when V-sized outputs are asked for, at popped vertices, at tentative ones beyond
    goal_dist, at unreached ones, at the wave source."""
    case, ctx, seeds, targets = c2_costs
    mesh = case.mesh
    sub = np.arange(0, 64)
    ctx.set_dijkstra_engine("tile_batch")
    rng = np.random.default_rng(77)
    faces = rng.choice(mesh.F, 40, replace=False)
    bary = rng.dirichlet(np.ones(3), size=40).astype(np.float32)
    plans = [0, 5, 17, 40, 48, 63]                                    # incl. NO_PATH_FOUND (48: the wave floods everything it can reach)
    ctx.set_resident_outputs(True)
    full = ctx.plan_dijkstra_batch(seeds[sub], targets[sub], goal_dist_offset=0.3, cost_limit=0.8, want_fields=True, path_cap=16384)   # vector maps resident
    want = {}
    for p in plans:
        extra = [np.array([seeds[p]] * 3, np.uint32)]                  # the wave source itself: no vector
        if full["codes"][p] == 0:
            d = full["dist"][p]
            far = np.nonzero(np.isfinite(d) & (d > d[targets[p]] + 0.3))[0]        # tentative values beyond goal_dist keep a predecessor
            if far.size:
                extra.append(np.array([far[0], far[far.size // 2], far[-1]], np.uint32))
        tri = [mesh.faces[f].astype(np.uint32) for f in faces] + extra
        br = [b for b in bary] + [np.array([0.2, 0.3, 0.5], np.float32)] * len(extra)
        want[p] = [(t, b, ctx.vector_at(t, b, slot=int(p))) for t, b in zip(tri, br)]
        vmap = ctx.download_output("vecmap", slot=int(p))
        for t, b, got in want[p][:8]:                                  # the resident map is the finalize pass's: spot-check against its download
            vm = vmap[t]
            has = ~(vm == 0).all(axis=1)
            exp = (vm[has] * b[has, None]).sum(axis=0) if has.any() else None
            assert (got is None) == (exp is None) and (got is None or np.allclose(got, exp, rtol=0, atol=1e-6))
    ctx.set_resident_outputs(False)
    lazy = ctx.plan_dijkstra_batch(seeds[sub], targets[sub], goal_dist_offset=0.3, cost_limit=0.8, want_fields=False, path_cap=16384)
    assert ctx.device_output(0, 4) == 0                                # no vector map anywhere
    assert [int(c) for c in lazy["codes"]] == [int(c) for c in full["codes"]]
    n_vec = 0
    for p in plans:
        for t, b, exp in want[p]:
            got = ctx.vector_at(t, b, slot=int(p))
            assert (got is None) == (exp is None), (p, t)
            if got is not None:
                assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (p, t)   # the same float operations: bit-equal
                n_vec += 1
    assert n_vec > 60
    ctx.set_dijkstra_engine("auto")
