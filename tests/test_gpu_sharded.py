"""Config 4 on the GPU: ONE plan on a mesh range-partitioned over `world` ranks.  The multi-GPU run is the driver's
(bench.py --gpus N --config C4); here several device contexts on the one GPU of the test box stand in for the
ranks: same kernels (k_tile_round restricted to the owned tiles, k_shard_pack / k_shard_apply / k_dij_finalize per
rank), the min-allreduce replaced by an elementwise minimum of the ranks' device buffers."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen, sharded
from tests.common import Case, terrain_case

pytestmark = pytest.mark.gpu


def engines(case, world, gpu_ctx_factory, cost_limit=1.0):
    out = []
    for r in range(world):
        ctx = gpu_ctx_factory()
        case.upload(ctx)
        out.append(sharded.GpuShardEngine(ctx, r, world, cost_limit))
    return out


@pytest.mark.parametrize("device_loop", [True, False])             # exchange loop resident on the device (stream events, termination
@pytest.mark.parametrize("world,offset", [(2, 0.3), (4, float("inf")), (3, 0.0), (3, -0.5)])   # words read every 8 exchanges) / host-checked
def test_sharded_plan_c1_bit_exact(gpu_ctx_factory, world, offset, device_loop):
    case = terrain_case(224, 1)
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target, goal_dist_offset=offset)
    eng = engines(case, world, gpu_ctx_factory)
    infos = [e.info for e in eng]
    assert infos[0]["t_lo"] == 0 and infos[-1]["t_hi"] == infos[0]["ntiles"]
    assert all(a["t_hi"] == b["t_lo"] for a, b in zip(infos, infos[1:]))
    res = sharded.plan_virtual_ranks(eng, seed, target, offset, rounds_per_exchange=4, max_exchanges=5000, device_loop=device_loop)
    assert res.code == ref.code == 0 and res.exchanges > 2 and (not device_loop or res.exchanges % 8 == 0)
    assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(res.pred, ref.pred) and np.array_equal(res.path, ref.path)


def test_sharded_plan_with_costs_invalid_and_unreachable(gpu_ctx_factory):
    mesh = meshgen.terrain(96, 0.1, 13)
    rng = np.random.default_rng(3)
    costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)
    invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    invalid[[s, t]] = 0
    costs[[s, t]] = 0
    case = Case(mesh, costs, 1.0, invalid)
    ref = case.om.dijkstra(case.weights, case.costs, s, t, invalid=invalid)
    res = sharded.plan_virtual_ranks(engines(case, 2, gpu_ctx_factory), s, t, max_exchanges=5000)
    assert res.code == ref.code
    assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(res.pred, ref.pred)
    # unreachable: cost limit below every neighbour of the target
    ref2 = case.om.dijkstra(case.weights, case.costs, s, t, invalid=invalid, cost_limit=-1.0)
    res2 = sharded.plan_virtual_ranks(engines(case, 2, gpu_ctx_factory, cost_limit=-1.0), s, t, max_exchanges=5000)
    assert res2.code == ref2.code == sharded.NO_PATH_FOUND
    assert np.array_equal(res2.dist.view(np.uint32), ref2.dist.view(np.uint32))


def test_sharded_plan_1m_two_ranks(gpu_ctx_factory):
    """BASELINE's 1M-vertex mesh cut in two: potential, predecessors and the vertex path of the reference."""
    case = Case(meshgen.terrain(1000, 0.1, 2))
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    res = sharded.plan_virtual_ranks(engines(case, 2, gpu_ctx_factory), seed, target, rounds_per_exchange=8, max_exchanges=5000)
    assert res.code == ref.code == 0
    assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(res.pred, ref.pred) and np.array_equal(res.path, ref.path)


# ---------------------------------------------------------------------------------------------------------------------
# partitioned DATA (north_star: "the mesh is range-partitioned across the 8 GPUs ... allreduce of halo-vertex distances only"):
# every context is created on ITS PART of the mesh only (mnav_shard_setup_partition)
# ---------------------------------------------------------------------------------------------------------------------
def part_engines(case, world, gpu_ctx_factory, cost_limit=1.0):
    owner = sharded.partition_vertices(case.mesh.xyz, world)
    out = []
    for r in range(world):
        part = sharded.extract_part(case.mesh.xyz, case.mesh.edges, owner, r, world)
        ctx = gpu_ctx_factory()
        sharded.PartitionedShardEngine.upload_part(ctx, part, case.costs, case.weights, case.invalid)
        out.append(sharded.PartitionedShardEngine(ctx, part, cost_limit))
    return out


@pytest.mark.parametrize("device_loop", [True, False])
@pytest.mark.parametrize("world,offset", [(2, 0.3), (4, float("inf")), (3, 0.0), (3, -0.5)])
def test_partitioned_plan_c1_bit_exact(gpu_ctx_factory, world, offset, device_loop):
    case = terrain_case(224, 1)
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target, goal_dist_offset=offset)
    eng = part_engines(case, world, gpu_ctx_factory)
    res = sharded.plan_virtual_ranks(eng, seed, target, offset, rounds_per_exchange=4, max_exchanges=5000, device_loop=device_loop)
    assert res.code == ref.code == 0 and res.exchanges > 2
    assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(res.pred, ref.pred) and np.array_equal(res.path, ref.path)
    # a second plan on the same engines, wave source and robot vertex inside ONE part (phantoms everywhere else)
    seed2, target2 = m.vertex_at(0.12, 0.1), m.vertex_at(0.2, 0.25)
    ref2 = case.om.dijkstra(case.weights, case.costs, seed2, target2, goal_dist_offset=offset)
    res2 = sharded.plan_virtual_ranks(eng, seed2, target2, offset, rounds_per_exchange=4, max_exchanges=5000, device_loop=device_loop)
    assert res2.code == ref2.code == 0
    assert np.array_equal(res2.dist.view(np.uint32), ref2.dist.view(np.uint32))
    assert np.array_equal(res2.pred, ref2.pred) and np.array_equal(res2.path, ref2.path)


def test_partitioned_plan_with_costs_invalid_and_unreachable(gpu_ctx_factory):
    mesh = meshgen.terrain(96, 0.1, 13)
    rng = np.random.default_rng(3)
    costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)
    invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    invalid[[s, t]] = 0
    costs[[s, t]] = 0
    case = Case(mesh, costs, 1.0, invalid)
    ref = case.om.dijkstra(case.weights, case.costs, s, t, invalid=invalid)
    res = sharded.plan_virtual_ranks(part_engines(case, 3, gpu_ctx_factory), s, t, max_exchanges=5000)
    assert res.code == ref.code
    assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(res.pred, ref.pred)
    assert np.array_equal(res.path, ref.path)
    ref2 = case.om.dijkstra(case.weights, case.costs, s, t, invalid=invalid, cost_limit=-1.0)
    eng2 = part_engines(case, 2, gpu_ctx_factory, cost_limit=-1.0)
    res2 = sharded.plan_virtual_ranks(eng2, s, t, max_exchanges=5000)
    assert res2.code == ref2.code == sharded.NO_PATH_FOUND
    assert np.array_equal(res2.dist.view(np.uint32), ref2.dist.view(np.uint32))
    assert sharded.plan_virtual_ranks(eng2, s, t, max_exchanges=5000, gather=False).code == sharded.NO_PATH_FOUND   # device walk: same verdict


def test_partitioned_plan_1m_four_ranks_and_device_footprint(gpu_ctx_factory):
    """BASELINE's 1M-vertex mesh in four parts: the reference's potential / predecessors / path, and a rank's device memory
    (mesh tables + per-vertex plan state) stays below 1.2 / world of what the whole mesh takes in one context."""
    case = Case(meshgen.terrain(1000, 0.1, 2))
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    world = 4
    eng = part_engines(case, world, gpu_ctx_factory)
    res = sharded.plan_virtual_ranks(eng, seed, target, rounds_per_exchange=8, max_exchanges=5000)
    assert res.code == ref.code == 0
    assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(res.pred, ref.pred) and np.array_equal(res.path, ref.path)
    # the path again with nothing mesh-sized leaving the devices: every part walks its segments itself (mnav_shard_walk)
    res_w = sharded.plan_virtual_ranks(eng, seed, target, rounds_per_exchange=8, max_exchanges=5000, gather=False)
    assert res_w.code == 0 and res_w.dist is None and np.array_equal(res_w.path, ref.path)
    far = m.vertex_at(0.5, 0.03)                                     # another robot vertex: the path crosses other interfaces
    ref_f = case.om.dijkstra(case.weights, case.costs, seed, far)
    res_f = sharded.plan_virtual_ranks(eng, seed, far, rounds_per_exchange=8, max_exchanges=5000, gather=False)
    assert res_f.code == ref_f.code == 0 and np.array_equal(res_f.path, ref_f.path)
    full = part_engines(case, 1, gpu_ctx_factory)                   # the same upload (no faces, two phantoms) of the whole mesh
    r1 = sharded.plan_virtual_ranks(full, seed, target, rounds_per_exchange=8, max_exchanges=5000)
    assert np.array_equal(r1.path, ref.path) and np.array_equal(r1.pred, ref.pred)
    whole_bytes = full[0].ctx.device_bytes()
    assert whole_bytes > 50 * m.V
    for e in eng:
        assert e.ctx.V <= 1.2 * m.V / world
        assert e.ctx.device_bytes() < 1.2 * whole_bytes / world, (e.ctx.device_bytes(), whole_bytes)


def test_partitioned_plan_negative_offset_on_a_flat_grid_full_of_ties(gpu_ctx_factory):
    """A negative goal_dist_offset stops the expansion AT the robot vertex (dijkstra_mesh_planner.cpp:293-300): among the vertices of
    exactly the robot's potential the vertex id decides who was expanded.  On a flat regular grid there are many of them; a part that
    does not hold the robot vertex compares with the robot's rank among its own ids (mnav_shard_set_goal_tie)."""
    from mesh_navigation_amd import meshgen
    from tests.common import Case
    m = meshgen.flat_grid(96, 0.1)
    case = Case(m)
    seed, target = m.vertex_at(0.15, 0.2), m.vertex_at(0.8, 0.75)
    for offset in (-0.05, -1.0):
        ref = case.om.dijkstra(case.weights, case.costs, seed, target, goal_dist_offset=offset)
        eng = part_engines(case, 3, gpu_ctx_factory)
        res = sharded.plan_virtual_ranks(eng, seed, target, offset, rounds_per_exchange=4, max_exchanges=5000)
        assert res.code == ref.code == 0
        assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(res.pred, ref.pred) and np.array_equal(res.path, ref.path)
