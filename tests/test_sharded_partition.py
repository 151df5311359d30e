"""Config 4 with PARTITIONED DATA (north_star: "the mesh is range-partitioned across the 8 GPUs ... allreduce of halo-vertex
distances only"): every process only ever sees its part of the mesh.  Host-side partition invariants, then the production
loop (sharded.run_sharded_plan / plan_virtual_ranks) over world_size-2 / -3 gloo processes with the CPU model of a part
(tests/shard_model.py::PartModelEngine): gathered potential, predecessors and the path walked ACROSS the processes must be
bit-equal to the oracle's single-process plan (dijkstra_mesh_planner.cpp:287-373)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from mesh_navigation_amd import meshgen, sharded
from tests.common import Case
from tests.shard_model import PartModelEngine


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    mesh = meshgen.terrain(40, 0.1, 17)
    rng = np.random.default_rng(4)
    costs = rng.uniform(0, 0.6, mesh.V).astype(np.float32)
    costs[rng.choice(mesh.V, 40, replace=False)] = 2.0             # above cost_limit: never act as sources
    return Case(mesh, costs, 0.5)


def _engine(case, rank, world, asynchronous=False):
    owner = sharded.partition_vertices(case.mesh.xyz, world)
    part = sharded.extract_part(case.mesh.xyz, case.mesh.edges, owner, rank, world)
    return PartModelEngine(part, part.local_edge_values(case.weights), part.local_costs(case.costs), asynchronous=asynchronous)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_partition_invariants(world):
    mesh = meshgen.terrain(100, 0.1, 3)
    owner = sharded.partition_vertices(mesh.xyz, world)
    counts = np.bincount(owner, minlength=world)
    assert counts.sum() == mesh.V and counts.min() >= mesh.V // world - 1 and counts.max() <= mesh.V // world + world
    e = mesh.edges.astype(np.int64)
    seen_edges = np.zeros(mesh.E, int)
    exch0 = None
    for r in range(world):
        p = sharded.extract_part(mesh.xyz, mesh.edges, owner, r, world)
        assert np.all(np.diff(p.gid.astype(np.int64)) > 0)                            # ascending global ids: ties break as on the whole mesh
        mine = p.owned[:p.gid.shape[0]].astype(bool)
        assert np.array_equal(p.gid[mine], np.nonzero(owner == r)[0])
        # every edge with an owned endpoint, in global order, and nothing else
        keep = (owner[e[:, 0]] == r) | (owner[e[:, 1]] == r)
        assert np.array_equal(p.edge_gid, np.nonzero(keep)[0])
        assert np.array_equal(p.gid[p.edges.astype(np.int64)], e[keep])
        seen_edges[p.edge_gid] += 1
        # the halo is exactly the 1-ring of the owned vertices
        nb = np.unique(np.concatenate([e[keep, 0], e[keep, 1]]))
        assert np.array_equal(np.union1d(nb, np.nonzero(owner == r)[0]), p.gid)
        # the interface list is the same everywhere; a part holds every interface vertex it owns or touches
        if exch0 is None:
            exch0 = p.exchange_global
        assert np.array_equal(exch0, p.exchange_global)
        held = p.exchange_vertex != sharded.NONE
        assert np.array_equal(p.gid[p.exchange_vertex[held].astype(np.int64)], p.exchange_global[held])
        assert set(p.gid[~mine].tolist()) <= set(p.exchange_global[held].tolist())   # every halo copy is exchanged
        assert p.n_local == p.gid.shape[0] + 2 and p.xyz.shape == (p.n_local, 3)
        assert p.n_local <= 1.2 * mesh.V / world + 2 * np.sqrt(mesh.V)                # the part, not the mesh
    assert seen_edges.min() >= 1 and seen_edges.max() <= 2                            # cut edges live on both sides


def _worker(rank, world, port, seed, target, offset, rpe, q, check_every):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = _case()
    eng = _engine(case, rank, world, asynchronous=check_every > 0)
    res = sharded.run_sharded_plan(eng, sharded.torch_allreduce_min(dist), seed, target, offset, rounds_per_exchange=rpe,
                                   check_every=max(1, check_every))
    if rank == world - 1:                                             # any rank holds the gathered result
        q.put((res.code, res.dist.tobytes(), res.pred.tobytes(), res.path.tolist(), res.exchanges))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,offset,rpe,check_every", [(2, 0.3, 4, 0), (3, float("inf"), 2, 0), (2, 0.0, 16, 0), (3, 0.3, 2, 4), (3, -0.4, 4, 0)])
def test_partitioned_plan_matches_oracle_gloo(world, offset, rpe, check_every):
    case = _case()
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.15), m.vertex_at(0.9, 0.85)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target, goal_dist_offset=offset)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, target, offset, rpe, q, check_every)) for r in range(world)]
    for p in procs:
        p.start()
    code, dbytes, pbytes, path, exchanges = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = np.frombuffer(dbytes, np.float32)
    pr = np.frombuffer(pbytes, np.uint32)
    assert code == ref.code == 0 and exchanges > 2
    assert np.array_equal(d.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(pr, ref.pred) and path == ref.path.tolist()


@pytest.mark.parametrize("world", [2, 4, 7])
def test_partitioned_virtual_ranks_and_unreachable_target(world):
    case = _case()
    m = case.mesh
    seed, target = m.vertex_at(0.2, 0.2), m.vertex_at(0.8, 0.7)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    engines = [_engine(case, r, world) for r in range(world)]
    res = sharded.plan_virtual_ranks(engines, seed, target, rounds_per_exchange=3)
    assert res.code == ref.code == 0
    assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(res.pred, ref.pred) and np.array_equal(res.path, ref.path)
    # a wall of vertices above the cost limit separates the target: NO_PATH_FOUND, like the reference (dijkstra :358)
    costs = case.costs.copy()
    n = int(np.sqrt(m.V))
    ids = np.arange(m.V).reshape(n, n)
    costs[ids[:, n // 2]] = 5.0
    costs[ids[:, n // 2 + 1]] = 5.0
    case2 = Case(m, costs, 0.5)
    ref2 = case2.om.dijkstra(case2.weights, case2.costs, seed, target)
    engines = [_engine(case2, r, world) for r in range(world)]
    res2 = sharded.plan_virtual_ranks(engines, seed, target, rounds_per_exchange=3)
    assert res2.code == ref2.code == sharded.NO_PATH_FOUND


@pytest.mark.parametrize("world", [3, 5])
def test_path_segments_walked_by_the_parts_themselves(world):
    """gather=False: nothing mesh-sized is assembled -- every part walks its own path segments (the contract of mnav_shard_walk,
    here the CPU model's walker) and publishes them; short segments (cap 7) force many hand-overs, also inside one part."""
    case = _case()
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.15), m.vertex_at(0.9, 0.85)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    engines = [_engine(case, r, world) for r in range(world)]
    res = sharded.plan_virtual_ranks(engines, seed, target, rounds_per_exchange=3, gather=False)
    assert res.code == ref.code == 0 and res.dist is None and res.pred is None
    assert np.array_equal(res.path, ref.path)
    walkers = [e.walker() for e in engines]
    code, path = sharded._walk_segments(
        lambda cur, first: np.minimum.reduce([sharded._segment_of(e.part, w, cur, seed, first, 7) for e, w in zip(engines, walkers)]),
        seed, target, m.V)
    assert code == 0 and np.array_equal(path, ref.path)
    # an unreachable robot vertex: the owner of the target reports it in the first segment
    costs = case.costs.copy()
    n = int(np.sqrt(m.V))
    ids = np.arange(m.V).reshape(n, n)
    costs[ids[:, n // 2]] = 5.0
    costs[ids[:, n // 2 + 1]] = 5.0
    case2 = Case(m, costs, 0.5)
    engines = [_engine(case2, r, world) for r in range(world)]
    res2 = sharded.plan_virtual_ranks(engines, m.vertex_at(0.2, 0.2), m.vertex_at(0.8, 0.7), rounds_per_exchange=3, gather=False)
    assert res2.code == sharded.NO_PATH_FOUND and res2.path.size == 0


def _worker_paths_only(rank, world, port, seed, target, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _engine(_case(), rank, world, asynchronous=True)
    res = sharded.run_sharded_plan(eng, sharded.torch_allreduce_min(dist), seed, target, 0.3, rounds_per_exchange=4, check_every=3, gather=False)
    q.put((rank, res.code, res.dist is None, res.path.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_partitioned_plan_paths_only_over_gloo():
    """the production call of bench.py --config C4 --gpus N: gather=False, the path walked across the processes through real
    int64 min-allreduces -- every rank ends up with the reference's path, nothing mesh-sized is exchanged"""
    case = _case()
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.15), m.vertex_at(0.9, 0.85)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_paths_only, args=(r, world, port, seed, target, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(g[0] for g in got) == list(range(world))
    for _, code, no_fields, path in got:
        assert code == ref.code == 0 and no_fields and path == ref.path.tolist()
