"""Config 4 on CPU: ONE plan on a range-partitioned mesh, world_size-2 (and 3) gloo processes.  Every process runs
the production loop (mesh_navigation_amd.sharded.run_sharded_plan) with real torch.distributed min-allreduces;
the device engine is replaced by its CPU model (tests/shard_model.py).  The gathered potential, predecessors and
vertex path must be bit-equal to the oracle's single-process plan."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from mesh_navigation_amd import meshgen, sharded
from tests.common import Case
from tests.shard_model import AsyncModelShardEngine, ModelShardEngine


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


def _worker(rank, world, port, seed, target, offset, rpe, q, check_every=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = _case()
    # check_every > 0: the device-resident loop (termination words read back once per block of exchanges)
    eng = (AsyncModelShardEngine if check_every else ModelShardEngine)(case.mesh, case.weights, case.costs, rank, world)
    res = sharded.run_sharded_plan(eng, sharded.torch_allreduce_min(dist), seed, target, offset, rounds_per_exchange=rpe,
                                   check_every=max(1, check_every))
    if rank == 0:
        q.put((res.code, res.dist.tobytes(), res.pred.tobytes(), res.path.tolist(), res.exchanges))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,offset,rpe,check_every", [(2, 0.3, 4, 0), (3, float("inf"), 2, 0), (2, 0.0, 16, 0), (2, 0.3, 4, 5), (3, 0.3, 2, 8)])
def test_sharded_single_plan_matches_oracle(world, offset, rpe, check_every):
    case = _case()
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.15), m.vertex_at(0.9, 0.85)       # the path crosses every strip
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
    if check_every:
        assert exchanges % check_every == 0                          # whole blocks: the loop only looks at the words between them
    assert np.array_equal(d.view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(pr, ref.pred) and path == ref.path.tolist()


def test_virtual_ranks_in_one_process_and_unreachable_target():
    """the lock-step variant of the loop (what the GPU test uses with several contexts on one GPU)"""
    case = _case()
    m = case.mesh
    seed, target = m.vertex_at(0.2, 0.2), m.vertex_at(0.8, 0.7)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    for cls, every in ((ModelShardEngine, 1), (AsyncModelShardEngine, 6)):      # host-checked loop / device-resident loop
        engines = [cls(m, case.weights, case.costs, r, 4) for r in range(4)]
        res = sharded.plan_virtual_ranks(engines, seed, target, rounds_per_exchange=3, check_every=every)
        assert res.code == ref.code == 0
        assert np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(res.pred, ref.pred)
        assert np.array_equal(res.path, ref.path)
    # a target nobody reaches: invalid ring around it
    inv = np.zeros(m.V, np.uint8)
    ring = np.unique(m.edges[(m.edges == target).any(1)].ravel())
    inv[ring] = 1
    inv[target] = 0
    case2 = Case(m, case.costs, 0.5, inv)
    ref2 = case2.om.dijkstra(case2.weights, case2.costs, seed, target, invalid=inv)
    engines = [ModelShardEngine(m, case2.weights, case2.costs, r, 2, invalid=inv) for r in range(2)]
    res2 = sharded.plan_virtual_ranks(engines, seed, target)
    assert res2.code == ref2.code == sharded.NO_PATH_FOUND
    assert np.array_equal(res2.dist.view(np.uint32), ref2.dist.view(np.uint32))


def test_rank_local_cancel_ends_the_plan_on_all_virtual_ranks():
    """a status raised on ONE rank (cancel = 1, error = 2) travels with the termination reduce: every rank returns the same
    code in the same exchange (no rank is left waiting in a collective)"""
    from mesh_navigation_amd import meshgen, sharded
    from tests.shard_model import ModelShardEngine
    from tests.common import Case
    case = Case(meshgen.terrain(48, 0.1, 5))
    engines = [ModelShardEngine(case.mesh, case.weights, case.costs, r, 3) for r in range(3)]
    orig = engines[1].rounds
    calls = {"n": 0}

    def cancelling_rounds(r):
        calls["n"] += 1
        if calls["n"] == 3:
            engines[1].status = 1                                       # what GpuShardEngine does when mnav_shard_rounds returns CANCELED
        return orig(r)

    engines[1].rounds = cancelling_rounds
    res = sharded.plan_virtual_ranks(engines, case.mesh.vertex_at(0.1, 0.1), case.mesh.vertex_at(0.9, 0.9), rounds_per_exchange=2)
    assert res.code == sharded.CANCELED and res.exchanges == 3


def test_rank_local_cancel_in_the_device_resident_loop():
    """the same with the loop that only looks at the termination words once per block of exchanges: the status rides on the
    reduced words of every exchange (MIN over -status keeps it), so the block that contains the cancel ends the plan on
    every rank -- at the block's end, never later"""
    case = Case(meshgen.terrain(48, 0.1, 5))
    engines = [AsyncModelShardEngine(case.mesh, case.weights, case.costs, r, 3) for r in range(3)]
    orig = engines[2].rounds
    calls = {"n": 0}

    def cancelling_rounds(r):
        calls["n"] += 1
        if calls["n"] == 6:
            engines[2].status = 1
        return orig(r)

    engines[2].rounds = cancelling_rounds
    res = sharded.plan_virtual_ranks(engines, case.mesh.vertex_at(0.1, 0.1), case.mesh.vertex_at(0.9, 0.9), rounds_per_exchange=2, check_every=4)
    assert res.code == sharded.CANCELED and res.exchanges == 8          # cancel in exchange 6, seen at the end of the block 5..8
    # a failure (status 2) wins over a cancel (status 1) when both happen in one block
    engines = [AsyncModelShardEngine(case.mesh, case.weights, case.costs, r, 2) for r in range(2)]
    o0, o1 = engines[0].rounds, engines[1].rounds
    n = {"a": 0, "b": 0}

    def r0(r):
        n["a"] += 1
        if n["a"] == 2:
            engines[0].status = 1
        return o0(r)

    def r1(r):
        n["b"] += 1
        if n["b"] == 3:
            engines[1].status = 2
        return o1(r)

    engines[0].rounds, engines[1].rounds = r0, r1
    res = sharded.plan_virtual_ranks(engines, case.mesh.vertex_at(0.1, 0.1), case.mesh.vertex_at(0.9, 0.9), rounds_per_exchange=2, check_every=4)
    assert res.code == sharded.INTERNAL_ERROR and res.exchanges == 4
