"""The N>1 path on CPU: world_size-2 gloo processes run the goal sharding / result gathering of
mesh_navigation_amd.multi (with the CPU *checker* standing in for the device planner, as the
checker, so the gathered results can be compared with a single-process run)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mesh_navigation_amd import meshgen, multi
from tests.common import Case


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, goals, target, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = Case(meshgen.terrain(24, 0.1, 31))

    def plan_batch(g, t):       # same return shape as MnavContext.plan_dijkstra_batch
        rs = [case.om.dijkstra(case.weights, case.costs, int(a), int(b)) for a, b in zip(g, t)]
        return dict(codes=np.array([r.code for r in rs], np.uint32), paths=[r.path for r in rs])

    targets = np.full(len(goals), target, np.uint32)
    local, gathered = multi.plan_goals_sharded(plan_batch, goals, targets, rank, world, dist, gather_to=0)
    total, tmax = multi.aggregate_throughput(len(local["codes"]), 0.5 + rank, dist)
    if rank == 0:
        q.put((gathered["codes"].tolist(), [p.tolist() for p in gathered["paths"]], total, tmax))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            ranges = [multi.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(180)
def test_two_rank_goal_sharding_matches_single_process():
    case = Case(meshgen.terrain(24, 0.1, 31))
    rng = np.random.default_rng(7)
    goals = rng.choice(case.mesh.V, size=9, replace=False).astype(np.uint32)
    target = case.mesh.vertex_at(0.9, 0.9)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, goals, target, q)) for r in range(2)]
    for p in procs:
        p.start()
    codes, paths, total, tmax = q.get(timeout=150)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total == 9 and tmax == 1.5                      # sum of plans, max over ranks
    for k, g in enumerate(goals):
        ref = case.om.dijkstra(case.weights, case.costs, int(g), int(target))
        assert codes[k] == ref.code and paths[k] == ref.path.tolist()
