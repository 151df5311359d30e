"""Config 4, partitioned data, the PRODUCTION classes under torch.distributed: two PROCESSES, each with its own context created on
its part of the mesh only (sharded.PartitionedShardEngine on mnav_shard_setup_partition), exchanging the interface distances through
a real all-reduce.  One GPU is all a test box has -- RCCL refuses two ranks on one device --, so the ranks share GPU 0 and the
collective is gloo on host-staged buffers; everything else (the device-resident loop with its stream links, the termination words,
the path walked across the processes on the device) is what `bench.py --config C4 --gpus N` runs.  The gathered potential,
predecessors and path must be the oracle's single-process plan bit for bit (dijkstra_mesh_planner.cpp:287-373)."""
import os
import socket

import numpy as np
import pytest

from mesh_navigation_amd import sharded
from tests.common import terrain_case

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _host_staged_allreduce_min(dist, torch):
    """in-place MIN all-reduce of a device tensor through the host (gloo): `.cpu()` waits for the stream the engine ordered its
    kernels on, the copy back is ordered on it again"""
    def f(x):
        if isinstance(x, np.ndarray):
            t = torch.from_numpy(x)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return
        h = x.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN)
        x.copy_(h)
    return f


def _worker(rank, world, port, seed, target, offset, device_loop, q):
    import torch
    import torch.distributed as dist
    from mesh_navigation_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        case = terrain_case(224, 1)
        owner = sharded.partition_vertices(case.mesh.xyz, world)
        part = sharded.extract_part(case.mesh.xyz, case.mesh.edges, owner, rank, world)
        ctx = capi.MnavContext(0)
        sharded.PartitionedShardEngine.upload_part(ctx, part, case.costs, case.weights)
        eng = sharded.PartitionedShardEngine(ctx, part)
        red = _host_staged_allreduce_min(dist, torch)
        out = []
        for (s, t) in ((seed, target), (case.mesh.vertex_at(0.12, 0.1), case.mesh.vertex_at(0.2, 0.25))):   # the second plan: both ends inside one part
            res = sharded.run_sharded_plan(eng, red, s, t, offset, rounds_per_exchange=4, max_exchanges=5000, device_loop=device_loop)
            out.append((res.code, res.dist.tobytes(), res.pred.tobytes(), res.path.tolist(), res.exchanges))
        if rank == world - 1:
            q.put(out)
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("device_loop,offset", [(True, 0.3), (False, -0.5)])
def test_partitioned_plan_over_two_processes_on_one_gpu(device_loop, offset):
    import torch.multiprocessing as mp
    case = terrain_case(224, 1)
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, target, offset, device_loop, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = q.get(timeout=480)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    for (s, t), (code, dbytes, pbytes, path, exchanges) in zip(((seed, target), (m.vertex_at(0.12, 0.1), m.vertex_at(0.2, 0.25))), out):
        ref = case.om.dijkstra(case.weights, case.costs, s, t, goal_dist_offset=offset)
        assert code == ref.code == 0
        assert np.array_equal(np.frombuffer(dbytes, np.float32).view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(np.frombuffer(pbytes, np.uint32), ref.pred) and path == ref.path.tolist()
    assert out[0][4] > 2
