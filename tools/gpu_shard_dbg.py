import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mesh_navigation_amd import capi, meshgen, sharded
from tests.common import terrain_case
case = terrain_case(224, 1)
m = case.mesh
seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
ref = case.om.dijkstra(case.weights, case.costs, seed, target, goal_dist_offset=0.3)
for world in (2,):
    for dl in (False, True):
        owner = sharded.partition_vertices(m.xyz, world)
        eng = []
        for r in range(world):
            part = sharded.extract_part(m.xyz, m.edges, owner, r, world)
            ctx = capi.MnavContext(0)
            sharded.PartitionedShardEngine.upload_part(ctx, part, case.costs, case.weights, None)
            eng.append(sharded.PartitionedShardEngine(ctx, part))
        res = sharded.plan_virtual_ranks(eng, seed, target, 0.3, rounds_per_exchange=4, max_exchanges=5000, device_loop=dl)
        print("world", world, "device_loop", dl, "code", res.code, "exch", res.exchanges, [(e.status, e.error) for e in eng], flush=True)
        if res.code == 0:
            print(" dist eq", np.array_equal(res.dist.view(np.uint32), ref.dist.view(np.uint32)), "pred eq", np.array_equal(res.pred, ref.pred), "path eq", np.array_equal(res.path, ref.path))
        else:
            # look at the local states
            for e in eng:
                d = e.dist.cpu().numpy(); n0 = e.part.gid.shape[0]
                g = ref.dist[e.part.gid]
                fin = np.isfinite(g) & (g <= ref.dist[target] + 0.3)
                own = e.part.owned[:n0].astype(bool)
                bad = fin & (d[:n0].view(np.uint32) != g.view(np.uint32))
                print("  rank", e.part.rank, "n0", n0, "popped", int(fin.sum()), "differ among popped: owned", int((bad & own).sum()), "halo", int((bad & ~own).sum()),
                      "lower than ref", int((fin & (d[:n0] < g)).sum()))
