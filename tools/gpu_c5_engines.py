"""C5 (64 concurrent goals, 1M mesh): round engine vs one workgroup per plan."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mesh_navigation_amd import capi, meshgen
mesh = meshgen.terrain(1000, 0.1, 2)
w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
robot = mesh.vertex_at(0.9, 0.9)
for nb in (32, 64, 96):
    goals = np.random.default_rng(5).choice(mesh.V, size=nb, replace=False).astype(np.uint32)
    tg = np.full(nb, robot, np.uint32)
    for eng in ("tiled", "persistent"):
        ctx.set_dijkstra_engine(eng)
        ts = []
        for k in range(8):
            t0 = time.perf_counter(); r = ctx.plan_dijkstra_batch(goals, tg, want_fields=False, path_cap=16384); dt = time.perf_counter() - t0
            if k >= 2: ts.append(dt)
        print(json.dumps({"batch": nb, "engine": eng, "ms": float(np.median(ts)) * 1e3, "plans_per_s": nb / float(np.median(ts))}), flush=True)
ctx.close()
