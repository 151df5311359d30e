"""Throughput of small / medium Dijkstra batches on the C2 mesh per engine (paths only): where does the tile-batch engine with the
quarter-wave solve overtake the per-plan engines?"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
N = int(os.environ.get("N", "1000"))
mesh = meshgen.terrain(N, 0.1, 2 if N == 1000 else 4)
w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
robot = mesh.vertex_at(0.9, 0.9)
rng = np.random.default_rng(5)
for B in [int(x) for x in os.environ.get("BS", "64,128,256,512,1024").split(",")]:
    g = rng.choice(mesh.V, size=B, replace=False).astype(np.uint32)
    row = {}
    for eng in os.environ.get("ENGINES", "tiled,persistent,tile_batch").split(","):
        ctx.set_dijkstra_engine(eng)
        best = 1e9
        for r in range(3):
            t0 = time.perf_counter()
            b = ctx.plan_dijkstra_batch(g, np.full(B, robot, np.uint32), want_fields=False, path_cap=16384, want_stats=False)
            dt = time.perf_counter() - t0
            assert (b["codes"] == 0).all()
            if r:
                best = min(best, dt)
        row[eng] = round(B / best, 0)
    print(B, row, flush=True)
