"""Where the asynchronous engine stops beating the tile-batch engine on the 1M mesh: ms per batch for both, paths only.
    python tools/gpu_async_crossover.py [grid=1000]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mesh_navigation_amd import capi, meshgen  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mesh = meshgen.terrain(N, 0.1, 21 if N <= 1000 else 4)
w = meshgen.edge_lengths(mesh)
robot = mesh.vertex_at(0.9, 0.9)
out = dict(grid=N)
for n in ((64, 96, 128, 192, 256, 384, 512) if N <= 1000 else (64, 96, 128, 192, 256)):
    goals = np.random.default_rng(5).choice(mesh.V, size=n, replace=False).astype(np.uint32)
    tg = np.full(n, robot, np.uint32)
    row = {}
    sig = None
    for eng in ("async", "tile_batch"):
        ctx = capi.MnavContext(0)
        ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
        ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
        ctx.set_dijkstra_engine(eng)
        if eng == "async":
            ctx.set_option("async_max_batch", 4096)
        ts = []
        for r in range(4):
            t0 = time.perf_counter()
            b = ctx.plan_dijkstra_batch(goals, tg, want_fields=False, path_cap=65536, want_stats=False)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert (b["codes"] == 0).all(), (eng, n, b["codes"][:8])
        s2 = int(sum(int(np.asarray(p, np.uint64).sum()) for p in b["paths"]))
        assert sig is None or sig == s2, (eng, n)
        sig = s2
        row[eng] = round(float(np.median(ts[1:])), 2)
        del ctx
    out[str(n)] = row
print(json.dumps(out))
