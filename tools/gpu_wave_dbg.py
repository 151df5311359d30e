import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
N = int(sys.argv[1]); B = int(sys.argv[2])
mesh = meshgen.terrain(N, 0.1, 2)
w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
rng = np.random.default_rng(5)
g = rng.choice(mesh.V, B, replace=False).astype(np.uint32)
t = np.full(B, mesh.vertex_at(0.9, 0.9), np.uint32)
for eng in ("persistent", "wave"):
    ctx.set_dijkstra_engine(eng)
    print("run", eng, flush=True)
    t0 = time.perf_counter()
    r = ctx.plan_dijkstra_batch(g, t, goal_dist_offset=0.3, want_fields=True, path_cap=4096)
    print(eng, "rc", r["rc"], "codes", set(r["codes"].tolist()), "ms", (time.perf_counter() - t0) * 1e3, r["stats"]["ms_step_kernels"], flush=True)
    if eng == "persistent": ref = r
print("dist equal", np.array_equal(ref["dist"].view(np.uint32), r["dist"].view(np.uint32)), "paths", all(np.array_equal(a, b) for a, b in zip(ref["paths"], r["paths"])))
