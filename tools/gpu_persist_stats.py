import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
mesh = meshgen.terrain(1000, 0.1, 2)
w = meshgen.edge_lengths(mesh); costs = np.zeros(mesh.V, np.float32)
ctx = capi.MnavContext(0); ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None); ctx.upload_costs(costs, w)
rng = np.random.default_rng(5); B = int(os.environ.get("B", "256"))
goals = rng.choice(mesh.V, size=B, replace=False).astype(np.uint32)
t = np.full(B, mesh.vertex_at(0.9, 0.9), np.uint32)
ctx.set_dijkstra_engine("persistent")
for _ in range(2):
    r = ctx.plan_dijkstra_batch(goals, t, want_fields=False, path_cap=8192)
st = r["stats"]
print("tile", os.environ.get("MNAV_TILE_SIZE"), "B", B, "acts/plan", st["evals"] / B, "sweeps/plan(max over plans)", st["bands"], "ms kernel", st["ms_step_kernels"], "ms prop", st["ms_propagation"], "ms total", st["ms_total"], "settled/plan", st["settled"] / B)
