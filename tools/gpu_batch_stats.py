import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mesh_navigation_amd import capi, meshgen
mesh = meshgen.terrain(1000, 0.1, 2); w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0); ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None); ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
robot = mesh.vertex_at(0.9, 0.9); rng = np.random.default_rng(5)
for k in range(3):
    g = rng.choice(mesh.V, 5120, replace=False).astype(np.uint32); t = np.full(5120, robot, np.uint32)
    t0 = time.perf_counter(); r = ctx.plan_dijkstra_batch(g, t, want_fields=False, path_cap=16384); dt = time.perf_counter() - t0
    st = r["stats"]; print(json.dumps({"wall_ms": dt * 1e3, **{k2: (round(v, 2) if isinstance(v, float) else v) for k2, v in st.items() if k2.startswith("ms_")}}), flush=True)
ctx.close()
