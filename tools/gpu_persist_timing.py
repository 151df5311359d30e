import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from mesh_navigation_amd import build as B
B.LIB = os.path.join(os.path.dirname(B.LIB), "libmnav_timing.so")
mesh = meshgen.terrain(1000, 0.1, 2)
w = meshgen.edge_lengths(mesh); costs = np.zeros(mesh.V, np.float32)
ctx = capi.MnavContext(0); L = ctx._L
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None); ctx.upload_costs(costs, w)
rng = np.random.default_rng(5); Bn = int(os.environ.get("B", "1024"))
goals = rng.choice(mesh.V, size=Bn, replace=False).astype(np.uint32)
t = np.full(Bn, mesh.vertex_at(0.9, 0.9), np.uint32)
ctx.set_dijkstra_engine("persistent")
buf = np.zeros((4096, 8), np.uint64)
for _ in range(2):
    r = ctx.plan_dijkstra_batch(goals, t, want_fields=False, path_cap=8192)
    n = L.mnav_debug_tile_timing(buf.ctypes.data_as(C.c_void_p), 4096)
tt = buf[:n].astype(np.int64)
d = np.diff(tt[:, :6], axis=1)
print("B", Bn, "n", n, "ms kernel", r["stats"]["ms_step_kernels"])
print("mean cycles [scan, header, stage, sweeps, epilogue]:", d.mean(axis=0).round(0), "total", d.sum(axis=1).mean())
print("sweeps mean", tt[:, 6].mean(), "cycles/sweep", d[:, 3].sum() / max(tt[:, 6].sum(), 1), "nl mean", tt[:, 7].mean())
