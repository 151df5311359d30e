import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from mesh_navigation_amd import build as B
B.LIB = os.path.join(os.path.dirname(B.LIB), "libmnav_timing.so")
from tests.common import Case
case = Case(meshgen.terrain(1000, 0.1, 2))
m = case.mesh
ctx = capi.MnavContext(0)
L = ctx._L
case.upload(ctx)
seed_v, target_v = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
buf = np.zeros((4096, 8), np.uint64)
for rep in range(2):
    o = ctx.plan_dijkstra(seed_v, target_v, want_fields=False)
    n = L.mnav_debug_tile_timing(buf.ctypes.data_as(C.c_void_p), 4096)
print(o.stats)
t = buf[:n].astype(np.int64)
d = np.diff(t[:, :7], axis=1)   # phases 0-1 prologue,1-2 scan,2-3 stage,3-4 queue init,4-5 sweeps,5-6 epilogue
sw = (t[:, 7] & 0xFFFFFFFF)
print("n", n, "mean cycles per phase [prologue, scan, stage, qinit, sweeps, epilogue]:", d.mean(axis=0).round(0))
print("median:", np.median(d, axis=0))
print("sweeps mean", sw.mean(), "cycles per sweep", (d[:, 4].sum() / max(sw.sum(), 1)))
print("total mean cycles", (t[:, 6] - t[:, 0]).mean())
