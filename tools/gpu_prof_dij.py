"""Scratch: a few C2 Dijkstra plans (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from tests.common import Case
N = int(os.environ.get("TUNE_N", "1000"))
case = Case(meshgen.terrain(N, 0.1, 2))
m = case.mesh
ctx = capi.MnavContext(0)
case.upload(ctx)
seed_v, target_v = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
for _ in range(5):
    o = ctx.plan_dijkstra(seed_v, target_v, want_fields=False)
print(o.stats)
if os.environ.get("PROF_CVP"):
    sp = m.xyz[seed_v] + np.array([0.03, 0.02, 0.0], np.float32)
    tp = m.xyz[target_v] + np.array([0.03, 0.02, 0.0], np.float32)
    sf, _ = case.om.containing_face(sp); tf, _ = case.om.containing_face(tp)
    for _ in range(3):
        oc = ctx.plan_cvp(sp, sf, tf, want_fields=False, want_vecmap=False)
    print(oc.stats)
