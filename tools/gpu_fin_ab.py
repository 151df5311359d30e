"""A/B of the finalize pass variants (libmnav built with -DMNAV_FIN_WAVES / -DMNAV_FIN_OCC): ms of the V-sized outputs of one
7168-plan batch on the 1M mesh (bench.py's headline step), one subprocess per library.   python tools/gpu_fin_ab.py lib1.so lib2.so ..."""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, time, json, numpy as np
sys.path.insert(0, ".")
from mesh_navigation_amd import capi, meshgen
mesh = meshgen.terrain(1000, 0.1, 21)
w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
ctx.set_resident_outputs(True)
B = 7168
goals = np.random.default_rng(5).choice(mesh.V, size=B, replace=False).astype(np.uint32)
tg = np.full(B, mesh.vertex_at(0.9, 0.9), np.uint32)
out = []
for r in range(3):
    b = ctx.plan_dijkstra_batch(goals, tg, want_fields=False, path_cap=65536, want_stats=False)
    st = b["stats"]
    out.append(dict(total=st["ms_total"], engine=st["ms_step_kernels"], prop=st["ms_propagation"]))
print(json.dumps(out[1:]))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, MNAV_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
