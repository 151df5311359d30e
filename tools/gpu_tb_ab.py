"""A/B of libmnav builds on the headline batch (7168 plans, 1M mesh, paths only): engine-run ms per library, one subprocess each.
   python tools/gpu_tb_ab.py lib1.so lib2.so ...   [env N, B]"""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, os, json, numpy as np
sys.path.insert(0, ".")
from mesh_navigation_amd import capi, meshgen
N = int(os.environ.get("N", "1000")); B = int(os.environ.get("B", "7168"))
mesh = meshgen.terrain(N, 0.1, 21 if N <= 1000 else 4)
w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
ctx.set_dijkstra_engine("tile_batch")
goals = np.random.default_rng(5).choice(mesh.V, size=B, replace=False).astype(np.uint32)
tg = np.full(B, mesh.vertex_at(0.9, 0.9), np.uint32)
out = []
for r in range(3):
    b = ctx.plan_dijkstra_batch(goals, tg, want_fields=False, path_cap=65536, want_stats=False)
    assert (b["codes"] == 0).all()
    out.append(round(b["stats"]["ms_step_kernels"], 2))
sig = int(sum(int(np.asarray(p, np.uint64).sum()) for p in b["paths"][:256]))
print(json.dumps(dict(engine_ms=out[1:], sig=sig)))
'''
for spec in sys.argv[1:]:                                             # lib.so[@VAR=VALUE[,VAR=VALUE]]
    lib, _, extra = spec.partition("@")
    env = dict(os.environ, MNAV_LIB=os.path.abspath(lib))
    for kv in filter(None, extra.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    print(os.path.basename(spec), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
