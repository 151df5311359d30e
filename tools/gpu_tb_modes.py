"""Engine-run ms of the tile-batch engine per batch in the two modes bench.py uses on the C2 mesh -- headline (outputs resident in HBM,
finalize pass) and vertex paths only --, batch after batch in one process, for one or more builds / environments:
   python tools/gpu_tb_modes.py lib.so[@VAR=VALUE[,VAR=VALUE]] ...       (round 6: where the other buffer's fill belongs)"""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, os, json, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from mesh_navigation_amd import capi, meshgen
B = int(os.environ.get("B", "7168"))
mesh = meshgen.terrain(1000, 0.1, 21)
w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
ctx.set_dijkstra_engine("tile_batch")
rng = np.random.default_rng(5)
robot = mesh.vertex_at(0.9, 0.9)
tg = np.full(B, robot, np.uint32)
out = {}
def run(tag, n, **kw):
    eng, wall = [], []
    for r in range(n):
        g = rng.choice(mesh.V, size=B, replace=False).astype(np.uint32)
        t0 = time.perf_counter()
        b = ctx.plan_dijkstra_batch(g, tg, goal_dist_offset=0.3, want_fields=False, path_cap=16384, **kw)
        wall.append(round((time.perf_counter() - t0) * 1e3, 1))
        assert (b["codes"] == 0).all()
        st = b["stats"] if "stats" in b and b["stats"] else ctx.stats()
        eng.append(round(st["ms_step_kernels"], 1))
    out[tag] = {"engine_ms": eng, "wall_ms": wall}
ctx.set_resident_outputs(True)
run("headline", 5)
ctx.set_resident_outputs(False)
run("paths_only", 5, want_stats=False)
print(json.dumps(out))
'''
if sys.argv[1:2] == ["--inline"]:                                     # in this process (under rocprofv3), library / switches from the environment
    exec(CHILD)
    sys.exit(0)
for spec in sys.argv[1:]:
    lib, _, extra = spec.partition("@")
    env = dict(os.environ, MNAV_LIB=os.path.abspath(lib))
    for kv in filter(None, extra.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    print(os.path.basename(spec), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
