import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mesh_navigation_amd import capi, meshgen
from bench import vertex_normals
mesh = meshgen.terrain(1000, 0.1, 3)
vnrm, _ = vertex_normals(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vnrm)
for thr in (0.3, 0.6, 0.3, 0.6, 0.6, 0.6):
    ctx.layer_steepness(0, thr)
    t = time.perf_counter(); st = ctx.layer_inflation(1, 0); print("thr", thr, st, "wall ms", (time.perf_counter() - t) * 1e3, flush=True)
ctx.close()
