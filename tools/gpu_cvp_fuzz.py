"""CVP on the C3-like 1M mesh (device-built costs): random plans against the CPU oracle, bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mesh_navigation_amd import capi, meshgen
from oracle import oracle as O
from bench import vertex_normals
N = int(os.environ.get("FUZZ_N", "1000")); NP = int(os.environ.get("FUZZ_PLANS", "24"))
mesh = meshgen.terrain(N, 0.1, 3)
vnrm, _ = vertex_normals(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vnrm)
ctx.layer_steepness(0, 0.6); ctx.layer_inflation(1, 0); ctx.combine_layers([0, 1], [1.0, 1.0], mode="avg", edge_cost_factor=1.0)
vc, w = ctx.download_costs()
om = O.OracleMesh(mesh.xyz, mesh.faces)
free = np.nonzero(vc < 0.5)[0]
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "11")))
first_face = np.full(mesh.V, -1, np.int64); fl = mesh.faces.ravel(); first_face[fl[::-1]] = (np.arange(fl.size)[::-1] // 3)
bad = 0
t0 = time.time()
for k in range(NP):
    g, r = rng.choice(free, 2, replace=False)
    sf, tf = int(first_face[g]), int(first_face[r])
    sp = mesh.xyz[mesh.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
    off = float(rng.choice([0.3, 0.3, 2.0, np.inf]))
    o = ctx.plan_cvp(sp, sf, tf, goal_dist_offset=off, want_fields=True, want_vecmap=False)
    ref = om.cvp(w, vc, vnrm, sp, sf, tf, goal_dist_offset=off)
    same = o.code == ref.code and np.array_equal(o.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(o.pred, ref.pred)
    if not same:
        bad += 1
        print("MISMATCH plan", k, "codes", o.code, ref.code, "dist diff", int((o.dist.view(np.uint32) != ref.dist.view(np.uint32)).sum()), flush=True)
    st = o.stats
    print(k, "off", off, "code", o.code, "ms", round(st["ms_total"], 1), "steps", st["steps"], "shrinks", st["band_shrinks"], "settled", st["settled"], "ok" if same else "BAD", flush=True)
print("plans", NP, "bad", bad, "wall", round(time.time() - t0, 1))
ctx.close()
