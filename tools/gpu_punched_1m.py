"""Scratch: parity and timing on a 1M-vertex terrain with 20 % of the faces punched out (deep cascades)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from tests.common import Case
t0 = time.time()
mesh = meshgen.punched(1000, 0.1, 7, drop=float(os.environ.get("DROP", "0.2")))
case = Case(mesh)
deg = np.bincount(mesh.edges.ravel(), minlength=mesh.V)
print("mesh", mesh.V, mesh.F, mesh.E, "face-less", int((deg == 0).sum()), "gen s", time.time() - t0, flush=True)
ctx = capi.MnavContext(0); case.upload(ctx)
s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
while deg[s] == 0: s += 1
while deg[t] == 0: t += 1
out = {}
t0 = time.time(); ref = case.om.dijkstra(case.weights, case.costs, s, t); out["oracle_dijkstra_ms"] = (time.time() - t0) * 1e3
for engine in ("tiled", "persistent"):
    ctx.set_dijkstra_engine(engine)
    o = ctx.plan_dijkstra(s, t); o = ctx.plan_dijkstra(s, t)
    out[f"dijkstra_{engine}"] = dict(code=int(o.code), ref_code=int(ref.code), bit_exact=bool(np.array_equal(o.dist.view(np.uint32), ref.dist.view(np.uint32))),
                                     pred_equal=bool(np.array_equal(o.pred, ref.pred)), path_equal=bool(np.array_equal(o.path, ref.path)), ms=o.stats["ms_total"])
    print(engine, out[f"dijkstra_{engine}"], flush=True)
ctx.set_dijkstra_engine("auto")
sf = int(np.where((mesh.faces == s).any(axis=1))[0][0]); tf = int(np.where((mesh.faces == t).any(axis=1))[0][0])
sp = mesh.xyz[mesh.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
t0 = time.time(); refc = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf); out["oracle_cvp_ms"] = (time.time() - t0) * 1e3
o = ctx.plan_cvp(sp, sf, tf); o = ctx.plan_cvp(sp, sf, tf)
upd = refc.pred != np.arange(mesh.V)
out["cvp"] = dict(code=int(o.code), ref_code=int(refc.code), dist_bit_diffs=int((o.dist.view(np.uint32) != refc.dist.view(np.uint32)).sum()),
                  pred_diffs=int((o.pred != refc.pred).sum()), dir_bit_diffs=int((o.direction[upd].view(np.uint32) != refc.direction[upd].view(np.uint32)).sum()),
                  reached=int(np.isfinite(refc.dist).sum()), ms=o.stats["ms_total"], steps=o.stats["steps"])
print("cvp", out["cvp"], "oracle ms", out["oracle_cvp_ms"], flush=True)
json.dump(out, open("gpurun_out/punched_1m.json", "w"), indent=1)
