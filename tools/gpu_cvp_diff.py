"""Scratch: locate the earliest CVP mismatch between the device and the oracle (C3 1M)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from tests.common import Case, layered_costs
N = int(os.environ.get("DIFF_N", "1000"))
base = Case(meshgen.terrain(N, 0.1, 3, amplitude=0.8))
costs, parts = layered_costs(base, "avg")
case = Case(base.mesh, costs, 1.0)
ctx = capi.MnavContext(0); case.upload(ctx)
m = case.mesh
free = np.where(costs < 0.5)[0]
def near(fi, fj):
    v = m.vertex_at(fi, fj)
    return int(free[((m.xyz[free, :2] - m.xyz[v, :2]) ** 2).sum(1).argmin()])
s, t = near(0.1, 0.1), near(0.9, 0.9)
sp = m.xyz[s] + np.array([0.02, 0.01, 0], np.float32); tp = m.xyz[t] + np.array([0.02, 0.01, 0], np.float32)
sf, _ = case.om.containing_face(sp); tf, _ = case.om.containing_face(tp)
ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf)
out = ctx.plan_cvp(sp, sf, tf)
fin = np.isfinite(ref.dist)
bad = np.where(fin & (out.dist != ref.dist))[0]
print("mismatching vertices", len(bad), "of", fin.sum())
order = bad[np.argsort(ref.dist[bad])]
rows = []
fe = m.face_edges
for v in order[:12]:
    row = dict(v=int(v), gpu=float(out.dist[v]), ref=float(ref.dist[v]), pred_gpu=int(out.pred[v]), pred_ref=int(ref.pred[v]),
               cut_gpu=int(out.cutface[v]), cut_ref=int(ref.cutface[v]), dir_gpu=float(out.direction[v]), dir_ref=float(ref.direction[v]))
    # faces around v with support values (both sides agree on supports if they are earlier)
    faces = np.where((m.faces == v).any(axis=1))[0]
    fl = []
    for f in faces:
        k3 = int(np.where(m.faces[f] == v)[0][0]); k1, k2 = (k3 + 1) % 3, (k3 + 2) % 3
        v1, v2 = int(m.faces[f][k1]), int(m.faces[f][k2])
        c = float(case.weights[fe[f][k1]]); b = float(case.weights[fe[f][k3]]); a = float(case.weights[fe[f][k2]])
        fl.append(dict(f=int(f), v1=v1, v2=v2, u1_ref=float(ref.dist[v1]), u2_ref=float(ref.dist[v2]), u1_gpu=float(out.dist[v1]), u2_gpu=float(out.dist[v2]), a=a, b=b, c=c,
                       cost=[float(case.costs[v1]), float(case.costs[v2])]))
    row["faces"] = fl
    rows.append(row)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/cvp_diff.json", "w"), indent=1)
print(json.dumps(rows[:2], indent=1))
