"""Scratch: how exact is the device CVP against the oracle on the test cases (bit counts)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from tests.common import Case, layered_costs, terrain_case

def report(name, case, sp, sf, tf, **kw):
    ctx = capi.MnavContext(0); case.upload(ctx)
    ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf, invalid=case.invalid, **kw)
    out = ctx.plan_cvp(sp, sf, tf, want_vecmap=True, **kw)
    upd = ref.pred != np.arange(case.mesh.V)
    nd = int((out.dist.view(np.uint32) != ref.dist.view(np.uint32)).sum())
    npred = int((out.pred != ref.pred).sum())
    ndir = int((out.direction[upd].view(np.uint32) != ref.direction[upd].view(np.uint32)).sum())
    ncut = int((out.cutface[upd] != ref.cutface[upd]).sum())
    vm = float(np.abs(out.vecmap[upd] - ref.vecmap[upd]).max()) if upd.any() else 0.0
    print(f"{name}: code {out.code}/{ref.code} dist bit diffs {nd} pred diffs {npred} direction bit diffs {ndir} (of {int(upd.sum())}) cutface diffs {ncut} vecmap max abs {vm:.2e}", flush=True)
    ctx.close()

def face_of(mesh, v): return int(np.where((mesh.faces == v).any(axis=1))[0][0])

case = terrain_case(224, 1); m = case.mesh
sp = m.xyz[m.vertex_at(0.1, 0.1)] + np.array([0.03, 0.02, 0], np.float32); tp = m.xyz[m.vertex_at(0.9, 0.9)] + np.array([0.01, 0.04, 0], np.float32)
sf, _ = case.om.containing_face(sp); tf, _ = case.om.containing_face(tp)
for off in (0.3, float("inf")): report(f"C1 off {off}", case, sp, sf, tf, goal_dist_offset=off)
case = terrain_case(1000, 2); m = case.mesh
sp = m.xyz[m.vertex_at(0.1, 0.1)] + np.array([0.03, 0.02, 0], np.float32); tp = m.xyz[m.vertex_at(0.9, 0.9)] + np.array([0.01, 0.04, 0], np.float32)
sf, _ = case.om.containing_face(sp); tf, _ = case.om.containing_face(tp)
report("C2 1M", case, sp, sf, tf)
mesh = meshgen.terrain(96, 0.1, 13)
rng = np.random.default_rng(3)
costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)
invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
invalid[[s, t]] = 0; costs[[s, t]] = 0
case = Case(mesh, costs, 1.0, invalid)
sp = mesh.xyz[s] + np.array([0.03, 0.02, 0], np.float32); tp = mesh.xyz[t] + np.array([0.03, 0.02, 0], np.float32)
sf, _ = case.om.containing_face(sp); tf, _ = case.om.containing_face(tp)
for off in (0.3, float("inf")): report(f"adversarial off {off}", case, sp, sf, tf, goal_dist_offset=off)
for seed, cut in ((5, 60), (9, 40)):
    mesh = meshgen.punched(96, 0.1, seed, drop=0.30, cut_column=cut)
    case = Case(mesh)
    deg = np.bincount(mesh.edges.ravel(), minlength=mesh.V)
    s, t = mesh.vertex_at(0.1, 0.2), mesh.vertex_at(0.5, 0.8)
    while deg[s] == 0: s += 1
    while deg[t] == 0: t += 1
    sf, tf = face_of(mesh, s), face_of(mesh, t)
    sp = mesh.xyz[mesh.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
    for off in (0.3, float("inf")): report(f"punched seed {seed} off {off}", case, sp, sf, tf, goal_dist_offset=off)
