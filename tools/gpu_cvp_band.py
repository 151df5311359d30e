"""Scratch: CVP latency / throughput vs band width on C3 (1M vertices, layered costs)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from tests.common import Case, layered_costs
base = Case(meshgen.terrain(1000, 0.1, 3, amplitude=0.8))
costs, parts = layered_costs(base, "avg")
case = Case(base.mesh, costs, 1.0)
ctx = capi.MnavContext(0); case.upload(ctx)
m = case.mesh
free = np.where(costs < 0.5)[0]
rng = np.random.default_rng(5)
def near(fi, fj):
    v = m.vertex_at(fi, fj)
    return int(free[((m.xyz[free, :2] - m.xyz[v, :2]) ** 2).sum(1).argmin()])
t = near(0.9, 0.9)
tp = m.xyz[t] + np.array([0.02, 0.01, 0], np.float32); tf, _ = case.om.containing_face(tp)
finite_w = case.weights[np.isfinite(case.weights)]
mean_w = float(finite_w.mean())
ref = None
out = {}
for B in (1, 64):
    seeds = rng.choice(free, size=B, replace=False)
    sps = (m.xyz[seeds] + np.array([0.02, 0.01, 0], np.float32)).astype(np.float32)
    sfs = np.array([case.om.containing_face(p)[0] for p in sps], np.uint32)
    tfs = np.full(B, tf, np.uint32)
    for mult in (3, 6, 12, 24, 48):
        ctx.set_band_width(mult * mean_w)
        b = ctx.plan_cvp_batch(sps, sfs, tfs, want_fields=(B == 1))
        t0 = time.time(); b = ctx.plan_cvp_batch(sps, sfs, tfs, want_fields=(B == 1)); dt = time.time() - t0
        st = b["stats"]
        same = None
        if B == 1:
            d = b["dist"][0]
            if ref is None: ref = d.copy()
            same = bool(np.array_equal(ref.view(np.uint32), d.view(np.uint32)))
        out[f"{B}_{mult}"] = dict(wall_ms=dt * 1e3, steps=st["steps"], ms_kern=st["ms_step_kernels"], evals=st["evals"], ok=int((b["codes"] == 0).sum()), same_as_first=same)
        print(B, mult, out[f"{B}_{mult}"], flush=True)
json.dump(out, open("gpurun_out/cvp_band.json", "w"), indent=1)
