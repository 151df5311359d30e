"""Scratch: CVP throughput in batches on the C3 configuration (1M vertices, layered costs)."""
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
out = {}
for B in (1, 8, 32, 64, 128):
    seeds = rng.choice(free, size=B, replace=False)
    sps = (m.xyz[seeds] + np.array([0.02, 0.01, 0], np.float32)).astype(np.float32)
    sfs = np.array([case.om.containing_face(p)[0] for p in sps], np.uint32)
    tfs = np.full(B, tf, np.uint32)
    b = ctx.plan_cvp_batch(sps, sfs, tfs)
    t0 = time.time(); b = ctx.plan_cvp_batch(sps, sfs, tfs); dt = time.time() - t0
    st = b["stats"]
    out[B] = dict(wall_ms=dt * 1e3, plans_per_s=B / dt, steps=st["steps"], ms_prop=st["ms_propagation"], ms_kern=st["ms_step_kernels"], codes_ok=int((b["codes"] == 0).sum()), algo=st["algorithmic_bytes"])
    print(B, out[B], flush=True)
t0 = time.time(); ref = case.om.cvp(case.weights, case.costs, case.vn, sps[0], int(sfs[0]), tf); print("oracle cvp ms", (time.time() - t0) * 1e3, ref.stats["t_propagation_ms"])
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "cvp_batch.json"), "w"), indent=1)
