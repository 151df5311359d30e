"""A/B of the CVP step kernels on the C3-like configuration (1M vertices, layered costs): 8-lane replay (MNAV_CVP_WIDE=0) against the
wide kernel (1): plans/s per batch size, and the potentials of both must be bit-identical."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from tests.common import Case, layered_costs
N = int(os.environ.get("N", "1000"))
base = Case(meshgen.terrain(N, 0.1, 3, amplitude=0.8))
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
for B in [int(x) for x in os.environ.get("BS", "1,16,128").split(",")]:
    seeds = rng.choice(free, size=B, replace=False)
    sps = (m.xyz[seeds] + np.array([0.02, 0.01, 0], np.float32)).astype(np.float32)
    sfs = np.array([case.om.containing_face(p)[0] for p in sps], np.uint32)
    tfs = np.full(B, tf, np.uint32)
    ref = None
    for wide in os.environ.get("WIDES", "0,1").split(","):
        os.environ["MNAV_CVP_WIDE"] = wide
        for g in os.environ.get("GS", "").split(","):
            if g:
                os.environ["MNAV_BLOCKS_PER_PLAN_WIDE"] = g
            b = ctx.plan_cvp_batch(sps, sfs, tfs, want_fields=(B <= 16))
            t0 = time.time(); b = ctx.plan_cvp_batch(sps, sfs, tfs, want_fields=(B <= 16)); dt = time.time() - t0
            st = b["stats"]
            eq = None
            if B <= 16:
                sig = [b["dist"][k].view(np.uint32).copy() for k in range(B)]
                if ref is None:
                    ref = sig
                eq = all(np.array_equal(x, y) for x, y in zip(ref, sig))
            Lb = capi.load()
            if hasattr(Lb, "mnav_debug_wide_timing"):
                import ctypes
                tt = (ctypes.c_ulonglong * 12)()
                Lb.mnav_debug_wide_timing(tt)
                names = ["pre+slots", "A loads", "A compute", "B replay", "post", "push", "park", "-"]
                print("   phase Gcycles:", {nm: round(tt[i] / 1e9, 2) for i, nm in enumerate(names[:7])}, flush=True)
                r = max(int(tt[8]), 1)
                print("   rounds %d: active entries per round %.1f, evaluated %.1f, serial-rule %.3f" % (r, tt[9] / r, tt[10] / r, tt[11] / r), flush=True)
            print(dict(B=B, wide=wide, G=g, wall_ms=round(dt * 1e3, 1), plans_per_s=round(B / dt, 1), steps=st["steps"], ms_kern=round(st["ms_step_kernels"], 1),
                       evals=st.get("evals"), ok=int((b["codes"] == 0).sum()), equal_to_first=eq), flush=True)
