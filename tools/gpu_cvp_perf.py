"""CVP throughput experiments on the GPU box: batch sizes / library variants (MNAV_LIB).  Prints one line per case."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime order, see tests/conftest.py)
torch.cuda.init()
from mesh_navigation_amd import capi, meshgen
from bench import vertex_normals

N = int(os.environ.get("PERF_N", "1000"))
mesh = meshgen.terrain(N, 0.1, 3)
vnrm, _ = vertex_normals(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vnrm)
ctx.layer_steepness(0, 0.6)
ctx.layer_inflation(1, 0)
ctx.combine_layers([0, 1], [1.0, 1.0], mode="avg", edge_cost_factor=1.0)
vc, w = ctx.download_costs()
free = np.nonzero(vc < 0.5)[0]
rng = np.random.default_rng(5)
first_face = np.full(mesh.V, -1, np.int64)
fl = mesh.faces.ravel()
first_face[fl[::-1]] = (np.arange(fl.size)[::-1] // 3)
def wave_seed(v):
    f = int(first_face[v])
    return mesh.xyz[mesh.faces[f]].astype(np.float64).mean(axis=0).astype(np.float32), f
robot = int(free[np.argmin(np.abs(mesh.xyz[free, 0] - 0.9 * N * 0.1) + np.abs(mesh.xyz[free, 1] - 0.9 * N * 0.1))])
tf = int(first_face[robot])
goals = rng.choice(free, size=1100, replace=False)
tag = os.environ.get("MNAV_LIB", "default").split("/")[-1]
mean_w = float(w[np.isfinite(w)].mean())
if os.environ.get("PERF_BAND_EDGES"):
    ctx.set_band_width(float(os.environ["PERF_BAND_EDGES"]) * mean_w)     # band width in mean edge weights (default 12)
for nb in [int(x) for x in os.environ.get("PERF_BATCHES", "1,128,512").split(",")]:
    seeds = [wave_seed(int(v)) for v in goals[:nb]]
    sps = np.stack([x[0] for x in seeds]); sfs = np.array([x[1] for x in seeds], np.uint32)
    ctx.plan_cvp_batch(sps, sfs, np.full(nb, tf, np.uint32))
    t0 = time.perf_counter()
    rb = ctx.plan_cvp_batch(sps, sfs, np.full(nb, tf, np.uint32))
    dt = time.perf_counter() - t0
    st = rb["stats"]
    Lb = capi.load()
    if hasattr(Lb, "mnav_debug_wide_timing"):
        import ctypes
        tt = (ctypes.c_ulonglong * 12)()
        Lb.mnav_debug_wide_timing(tt)
        names = ["pre+slots", "A loads", "A compute", "B replay", "post", "push", "park", "-"]
        print("   phase Gcycles:", {nm: round(tt[i] / 1e9, 2) for i, nm in enumerate(names[:7])}, flush=True)
        r = max(int(tt[8]), 1)
        print("   rounds %d: active entries per round %.1f, evaluated %.1f, serial-rule %.3f" % (r, tt[9] / r, tt[10] / r, tt[11] / r), flush=True)
    print(json.dumps({"band_edges": os.environ.get("PERF_BAND_EDGES", "12"), "lib": tag, "batch": nb, "plans_per_s": nb / dt, "ms": dt * 1e3, "steps": st["steps"], "launches": st["launches"],
                      "evals_per_plan": st["evals"] / nb, "band_shrinks": st["band_shrinks"], "settled_per_plan": st["settled"] / nb, "ms_step_kernels": st["ms_step_kernels"],
                      "codes": sorted(set(int(c) for c in rb["codes"]))}), flush=True)
ctx.close()
