"""Wave-per-plan engine vs the workgroup-per-plan engine on the C2 mesh: correctness (paths, codes, potential of a
few plans) and throughput.  python tools/gpu_wave_engine.py [batch] [offset]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
off = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
N = int(os.environ.get("MNAV_BENCH_N", "1000"))
mesh = meshgen.terrain(N, 0.1, 2)
w = meshgen.edge_lengths(mesh)
ctx = capi.MnavContext(0)
ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
rng = np.random.default_rng(5)
g = rng.choice(mesh.V, B, replace=False).astype(np.uint32)
t = np.full(B, mesh.vertex_at(0.9, 0.9), np.uint32)
res = {}
out = {}
ENGS = ("wave",) if os.environ.get("MNAV_ONLY_WAVE") else ("persistent",) if os.environ.get("MNAV_ONLY_PERSISTENT") else ("persistent", "wave")
for eng in ENGS:
    ctx.set_dijkstra_engine(eng)
    r = ctx.plan_dijkstra_batch(g, t, goal_dist_offset=off, path_cap=16384)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = ctx.plan_dijkstra_batch(g, t, goal_dist_offset=off, path_cap=16384)
        best = min(best, time.perf_counter() - t0)
    res[eng] = r
    st = r["stats"]
    out[eng] = dict(acts_per_plan=st["evals"] / B, plans_per_s=B / best, ms=best * 1e3, kernel_ms=st["ms_step_kernels"], prop_ms=st["ms_propagation"],
                    ok=bool((r["codes"] == 0).all()), gbps=st["algorithmic_bytes"] / (st["ms_step_kernels"] * 1e-3) / 1e9)
if "persistent" in res and "wave" in res:
    out["paths_equal"] = bool(all(np.array_equal(a, b) for a, b in zip(res["persistent"]["paths"], res["wave"]["paths"])))
if os.environ.get("MNAV_ONLY_PERSISTENT"):
    out["tile"] = os.environ.get("MNAV_TILE_SIZE", "512"); print(json.dumps(out)); sys.exit(0)
# full potential of a few plans: bit-equal between the engines
ctx.set_dijkstra_engine("persistent"); a = ctx.plan_dijkstra_batch(g[:160], t[:160], goal_dist_offset=off, want_fields=True, path_cap=16384)
ctx.set_dijkstra_engine("wave"); b = ctx.plan_dijkstra_batch(g[:160], t[:160], goal_dist_offset=off, want_fields=True, path_cap=16384)
out["dist_equal"] = bool(np.array_equal(a["dist"].view(np.uint32), b["dist"].view(np.uint32)))
out["pred_equal"] = bool(np.array_equal(a["pred"], b["pred"]))
out["wave_tile"] = os.environ.get("MNAV_WAVE_TILE", "128"); out["batch"] = B; out["offset"] = off
print(json.dumps(out))
