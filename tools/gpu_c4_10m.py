"""BASELINE config C4 on ONE GPU: 10M-vertex terrain (N=3163, seed 4), Dijkstra: parity with the
oracle + size-independent properties + timings.  Writes gpurun_out/c4_10m.json."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from oracle import oracle as O
out = {}
t = time.time(); mesh = meshgen.terrain(3163, 0.1, 4); out["gen_s"] = time.time() - t
print("mesh", mesh.V, mesh.F, mesh.E, out["gen_s"], flush=True)
w = meshgen.edge_lengths(mesh); costs = np.zeros(mesh.V, np.float32)
ctx = capi.MnavContext(0)
t = time.time(); ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None); ctx.upload_costs(costs, w); out["upload_s"] = time.time() - t
print("upload", out["upload_s"], flush=True)
s, tg = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
o = ctx.plan_dijkstra(s, tg, want_fields=True)
for rep in range(3):
    o2 = ctx.plan_dijkstra(s, tg, want_fields=False)
out["early_exit_stats"] = o2.stats
of = ctx.plan_dijkstra(s, tg, goal_dist_offset=float("inf"), want_fields=True)
out["full_field_stats"] = ctx.plan_dijkstra(s, tg, goal_dist_offset=float("inf"), want_fields=False).stats
d, e = of.dist, mesh.edges
out["full_field_fixed_point"] = bool((d[e[:, 0]] <= d[e[:, 1]] + w).all() and (d[e[:, 1]] <= d[e[:, 0]] + w).all() and d[s] == 0 and np.isfinite(d).all())
nz = np.arange(mesh.V) != s
out["pred_descends"] = bool((d[of.pred[nz]] < d[nz]).all())
out["algorithmic_bytes_full"] = of.stats["algorithmic_bytes"]
t = time.time(); om = O.OracleMesh(mesh.xyz, mesh.faces); out["oracle_mesh_s"] = time.time() - t
ref = om.dijkstra(w, costs, s, tg)
out["oracle_ms"] = ref.stats["t_propagation_ms"] + ref.stats["t_init_ms"] + ref.stats["t_backtrack_ms"]
out["bit_exact_dist"] = bool(np.array_equal(o.dist.view(np.uint32), ref.dist.view(np.uint32)))
out["pred_equal"] = bool(np.array_equal(o.pred, ref.pred)); out["path_equal"] = bool(np.array_equal(o.path, ref.path)); out["path_len"] = int(len(ref.path))
reff = om.dijkstra(w, costs, s, tg, goal_dist_offset=np.inf)
out["full_bit_exact"] = bool(np.array_equal(of.dist.view(np.uint32), reff.dist.view(np.uint32)) and np.array_equal(of.pred, reff.pred))
out["oracle_full_ms"] = reff.stats["t_propagation_ms"]
rng = np.random.default_rng(5); B = 16
goals = rng.choice(mesh.V, size=B, replace=False).astype(np.uint32)
b = ctx.plan_dijkstra_batch(goals, np.full(B, tg, np.uint32), want_fields=False, path_cap=32768)
t = time.time(); b = ctx.plan_dijkstra_batch(goals, np.full(B, tg, np.uint32), want_fields=False, path_cap=32768); out["batch16_wall_ms"] = (time.time() - t) * 1e3
out["batch16_stats"] = b["stats"]; out["batch16_codes_ok"] = bool((b["codes"] == 0).all())
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/c4_10m.json", "w"), indent=1, default=str)
print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}, indent=1))
for k in ("early_exit_stats", "full_field_stats", "batch16_stats"):
    print(k, {a: out[k][a] for a in ("steps", "evals", "bands", "settled", "ms_propagation", "ms_path", "ms_total", "algorithmic_bytes")})
