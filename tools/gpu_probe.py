"""Scratch GPU probe: parity + timing at C1 / C2, results to gpurun_out/probe.json."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen  # noqa: E402
from tests.common import Case  # noqa: E402

out = {}
os.makedirs("gpurun_out", exist_ok=True)
for name, N, seed in (("C1", 224, 1), ("C2", 1000, 2)):
    t = time.time()
    case = Case(meshgen.terrain(N, 0.1, seed))
    m = case.mesh
    print(name, "gen+oracle mesh", time.time() - t, flush=True)
    ctx = capi.MnavContext(0)
    t = time.time(); case.upload(ctx); print("upload", time.time() - t, flush=True)
    seed_v, target_v = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    t = time.time(); ref = case.om.dijkstra(case.weights, case.costs, seed_v, target_v); t_cpu = time.time() - t
    res = {}
    ctx.set_dijkstra_engine("band")
    o = ctx.plan_dijkstra(seed_v, target_v, want_fields=False)
    o = ctx.plan_dijkstra(seed_v, target_v, want_fields=False)
    res["dij_band_stats"] = o.stats
    ctx.set_dijkstra_engine("tiled")
    for off in (0.0, 0.05, 1.0, float("inf")):
        r2 = case.om.dijkstra(case.weights, case.costs, seed_v, target_v, goal_dist_offset=off)
        o2 = ctx.plan_dijkstra(seed_v, target_v, goal_dist_offset=off, want_fields=True)
        res[f"dij_off{off}_exact"] = [bool(np.array_equal(o2.dist.view(np.uint32), r2.dist.view(np.uint32))), int((o2.pred != r2.pred).sum()), bool(np.array_equal(o2.path, r2.path))]
    for rep in range(4):
        o = ctx.plan_dijkstra(seed_v, target_v, want_fields=(rep == 0), want_vecmap=(rep == 0))
        if rep == 0:
            res["dij_dist_bitexact"] = bool(np.array_equal(o.dist.view(np.uint32), ref.dist.view(np.uint32)))
            res["dij_pred_diff"] = int((o.pred != ref.pred).sum())
            res["dij_path_equal"] = bool(np.array_equal(o.path, ref.path))
            res["dij_code"] = (int(o.code), int(ref.code))
        res[f"dij_stats_{rep}"] = o.stats
    res["dij_cpu_ms"] = ref.stats["t_propagation_ms"]
    res["dij_cpu_wall_ms"] = t_cpu * 1e3
    # full field
    o = ctx.plan_dijkstra(seed_v, target_v, goal_dist_offset=float("inf"), want_fields=False)
    res["dij_fullfield_stats"] = o.stats
    # cvp
    sp = m.xyz[seed_v] + np.array([0.03, 0.02, 0.0], np.float32)
    tp = m.xyz[target_v] + np.array([0.03, 0.02, 0.0], np.float32)
    sf, _ = case.om.containing_face(sp) if N < 500 else (None, None)
    if sf is None:
        # brute-force nearest vertex is O(V): fine
        sf, _ = case.om.containing_face(sp)
    tf, _ = case.om.containing_face(tp)
    t = time.time(); refc = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf); t_cpu = time.time() - t
    for rep in range(3):
        oc = ctx.plan_cvp(sp, sf, tf, want_fields=(rep == 0), want_vecmap=(rep == 0))
        if rep == 0:
            fin = np.isfinite(refc.dist)
            res["cvp_same_reached"] = bool(np.array_equal(np.isfinite(oc.dist), fin))
            if res["cvp_same_reached"]:
                rel = np.abs(oc.dist[fin] - refc.dist[fin]) / np.maximum(refc.dist[fin], 1e-12)
                res["cvp_maxrel"] = float(rel.max())
                res["cvp_bitdiff"] = int((oc.dist[fin] != refc.dist[fin]).sum())
            res["cvp_pred_diff"] = int((oc.pred != refc.pred).sum())
            res["cvp_dir_maxabs"] = float(np.abs(oc.direction - refc.direction).max())
            res["cvp_cut_diff"] = int((oc.cutface != refc.cutface).sum())
            res["cvp_vec_maxabs"] = float(np.abs(oc.vecmap - refc.vecmap).max())
            res["cvp_code"] = (int(oc.code), int(refc.code))
        res[f"cvp_stats_{rep}"] = oc.stats
    res["cvp_cpu_ms"] = refc.stats["t_propagation_ms"]
    # batch
    rng = np.random.default_rng(5)
    for B in (8, 64):
        goals = rng.choice(m.V, size=B, replace=False).astype(np.uint32)
        targets = np.full(B, target_v, np.uint32)
        b = ctx.plan_dijkstra_batch(goals, targets, want_fields=False, path_cap=8192)
        b = ctx.plan_dijkstra_batch(goals, targets, want_fields=False, path_cap=8192)
        res[f"batch{B}_stats"] = b["stats"]
        res[f"batch{B}_codes"] = [int(c) for c in b["codes"][:4]]
    out[name] = res
    print(json.dumps({name: res}, indent=1, default=str), flush=True)
    ctx.close()
json.dump(out, open("gpurun_out/probe.json", "w"), indent=1, default=str)
