"""Scratch: sweep tile size / band width of the tiled SSSP engine on C2 (and C4 optionally)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen
from tests.common import Case

N = int(os.environ.get("TUNE_N", "1000"))
case = Case(meshgen.terrain(N, 0.1, 2))
m = case.mesh
seed_v, target_v = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
ref = case.om.dijkstra(case.weights, case.costs, seed_v, target_v)
rows = []
for ts in (1024, 2048):
    os.environ["MNAV_TILE_SIZE"] = str(ts)
    ctx = capi.MnavContext(0)
    case.upload(ctx)
    for mult in (2.0, 4.0, 8.0):
        os.environ["MNAV_TILE_BAND"] = str(0.115 * np.sqrt(ts) * mult)
        case.upload(ctx)  # re-reads MNAV_TILE_BAND
        o = ctx.plan_dijkstra(seed_v, target_v, want_fields=True)
        ok = bool(np.array_equal(o.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(o.pred, ref.pred))
        best = 1e9
        for _ in range(3):
            o = ctx.plan_dijkstra(seed_v, target_v, want_fields=False)
            best = min(best, o.stats["ms_propagation"])
        rows.append(dict(tile=ts, mult=mult, ok=ok, ms=round(best, 3), rounds=o.stats["steps"], acts=o.stats["evals"], sweeps=o.stats["bands"]))
        print(rows[-1], flush=True)
    ctx.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/tune_tiles.json", "w"), indent=1)
