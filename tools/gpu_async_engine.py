"""GPU check + timing of the asynchronous tile engine (engine 'async', mnav_async.h) against the oracle and the other engines.
The engine was written after round 4's GPU budget was spent: this is the FIRST thing to run on hardware before it may be chosen by
'auto' or enter the -m gpu suite.  Run under a timeout (the kernel gives up by itself after MNAV_ASYNC_MAX_S, default 10 s):

    MNAV_VERBOSE=1 timeout 300 python tools/gpu_async_engine.py [grid] [batch]

Prints one JSON line: parity (potential / predecessors / path bit-exact vs the oracle, single plans and a batch, fields and
paths-only) and ms per call for tiled / tile_batch / async at 1, 8, 64 plans."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mesh_navigation_amd import capi, meshgen  # noqa: E402
from tests.common import Case  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    m = meshgen.terrain(n, 0.1, 21)
    rng = np.random.default_rng(3)
    case = Case(m, rng.uniform(0.0, 0.6, m.V).astype(np.float32), 1.0)
    ctx = capi.MnavContext(0)
    case.upload(ctx)
    out = dict(V=int(m.V))
    robot = m.vertex_at(0.85, 0.8)
    goals = rng.choice(m.V, nb, replace=False).astype(np.uint32)
    goals = goals[goals != robot]
    targets = np.full(goals.shape[0], robot, np.uint32)
    # ---- parity: single plans (three offsets), then a batch with fields and paths only
    ctx.set_dijkstra_engine("async")
    bad = []
    for off in (0.3, 0.0, float("inf"), -0.2):
        for g in goals[:3]:
            ref = case.om.dijkstra(case.weights, case.costs, int(g), robot, goal_dist_offset=off)
            o = ctx.plan_dijkstra(int(g), robot, goal_dist_offset=off)
            ok = (o.code == ref.code and np.array_equal(o.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(o.pred, ref.pred)
                  and np.array_equal(o.path, ref.path))
            if not ok:
                bad.append(("single", off, int(g)))
    refs = [case.om.dijkstra(case.weights, case.costs, int(g), robot) for g in goals[:16]]
    b = ctx.plan_dijkstra_batch(goals, targets, want_fields=True)
    for k, ref in enumerate(refs):
        if not (b["codes"][k] == ref.code and np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32))
                and np.array_equal(b["pred"][k], ref.pred) and np.array_equal(b["paths"][k], ref.path)):
            bad.append(("batch_fields", k))
    b = ctx.plan_dijkstra_batch(goals, targets)
    for k, ref in enumerate(refs):
        if not np.array_equal(b["paths"][k], ref.path):
            bad.append(("batch_paths", k))
    out["parity_failures"] = bad
    # ---- timing
    for engine in ("tiled", "tile_batch", "async"):
        ctx.set_dijkstra_engine(engine)
        for k in (1, 8, goals.shape[0]):
            ctx.plan_dijkstra_batch(goals[:k], targets[:k])
            ts = []
            for _ in range(5):
                t = time.perf_counter()
                ctx.plan_dijkstra_batch(goals[:k], targets[:k])
                ts.append((time.perf_counter() - t) * 1e3)
            out[f"{engine}_{k}_plans_ms"] = round(float(np.median(ts)), 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
