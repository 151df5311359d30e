"""Timing of the asynchronous tile engine (engine 'async', mnav_async.h: ticket queue) against the tile rounds, with its knobs:
banded solves, workgroups per plan.  One JSON line per mesh.

    timeout 600 python tools/gpu_async_tune.py [grid=1000] [reps=5]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mesh_navigation_amd import capi, meshgen  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    mesh = meshgen.terrain(N, 0.1, 21 if N <= 1000 else 4)
    w = meshgen.edge_lengths(mesh)
    robot = mesh.vertex_at(0.9, 0.9)
    goals = np.random.default_rng(5).choice(mesh.V, size=128, replace=False).astype(np.uint32)
    out = dict(grid=N, V=int(mesh.V))
    ref_paths = None
    variants = [("tiled", {}), ("async", {}), ("async_wg512", dict(async_wg_per_plan=512)), ("async_band2", dict(async_band_mult=2.0)),
                ("async_band8", dict(async_band_mult=8.0)), ("async_band3", dict(async_band_mult=3.0)), ("async_band6", dict(async_band_mult=6.0)),
                ("async_cu5", dict(async_wg_per_cu=5)), ("async_cu3", dict(async_wg_per_cu=3)), ("async_cu2", dict(async_wg_per_cu=2)),
                ("async_band1.5", dict(async_band_mult=1.5)), ("async_band1", dict(async_band_mult=1.0))]
    if len(sys.argv) > 3:
        variants = [v for v in variants if v[0] in sys.argv[3].split(",")]
    for label, opts in variants:
        ctx = capi.MnavContext(0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
        ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
        ctx.set_dijkstra_engine("tiled" if label == "tiled" else "async")
        res = {}
        for nb in (1, 8, 47, 96):
            tg = np.full(nb, robot, np.uint32)
            ctx.plan_dijkstra_batch(goals[:nb], tg)
            ts, dev = [], []
            for r in range(reps):
                g = goals[r:r + nb] if nb == 1 else goals[:nb]
                t0 = time.perf_counter()
                b = ctx.plan_dijkstra_batch(g, tg)
                ts.append((time.perf_counter() - t0) * 1e3)
                dev.append(b["stats"]["ms_total"])
                assert (b["codes"] == 0).all(), label
            res[f"{nb}_wall_ms"] = round(float(np.median(ts)), 3)
            res[f"{nb}_dev_ms"] = round(float(np.median(dev)), 3)
            if nb == 47:
                sig = [int(np.asarray(p, np.uint64).sum()) for p in b["paths"]]
                if ref_paths is None:
                    ref_paths = sig
                res["paths_equal_tiled"] = sig == ref_paths
        # the plan makePlan runs: fields + vector map
        try:
            for g in goals[:6]:
                o = ctx.plan_dijkstra(int(g), robot, want_fields=True, want_vecmap=True)
            res["1_full_outputs_dev_ms"] = round(o.stats["ms_total"], 3)
        except RuntimeError as e:
            res["full_outputs_error"] = f"goal {int(g)}: {e}"
        out[label] = res
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
