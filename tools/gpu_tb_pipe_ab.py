"""A/B of the pipelined sweeps of the tile-batch engine (MNAV_TB_PIPE=1 / 2: streams with forward marks + k_tb_solve_q<T, 1 / 2>, mnav_tb.h)
against the plain sweep, in one process: three contexts on the same mesh, the same batch through both, paths compared, engine-run time
and roofline fraction printed.  The pipelined variant was written after round 4's GPU minutes were spent: this is its first run.

    timeout 600 python tools/gpu_tb_pipe_ab.py [grid=1000] [batch=7168] [reps=3]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 7168
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    mesh = meshgen.terrain(N, 0.1, 21)
    w = meshgen.edge_lengths(mesh)
    robot = mesh.vertex_at(0.9, 0.9)
    goals = np.random.default_rng(5).choice(mesh.V, size=B, replace=False).astype(np.uint32)
    targets = np.full(B, robot, np.uint32)
    out, sigs = {}, {}
    for label, env in (("plain", "0"), ("pipelined", "1"), ("pipelined_across_chunks", "2")):
        os.environ["MNAV_TB_PIPE"] = env                                 # read when the context builds its streams (first batch)
        ctx = capi.MnavContext(0)
        ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
        ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
        ctx.set_dijkstra_engine("tile_batch")
        res = []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            b = ctx.plan_dijkstra_batch(goals, targets, want_fields=False, path_cap=65536, want_stats=False)
            dt = time.perf_counter() - t0
            assert (b["codes"] == 0).all(), label
            st = ctx.stats()
            if r:
                res.append(dict(wall_ms=dt * 1e3, engine_ms=st["ms_step_kernels"], steps=st["steps"], algo=st["algorithmic_bytes"]))
        best = min(res, key=lambda x: x["engine_ms"])
        best["frac"] = best["algo"] / best["engine_ms"] / 1e6 / 8000.0
        lens = np.array([len(p) for p in b["paths"]])
        sigs[label] = (int(lens.sum()), [int(np.asarray(p, np.uint64).sum()) for p in b["paths"][:512]])
        out[label] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in best.items()}
        del ctx
    out["paths_identical"] = sigs["plain"] == sigs["pipelined"] == sigs["pipelined_across_chunks"]
    out["speedup_engine"] = round(out["plain"]["engine_ms"] / out["pipelined"]["engine_ms"], 3)
    out["speedup_engine_across_chunks"] = round(out["plain"]["engine_ms"] / out["pipelined_across_chunks"]["engine_ms"], 3)
    print(json.dumps(dict(grid=N, batch=B, **out)))


if __name__ == "__main__":
    main()
