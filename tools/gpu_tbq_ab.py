"""Timing probe of the tile-batch engine on the C2 (1M) and C4 (10M) batch workloads: engine-run time, roofline fraction, phase
cycles with a -DMNAV_TB_TIMING build (MNAV_LIB), band widths (BANDS).  Round 4 used it as the A/B of the one-tile-per-wave solve
against the quarter-wave solve (the GRANS axis: the old kernel is gone, the values are only labels now); the paths of all runs of a
workload must be identical."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh_navigation_amd import capi, meshgen  # noqa: E402


def phases(tag):
    L = capi.load()
    if hasattr(L, "mnav_debug_tb_timing"):
        import ctypes
        tt = (ctypes.c_ulonglong * 8)()
        L.mnav_debug_tb_timing(tt)
        tot = float(sum(tt)) or 1.0
        names = ["fetch", "load", "pre", "sweeps", "writeback", "post", "export", "-"]
        print(tag, "phase Gcycles:", {n: round(tt[i] / 1e9, 2) for i, n in enumerate(names)}, "total %.2f" % (tot / 1e9), file=sys.stderr, flush=True)


def run(N, seed, B, reps, grans, bands=("",)):
    mesh = meshgen.terrain(N, 0.1, seed)
    w = meshgen.edge_lengths(mesh)
    ctx = capi.MnavContext(0)
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
    ctx.upload_costs(np.zeros(mesh.V, np.float32), w)
    robot = mesh.vertex_at(0.9, 0.9)
    g = np.random.default_rng(5).choice(mesh.V, size=B, replace=False).astype(np.uint32)
    ctx.set_dijkstra_engine("tile_batch")
    out, ref = {}, None
    for gran, band in [(g_, b_) for g_ in grans for b_ in bands]:
        os.environ["MNAV_TB_GRAN"] = str(gran)
        if band:
            os.environ["MNAV_TB_BAND_MULT"] = band
        phases("(reset)")
        res = []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            b = ctx.plan_dijkstra_batch(g, np.full(B, robot, np.uint32), want_fields=False, path_cap=65536, want_stats=False)
            dt = time.perf_counter() - t0
            assert (b["codes"] == 0).all()
            st = ctx.stats()
            if r:
                res.append(dict(wall_ms=dt * 1e3, prop_ms=st["ms_propagation"], kern_ms=st["ms_step_kernels"], steps=st["steps"], algo=st["algorithmic_bytes"]))
        best = min(res, key=lambda x: x["kern_ms"])
        best["frac"] = best["algo"] / best["kern_ms"] / 1e6 / 8000.0
        best["plans_per_s"] = B / best["wall_ms"] * 1e3
        lens = np.array([len(p) for p in b["paths"]])
        sig = (int(lens.sum()), int(sum(int(np.asarray(p, np.uint64).sum()) for p in b["paths"][:256])))
        if ref is None:
            ref = sig
        best["paths_equal_first"] = sig == ref
        out[str(gran) + (("/band" + band) if band else "")] = best
        phases("N %d gran %d band %s" % (N, gran, band))
    ctx.close()
    return out


if __name__ == "__main__":
    which = os.environ.get("WHICH", "c2,c4").split(",")
    grans = [int(x) for x in os.environ.get("GRANS", "64,16").split(",")]
    bands = tuple(os.environ.get("BANDS", "").split(",")) if os.environ.get("BANDS") else ("",)
    if "c2" in which:
        print(json.dumps({"C2": run(1000, 2, int(os.environ.get("B2", "7168")), int(os.environ.get("REPS2", "2")), grans, bands)}), flush=True)
    if "c4" in which:
        print(json.dumps({"C4": run(3163, 4, int(os.environ.get("B4", "4096")), 1, grans, bands)}), flush=True)
